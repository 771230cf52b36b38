// common-model.h — host scene-build API: Camera, materials, OOStore.
// Same class names, constructor arguments and defaults as the reference's
// src/common-model.h (Camera :91-113, Material/Lambertian/Metal/Dielectric
// :115-151, OOStore :153-167).  These are construction-time data holders: the hit
// and scatter arithmetic of the reference's common-model.cpp runs on the GPU
// (csrc/rtow_trace_body.h), so the classes expose accessors instead of
// hit()/scatter().
#pragma once
#include <algorithm>
#include <memory>
#include <optional>
#include <vector>

#include "vec3.h"

namespace rtweekend::detail {

using time_t = double;

class Camera {
 public:
  static constexpr auto focal_length = 1.0;
  Camera(point lookfrom, point lookat, vec3 vup, double fov, double aspect_ratio, double aperture,
         std::optional<double> focus_dist = std::nullopt, time_t t0 = 0, time_t t1 = 0);

  // accessors the reference lacks (its members are private, src/common-model.h:104-112);
  // the flattening step needs them
  const point &origin() const { return origin_; }
  const vec3 &w() const { return w_; }
  const vec3 &u() const { return u_; }
  const vec3 &v() const { return v_; }
  const vec3 &horizontal() const { return horizontal_; }
  const vec3 &vertical() const { return vertical_; }
  const point &lower_left_corner() const { return lower_left_corner_; }
  double lens_radius() const { return lens_radius_; }
  time_t t0() const { return t0_; }
  time_t t1() const { return t1_; }

 private:
  point origin_;
  vec3 w_, u_, v_;
  vec3 horizontal_{};
  vec3 vertical_{};
  point lower_left_corner_{};
  double lens_radius_;
  double t0_, t1_;
};

class Material {
 public:
  enum class Kind { lambertian = 0, metal = 1, dielectric = 2 };
  [[nodiscard]] virtual Kind kind() const = 0;
  virtual ~Material() = default;
};

struct Lambertian : public Material {
  explicit Lambertian(const color &albedo) : albedo{albedo} {}
  Kind kind() const override { return Kind::lambertian; }
  color albedo;
};

struct Metal : public Material {
  explicit Metal(const color &albedo, double fuzz = 0)
      : albedo{albedo}, fuzz{std::clamp(fuzz, 0.0, 1.0)} {}
  Kind kind() const override { return Kind::metal; }
  color albedo;
  double fuzz;
};

struct Dielectric : public Material {
  explicit Dielectric(double index_of_refraction, double fuzz = 0)
      : ir{index_of_refraction}, fuzz{std::clamp(fuzz, 0.0, 1.0)} {}
  Kind kind() const override { return Kind::dielectric; }
  double ir;
  double fuzz;
};

// Heap store with stable references: add<Derived>(args...) returns Derived&.
template <typename Base>
class OOStore : private std::vector<std::unique_ptr<Base>> {
  using IBase = std::vector<std::unique_ptr<Base>>;

 public:
  template <typename Derived, typename... Args>
  Derived &add(Args &&...args) {
    IBase::push_back(std::make_unique<Derived>(std::forward<Args>(args)...));
    return static_cast<Derived &>(*IBase::back());
  }
  using IBase::begin, IBase::end;
  using IBase::cbegin, IBase::cend, IBase::size, IBase::data;
};

}  // namespace rtweekend::detail

namespace rtweekend {
using detail::Camera;
using detail::Dielectric;
using detail::Lambertian;
using detail::Material;
using detail::Metal;
using detail::OOStore;
}  // namespace rtweekend
