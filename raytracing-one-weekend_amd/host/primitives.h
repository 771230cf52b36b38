// primitives.h — Sphere, MovingSphere, Triangle and their stores.
// One header covers both of the reference's interchangeable primitive models:
// the classes have the constructors of src/oo-primitives.h:26-88 (≡
// src/variant-primitives.h:22-82); PrimitiveStore_t is the OO store
// (src/oo-primitives.h:102), VariantStore is the variant store
// (src/variant-primitives.h:84-102) and World is the legacy spelling from
// src/vmodel.h:250-253.  On the device all of them become the tagged
// class-major record arrays of include/rtow.h.
#pragma once
#include <variant>
#include <vector>

#include "common-model.h"

namespace rtweekend::detail {

class Primitive {
 public:
  enum class Kind { sphere = 0, moving_sphere = 1, triangle = 2 };
  explicit Primitive(const Material &m) : material_{&m} {}
  virtual ~Primitive() = default;
  [[nodiscard]] virtual Kind kind() const = 0;
  [[nodiscard]] const Material &material() const { return *material_; }

 private:
  const Material *material_;
};

class Sphere : public Primitive {
 public:
  Sphere(point center, double radius, const Material &material)
      : Primitive{material}, center_{center}, radius_{radius} {}
  Kind kind() const override { return Kind::sphere; }
  const point &center() const { return center_; }
  [[nodiscard]] const double &radius() const { return radius_; }

 private:
  point center_;
  double radius_;
};

class MovingSphere : public Primitive {
 public:
  MovingSphere(point c0, point c1, double radius, const Material &material)
      : Primitive{material}, center0_{c0}, center1_{c1}, t0_{0.0}, t1_{1.0}, radius_{radius} {}
  Kind kind() const override { return Kind::moving_sphere; }
  const point &center() const { return center0_; }
  const point &center1() const { return center1_; }
  [[nodiscard]] const double &radius() const { return radius_; }
  time_t t0() const { return t0_; }
  time_t t1() const { return t1_; }

 private:
  point center0_, center1_;
  time_t t0_, t1_;
  double radius_;
};

class Triangle : public Primitive {
 public:
  Triangle(point a, point b, point c, const Material &material)
      : Primitive{material}, a_{a}, b_{b}, c_{c} {}
  Kind kind() const override { return Kind::triangle; }
  const point &a() const { return a_; }
  const point &b() const { return b_; }
  const point &c() const { return c_; }

 private:
  point a_, b_, c_;
};

template <typename... T>
class VariantStore : private std::vector<std::variant<T...>> {
 public:
  using value_type = std::variant<T...>;

 private:
  using IBase = std::vector<value_type>;

 public:
  template <typename U, typename... Args>
  U &add(Args &&...args) {
    IBase::push_back(U{std::forward<Args>(args)...});
    return std::get<U>(IBase::back());
  }
  using IBase::begin, IBase::end;
  using IBase::cbegin, IBase::cend, IBase::size, IBase::data;
};

struct World : public VariantStore<Sphere, MovingSphere> {  // src/vmodel.h:250-253
  using Store = VariantStore<Sphere, MovingSphere>;
  OOStore<Material> boutique;
};

}  // namespace rtweekend::detail

namespace rtweekend {
using PrimitiveStore_t = detail::OOStore<detail::Primitive>;
using MaterialStore_t = detail::OOStore<detail::Material>;
using VariantPrimitiveStore_t = detail::VariantStore<detail::Sphere, detail::MovingSphere, detail::Triangle>;  // src/variant-primitives.h:104
using detail::MovingSphere;
using detail::Sphere;
using detail::Triangle;
using detail::VariantStore;
using detail::World;
}  // namespace rtweekend
