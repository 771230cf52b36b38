// render.h — Config, Scene and render(): the boundary the device path sits behind.
// Same names, fields and defaults as the reference's src/render.h:11-35.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <optional>
#include <string>

#include "primitives.h"

struct rtow_scene_t;

namespace rtweekend::detail {

struct Config {
  int number_of_balls_sqrt = 11;
  double aspect_ratio = 3.0 / 2.0;
  int image_width = 200;
  int samples_per_pixel = 20;
  bool moving_spheres = true;
  int max_child_rays = 20;
  int nthreads = 4;  // on the device: number of sample streams per pixel (same spp rounding)
  std::optional<std::string> model = {};
};

// Knobs of the device path that have no counterpart in the reference's Config.
struct DeviceOptions {
  int device = 0;
  uint64_t seed = 1;
  int precision = 1;  // RTOW_F64_FAST; 0 = RTOW_F64_STRICT
  int kernel = 0;     // RTOW_KERNEL_AUTO
  int builder = -1;   // -1: the context's default (RTOW_BUILDER env); 0 host SAH, 1 device (binned SAH; RTOW_DEVICE_TREE=ploc|radix for the earlier trees), 2 auto
  bool binary_ppm = false;  // P6 (write_color runs on the device) instead of the reference's P3 text
  // OBJ input beyond the reference's (which reads shapes[0] only and throws on a face that is not a
  // triangle, src/main.cpp:115-133): every shape of the file, polygons fan-triangulated
  bool general_obj = false;
  // --gpus N: strips of 8 rows dealt round-robin to devices device .. device+N-1, one host thread and one
  // context per device (rtow_render_multi): ONE ncclGather of the strip buffers to the first device, one
  // device-to-host copy (bench.py is the one-process-per-GPU form of the same partition).
  // gpus_same_device: all N contexts on `device`, every rank copies its own strips to the host — RCCL
  // cannot span one device twice; this tests the partition on a one-GPU box.  rccl: take the RCCL path
  // even with one device (a one-rank communicator; exercises the collective where only one GPU exists).
  int gpus = 1;
  bool gpus_same_device = false;
  bool rccl = false;
  // which of the reference's scene models the scene scripts build (the reference picks at compile time):
  // 0 OO primitives (default, src/oo-primitives.h), 1 variant primitives (src/variant-primitives.h),
  // 2 the World of src/vmodel.h (spheres only)
  int primitives_model = 0;
};
DeviceOptions &device_options();

// Scene (src/render.h:22-33) over either of the reference's two interchangeable primitive stores:
// the OO store (src/oo-primitives.h:102) or the variant store (src/variant-primitives.h:104,
// RTWEEKEND_USE_VARIANT_PRIMITIVES in the reference's build).
template <class PrimStore>
class BasicScene {
  mutable PrimStore primitives_;
  MaterialStore_t boutique_;
  Camera cam_;

 public:
  template <typename T>
  explicit BasicScene(T &&camera) : cam_{std::forward<T>(camera)} {}
  auto &camera() const { return cam_; }
  auto &primitives() { return primitives_; }
  const auto &primitives() const { return primitives_; }
  auto &boutique() { return boutique_; }
  const auto &boutique() const { return boutique_; }
};
using Scene = BasicScene<PrimitiveStore_t>;
using VariantScene = BasicScene<VariantPrimitiveStore_t>;

// The legacy model of src/vmodel.h:250-253 — a World of spheres that owns its materials — has no camera of
// its own; this pairs one with it so that the scene scripts and flatten() treat all three models alike.
struct WorldScene {
  World world;
  Camera cam;
  template <typename T>
  explicit WorldScene(T &&camera) : cam{std::forward<T>(camera)} {}
  auto &camera() const { return cam; }
  auto &primitives() { return static_cast<World::Store &>(world); }
  const auto &primitives() const { return static_cast<const World::Store &>(world); }
  auto &boutique() { return world.boutique; }
  const auto &boutique() const { return world.boutique; }
};

// Flattened copy of a scene (owns the arrays an rtow_scene_t points into).  The three models flatten to
// the same rtow_scene_t when they were built by the same calls (tests/test_host_api.py).
struct FlatScene;
FlatScene *flatten(const Scene &world);
FlatScene *flatten(const VariantScene &world);
FlatScene *flatten(const World &world, const Camera &camera);
const rtow_scene_t *flat_view(const FlatScene *f);
rtow_scene_t *flat_release(FlatScene *f);  // heap rtow_scene_t owning malloc'ed arrays
void flat_free(FlatScene *f);

// Renders on the GPU through the C-ABI and writes the P3 PPM to std::cout,
// diagnostics to std::cerr (src/render.cpp:135-191).  Throws std::runtime_error
// if the device path fails — there is no CPU fallback.
void render(const Scene &world, const Config &cfg);
void render(const VariantScene &world, const Config &cfg);                 // the reference's Scene over variant primitives
void render(const World &world, const Camera &camera, const Config &cfg);  // src/vmodel.h's World

std::ostream &operator<<(std::ostream &o, const Config &c);

// The two scene scripts of the reference's main.cpp, on each of the scene models.
Scene lots_of_balls(const Config &cfg);  // src/main.cpp:23-83
Scene foo(const Config &cfg);            // src/main.cpp:85-136
VariantScene lots_of_balls_variant(const Config &cfg);
VariantScene foo_variant(const Config &cfg);
WorldScene lots_of_balls_world(const Config &cfg);

}  // namespace rtweekend::detail

namespace rtweekend {
using detail::Config;
using detail::DeviceOptions;
using detail::device_options;
using detail::render;
using detail::Scene;
using detail::VariantScene;
using detail::WorldScene;
}  // namespace rtweekend
