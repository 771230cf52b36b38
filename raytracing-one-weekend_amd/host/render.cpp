// render.cpp — host side of the boundary: Camera set-up, Scene flattening, the
// call into the HIP path through the C-ABI, and the reference's PPM writer.
#include "render.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <variant>
#include <vector>

#include <unistd.h>

#include "../../include/rtow.h"

namespace rtweekend::detail {

// src/common-model.cpp:136-154 — same expressions, same order.
Camera::Camera(point lookfrom, point lookat, vec3 vup, double fov, double aspect_ratio,
               double aperture, std::optional<double> focus_dist, time_t t0, time_t t1)
    : origin_{lookfrom},
      w_{normalize(lookfrom - lookat)},
      u_{normalize(cross(vup, w_))},
      v_{normalize(cross(w_, u_))},
      lens_radius_{aperture / 2},
      t0_{t0},
      t1_{t1} {
  const double pi = 3.141592653589793238462643383279502884;
  auto viewport_height = 2.0 * std::tan(fov * pi / 180 / 2);
  auto viewport_width = aspect_ratio * viewport_height;
  auto fd = focus_dist ? focus_dist.value() : length(lookfrom - lookat);
  horizontal_ = fd * viewport_width * u_;
  vertical_ = fd * viewport_height * v_;
  lower_left_corner_ = origin_ - horizontal_ / 2.0 - vertical_ / 2.0 - fd * w_;
}

DeviceOptions &device_options() {
  static DeviceOptions o;
  return o;
}

// ------------------------------------------------------------------ flatten ---
struct FlatScene {
  rtow_scene_t s{};
  std::vector<double> sg, mg, tg;
  std::vector<int32_t> sm, mm, tm, pk, pi;
  std::vector<rtow_material_t> mats;
  void bind() {
    s.n_spheres = (int32_t)sm.size();
    s.sphere_geom = sg.data();
    s.sphere_mat = sm.data();
    s.n_moving = (int32_t)mm.size();
    s.moving_geom = mg.data();
    s.moving_mat = mm.data();
    s.n_triangles = (int32_t)tm.size();
    s.triangle_geom = tg.data();
    s.triangle_mat = tm.data();
    s.n_materials = (int32_t)mats.size();
    s.materials = mats.data();
    s.n_prims = (int32_t)pk.size();
    s.prim_kind = pk.data();
    s.prim_index = pi.data();
  }
};

static void put3(double *d, const vec3 &v) {
  d[0] = v.x;
  d[1] = v.y;
  d[2] = v.z;
}

static void put_camera(FlatScene *f, const Camera &c) {
  put3(f->s.camera.origin, c.origin());
  put3(f->s.camera.u, c.u());
  put3(f->s.camera.v, c.v());
  put3(f->s.camera.w, c.w());
  put3(f->s.camera.horizontal, c.horizontal());
  put3(f->s.camera.vertical, c.vertical());
  put3(f->s.camera.lower_left_corner, c.lower_left_corner());
  f->s.camera.lens_radius = c.lens_radius();
  f->s.camera.t0 = c.t0();
  f->s.camera.t1 = c.t1();
}

// one primitive -> its class's record array (insertion order kept in prim_kind / prim_index)
static void put_primitive(FlatScene *f, const Sphere &s, int32_t mi) {
  f->pk.push_back(RTOW_PRIM_SPHERE);
  f->pi.push_back((int32_t)f->sm.size());
  f->sg.insert(f->sg.end(), {s.center().x, s.center().y, s.center().z, s.radius()});
  f->sm.push_back(mi);
}
static void put_primitive(FlatScene *f, const MovingSphere &s, int32_t mi) {
  f->pk.push_back(RTOW_PRIM_MOVING_SPHERE);
  f->pi.push_back((int32_t)f->mm.size());
  f->mg.insert(f->mg.end(), {s.center().x, s.center().y, s.center().z, s.center1().x, s.center1().y, s.center1().z,
                             s.radius(), 0.0});
  f->mm.push_back(mi);
}
static void put_primitive(FlatScene *f, const Triangle &t, int32_t mi) {
  f->pk.push_back(RTOW_PRIM_TRIANGLE);
  f->pi.push_back((int32_t)f->tm.size());
  f->tg.insert(f->tg.end(), {t.a().x, t.a().y, t.a().z, t.b().x, t.b().y, t.b().z, t.c().x, t.c().y, t.c().z});
  f->tm.push_back(mi);
}
// the two store flavours: OO (unique_ptr<Primitive>, dispatch on kind()) and variant (std::visit,
// src/variant-primitives.h:107-113)
static const Primitive &as_primitive(const std::unique_ptr<Primitive> &p) { return *p; }
template <class... T>
static const Primitive &as_primitive(const std::variant<T...> &v) {
  return std::visit([](const auto &x) -> const Primitive & { return x; }, v);
}
static void put_any(FlatScene *f, const std::unique_ptr<Primitive> &p, int32_t mi) {
  switch (p->kind()) {
    case Primitive::Kind::sphere: put_primitive(f, static_cast<const Sphere &>(*p), mi); break;
    case Primitive::Kind::moving_sphere: put_primitive(f, static_cast<const MovingSphere &>(*p), mi); break;
    case Primitive::Kind::triangle: put_primitive(f, static_cast<const Triangle &>(*p), mi); break;
  }
}
template <class... T>
static void put_any(FlatScene *f, const std::variant<T...> &v, int32_t mi) {
  std::visit([&](const auto &x) { put_primitive(f, x, mi); }, v);
}

template <class PrimStore>
static FlatScene *flatten_parts(const Camera &cam, const PrimStore &prims, const MaterialStore_t &boutique) {
  auto f = std::make_unique<FlatScene>();
  put_camera(f.get(), cam);
  std::unordered_map<const Material *, int32_t> mat_index;
  for (auto it = boutique.cbegin(); it != boutique.cend(); ++it) {
    const Material *m = it->get();
    rtow_material_t r;
    std::memset(&r, 0, sizeof r);
    r.kind = static_cast<int32_t>(m->kind());
    switch (m->kind()) {
      case Material::Kind::lambertian:
        put3(r.albedo, static_cast<const Lambertian *>(m)->albedo);
        break;
      case Material::Kind::metal:
        put3(r.albedo, static_cast<const Metal *>(m)->albedo);
        r.fuzz = static_cast<const Metal *>(m)->fuzz;
        break;
      case Material::Kind::dielectric:
        r.ir = static_cast<const Dielectric *>(m)->ir;
        r.fuzz = static_cast<const Dielectric *>(m)->fuzz;
        break;
    }
    mat_index[m] = (int32_t)f->mats.size();
    f->mats.push_back(r);
  }
  for (auto it = prims.cbegin(); it != prims.cend(); ++it) {
    auto found = mat_index.find(&as_primitive(*it).material());
    if (found == mat_index.end())
      throw std::runtime_error("primitive refers to a material that is not in the scene's boutique");
    put_any(f.get(), *it, found->second);
  }
  f->bind();
  return f.release();
}

FlatScene *flatten(const Scene &world) { return flatten_parts(world.camera(), world.primitives(), world.boutique()); }
FlatScene *flatten(const VariantScene &world) {
  return flatten_parts(world.camera(), world.primitives(), world.boutique());
}
FlatScene *flatten(const World &world, const Camera &camera) {
  return flatten_parts(camera, static_cast<const World::Store &>(world), world.boutique);
}

const rtow_scene_t *flat_view(const FlatScene *f) { return &f->s; }
void flat_free(FlatScene *f) { delete f; }

template <class T>
static T *dup_array(const std::vector<T> &v) {
  T *p = static_cast<T *>(std::malloc(sizeof(T) * (v.empty() ? 1 : v.size())));
  if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

rtow_scene_t *flat_release(FlatScene *f) {
  auto *s = static_cast<rtow_scene_t *>(std::malloc(sizeof(rtow_scene_t)));
  *s = f->s;
  s->sphere_geom = dup_array(f->sg);
  s->sphere_mat = dup_array(f->sm);
  s->moving_geom = dup_array(f->mg);
  s->moving_mat = dup_array(f->mm);
  s->triangle_geom = dup_array(f->tg);
  s->triangle_mat = dup_array(f->tm);
  s->materials = dup_array(f->mats);
  s->prim_kind = dup_array(f->pk);
  s->prim_index = dup_array(f->pi);
  delete f;
  return s;
}

// ------------------------------------------------------------------- render ---
// src/render.cpp:11-20
static void write_color(std::string &out, double r, double g, double b, int samples_per_pixel) {
  const double d = static_cast<double>(samples_per_pixel);
  const double c[3] = {std::sqrt(r / d), std::sqrt(g / d), std::sqrt(b / d)};
  char buf[48];
  int n = std::snprintf(buf, sizeof buf, "%d %d %d\n",
                        static_cast<int>(256 * std::clamp(c[0], 0.0, 0.999)),
                        static_cast<int>(256 * std::clamp(c[1], 0.0, 0.999)),
                        static_cast<int>(256 * std::clamp(c[2], 0.0, 0.999)));
  out.append(buf, (size_t)n);
}

std::string ppm_text(const double *rgb_sums, int width, int height, int spp_effective) {
  std::string s;
  s.reserve((size_t)width * height * 12 + 32);
  s += "P3\n" + std::to_string(width) + ' ' + std::to_string(height) + "\n255\n";  // :182
  for (size_t p = 0; p < (size_t)width * height; ++p)
    write_color(s, rgb_sums[p * 3], rgb_sums[p * 3 + 1], rgb_sums[p * 3 + 2], spp_effective);
  return s;
}

static void render_flat(FlatScene *flat, const Config &cfg);
void render(const Scene &world, const Config &cfg) { render_flat(flatten(world), cfg); }
void render(const VariantScene &world, const Config &cfg) { render_flat(flatten(world), cfg); }
void render(const World &world, const Camera &camera, const Config &cfg) { render_flat(flatten(world, camera), cfg); }

// (takes ownership of `flat`)
static void render_flat(FlatScene *flat, const Config &cfg) {
  int image_height = static_cast<int>(cfg.image_width / cfg.aspect_ratio);  // src/render.cpp:137
  namespace khr = std::chrono;
  const DeviceOptions &opt = device_options();
  std::cerr << "Started rendering with " << cfg.nthreads << " sample streams on HIP device "
            << opt.device << (opt.gpus > 1 ? " (+" + std::to_string(opt.gpus - 1) + " more)" : std::string()) << "\n";
  khr::time_point start{khr::high_resolution_clock::now()};

  rtow_config_t rc;
  std::memset(&rc, 0, sizeof rc);
  rc.image_width = cfg.image_width;
  rc.image_height = image_height;
  rc.samples_per_pixel = cfg.samples_per_pixel;
  rc.nstreams = cfg.nthreads;
  rc.max_child_rays = cfg.max_child_rays;
  rc.precision = opt.precision;
  rc.kernel = opt.kernel;
  rc.rank = 0;
  rc.nranks = 1;
  rc.tile_rows = 8;
  rc.seed = opt.seed;

  const size_t nvalues = (size_t)cfg.image_width * (image_height > 0 ? image_height : 0) * 3;
  std::vector<double> image(opt.binary_ppm ? 0 : nvalues);
  std::vector<unsigned char> rgb8(opt.binary_ppm ? nvalues : 0);
  rtow_stats_t st;
  std::memset(&st, 0, sizeof st);
  rtow_build_info_t bi;
  std::memset(&bi, 0, sizeof bi);
  const int ngpus = std::max(opt.gpus, 1);
  if (ngpus > 1 || opt.rccl) {
    // several devices (or --rccl): rtow_render_multi — one host thread and context per device, strips dealt
    // round-robin, one ncclGather to the first device + one D2H (distinct devices), or one D2H per rank
    // (--gpus-same-device: RCCL cannot span a device twice)
    std::vector<int32_t> devs((size_t)ngpus);
    for (int r = 0; r < ngpus; ++r) devs[(size_t)r] = opt.gpus_same_device ? opt.device : opt.device + r;
    const int use_rccl = opt.gpus_same_device && ngpus > 1 ? 0 : 1;
    // RCCL prints a version banner on stdout when it initialises; stdout is the PPM (src/render.cpp:182-186),
    // so the banner is sent to stderr while the communicator is created (rtow_multi_create; nothing else of
    // this process writes to stdout in the meantime)
    std::cout.flush();
    std::fflush(stdout);
    const int saved_stdout = dup(1);
    if (saved_stdout >= 0) (void)dup2(2, 1);
    rtow_multi *multi = nullptr;
    int err = rtow_multi_create(ngpus, devs.data(), use_rccl, &multi);
    std::fflush(stdout);
    if (saved_stdout >= 0) {
      (void)dup2(saved_stdout, 1);
      close(saved_stdout);
    }
    std::string msg = err == RTOW_OK ? std::string() : std::string(rtow_last_error());
    if (err == RTOW_OK && opt.builder >= 0 && (err = rtow_multi_set_builder(multi, opt.builder))) msg = rtow_last_error();
    if (err == RTOW_OK && nvalues) {
      // --p6: write_color runs on the device of the rank that owns the pixel; bytes are gathered and copied
      if ((err = rtow_multi_upload(multi, flat_view(flat))) ||
          (err = opt.binary_ppm ? rtow_multi_render_rgb8(multi, &rc, rgb8.data(), &st)
                                : rtow_multi_render(multi, &rc, image.data(), &st)))
        msg = rtow_last_error();
      else
        (void)rtow_multi_build_info(multi, &bi);
    }
    rtow_multi_destroy(multi);
    flat_free(flat);
    if (err != RTOW_OK) throw std::runtime_error("HIP render failed (" + std::to_string(err) + "): " + msg);
  } else {
    rtow_ctx *ctx = nullptr;
    int err = rtow_ctx_create(opt.device, &ctx);
    if (err == RTOW_OK && opt.builder >= 0) err = rtow_ctx_set_builder(ctx, opt.builder);
    if (err == RTOW_OK && nvalues)
      err = opt.binary_ppm ? rtow_render_rgb8(ctx, flat_view(flat), &rc, rgb8.data(), &st)
                           : rtow_render(ctx, flat_view(flat), &rc, image.data(), &st);
    const std::string msg = err == RTOW_OK ? std::string() : std::string(rtow_last_error());
    if (err == RTOW_OK && nvalues) (void)rtow_build_info(ctx, &bi);
    rtow_ctx_destroy(ctx);
    flat_free(flat);
    if (err != RTOW_OK) throw std::runtime_error("HIP render failed (" + std::to_string(err) + "): " + msg);
  }

  const int spp_eff = cfg.samples_per_pixel / cfg.nthreads * cfg.nthreads;  // src/render.cpp:185
  if (opt.binary_ppm) {
    std::cout << "P6\n" << cfg.image_width << ' ' << image_height << "\n255\n";
    std::cout.write(reinterpret_cast<const char *>(rgb8.data()), (std::streamsize)rgb8.size());
  } else {
    std::cout << ppm_text(image.data(), cfg.image_width, image_height, spp_eff);
  }

  auto took = khr::high_resolution_clock::now() - start;
  if (bi.ref_tree_nodes > 0)  // --kernel reftree: the reference's own tree and its diagnostic (src/render.cpp:148)
    std::cerr << "Total BVH stupid volume: " << bi.ref_tree_stupid_volume << "\n"
              << "Reference tree: " << bi.ref_tree_nodes << " nodes, built in " << bi.ref_tree_build_ms << " ms\n";
  // what the device walked (the reference prints its tree's diagnostic here, src/render.cpp:148; only the
  // structures this render's kernel reads are built)
  if (bi.bvh_image_bytes > 0)
    std::cerr << "BVH image: " << bi.bvh_nodes << " nodes, " << bi.bvh_image_bytes << " bytes, built on the "
              << (bi.builder == RTOW_BUILDER_DEVICE_LBVH ? "device" : "host (SAH)") << " in " << bi.bvh_build_ms
              << " ms" << (bi.bvh4_nodes > 0 ? "; 4-wide image: " + std::to_string(bi.bvh4_nodes) + " nodes of " + std::to_string(bi.bvh4_node_bytes) + " bytes"
                                   : std::string()) << "\n";
  if (bi.grid_image_bytes > 0)
    std::cerr << "Grid image: " << bi.grid_image_bytes << " bytes, built in " << bi.grid_build_ms << " ms\n";
  std::cerr << "Traced " << st.samples << " samples, " << st.segments << " ray segments; kernel "
            << st.kernel_ms << " ms ("
            << (st.kernel_ms > 0 ? st.samples / st.kernel_ms / 1e3 : 0.0) << " Msamples/s)\n";
  // The reference counts scanlines down on stderr as each of its threads works through the image
  // (src/render.cpp:154).  Here every row was finished by the one call above, so the countdown has a single
  // state left to show — its last one; `\nDone in` follows it exactly as in the reference (:188-189).
  std::cerr << "\rScanlines remaining: " << 0 << ' ' << std::flush;
  std::cerr << "\nDone in " << khr::duration_cast<khr::milliseconds>(took).count() << "ms\n";
}

// src/render.cpp:193-203
std::ostream &operator<<(std::ostream &o, const Config &c) {
  return o << "Config {\n"
           << "aspect_ratio: " << c.aspect_ratio << "\n"
           << "number_of_balls_sqrt: " << c.number_of_balls_sqrt << "\n"
           << "moving_spheres: " << c.moving_spheres << "\n"
           << "image_width: " << c.image_width << "\n"
           << "samples_per_pixel: " << c.samples_per_pixel << "\n"
           << "max_child_rays: " << c.max_child_rays << "\n"
           << "nthreads: " << c.nthreads << "\n"
           << "}\n";
}

}  // namespace rtweekend::detail
