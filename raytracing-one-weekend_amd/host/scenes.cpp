// scenes.cpp — the two scene scripts of the reference's main.cpp, on the host
// scene-build API.  Draw order follows the reference as built by g++ (function
// arguments evaluated right to left), written as explicit statements so the
// scene does not depend on the compiler.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_rng.h"
#include "render.h"

namespace rt = rtweekend;

namespace rtweekend::detail {

// src/main.cpp:23-83, on any of the scene models (render.h): the same calls in the same order, so the three
// flatten to the same records
template <class SceneT>
static SceneT lots_of_balls_on(const Config &cfg) {
  rt::Camera cam{rt::point(13, 2, 3), rt::point(0, 0, 0), rt::vec3(0, 1, 0), 20.0,
                 cfg.aspect_ratio,    0.1,               10.0,              0,
                 1};
  SceneT world{cam};
  auto &boutique = world.boutique();

  auto &ground_material = boutique.template add<rt::Lambertian>(rt::color{0.5, 0.5, 0.5});
  world.primitives().template add<rt::Sphere>(rt::point{0, -1000, 0}, 1000.0, ground_material);

  const int nsqrt = cfg.number_of_balls_sqrt;
  for (int a = -nsqrt; a < nsqrt; a++) {
    for (int b = -nsqrt; b < nsqrt; b++) {
      const double choose_mat = rt::random_double();
      // center(a + 0.9*rd(), 0.2, b + 0.9*rd()): the z argument draws first (src/main.cpp:46)
      const double cz = b + 0.9 * rt::random_double();
      const double cx = a + 0.9 * rt::random_double();
      const rt::point center(cx, 0.2, cz);

      if (length(center - rt::point{4, 0.2, 0}) > 0.9) {
        if (choose_mat < 0.8) {  // diffuse
          const rt::color a1 = rt::random_vec3();
          const rt::color a2 = rt::random_vec3();
          auto &mat = boutique.template add<rt::Lambertian>(a1 * a2);
          if (cfg.moving_spheres) {
            const rt::point center2 = center + rt::point(0, rt::random_double(0, .5), 0);
            world.primitives().template add<rt::MovingSphere>(center, center2, 0.2, mat);
          } else {
            world.primitives().template add<rt::Sphere>(center, 0.2, mat);
          }
        } else if (choose_mat < 0.95) {  // metal
          const rt::color albedo = rt::random_vec3(0.5, 1);
          const double fuzz = rt::random_double(0, 0.5);
          auto &mat = boutique.template add<rt::Metal>(albedo, fuzz);
          world.primitives().template add<rt::Sphere>(center, 0.2, mat);
        } else {  // glass
          auto &mat = boutique.template add<rt::Dielectric>(1.5);
          world.primitives().template add<rt::Sphere>(center, 0.2, mat);
        }
      }
    }
  }

  auto &glass = boutique.template add<rt::Dielectric>(1.5);
  auto &reddish = boutique.template add<rt::Lambertian>(rt::color{0.4, 0.2, 0.1});
  auto &reddish_metal = boutique.template add<rt::Metal>(rt::color{0.7, 0.6, 0.5});
  world.primitives().template add<rt::Sphere>(rt::point(0, 1, 0), 1.0, glass);
  world.primitives().template add<rt::Sphere>(rt::point(-4, 1, 0), 1.0, reddish);
  world.primitives().template add<rt::Sphere>(rt::point(4, 1, 0), 1.0, reddish_metal);
  return world;
}
Scene lots_of_balls(const Config &cfg) { return lots_of_balls_on<Scene>(cfg); }
VariantScene lots_of_balls_variant(const Config &cfg) { return lots_of_balls_on<VariantScene>(cfg); }
WorldScene lots_of_balls_world(const Config &cfg) { return lots_of_balls_on<WorldScene>(cfg); }

// Minimal Wavefront OBJ reader standing in for tinyobjloader 1.0.6 (not in this
// image): vertex positions and the faces of the first shape.
struct ObjMesh {
  std::vector<rt::point> vertices;
  std::vector<std::vector<long>> faces;  // 0-based vertex indices
};

static ObjMesh load_obj(const std::string &path, bool all_shapes) {
  std::FILE *f = std::fopen(path.c_str(), "r");
  if (!f) throw std::runtime_error("Can't load because cannot open " + path);
  ObjMesh m;
  char line[8192];
  bool seen_face = false;
  while (std::fgets(line, sizeof line, f)) {
    char *p = line;
    while (*p == ' ' || *p == '\t') ++p;
    const bool sep = p[1] == ' ' || p[1] == '\t';
    if (p[0] == 'v' && sep) {
      char *q = p + 1;
      const double x = std::strtod(q, &q);
      const double y = std::strtod(q, &q);
      const double z = std::strtod(q, &q);
      m.vertices.emplace_back(x, y, z);
    } else if (p[0] == 'f' && sep) {
      seen_face = true;
      std::vector<long> idx;
      char *q = p + 1;
      for (;;) {
        while (*q == ' ' || *q == '\t') ++q;
        if (*q == '\0' || *q == '\n' || *q == '\r') break;
        char *e = nullptr;
        const long vi = std::strtol(q, &e, 10);
        if (e == q) break;
        const long zero_based = vi > 0 ? vi - 1 : static_cast<long>(m.vertices.size()) + vi;
        if (zero_based < 0 || zero_based >= static_cast<long>(m.vertices.size())) {
          std::fclose(f);
          throw std::runtime_error("Can't load because of a bad vertex index in " + path);
        }
        idx.push_back(zero_based);
        q = e;
        while (*q && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;  // skip /vt/vn
      }
      m.faces.push_back(std::move(idx));
    } else if ((p[0] == 'o' || p[0] == 'g') && sep && seen_face && !all_shapes) {
      break;  // only shapes[0], src/main.cpp:115
    }
  }
  std::fclose(f);
  return m;
}

// src/main.cpp:85-136
template <class SceneT>
static SceneT foo_on(const Config &cfg) {
  rt::random_int();

  rt::Camera cam{rt::point(1, 0, -1), rt::point(0, 0, 0), rt::vec3(0, 1, 0), 35.0,
                 cfg.aspect_ratio,    0.01,              std::nullopt,      0,
                 1};
  SceneT world{cam};
  auto &boring_material = world.boutique().template add<rt::Lambertian>(rt::color{0.5, 0.5, 0.5});

  // (RTOW_GENERAL_OBJ=1: the same for hosts that build scenes through rtow_host_scene_obj)
  const char *genv = std::getenv("RTOW_GENERAL_OBJ");
  const bool general = device_options().general_obj || (genv && genv[0] == '1');
  const ObjMesh mesh = load_obj(cfg.model.value(), general);
  for (const auto &face : mesh.faces) {
    if (face.size() == 3) {
      world.primitives().template add<rt::Triangle>(mesh.vertices[face[0]], mesh.vertices[face[1]],
                                           mesh.vertices[face[2]], boring_material);
    } else if (face.size() > 3) {
      // The reference calls tinyobj 1.0.6's LoadObj with its default triangulate = true
      // (src/main.cpp:109), so polygons reach its loop already cut into the fan (v0, vi, vi+1) and its
      // "isn't a triangle" branch (:130) is only reachable for faces of fewer than three vertices.
      for (size_t i = 1; i + 1 < face.size(); ++i)
        world.primitives().template add<rt::Triangle>(mesh.vertices[face[0]], mesh.vertices[face[i]],
                                             mesh.vertices[face[i + 1]], boring_material);
    } else {
      throw std::runtime_error("Oops found a face that isn't a triangle");
    }
  }
  std::fprintf(stderr, "Scene has %zu triangles\n", world.primitives().size());
  return world;
}
Scene foo(const Config &cfg) { return foo_on<Scene>(cfg); }
VariantScene foo_variant(const Config &cfg) { return foo_on<VariantScene>(cfg); }

}  // namespace rtweekend::detail
