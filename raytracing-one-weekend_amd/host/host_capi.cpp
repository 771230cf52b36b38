// host_capi.cpp — C entry points over the host scene-build API (include/rtow.h,
// "host-side scene construction").  No GPU involved.
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "../../include/rtow.h"
#include "../csrc/rtow_reftree.h"
#include "scene_rng.h"
#include "render.h"

namespace rtweekend::detail {
std::string ppm_text(const double *rgb_sums, int width, int height, int spp_effective);
}

namespace rt = rtweekend;

extern "C" {

void rtow_host_rng_reset(void) { rt::detail::reseed_default(); }

static rt::Config config_from(const rtow_host_config_t *hc) {
  rt::Config cfg;
  cfg.number_of_balls_sqrt = hc->number_of_balls_sqrt;
  cfg.aspect_ratio = hc->aspect_ratio;
  cfg.moving_spheres = hc->moving_spheres != 0;
  return cfg;
}

int rtow_host_scene_cover_model(const rtow_host_config_t *hc, int32_t model, rtow_scene_t **out) {
  if (!hc || !out) return RTOW_EINVAL;
  try {
    switch (model) {
      case RTOW_MODEL_OO: {
        rt::Scene world = rt::detail::lots_of_balls(config_from(hc));
        *out = rt::detail::flat_release(rt::detail::flatten(world));
        return RTOW_OK;
      }
      case RTOW_MODEL_VARIANT: {
        rt::VariantScene world = rt::detail::lots_of_balls_variant(config_from(hc));
        *out = rt::detail::flat_release(rt::detail::flatten(world));
        return RTOW_OK;
      }
      case RTOW_MODEL_WORLD: {
        rt::WorldScene ws = rt::detail::lots_of_balls_world(config_from(hc));
        *out = rt::detail::flat_release(rt::detail::flatten(ws.world, ws.cam));
        return RTOW_OK;
      }
      default: return RTOW_EINVAL;
    }
  } catch (...) {
    return RTOW_EINVAL;
  }
}
int rtow_host_scene_cover(const rtow_host_config_t *hc, rtow_scene_t **out) {
  return rtow_host_scene_cover_model(hc, RTOW_MODEL_OO, out);
}

int rtow_host_scene_obj_model(const rtow_host_config_t *hc, const char *obj_path, int32_t model, rtow_scene_t **out) {
  if (!hc || !out || !obj_path) return RTOW_EINVAL;
  try {
    rt::Config cfg = config_from(hc);
    cfg.model = std::string(obj_path);
    switch (model) {
      case RTOW_MODEL_OO: {
        rt::Scene world = rt::detail::foo(cfg);
        *out = rt::detail::flat_release(rt::detail::flatten(world));
        return RTOW_OK;
      }
      case RTOW_MODEL_VARIANT: {
        rt::VariantScene world = rt::detail::foo_variant(cfg);
        *out = rt::detail::flat_release(rt::detail::flatten(world));
        return RTOW_OK;
      }
      default: return RTOW_EINVAL;  // (src/vmodel.h's World holds spheres only)
    }
  } catch (...) {
    return RTOW_EINVAL;
  }
}
int rtow_host_scene_obj(const rtow_host_config_t *hc, const char *obj_path, rtow_scene_t **out) {
  return rtow_host_scene_obj_model(hc, obj_path, RTOW_MODEL_OO, out);
}

void rtow_host_scene_free(rtow_scene_t *s) {
  if (!s) return;
  std::free((void *)s->sphere_geom);
  std::free((void *)s->sphere_mat);
  std::free((void *)s->moving_geom);
  std::free((void *)s->moving_mat);
  std::free((void *)s->triangle_geom);
  std::free((void *)s->triangle_mat);
  std::free((void *)s->materials);
  std::free((void *)s->prim_kind);
  std::free((void *)s->prim_index);
  std::free(s);
}

int rtow_host_ppm(const double *rgb_sums, int32_t width, int32_t height, int32_t spp_effective,
                  char **out_text, uint64_t *out_len) {
  if (!rgb_sums || !out_text || !out_len || width <= 0 || height <= 0) return RTOW_EINVAL;
  try {
    std::string s = rt::detail::ppm_text(rgb_sums, width, height, spp_effective);
    char *p = static_cast<char *>(std::malloc(s.size() + 1));
    if (!p) return RTOW_ENOMEM;
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *out_text = p;
    *out_len = s.size();
    return RTOW_OK;
  } catch (...) {
    return RTOW_ENOMEM;
  }
}

void rtow_host_free(void *p) { std::free(p); }

// The reference's own tree over a flattened scene (csrc/rtow_reftree.h — what RTOW_KERNEL_REFTREE walks), built
// on the host without a GPU: node count, depth and the reference's "Total BVH stupid volume" diagnostic.
int rtow_host_reftree_info(const rtow_scene_t *scene, int32_t *n_nodes, int32_t *depth, double *stupid_volume) {
  if (!scene || scene->n_prims <= 0) return RTOW_EINVAL;
  try {
    rtow::RefTree t;
    rtow::build_reftree(scene, t);
    if (n_nodes) *n_nodes = t.n_nodes;
    if (depth) *depth = t.depth;
    if (stupid_volume) *stupid_volume = t.stupid_volume;
    return t.ok ? RTOW_OK : RTOW_EINVAL;
  } catch (...) {
    return RTOW_ENOMEM;
  }
}

}  // extern "C"
