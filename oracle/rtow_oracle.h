/*
 * rtow_oracle.h — C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  The oracle is a CPU restatement of the reference's
 * per-pixel sample loop (src/render.cpp, src/common-model.cpp,
 * src/random-utils.cpp of joaotavora/raytracing-one-weekend).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only
 * as the checker / the timed CPU baseline — never on the product path.
 *
 * Parity pin: the reference has no tests or golden files of its own and cannot
 * be built in this image (it needs glm, absent; stand-in headers are not
 * allowed).  The oracle is pinned by the five PPM md5 sums recorded in
 * SURVEY.md §8c, which were produced from the reference's own sources; see
 * tests/test_oracle_golden.py and DESIGN.md §3.
 */
#ifndef RTOW_ORACLE_H
#define RTOW_ORACLE_H

#include "../include/rtow.h"

#ifdef __cplusplus
extern "C" {
#endif

/* RNG policies */
#define ORC_RNG_MT19937 0 /* the reference's process-global std::mt19937 stream */
#define ORC_RNG_PHILOX 1  /* the device path's counter-based stream             */

typedef struct orc_stats_t {
  uint64_t samples, segments, prim_tests, node_tests, rng_doubles;
  double bvh_stupid_volume; /* src/render.cpp:36-50, printed at :148 */
  int32_t bvh_nodes, bvh_leaves;
} orc_stats_t;

/* Reset the global mt19937 to its default seed (5489). */
void orc_mt_reset(void);
/* Burn n random_double() draws from the global stream. */
void orc_mt_burn(uint64_t n);

/* Scene scripts of src/main.cpp (consume the global mt19937 stream). */
int orc_scene_cover(int number_of_balls_sqrt, double aspect_ratio, int moving_spheres,
                    rtow_scene_t **out);
int orc_scene_obj(const char *path, double aspect_ratio, rtow_scene_t **out);
void orc_scene_free(rtow_scene_t *s);

/* render(): fills `rgb_sums` (local_rows*W*3 doubles; cfg->rank/nranks/tile_rows
 * select the rows exactly as rtow_render_device does).  nthreads > 1 is only
 * meaningful (and race-free) with ORC_RNG_PHILOX. */
int orc_render(const rtow_scene_t *scene, const rtow_config_t *cfg, int rng_mode, int nthreads,
               double *rgb_sums, orc_stats_t *stats);

/* The same render with the closest hit found through a tree of the checker's own when accel != 0
 * (SAH build over exact f64 bounds, near-first descent) instead of the reference's median-split
 * tree: same primitive tests, same image (tests/test_oracle_units.py compares the two bit for bit),
 * ~100x faster on the 96,800-triangle mesh.  stats->node_tests / prim_tests then count the
 * checker's tree, not the reference's. */
int orc_render_ex(const rtow_scene_t *scene, const rtow_config_t *cfg, int rng_mode, int nthreads, int accel,
                  double *rgb_sums, orc_stats_t *stats);

/* write_color + P3 PPM text (src/render.cpp:11-20,182-186); free with orc_free. */
int orc_ppm(const double *rgb_sums, int width, int height, int spp_effective, char **out_text,
            uint64_t *out_len);
void orc_free(void *p);

/* Optional log of every traced segment (12 doubles each: pixel, sample, segment,
 * origin xyz, direction xyz, time, t_hit or inf, class index of the hit primitive);
 * single-threaded Philox renders only.  Used by tests/tools/sim_traversal.py. */
void orc_set_raylog(double *buf, uint64_t capacity_records);
uint64_t orc_raylog_count(void);

/* Function-level probes (for unit tests). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4], int rounds);
int orc_philox_rounds(void);
/* value k of request `request`: kind 0 = the block of a new sample (k = 0, 1 jitter; 2 shutter time; 3, 4 the lens
 * point x, y), kind 2 = the block of a bounce (k = 0..2 the point of the unit ball; 3 the dielectric coin) */
double orc_philox_request(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t request, int kind,
                          int k);
double orc_mt_random_double(double a, double b);
/* returns 1 on hit and writes t, point[3], normal[3], front_facing */
int orc_sphere_hit(const double center[3], double radius, const double ro[3], const double rd[3],
                   double tmin, double tmax, double *t, double *p, double *n, int *front);
int orc_triangle_hit(const double a[3], const double b[3], const double c[3], const double ro[3],
                     const double rd[3], double tmin, double tmax, double *t, double *p, double *n);
int orc_aabb_hit(const double bmin[3], const double bmax[3], const double ro[3],
                 const double rd[3], double tmin, double tmax);

#ifdef __cplusplus
}
#endif
#endif
