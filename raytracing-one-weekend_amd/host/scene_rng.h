// scene_rng.h — random numbers for SCENE CONSTRUCTION on the host.
//
// The reference builds its scenes with one process-global, default-seeded std::mt19937
// (src/random-utils.cpp:6-17) and the render then continues on the same stream.  Here the
// render is counter-based on the device, so this generator only feeds the scene scripts
// (host/scenes.cpp) — which must draw exactly what the reference draws, or the scenes
// would differ.  SceneRng makes that stream an explicit object; the free functions below
// keep the reference's spellings (random_double / random_int / random_vec3) for callers.
#pragma once
#include <cstdint>
#include <random>

#include "vec3.h"

namespace rtweekend::detail {

class SceneRng {
 public:
  static SceneRng &global();        // the process-wide stream the scene scripts share
  void reseed_default();            // back to std::mt19937's default seed (5489)
  // [a, b): what libstdc++'s uniform_real_distribution<double> yields from a 32-bit engine —
  // two draws, low word first, (lo + hi*2^32) / 2^64, then *(b-a)+a — written out so the
  // scene does not depend on the C++ library in use
  double uniform(double a, double b);
  // [a, b] integers: libstdc++ 11's uniform_int_distribution (Lemire's method)
  int uniform_int(int a, int b);

 private:
  std::mt19937 engine_;
};

inline double random_double(double a = 0, double b = 1.0) { return SceneRng::global().uniform(a, b); }
inline int random_int(int a = 0, int b = 1) { return SceneRng::global().uniform_int(a, b); }
// components in draw order x, y, z (the reference brace-initialises, src/random-utils.cpp:19-22)
inline color random_vec3(double min = 0, double max = 1.0) {
  SceneRng &g = SceneRng::global();
  const double x = g.uniform(min, max);
  const double y = g.uniform(min, max);
  const double z = g.uniform(min, max);
  return color{x, y, z};
}
inline void reseed_default() { SceneRng::global().reseed_default(); }

}  // namespace rtweekend::detail

namespace rtweekend {
using detail::random_double;
using detail::random_int;
using detail::random_vec3;
}  // namespace rtweekend
