// random-utils.h — host-side random numbers for SCENE CONSTRUCTION only.
// Same names and meaning as the reference's src/random-utils.h:9-26: one
// process-global default-seeded std::mt19937 and uniform doubles in [a,b).  The
// render itself does not use this stream (the device path is counter-based).
#pragma once
#include "vec3.h"

namespace rtweekend::detail {
double random_double(double a = 0, double b = 1.0);
int random_int(int a = 0, int b = 1);
color random_vec3(double min = 0, double max = 1.0);
void reseed_default();  // back to std::mt19937's default seed (5489)
}  // namespace rtweekend::detail

namespace rtweekend {
using detail::random_double;
using detail::random_int;
using detail::random_vec3;
}  // namespace rtweekend
