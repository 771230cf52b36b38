#include "random-utils.h"

#include <cstdint>
#include <random>

namespace rtweekend::detail {

static std::mt19937 &gen() {  // src/random-utils.cpp:6-9
  static std::mt19937 the_generator;
  return the_generator;
}

void reseed_default() { gen().seed(std::mt19937::default_seed); }

// What libstdc++'s uniform_real_distribution<double> does with a 32-bit engine
// (src/random-utils.cpp:11-13): two draws, low word first, (lo + hi*2^32) / 2^64,
// then *(b-a)+a.  Written out so the scene does not depend on the C++ library.
double random_double(double a, double b) {
  const double lo = static_cast<double>(gen()());
  const double hi = static_cast<double>(gen()());
  double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
  if (r >= 1.0) r = 0x1.fffffffffffffp-1;
  return r * (b - a) + a;
}

// uniform_int_distribution{a,b} as libstdc++ 11 implements it for a 32-bit
// engine (Lemire's nearly-divisionless method) — src/random-utils.cpp:15-17.
int random_int(int a, int b) {
  const uint64_t range = static_cast<uint64_t>(static_cast<int64_t>(b) - a) + 1;  // <= 2^32
  if (range == (1ull << 32)) return a + static_cast<int>(gen()());
  uint64_t product = static_cast<uint64_t>(gen()()) * range;
  uint32_t low = static_cast<uint32_t>(product);
  if (low < range) {
    const uint32_t threshold = static_cast<uint32_t>(-static_cast<uint32_t>(range)) % static_cast<uint32_t>(range);
    while (low < threshold) {
      product = static_cast<uint64_t>(gen()()) * range;
      low = static_cast<uint32_t>(product);
    }
  }
  return a + static_cast<int>(product >> 32);
}

// x, y, z in draw order (brace-init, src/random-utils.cpp:19-22)
color random_vec3(double min, double max) {
  const double x = random_double(min, max);
  const double y = random_double(min, max);
  const double z = random_double(min, max);
  return color{x, y, z};
}

}  // namespace rtweekend::detail
