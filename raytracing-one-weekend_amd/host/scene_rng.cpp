#include "scene_rng.h"

namespace rtweekend::detail {

SceneRng &SceneRng::global() {
  static SceneRng the_stream;
  return the_stream;
}

void SceneRng::reseed_default() { engine_.seed(std::mt19937::default_seed); }

double SceneRng::uniform(double a, double b) {
  const double lo = static_cast<double>(engine_());
  const double hi = static_cast<double>(engine_());
  double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
  if (r >= 1.0) r = 0x1.fffffffffffffp-1;  // generate_canonical never returns 1
  return r * (b - a) + a;
}

int SceneRng::uniform_int(int a, int b) {
  const uint64_t range = static_cast<uint64_t>(static_cast<int64_t>(b) - a) + 1;  // <= 2^32
  if (range == (1ull << 32)) return a + static_cast<int>(engine_());
  uint64_t product = static_cast<uint64_t>(engine_()) * range;
  uint32_t low = static_cast<uint32_t>(product);
  if (low < range) {
    const uint32_t threshold =
        static_cast<uint32_t>(-static_cast<uint32_t>(range)) % static_cast<uint32_t>(range);
    while (low < threshold) {
      product = static_cast<uint64_t>(engine_()) * range;
      low = static_cast<uint32_t>(product);
    }
  }
  return a + static_cast<int>(product >> 32);
}

}  // namespace rtweekend::detail
