// Host-side timing of the mesh build's phases (not part of the product):
//   g++ -O3 -std=c++20 -pthread -I../../../raytracing-one-weekend_amd/csrc hb.cpp -o hb && ./hb mesh.obj
// To time the builder as it was before (node pairs from one atomic counter):
//   mkdir old && git show 7d9cecf^:raytracing-one-weekend_amd/csrc/rtow_bvh.h > old/rtow_bvh.h &&
//   git show 7d9cecf^:raytracing-one-weekend_amd/csrc/rtow_bvh4.h > old/rtow_bvh4.h && g++ ... -Iold hb.cpp
// Measured on the GPU box's 16-core share, 96.8k-triangle mesh (scripts/make_mesh.py 10): before 19-20 ms build_bvh +
// 16 ms image (serial build 41.6 ms); after 8.5-9.5 + 6.2 ms.
#include <chrono>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdint>
#include <algorithm>
#include <atomic>
#include <future>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include "rtow_bvh.h"
#include "rtow_bvh4.h"
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  std::vector<double> tri, sph, sph_r, mov, v;
  std::ifstream f(argv[1]);
  std::string line;
  while (std::getline(f, line)) {
    if (line.size() > 2 && line[0] == 'v' && line[1] == ' ') { std::istringstream s(line.substr(2)); double x, y, z; s >> x >> y >> z; v.insert(v.end(), {x, y, z}); }
    else if (line.size() > 2 && line[0] == 'f' && line[1] == ' ') {
      std::istringstream s(line.substr(2)); std::string tok; int id[3], k = 0;
      while (k < 3 && (s >> tok)) id[k++] = std::stoi(tok.substr(0, tok.find('/'))) - 1;
      if (k < 3) continue;
      double t[12]; const double *a = &v[id[0] * 3], *b = &v[id[1] * 3], *c = &v[id[2] * 3];
      for (int q = 0; q < 3; ++q) { t[q] = a[q]; t[3 + q] = b[q] - a[q]; t[6 + q] = c[q] - a[q]; }
      t[9] = t[4] * t[8] - t[5] * t[7]; t[10] = t[5] * t[6] - t[3] * t[8]; t[11] = t[3] * t[7] - t[4] * t[6];
      tri.insert(tri.end(), t, t + 12);
    }
  }
  const int nt = (int)(tri.size() / 12);
  std::vector<int32_t> pmat(nt, 0); std::vector<unsigned char> mats(48, 0);
  double org[3] = {1, 2, 3};
  for (int rep = 0; rep < 5; ++rep) {
    rtow::HostBvh bvh;
    double t0 = now_ms();
    rtow::build_bvh(sph, sph_r, mov, tri, bvh, 2, 1.5, 0.0, 1.0, rep == 4 ? 1 << 30 : 4096);
    double t1 = now_ms();
    rtow::Bvh4Image img4;
    rtow::make_bvh4_image(bvh, tri, pmat, mats, org, img4, true);
    double t2 = now_ms();
    bool ok = rtow::validate_bvh4_image(img4, (size_t)nt);
    double t3 = now_ms();
    std::printf("%s nt %d: build_bvh %.2f ms, make_bvh4_image %.2f ms, validate %.2f ms (ok %d, nodes %d, depth %d)\n", rep == 4 ? "serial  " : "threaded", nt, t1 - t0, t2 - t1, t3 - t2, (int)ok, img4.n_nodes, bvh.depth);
  }
}
