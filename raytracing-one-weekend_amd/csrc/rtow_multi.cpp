// rtow_multi.cpp — rtow_render_multi: one frame over several HIP devices from ONE process.
//
// The reference fans its render out over `nthreads` workers and adds their partial images in
// launch order (src/render.cpp:169-180).  Here a worker is a device: one host thread and one
// context per device, the image cut into strips of cfg->tile_rows rows dealt round-robin (the
// partition of include/rtow.h, rank = position in `device_ids`), every device traces its strips
// into a [max_rows][W][3] f64 buffer, and ONE ncclGather (RCCL over xGMI) brings the buffers to
// the first device, from where ONE device-to-host copy delivers them; the host then puts every
// strip row in its place.  Any N gives the one-device image bit for bit (a pixel's value does not
// depend on who traces it: counter-based RNG, fixed sample order).
//
// RCCL is loaded on demand (dlopen of librccl.so) so that librtow.so itself does not depend on it:
// the one-process-per-GPU form (bench.py, torch.distributed) brings its own RCCL.  With
// use_rccl == 0 every device copies its strips to the host itself (no collective): the form
// `rtweekend --gpus N --gpus-same-device` uses to test the partition on a one-GPU box, where a
// communicator over duplicate devices cannot exist.
//
// Measured: only with ONE device in the communicator (the GPU boxes of this project have one);
// the N > 1 RCCL path is correct by construction and covered as far as one device allows
// (tests/test_gpu_parity.py::test_rtweekend_rccl_path_with_one_rank).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtow.h"

namespace rtow {
int set_last_error(int code, const char *fmt, ...);  // rtow_capi.cpp
}

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGather) Gather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string &why) {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
      return false;
    }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    Gather = (decltype(Gather))dlsym(lib, "ncclGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !Gather || !GetErrorString) {
      why = "librccl.so lacks ncclCommInitAll / ncclGather";
      return false;
    }
    return true;
  }
};

struct Rank {
  rtow_ctx *ctx = nullptr;
  void *d_local = nullptr;
  hipStream_t stream = nullptr;
  int err = RTOW_OK;
  std::string msg;
  rtow_stats_t stats{};
};

}  // namespace

extern "C" int rtow_render_multi(int32_t n_devices, const int32_t *device_ids, const rtow_scene_t *scene,
                                 const rtow_config_t *cfg, double *rgb_sums_host, rtow_stats_t *stats,
                                 int32_t use_rccl) {
  if (n_devices < 1 || n_devices > 64 || !device_ids) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: 1..64 devices");
  if (!scene || !cfg || !rgb_sums_host) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: NULL argument");
  if (cfg->accumulate) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: accumulate is not supported");
  rtow_config_t probe = *cfg;
  probe.rank = 0;
  probe.nranks = n_devices;
  int max_rows = rtow_local_rows(&probe);  // rank 0 owns the first strip: nobody has more rows
  if (max_rows < 0) return max_rows;
  const int W = cfg->image_width, H = cfg->image_height;
  const size_t count = (size_t)max_rows * (size_t)W * 3;  // doubles per rank in the gather
  if (count == 0) return RTOW_OK;

  Rccl rccl;
  std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
  if (use_rccl) {
    std::string why;
    if (!rccl.load(why)) return rtow::set_last_error(RTOW_ENODEV, "%s", why.c_str());
    std::vector<int> devs(device_ids, device_ids + n_devices);
    const ncclResult_t r = rccl.CommInitAll(comms.data(), n_devices, devs.data());
    if (r != ncclSuccess) {
      const int rc = rtow::set_last_error(RTOW_EHIP, "ncclCommInitAll over %d devices: %s", n_devices, rccl.GetErrorString(r));
      dlclose(rccl.lib);
      return rc;
    }
  }

  std::vector<Rank> ranks((size_t)n_devices);
  void *d_gather = nullptr;  // on the first device: [rank][max_rows][W][3]
  std::vector<double> staging((size_t)n_devices * count);

  auto work = [&](int r) {
    Rank &me = ranks[(size_t)r];
    auto fail = [&](int code, const char *what) {
      me.err = code;
      me.msg = std::string(what) + ": " + rtow_last_error();
    };
    rtow_config_t mine = *cfg;
    mine.rank = r;
    mine.nranks = n_devices;
    int rc = rtow_ctx_create(device_ids[r], &me.ctx);
    if (rc) return fail(rc, "rtow_ctx_create");
    if ((rc = rtow_scene_upload(me.ctx, scene))) return fail(rc, "rtow_scene_upload");
    auto hip = [&](hipError_t e, const char *what) {
      if (e == hipSuccess) return true;
      me.err = RTOW_EHIP;
      me.msg = std::string(what) + ": " + hipGetErrorString(e);
      return false;
    };
    if (!hip(hipSetDevice(device_ids[r]), "hipSetDevice")) return;
    if (!hip(hipStreamCreate(&me.stream), "hipStreamCreate")) return;
    if (!hip(hipMalloc(&me.d_local, count * sizeof(double)), "hipMalloc")) return;
    if (!hip(hipMemsetAsync(me.d_local, 0, count * sizeof(double), me.stream), "hipMemsetAsync")) return;  // ranks with fewer rows
    if (r == 0 && use_rccl && !hip(hipMalloc(&d_gather, (size_t)n_devices * count * sizeof(double)), "hipMalloc")) return;
    if (rtow_local_rows(&mine) > 0) {
      if ((rc = rtow_render_device(me.ctx, &mine, me.d_local, me.stream, &me.stats))) return fail(rc, "rtow_render_device");
    }
    if (!use_rccl) {  // every device delivers its own strips
      if (!hip(hipMemcpyAsync(staging.data() + (size_t)r * count, me.d_local, count * sizeof(double), hipMemcpyDeviceToHost, me.stream), "hipMemcpyAsync"))
        return;
      hip(hipStreamSynchronize(me.stream), "hipStreamSynchronize");
    }
  };
  {
    std::vector<std::thread> th;
    for (int r = 0; r < n_devices; ++r) th.emplace_back(work, r);
    for (auto &t : th) t.join();
  }
  int rc = RTOW_OK;
  for (int r = 0; r < n_devices && rc == RTOW_OK; ++r)
    if (ranks[(size_t)r].err) rc = rtow::set_last_error(ranks[(size_t)r].err, "rank %d (device %d): %s", r, device_ids[r], ranks[(size_t)r].msg.c_str());

  if (rc == RTOW_OK && use_rccl) {
    // the one collective: every rank's strip buffer to the first device.  One thread per device calls
    // ncclGather on its own communicator and stream (the calls of a collective may come from different
    // threads; each blocks only its own stream).
    std::vector<ncclResult_t> res((size_t)n_devices, ncclSuccess);
    std::vector<std::thread> th;
    for (int r = 0; r < n_devices; ++r)
      th.emplace_back([&, r] {
        (void)hipSetDevice(device_ids[r]);
        res[(size_t)r] = rccl.Gather(ranks[(size_t)r].d_local, r == 0 ? d_gather : nullptr, count, ncclDouble, 0, comms[(size_t)r],
                                     ranks[(size_t)r].stream);
        if (res[(size_t)r] == ncclSuccess && hipStreamSynchronize(ranks[(size_t)r].stream) != hipSuccess) res[(size_t)r] = ncclUnhandledCudaError;
      });
    for (auto &t : th) t.join();
    for (int r = 0; r < n_devices && rc == RTOW_OK; ++r)
      if (res[(size_t)r] != ncclSuccess) rc = rtow::set_last_error(RTOW_EHIP, "ncclGather on rank %d: %s", r, rccl.GetErrorString(res[(size_t)r]));
    if (rc == RTOW_OK) {  // the one device-to-host copy
      (void)hipSetDevice(device_ids[0]);
      const hipError_t e = hipMemcpy(staging.data(), d_gather, (size_t)n_devices * count * sizeof(double), hipMemcpyDeviceToHost);
      if (e != hipSuccess) rc = rtow::set_last_error(RTOW_EHIP, "D2H of the gathered strips: %s", hipGetErrorString(e));
    }
  }

  if (rc == RTOW_OK) {
    // strips back to their rows: rank r's i-th local row is global row rows_r[i]
    const size_t row_values = (size_t)W * 3;
    std::vector<int32_t> row_ids((size_t)std::max(max_rows, 1));
    for (int r = 0; r < n_devices; ++r) {
      rtow_config_t mine = *cfg;
      mine.rank = r;
      mine.nranks = n_devices;
      const int rows = rtow_local_row_list(&mine, row_ids.data(), max_rows);
      for (int i = 0; i < rows; ++i)
        std::memcpy(rgb_sums_host + (size_t)row_ids[(size_t)i] * row_values, staging.data() + (size_t)r * count + (size_t)i * row_values,
                    row_values * sizeof(double));
    }
    (void)H;
    if (stats) {
      std::memset(stats, 0, sizeof *stats);
      for (int r = 0; r < n_devices; ++r) {
        const rtow_stats_t &s = ranks[(size_t)r].stats;
        stats->samples += s.samples;
        stats->segments += s.segments;
        stats->prim_tests += s.prim_tests;
        stats->node_tests += s.node_tests;
        stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms);
        stats->total_ms = std::max(stats->total_ms, s.total_ms);
        stats->local_rows += s.local_rows;
        if (s.kernel_used) stats->kernel_used = s.kernel_used;
      }
    }
  }

  for (int r = 0; r < n_devices; ++r) {
    Rank &me = ranks[(size_t)r];
    (void)hipSetDevice(device_ids[r]);
    if (me.d_local) (void)hipFree(me.d_local);
    if (r == 0 && d_gather) (void)hipFree(d_gather);
    if (me.stream) (void)hipStreamDestroy(me.stream);
    if (use_rccl && comms[(size_t)r]) (void)rccl.CommDestroy(comms[(size_t)r]);
    rtow_ctx_destroy(me.ctx);
  }
  if (rccl.lib) dlclose(rccl.lib);
  return rc;
}
