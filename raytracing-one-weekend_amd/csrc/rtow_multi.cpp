// rtow_multi.cpp — one frame over several HIP devices from ONE process: the persistent handle
// `rtow_multi` and the one-shot rtow_render_multi on top of it.
//
// The reference fans its render out over `nthreads` workers and adds their partial images in
// launch order (src/render.cpp:169-180).  Here a worker is a device: one host thread and one
// context per device, the image cut into strips of cfg->tile_rows rows dealt round-robin (the
// partition of include/rtow.h, rank = position in `device_ids`), every device traces its strips
// into a [max_rows][W][3] buffer — f64 radiance sums (rtow_multi_render) or, with write_color run by the rank that
// owns the pixel (rtow_multi_render_rgb8; src/render.cpp:11-20,182-186), the 8-bit values of the PPM — and ONE
// ncclGather (RCCL over xGMI), enqueued on each rank's stream right behind its trace kernel, brings the buffers to
// the first device.  There a small kernel puts every strip row in its place (csrc/rtow_reduce.hip, rtow_place_rows)
// and ONE device-to-host copy delivers the finished image straight into the caller's buffer: no host pass over the
// pixels.  Any N gives the one-device image bit for bit (a pixel's value does not depend on who traces it:
// counter-based RNG, fixed sample order; every pixel has one owner, so its byte is the owner's byte).
//
// What lives in the handle, created once (rtow_multi_create) and reused by every frame: the
// contexts (with their scene images and workspaces), one stream per device, the strip buffers,
// the gather buffer on the first device, the pinned host staging buffer, the RCCL communicator
// (ncclCommInitAll takes tens of milliseconds — several frames of the cover scene) and the worker
// threads themselves.  rtow_multi_upload replaces the scene on every device (the acceleration
// structures are built once per device, concurrently); rtow_multi_render is then launches, one
// collective and one copy.
//
// RCCL is loaded on demand (dlopen of librccl.so) so that librtow.so itself does not depend on it:
// the one-process-per-GPU form (bench.py, torch.distributed) brings its own RCCL.  With
// use_rccl == 0 every device copies its strips to the host itself (no collective): the form
// `rtweekend --gpus N --gpus-same-device` uses to test the partition on a one-GPU box, where a
// communicator over duplicate devices cannot exist.
//
// Measured: only with ONE device in the communicator (the GPU boxes of this project have one);
// the N > 1 RCCL path is correct by construction, rehearsed as far as one device allows
// (tests/test_gpu_parity.py::test_rtweekend_rccl_path_with_one_rank) and checked bit for bit by
// tests/test_gpu_two_devices.py on the first box that shows two devices.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtow.h"

namespace rtow {
int set_last_error(int code, const char *fmt, ...);  // rtow_capi.cpp
int ctx_check_dropped(rtow_ctx *c);                  // rtow_capi.cpp: samples given up at the end-of-launch bound
int ctx_mirror_dropped(rtow_ctx *c, void *stream);   //   ... queue the copy of the sticky word into pinned memory
int ctx_check_dropped_mirrored(rtow_ctx *c);         //   ... read the mirror (the stream has been waited for)
int launch_place_rows(const void *gathered, void *image, uint32_t n_ranks, uint32_t max_rows, uint32_t row_bytes,
                      uint32_t height, uint32_t tile_rows, void *stream);  // rtow_reduce.hip
}

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclGather) Gather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string &why) {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      // RTLD_NODELETE: never unloaded.  librccl.so registers its own device code objects with the HIP runtime when it is
      // loaded and unregisters them from handlers that run when it is unloaded; unloading it in the middle of a process
      // that goes on using HIP (or at exit, in an order nobody controls) is a teardown this library does not need.
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
      if (lib) break;
    }
    if (!lib) {
      why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?");
      return false;
    }
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
    Gather = (decltype(Gather))dlsym(lib, "ncclGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !CommAbort || !Gather || !GetErrorString) {
      why = "librccl.so lacks ncclCommInitAll / ncclGather";
      return false;
    }
    return true;
  }
};

struct Rank {
  int device = 0;
  rtow_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  void *d_local = nullptr;   // this rank's strips: [max_rows][W][3] f64 sums or bytes
  size_t local_bytes = 0;
  hipEvent_t strips_ev = nullptr;  // everything this rank queues for a frame (trace, its side of the gather or its copy of
                                   // the strips to the first device, the mirror of its dropped-samples word) has been queued
  ncclComm_t comm = nullptr;
  int err = RTOW_OK;
  std::string msg;
  rtow_stats_t stats{};
  bool hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    err = RTOW_EHIP;
    msg = std::string(what) + ": " + hipGetErrorString(e);
    return false;
  }
  void fail_with_last(int code, const char *what) {
    err = code;
    msg = std::string(what) + ": " + rtow_last_error();
  }
};

template <class F>
int guarded(const char *what, F &&f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc &) {
    return rtow::set_last_error(RTOW_ENOMEM, "%s: out of host memory", what);
  } catch (const std::exception &e) {
    return rtow::set_last_error(RTOW_EINVAL, "%s: %s", what, e.what());
  } catch (...) {
    return rtow::set_last_error(RTOW_EINVAL, "%s: unknown C++ exception", what);
  }
}

}  // namespace

// One worker thread per device, alive as long as the handle: run(job) executes job(rank) on every
// worker and returns when all are done (a frame costs two condition-variable round trips, not N
// thread creations).
struct rtow_multi {
  int n = 0;
  bool use_rccl = false;
  bool have_scene = false;
  Rccl rccl;
  std::vector<Rank> ranks;
  void *d_gather = nullptr;  // on the first device: [rank][max_rows][W][3]
  size_t gather_bytes = 0;
  void *d_image = nullptr;   // on the first device: [H][W][3], rows in place — what the one device-to-host copy reads
  size_t image_bytes = 0;
  bool comm_dead = false;    // a collective failed half-issued and the communicators were aborted: the handle renders no more
  int fail_gather_rank = -1; // RTOW_MULTI_FAIL_GATHER (tests): this rank's gather enqueue reports a failure without enqueuing
  bool d2h_blocking = false; // RTOW_MULTI_D2H=blocking: wait for the first device's stream, then the runtime's blocking copy
  // One frame = ONE worker hand-off.  Inside it the workers meet once (`arrived`): after every rank has queued its trace
  // launch and before any queues its side of the collective, so that a rank that failed keeps all of them from enqueuing.
  std::atomic<int> arrived{0};
  std::atomic<int> failed{0};
  // where the last frame's time went (rtow_multi_frame_breakdown): host clock + events on the first device's stream
  hipEvent_t frame_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // before trace, after trace+reduce, after the gather, after placement + copy
  double last_ms[RTOW_MULTI_BREAKDOWN_FIELDS] = {};
  // workers
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void(int)> job;
  unsigned long long generation = 0;
  std::atomic<int> pending{0};  // workers still inside the current job (atomic: run() spins on it before it sleeps)
  bool quit = false;

  void worker(int r) {
    unsigned long long seen = 0;
    for (;;) {
      std::function<void(int)> mine;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
        mine = job;
      }
      try {
        mine(r);
      } catch (const std::exception &e) {
        ranks[(size_t)r].err = RTOW_EINVAL;
        ranks[(size_t)r].msg = e.what();
      } catch (...) {
        ranks[(size_t)r].err = RTOW_EINVAL;
        ranks[(size_t)r].msg = "unknown C++ exception";
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) cv_done.notify_all();
      }
    }
  }
  void run(std::function<void(int)> f) {
    std::unique_lock<std::mutex> lk(mu);
    job = std::move(f);
    pending.store(n, std::memory_order_release);
    ++generation;
    cv_go.notify_all();
    // A frame's job is a few launches per rank — tens of microseconds — so the caller looks at the counter for a short
    // while before it goes to sleep on the condition variable: being woken costs about as much as the job takes
    // (round 5: the one-device handle's hand-off 0.072 -> 0.058 ms per frame; what is left is the workers' own wake-up and their launches).  Scene uploads (milliseconds) fall through
    // to the wait.
    lk.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    while (pending.load(std::memory_order_acquire) != 0 &&
           std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(150)) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    lk.lock();
    cv_done.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
  }
  // first error of any rank -> thread-local message of the caller; clears the per-rank state
  int collect(const char *what) {
    int rc = RTOW_OK;
    for (int r = 0; r < n; ++r) {
      Rank &k = ranks[(size_t)r];
      if (k.err && rc == RTOW_OK)
        rc = rtow::set_last_error(k.err, "%s: rank %d (device %d): %s", what, r, k.device, k.msg.c_str());
      k.err = RTOW_OK;
      k.msg.clear();
    }
    return rc;
  }
};

extern "C" {

void rtow_multi_destroy(rtow_multi *m) {
  if (!m) return;
  if (!m->threads.empty()) {
    {
      std::lock_guard<std::mutex> lk(m->mu);
      m->quit = true;
    }
    m->cv_go.notify_all();
    for (auto &t : m->threads) t.join();
  }
  for (int r = 0; r < (int)m->ranks.size(); ++r) {
    Rank &k = m->ranks[(size_t)r];
    (void)hipSetDevice(k.device);
    if (k.stream) (void)hipStreamSynchronize(k.stream);
    if (k.d_local) (void)hipFree(k.d_local);
    if (r == 0 && m->d_gather) (void)hipFree(m->d_gather);
    if (r == 0 && m->d_image) (void)hipFree(m->d_image);
    if (k.strips_ev) (void)hipEventDestroy(k.strips_ev);
    if (k.comm) (void)m->rccl.CommDestroy(k.comm);
    if (k.stream) (void)hipStreamDestroy(k.stream);
    rtow_ctx_destroy(k.ctx);
  }
  if (m->rccl.lib) dlclose(m->rccl.lib);  // (drops the reference; RTLD_NODELETE keeps the library mapped)
  for (hipEvent_t &e : m->frame_ev)
    if (e) (void)hipEventDestroy(e);
  delete m;
}

int rtow_multi_create(int32_t n_devices, const int32_t *device_ids, int32_t use_rccl, rtow_multi **out) {
  if (!out) return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_create: out is NULL");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64 || !device_ids)
    return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_create: 1..64 devices");
  return guarded("rtow_multi_create", [&]() -> int {
    rtow_multi *m = new rtow_multi();
    m->n = n_devices;
    m->use_rccl = use_rccl != 0;
    m->ranks.resize((size_t)n_devices);
    for (int r = 0; r < n_devices; ++r) m->ranks[(size_t)r].device = device_ids[r];
    if (m->use_rccl) {
      std::string why;
      if (!m->rccl.load(why)) {
        const int rc = rtow::set_last_error(RTOW_ENODEV, "%s", why.c_str());
        rtow_multi_destroy(m);
        return rc;
      }
      std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
      std::vector<int> devs(device_ids, device_ids + n_devices);
      const ncclResult_t res = m->rccl.CommInitAll(comms.data(), n_devices, devs.data());
      if (res != ncclSuccess) {
        const int rc = rtow::set_last_error(RTOW_EHIP, "ncclCommInitAll over %d devices: %s", n_devices,
                                            m->rccl.GetErrorString(res));
        rtow_multi_destroy(m);
        return rc;
      }
      for (int r = 0; r < n_devices; ++r) m->ranks[(size_t)r].comm = comms[(size_t)r];
    }
    if (const char *e = std::getenv("RTOW_MULTI_FAIL_GATHER")) m->fail_gather_rank = std::atoi(e);
    if (const char *e = std::getenv("RTOW_MULTI_D2H")) m->d2h_blocking = std::strcmp(e, "blocking") == 0;
    try {
      for (int r = 0; r < n_devices; ++r) m->threads.emplace_back([m, r] { m->worker(r); });
    } catch (...) {  // a thread could not be started: the ones that run are joined by destroy
      rtow_multi_destroy(m);
      throw;
    }
    m->run([m](int r) {
      Rank &me = m->ranks[(size_t)r];
      const int rc = rtow_ctx_create(me.device, &me.ctx);
      if (rc) return me.fail_with_last(rc, "rtow_ctx_create");
      if (!me.hip(hipSetDevice(me.device), "hipSetDevice")) return;
      if (!me.hip(hipStreamCreate(&me.stream), "hipStreamCreate")) return;
      me.hip(hipEventCreateWithFlags(&me.strips_ev, hipEventDisableTiming), "hipEventCreate");
    });
    const int rc = m->collect("rtow_multi_create");
    if (rc) {
      const std::string keep = rtow_last_error();
      rtow_multi_destroy(m);
      return rtow::set_last_error(rc, "%s", keep.c_str());
    }
    *out = m;
    return RTOW_OK;
  });
}

int rtow_multi_set_builder(rtow_multi *m, int32_t builder) {
  if (!m) return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_set_builder: handle is NULL");
  for (Rank &k : m->ranks) {
    const int rc = rtow_ctx_set_builder(k.ctx, builder);
    if (rc) return rc;
  }
  return RTOW_OK;
}

int rtow_multi_upload(rtow_multi *m, const rtow_scene_t *scene) {
  if (!m || !scene) return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_upload: NULL argument");
  return guarded("rtow_multi_upload", [&]() -> int {
    m->have_scene = false;
    m->run([m, scene](int r) {
      Rank &me = m->ranks[(size_t)r];
      const int rc = rtow_scene_upload(me.ctx, scene);
      if (rc) me.fail_with_last(rc, "rtow_scene_upload");
    });
    const int rc = m->collect("rtow_multi_upload");
    m->have_scene = rc == RTOW_OK;
    return rc;
  });
}

int rtow_multi_build_info(rtow_multi *m, rtow_build_info_t *out) {
  if (!m || !out) return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_build_info: NULL argument");
  return rtow_build_info(m->ranks[0].ctx, out);  // every device builds the same structures
}

// One frame.  rgb8 == false: W*H*3 f64 radiance sums; true: W*H*3 bytes, write_color by the owning rank.
static int multi_render(rtow_multi *m, const rtow_config_t *cfg, void *host_out, rtow_stats_t *stats, bool rgb8) {
  const char *what = rgb8 ? "rtow_multi_render_rgb8" : "rtow_multi_render";
  if (!m || !cfg || !host_out) return rtow::set_last_error(RTOW_EINVAL, "%s: NULL argument", what);
  if (!m->have_scene) return rtow::set_last_error(RTOW_ENOSCENE, "%s: no scene uploaded", what);
  if (cfg->accumulate) return rtow::set_last_error(RTOW_EINVAL, "%s: accumulate is not supported", what);
  if (m->comm_dead)
    return rtow::set_last_error(RTOW_EHIP, "%s: the communicators of this handle were aborted after a failed collective; "
                                           "create a new handle", what);
  return guarded(what, [&]() -> int {
    const int n = m->n;
    rtow_config_t probe = *cfg;
    probe.rank = 0;
    probe.nranks = n;
    const int max_rows = rtow_local_rows(&probe);  // rank 0 owns the first strip: nobody has more rows
    if (max_rows < 0) return max_rows;
    const int W = cfg->image_width, H = cfg->image_height;
    const size_t es = rgb8 ? 1 : sizeof(double);
    const size_t count = (size_t)max_rows * (size_t)W * 3;  // values per rank in the gather
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (count == 0) return RTOW_OK;
    const size_t bytes = count * es, row_bytes = (size_t)W * 3 * es, image_bytes = (size_t)H * row_bytes;
    // buffers of the handle on the first device, grown on demand (a frame of the same shape allocates nothing)
    {
      Rank &r0 = m->ranks[0];
      hipError_t e = hipSetDevice(r0.device);
      if (e == hipSuccess && m->gather_bytes < (size_t)n * bytes) {
        if (m->d_gather) (void)hipFree(m->d_gather);
        m->d_gather = nullptr;
        m->gather_bytes = 0;
        if ((e = hipMalloc(&m->d_gather, (size_t)n * bytes)) == hipSuccess) m->gather_bytes = (size_t)n * bytes;
      }
      if (e == hipSuccess && m->image_bytes < image_bytes) {
        if (m->d_image) (void)hipFree(m->d_image);
        m->d_image = nullptr;
        m->image_bytes = 0;
        if ((e = hipMalloc(&m->d_image, image_bytes)) == hipSuccess) m->image_bytes = image_bytes;
      }
      if (e != hipSuccess) return rtow::set_last_error(RTOW_EHIP, "%s: buffers on device %d: %s", what, r0.device, hipGetErrorString(e));
    }
    const bool rccl = m->use_rccl;
    const bool want_stats = stats != nullptr;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    const clk::time_point t0 = clk::now();
    for (hipEvent_t &e : m->frame_ev)
      if (!e && hipEventCreate(&e) != hipSuccess) return rtow::set_last_error(RTOW_EHIP, "%s: hipEventCreate", what);
    m->arrived.store(0, std::memory_order_relaxed);
    m->failed.store(0, std::memory_order_relaxed);
    // ONE hand-off to the workers per frame.  Each worker: (1) queue this rank's trace (+ reduce / write_color) on its
    // stream; (2) meet the others; (3) unless some rank failed, queue its side of the one collective (or, without RCCL,
    // its copy of the strips to the first device), the mirror of its dropped-samples word, and an event behind all of it.
    m->run([m, cfg, n, count, bytes, rccl, want_stats, rgb8](int r) {
      Rank &me = m->ranks[(size_t)r];
      auto phase1 = [&]() {
        if (!me.hip(hipSetDevice(me.device), "hipSetDevice")) return;
        if (me.local_bytes < bytes) {
          if (me.d_local) (void)hipFree(me.d_local);
          me.d_local = nullptr;
          me.local_bytes = 0;
          if (!me.hip(hipMalloc(&me.d_local, bytes), "hipMalloc")) return;
          me.local_bytes = bytes;
        }
        rtow_config_t mine = *cfg;
        mine.rank = r;
        mine.nranks = n;
        const int rows = rtow_local_rows(&mine);
        std::memset(&me.stats, 0, sizeof me.stats);
        if (r == 0 && !me.hip(hipEventRecord(m->frame_ev[0], me.stream), "hipEventRecord")) return;
        // ranks with fewer rows than rank 0: the rest of their strip buffer is defined (zero) for the gather
        if ((size_t)rows * (size_t)cfg->image_width * 3 < count &&
            !me.hip(hipMemsetAsync(me.d_local, 0, bytes, me.stream), "hipMemsetAsync"))
          return;
        if (rows > 0) {
          const int rc = rgb8 ? rtow_render_device_rgb8(me.ctx, &mine, me.d_local, me.stream, want_stats ? &me.stats : nullptr)
                              : rtow_render_device(me.ctx, &mine, me.d_local, me.stream, want_stats ? &me.stats : nullptr);
          if (rc) return me.fail_with_last(rc, rgb8 ? "rtow_render_device_rgb8" : "rtow_render_device");
        }
        if (r == 0) me.hip(hipEventRecord(m->frame_ev[1], me.stream), "hipEventRecord");
      };
      try {
        phase1();
      } catch (const std::exception &e) {  // every worker must reach the meeting point
        me.err = RTOW_EINVAL;
        me.msg = e.what();
      } catch (...) {
        me.err = RTOW_EINVAL;
        me.msg = "unknown C++ exception";
      }
      if (me.err) m->failed.fetch_add(1, std::memory_order_relaxed);
      m->arrived.fetch_add(1, std::memory_order_acq_rel);
      while (m->arrived.load(std::memory_order_acquire) < n) std::this_thread::yield();  // all n workers run this job
      if (m->failed.load(std::memory_order_relaxed) != 0) return;  // nobody enqueues a collective some rank cannot join
      if (rccl) {
        // the one collective, stream-ordered behind every rank's kernels: each rank's strip buffer to the first device.
        // One thread per device calls ncclGather on its own communicator and stream.
        const ncclResult_t res = r == m->fail_gather_rank
                                     ? ncclInternalError  // (test hook: this rank's side of the collective is never enqueued)
                                     : m->rccl.Gather(me.d_local, r == 0 ? m->d_gather : nullptr, count, rgb8 ? ncclUint8 : ncclDouble,
                                                      0, me.comm, me.stream);
        if (res != ncclSuccess) {
          me.err = RTOW_EHIP;
          me.msg = std::string("ncclGather: ") + m->rccl.GetErrorString(res);
          m->failed.fetch_add(1, std::memory_order_relaxed);
          return;
        }
      } else {
        // no collective (ranks on one device, where a communicator cannot exist): every rank copies its strips into
        // its slot of the first device's gather buffer
        unsigned char *slot = (unsigned char *)m->d_gather + (size_t)r * bytes;
        const int dev0 = m->ranks[0].device;
        if (!me.hip(me.device == dev0 ? hipMemcpyAsync(slot, me.d_local, bytes, hipMemcpyDeviceToDevice, me.stream)
                                      : hipMemcpyPeerAsync(slot, dev0, me.d_local, me.device, bytes, me.stream),
                    "copy of the strips to the first device"))
          return;
      }
      if (r == 0 && !me.hip(hipEventRecord(m->frame_ev[2], me.stream), "hipEventRecord")) return;
      const int drc = rtow::ctx_mirror_dropped(me.ctx, me.stream);
      if (drc) return me.fail_with_last(drc, "mirror of the dropped-samples word");
      if (r != 0) me.hip(hipEventRecord(me.strips_ev, me.stream), "hipEventRecord");
    });
    const double t_handoff = ms_since(t0);
    const bool gather_failed = rccl && m->failed.load(std::memory_order_relaxed) != 0;
    bool gather_half_issued = false;
    if (gather_failed)  // did any rank enqueue its side?  (phase-1 failures leave every rank before the collective)
      for (int r = 0; r < n; ++r) gather_half_issued |= m->ranks[(size_t)r].msg.rfind("ncclGather", 0) == 0;
    int rc = m->collect(what);
    if (rc != RTOW_OK && gather_half_issued) {
      // A collective that one rank could not enqueue never completes on the ranks that did: waiting for their
      // streams would hang.  Abort every communicator (in-flight kernels exit), do not wait, and refuse further
      // frames on this handle: the error is an error code, never a wait.
      const std::string keep = rtow_last_error();
      for (Rank &k : m->ranks) {
        if (k.comm) {
          (void)hipSetDevice(k.device);
          (void)m->rccl.CommAbort(k.comm);
          k.comm = nullptr;
        }
      }
      m->comm_dead = true;
      return rtow::set_last_error(rc, "%s (gather; communicators aborted)", keep.c_str());
    }
    Rank &r0 = m->ranks[0];
    const void *d_frame = m->d_gather;  // one rank: its strips ARE the image, rows in place
    if (rc == RTOW_OK) {
      // on the first device, behind the strips' arrival (and behind everything the other ranks queued): rows into
      // place, then the ONE device-to-host copy, straight into the caller's buffer
      hipError_t e = hipSetDevice(r0.device);
      for (int r = 1; r < n && e == hipSuccess; ++r) e = hipStreamWaitEvent(r0.stream, m->ranks[(size_t)r].strips_ev, 0);
      if (e == hipSuccess && n > 1) {
        const int lrc = rtow::launch_place_rows(m->d_gather, m->d_image, (uint32_t)n, (uint32_t)max_rows, (uint32_t)row_bytes,
                                                (uint32_t)H, (uint32_t)cfg->tile_rows, r0.stream);
        if (lrc != 0) e = (hipError_t)lrc;
        d_frame = m->d_image;
      }
      if (e != hipSuccess) rc = rtow::set_last_error(RTOW_EHIP, "%s: placement on device %d: %s", what, r0.device, hipGetErrorString(e));
    }
    if (rc != RTOW_OK) {
      // a failed frame still leaves every stream idle before it returns
      const std::string keep = rtow_last_error();
      for (int r = 0; r < n; ++r) {
        (void)hipSetDevice(m->ranks[(size_t)r].device);
        (void)hipStreamSynchronize(m->ranks[(size_t)r].stream);
      }
      return rtow::set_last_error(rc, "%s", keep.c_str());
    }
    // The first device's stream is the only one to wait for: it is ordered behind every other rank's event.  The copy
    // goes straight into the caller's (pageable) buffer.
    const clk::time_point t1 = clk::now();
    double t_wait = 0.0;
    hipError_t e = hipSuccess;
    if (m->d2h_blocking) {
      e = hipStreamSynchronize(r0.stream);
      t_wait = ms_since(t1);
      if (e == hipSuccess) e = hipMemcpy(host_out, d_frame, image_bytes, hipMemcpyDeviceToHost);
      if (e == hipSuccess) e = hipEventRecord(m->frame_ev[3], r0.stream);
    } else {
      e = hipMemcpyAsync(host_out, d_frame, image_bytes, hipMemcpyDeviceToHost, r0.stream);
      if (e == hipSuccess) e = hipEventRecord(m->frame_ev[3], r0.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(r0.stream);
    if (e != hipSuccess)
      return rtow::set_last_error(RTOW_EHIP, "%s: copy of the frame from device %d: %s", what, r0.device, hipGetErrorString(e));
    const double t_done = ms_since(t0);
    for (int r = 0; r < n; ++r)  // never RTOW_OK with samples dropped (the end-of-launch bound of the trace kernel)
      if ((rc = rtow::ctx_check_dropped_mirrored(m->ranks[(size_t)r].ctx))) return rc;
    {
      double *b = m->last_ms;
      float f = 0.f;
      b[RTOW_MB_HANDOFF_ENQUEUE] = t_handoff;
      b[RTOW_MB_PLACE_ENQUEUE] = std::chrono::duration<double, std::milli>(t1 - t0).count() - t_handoff;
      b[RTOW_MB_WAIT_AND_COPY] = t_done - std::chrono::duration<double, std::milli>(t1 - t0).count();
      b[RTOW_MB_WAIT_ONLY] = t_wait;
      b[RTOW_MB_DEV_TRACE] = hipEventElapsedTime(&f, m->frame_ev[0], m->frame_ev[1]) == hipSuccess ? f : -1.0;
      b[RTOW_MB_DEV_GATHER] = hipEventElapsedTime(&f, m->frame_ev[1], m->frame_ev[2]) == hipSuccess ? f : -1.0;
      b[RTOW_MB_DEV_PLACE_COPY] = hipEventElapsedTime(&f, m->frame_ev[2], m->frame_ev[3]) == hipSuccess ? f : -1.0;
      b[RTOW_MB_TOTAL] = ms_since(t0);
    }
    if (stats) {
      for (int r = 0; r < n; ++r) {
        const rtow_stats_t &s = m->ranks[(size_t)r].stats;
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
    return RTOW_OK;
  });
}

int rtow_multi_render(rtow_multi *m, const rtow_config_t *cfg, double *rgb_sums_host, rtow_stats_t *stats) {
  return multi_render(m, cfg, rgb_sums_host, stats, false);
}

int rtow_multi_render_rgb8(rtow_multi *m, const rtow_config_t *cfg, unsigned char *rgb8_host, rtow_stats_t *stats) {
  if (cfg && cfg->nstreams > 0 && cfg->samples_per_pixel / cfg->nstreams <= 0)
    return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_render_rgb8: no effective samples (samples_per_pixel < nstreams)");
  return multi_render(m, cfg, rgb8_host, stats, true);
}

int rtow_multi_frame_breakdown(rtow_multi *m, double *out, int32_t n) {
  if (!m || !out || n < 0) return rtow::set_last_error(RTOW_EINVAL, "rtow_multi_frame_breakdown: NULL argument");
  for (int i = 0; i < n && i < RTOW_MULTI_BREAKDOWN_FIELDS; ++i) out[i] = m->last_ms[i];
  return n < RTOW_MULTI_BREAKDOWN_FIELDS ? n : RTOW_MULTI_BREAKDOWN_FIELDS;
}

// The one-shot form: everything above for a single frame (set-up dominates it: use the handle for more than one).
int rtow_render_multi(int32_t n_devices, const int32_t *device_ids, const rtow_scene_t *scene,
                      const rtow_config_t *cfg, double *rgb_sums_host, rtow_stats_t *stats, int32_t use_rccl) {
  if (n_devices < 1 || n_devices > 64 || !device_ids) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: 1..64 devices");
  if (!scene || !cfg || !rgb_sums_host) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: NULL argument");
  if (cfg->accumulate) return rtow::set_last_error(RTOW_EINVAL, "rtow_render_multi: accumulate is not supported");
  rtow_multi *m = nullptr;
  int rc = rtow_multi_create(n_devices, device_ids, use_rccl, &m);
  if (rc == RTOW_OK) rc = rtow_multi_upload(m, scene);
  if (rc == RTOW_OK) rc = rtow_multi_render(m, cfg, rgb_sums_host, stats);
  if (rc != RTOW_OK) {
    const std::string keep = rtow_last_error();
    rtow_multi_destroy(m);
    return rtow::set_last_error(rc, "%s", keep.c_str());
  }
  rtow_multi_destroy(m);
  return RTOW_OK;
}

}  // extern "C"
