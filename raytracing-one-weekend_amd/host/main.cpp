// main.cpp — `rtweekend` command line, flag-compatible with the reference's
// src/main.cpp:138-170 (CLI11 is not in this image: a small parser of the same
// short/long flags), plus the device knobs the reference has no notion of.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>

#include "render.h"

namespace rt = rtweekend;

// A fatal signal names its frames.  Round 4 saw ONE `rtweekend` run die with SIGSEGV after the complete image and
// "Done in" had been written (gpurun_out/r04j_tests.log) and had nothing to say about where: no core, no backtrace.
// The handler writes the phase the program was in and the raw backtrace (module + offset per frame: enough for
// addr2line / llvm-symbolizer) to stderr, then re-raises with the default action, so the exit status stays the
// signal's.  Only async-signal-safe calls after installation (backtrace() is called once up front so that its lazy
// libgcc load does not happen inside the handler).
static volatile sig_atomic_t g_phase = 0;  // 0 flags, 1 scene + render, 2 main returning, 3 the program's own at-exit handlers done
static const char *const kPhaseText[] = {
    "parsing flags", "building the scene / rendering", "after render(): returning from main",
    "inside exit(): this program's handlers are done; at-exit handlers of the libraries loaded before main are running"};

static void on_fatal_signal(int sig) {
  auto put = [](const char *s) { (void)!write(2, s, std::strlen(s)); };
  put("rtweekend: fatal signal ");
  char num[4] = {(char)('0' + sig / 10 % 10), (char)('0' + sig % 10), '\n', 0};
  put(num);
  put("rtweekend: phase: ");
  put(kPhaseText[g_phase < 0 || g_phase > 3 ? 0 : g_phase]);
  put("\nrtweekend: backtrace (module(+offset)):\n");
  void *frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

static void install_fatal_signal_handler() {
  void *warm[2];
  (void)backtrace(warm, 2);
  struct sigaction sa;
  std::memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fatal_signal;
  sa.sa_flags = SA_RESETHAND | SA_NODEFER;
  sigemptyset(&sa.sa_mask);
  for (int sig : {SIGSEGV, SIGBUS, SIGFPE, SIGILL, SIGABRT}) sigaction(sig, &sa, nullptr);
  // registered after every handler the libraries registered while loading, before any registered while rendering:
  // it runs when the latter are done
  std::atexit([] { g_phase = 3; });
}

static void usage(const char *argv0) {
  std::cout << "Raytracing one weekend/week/restoflife (MI355X HIP path)\n"
               "Usage: "
            << argv0
            << " [OPTIONS]\n\n"
               "  -h,--help                   Print this help message and exit\n"
               "  -t,--threads INT            Number of sample streams (the reference's threads)\n"
               "  -w,--image-width INT        Image width\n"
               "  -s,--samples-per-pixel INT  Samples per pixel\n"
               "  -c,--max-child-rays INT     Max child rays\n"
               "  -a,--aspect-ratio FLOAT     Aspect ratio\n"
               "  -n,--balls_sqrt INT         Number of balls sqrt\n"
               "  -m,--moving-spheres         Moving spheres\n"
               "  -q,--quick                  Quickie\n"
               "  --dry-run                   Dry run\n"
               "  -l,--load TEXT              Model to load\n"
               "  --device INT                HIP device index (default 0)\n"
               "  --seed UINT                 render seed of the counter-based RNG (default 1)\n"
               "  --precision strict|fast|f32 f64 without / with FMA contraction (default fast); f32 = binary32 preview\n"
               "  --kernel auto|brute|bvh|grid|bvh4|reftree  closest-hit strategy (default auto; reftree: the\n"
               "                              reference's own median-split tree and f64 box test, strict arithmetic)\n"
               "  --primitives oo|variant|world  scene model the scripts build (the reference picks at compile time;\n"
               "                              world = src/vmodel.h, spheres only)\n"
               "  --builder host|device|auto  BVH build: binned SAH on the host, the same on the GPU, or (default) whichever delivers the frame sooner\n"
               "  --p6                        binary P6 output, write_color on the device\n"
               "  --general-obj               with -l: load every shape, not only the first\n"
               "  --gpus INT                  tile-split over INT devices (--device is the first), strips gathered\n"
               "                              to the first device with one ncclGather (RCCL)\n"
               "  --rccl                      take the RCCL gather path even with one device\n";
}

int main(int argc, char *argv[]) {
  install_fatal_signal_handler();
  rt::Config cfg{};
  bool dry_run = false;
  rt::DeviceOptions &opt = rt::device_options();

  auto is = [](const char *a, const char *s, const char *l) {
    return std::strcmp(a, s) == 0 || std::strcmp(a, l) == 0;
  };
  try {
    for (int i = 1; i < argc; ++i) {
      const char *a = argv[i];
      auto value = [&]() -> const char * {
        if (i + 1 >= argc) throw std::runtime_error(std::string(a) + ": 1 required TEXT missing");
        return argv[++i];
      };
      if (is(a, "-h", "--help")) {
        usage(argv[0]);
        return 0;
      } else if (is(a, "-t", "--threads")) {
        cfg.nthreads = std::stoi(value());
      } else if (is(a, "-w", "--image-width")) {
        cfg.image_width = std::stoi(value());
      } else if (is(a, "-s", "--samples-per-pixel")) {
        cfg.samples_per_pixel = std::stoi(value());
      } else if (is(a, "-c", "--max-child-rays")) {
        cfg.max_child_rays = std::stoi(value());
      } else if (is(a, "-a", "--aspect-ratio")) {
        cfg.aspect_ratio = std::stod(value());
      } else if (is(a, "-n", "--balls_sqrt")) {
        cfg.number_of_balls_sqrt = std::stoi(value());
      } else if (is(a, "-m", "--moving-spheres")) {
        cfg.moving_spheres = true;
      } else if (is(a, "-q", "--quick")) {
        // parsed and ignored, like the reference (src/main.cpp:142,157)
      } else if (std::strcmp(a, "--dry-run") == 0) {
        dry_run = true;
      } else if (is(a, "-l", "--load")) {
        cfg.model = std::string(value());
      } else if (std::strcmp(a, "--device") == 0) {
        opt.device = std::stoi(value());
      } else if (std::strcmp(a, "--seed") == 0) {
        opt.seed = std::stoull(value());
      } else if (std::strcmp(a, "--precision") == 0) {
        const std::string v = value();
        if (v == "strict") opt.precision = 0;
        else if (v == "fast") opt.precision = 1;
        else if (v == "f32") opt.precision = 2;
        else throw std::runtime_error("--precision: strict|fast|f32");
      } else if (std::strcmp(a, "--p6") == 0) {
        opt.binary_ppm = true;
      } else if (std::strcmp(a, "--general-obj") == 0) {
        opt.general_obj = true;
      } else if (std::strcmp(a, "--gpus") == 0) {
        opt.gpus = std::stoi(value());
        if (opt.gpus < 1 || opt.gpus > 64) throw std::runtime_error("--gpus: 1..64");
      } else if (std::strcmp(a, "--gpus-same-device") == 0) {  // test hook: every rank on --device
        opt.gpus_same_device = true;
      } else if (std::strcmp(a, "--rccl") == 0) {  // gather through RCCL even with one device
        opt.rccl = true;
      } else if (std::strcmp(a, "--builder") == 0) {
        const std::string v = value();
        if (v == "host") opt.builder = 0;
        else if (v == "device") opt.builder = 1;
        else if (v == "auto") opt.builder = 2;
        else throw std::runtime_error("--builder: host|device|auto");
      } else if (std::strcmp(a, "--kernel") == 0) {
        const std::string v = value();
        if (v == "auto") opt.kernel = 0;
        else if (v == "brute") opt.kernel = 1;
        else if (v == "bvh") opt.kernel = 2;
        else if (v == "grid") opt.kernel = 3;
        else if (v == "bvh4") opt.kernel = 4;
        else if (v == "reftree") opt.kernel = 5;
        else throw std::runtime_error("--kernel: auto|brute|bvh|grid|bvh4|reftree");
      } else if (std::strcmp(a, "--primitives") == 0) {
        const std::string v = value();
        if (v == "oo") opt.primitives_model = 0;
        else if (v == "variant") opt.primitives_model = 1;
        else if (v == "world") opt.primitives_model = 2;
        else throw std::runtime_error("--primitives: oo|variant|world");
      } else {
        throw std::runtime_error(std::string("The following argument was not expected: ") + a);
      }
    }
  } catch (const std::exception &e) {
    std::cerr << e.what() << "\nRun with --help for more information.\n";
    return 109;  // CLI11's exit code for parse errors of this kind
  }

  if (dry_run) {
    std::cout << cfg;
    return 0;
  }
  if (const char *e = std::getenv("RTOW_TEST_RAISE"))  // test hook: the handler above, without a GPU
    raise(std::atoi(e));
  g_phase = 1;
  try {
    if (cfg.model) {
      if (opt.primitives_model == 2) throw std::runtime_error("--primitives world: src/vmodel.h's World holds spheres only");
      if (opt.primitives_model == 1)
        rt::render(rt::detail::foo_variant(cfg), cfg);
      else
        rt::render(rt::detail::foo(cfg), cfg);
    } else if (opt.primitives_model == 2) {
      const rt::WorldScene ws = rt::detail::lots_of_balls_world(cfg);
      rt::render(ws.world, ws.cam, cfg);
    } else if (opt.primitives_model == 1) {
      rt::render(rt::detail::lots_of_balls_variant(cfg), cfg);
    } else {
      rt::render(rt::detail::lots_of_balls(cfg), cfg);
    }
  } catch (const std::exception &e) {  // (the reference lets these escape: terminate)
    std::cerr << "rtweekend: " << e.what() << "\n";
    return 1;
  }
  // Like the reference's main (src/main.cpp:165-170): return.  Every context is destroyed (rtow_ctx_destroy waits for
  // the device before and after it releases its memory), nothing of this program or of librtow.so calls HIP from a
  // destructor; what still runs in exit() is listed in INTEGRATION.md §3.
  g_phase = 2;
  return 0;
}
