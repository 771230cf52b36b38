#!/usr/bin/env python3
"""bench.py — Msamples/s of the path-tracing hot path on the RTOW cover scene.

A "step" is one full render of the workload: every rank traces its strips of the
image with the HIP kernels (scene already resident in HBM), and for N > 1 the
framebuffer strips are gathered to rank 0 with ONE torch.distributed gather
(backend nccl = RCCL over xGMI) and reassembled on the device.

Workload (BASELINE.json configs):
  N = 1 : configs[1]  cover scene (486 spheres), 1200x800, 100 spp, 50 bounces
  N > 1 : configs[2]  same scene, 500 spp, tile-split over N GPUs + RCCL gather
Both use 10 samples per work item (nstreams = spp / 10), f64 arithmetic (the
reference is all-fp64), the fast (FMA-contracted) kernel build and seed 1.
`--workload moving|suzanne|mesh100k` times another BASELINE config with the same code
(configs[3] = suzanne 1920x1080x256 spp, configs[4] = 96,800-triangle mesh 1920x1080x1024 spp);
the default N = 1 run also appends them, shortened to a few steps, as `other_configs`.

Launch: `python bench.py --gpus N` starts N ranks itself when it was not started by
torchrun (WORLD_SIZE unset): the parent spawns N fresh worker processes BEFORE anything
touches the GPU, relays rank 0's JSON line and exits non-zero if fewer than N devices are
visible.  Under `python -m torch.distributed.run … bench.py --gpus N` the ranks are taken
from the environment.  `--backend gloo` rehearses the N > 1 path on a one-GPU box (every
rank on cuda:0, collectives through host memory; at most 6 ranks).

stdout of rank 0 carries ONE compact JSON line (under 4 KB: the contract's keys, `roofline`, `cpu_baseline` and a
handful of scalars — raytracing-one-weekend_amd/benchline.py).  The full record of the run goes to `--details-out`
(default gpurun_out/bench_details.json) and to stderr, with
  roofline      — the bound that BINDS: "valu_issue".  `frac` = issue utilisation x lane activity of the
                  trace kernel (share of the chip's vector lane-slots that carried an active lane's
                  instruction), from the committed PMC passes of this very kernel source (null when the
                  committed profile belongs to another build); `traffic` = HBM bytes per launch (PMC);
                  `kernel_ms` = the kernel's mean duration, HIP events on the launch stream inside the
                  timed region.  `hbm_equivalent_streaming` keeps the metric's own figure, SURVEY.md §8d's
                  EQUIVALENT STREAMING bandwidth (the bytes a brute-force kernel would stream —
                  ceil(segments/64) * N_prim * record bytes + framebuffer — over the kernel's duration,
                  against the 8 TB/s HBM peak): nominal for a kernel that culls and reads LDS (it passes 1
                  on the meshes), which is why it is not what `roofline.frac` says.
  roofline_valu — the bound that binds: VALU issue.  `achieved` = the f64/f32 flops of the
                  walk's own counted node and primitive tests per second (live), against the
                  78.6 TFLOP/s f64 vector peak; `issue_utilisation` and `lane_activity` come
                  from the committed PMC file named in `pmc_file` (same kernel, same workload).
  end_to_end    — SURVEY.md §8d's region for one call of the host-buffer entry point
                  (rtow_render: scene upload incl. acceleration build, kernels, D2H of f64 sums).
  cpu_baseline  — oracle/ (the CPU restatement of the reference's sample loop, same
                  Philox stream) timed on this host's cores on a bounded sample of
                  the same workload (same scene and resolution, fewer spp).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import benchline  # noqa: E402
import rtow  # noqa: E402
import tiles  # noqa: E402

SEED = 1
# Samples per work item (= spp / nstreams).  Short items shorten the end-of-launch tail, long items
# save partial-sum traffic and item bookkeeping.  With the tail measures of the end of round 1
# (queue ending on cheap rows, sample donation, no polling of the empty queue) the optimum moved
# from 4 to 10: C2 (100 spp) 2 -> 8.44, 4 -> 9.38, 5 -> 9.46, 10 -> 9.54, 20 -> 8.96 Gsamples/s;
# 500 spp on one GPU 4 -> 9.66, 10 -> 10.11, 20 -> 10.19 (scripts/spi_sweep_500.sh).
SAMPLES_PER_ITEM = int(os.environ.get("RTOW_BENCH_SPI", "10"))
MESH_SAMPLES_PER_ITEM = int(os.environ.get("RTOW_BENCH_MESH_SPI", "16"))  # 8 / 16 / 32: 3.99 / 4.02 / 3.99 (C4), 1.966 / 1.974 / 1.964 (C5)
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_VALU_PEAK_TF = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
N_SIMD = 1024
# SURVEY.md §8d record sizes (f32 figure doubled for the f64 records this path computes on)
BYTES_PER_PRIM = {"sphere": 32, "moving": 56, "triangle": 72}
FLOPS_SPHERE_TEST = 23   # to the discriminant reject (src/common-model.cpp:70-75)
FLOPS_TRI_TEST = 59      # src/common-model.cpp:106-115
MAX_REHEARSAL_RANKS = 6  # processes that may share one GPU on the pool's boxes

WORKLOADS = {
    # name: (scene, width, aspect, spp, bounces, samples per item, BASELINE config)
    "cover": ("cover", 1200, 1.5, 100, 50, SAMPLES_PER_ITEM, "configs[1]"),
    "moving": ("moving", 1200, 1.5, 100, 50, SAMPLES_PER_ITEM, "configs[1] with moving spheres (the reference's default)"),
    "suzanne": ("suzanne", 1920, 16 / 9, 256, 20, MESH_SAMPLES_PER_ITEM, "configs[3]"),
    "mesh100k": ("mesh100k", 1920, 16 / 9, 1024, 20, MESH_SAMPLES_PER_ITEM, "configs[4]"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--width", type=int, default=0, help="override the image width (a custom config: profiling of the "
                                                         "streaming stress leg, --workload mesh100k --kernel brute --spp 1 --width 960)")
    ap.add_argument("--workload", choices=list(WORKLOADS), default="cover")
    ap.add_argument("--no-scaling-base", action="store_true",
                    help="N=1: skip the extra configs[2] (500 spp) measurement on this GPU")
    ap.add_argument("--no-scale-projection", action="store_true",
                    help="N=1: skip timing every rank's share of configs[2] at N = 2/4/8 on this GPU")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N=1: skip the shortened runs of the other BASELINE configs")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-reference-boundary", action="store_true",
                    help="N=1: skip the lines for nstreams = 4 (the reference's default thread count)")
    ap.add_argument("--precision", choices=["fast", "strict", "f32"], default="fast",
                    help="fast/strict: binary64 (the metric's arithmetic); f32: the preview build, never the headline")
    ap.add_argument("--kernel", choices=["auto", "brute", "bvh", "grid", "bvh4", "reftree"], default="auto",
                    help="reftree: the reference's own tree and box test (needs --precision strict; an exactness mode)")
    ap.add_argument("--moving", action="store_true", help="same as --workload moving")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", choices=["tiles", "samples"], default="tiles",
                    help="N>1 decomposition: strips of rows + one gather (default, bit-identical for any N) "
                         "or the reference's own: full frame per rank with spp/N of the samples, frames summed")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 path with every rank on cuda:0 (1-GPU box)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dump-image", default="", help="rank 0: write the gathered f64 sums of the last step (.npy)")
    ap.add_argument("--details-out", default="",
                    help="rank 0: where the full record of the run goes (per-rank tables, per-leg rooflines, builders, "
                         "multi-device handle); default gpurun_out/bench_details.json, the temp directory if that "
                         "cannot be written.  stdout carries only the compact line (benchline.py)")
    return ap.parse_args()


# ------------------------------------------------------------------ launching N ranks ---
def launch_ranks(a) -> int:
    """Parent of an N-rank run that was not started by torchrun.  Nothing here initialises the
    GPU (device_count() does not, on this image): the workers are fresh processes."""
    rehearsal = a.backend == "gloo"
    n_dev = torch.cuda.device_count()
    need = 1 if rehearsal else a.gpus
    if n_dev < need:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} needs {need} HIP device(s), {n_dev} visible "
                         f"(no CPU fallback; `--backend gloo` rehearses N ranks on one GPU)\n")
        return 2
    if rehearsal and a.gpus > MAX_REHEARSAL_RANKS:
        sys.stderr.write(f"bench.py: a gloo rehearsal puts every rank on cuda:0; at most {MAX_REHEARSAL_RANKS} ranks\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll ALL ranks: if any dies at start-up (device not visible, import error) the others would sit in
    # init_process_group / barrier until torch's timeout; end them and fail now instead
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = []
    while True:
        rcs = [p.poll() for p in procs]
        failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed or all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=20)
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write(f"bench.py: ranks failed: {failed}\n")
        return 1
    return 0


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(scene, label, W, height, depth, seconds):
    """Time the oracle (kind 'port') on this host: same scene/resolution, reduced spp."""
    sys.path.insert(0, str(ROOT / "tests"))
    import orc

    cores = host_cores()
    cal = rtow.make_config(W, height, 2, 1, depth, seed=SEED)
    orc.render(scene, rtow.make_config(64, 48, 1, 1, depth, seed=SEED), orc.RNG_PHILOX, nthreads=cores)  # warm up
    t0 = time.perf_counter()
    _, cst = orc.render(scene, cal, orc.RNG_PHILOX, nthreads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    rate = cst.samples / dt  # samples per second of the oracle on this host
    spp = int(max(1, min(256, seconds * rate / (W * height))))
    cfg = rtow.make_config(W, height, spp, 1, depth, seed=SEED)
    t0 = time.perf_counter()
    _, st = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": round(st.samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores,
        "kind": "port",
        "sample": f"{label} {W}x{height}, {spp} spp of the workload's spp, {depth} bounces, "
                  f"oracle/ (Philox stream) on {cores} threads, {dt:.1f} s",
    }


# ------------------------------------------------------------------------- workloads ---
def make_scene(kind):
    """(HostScene, label).  Mesh scenes come from the committed suzanne fixture; the 96,800-triangle
    mesh is its 10x10 subdivision (scripts/make_mesh.py; the reference's dragon.obj is absent)."""
    if kind in ("cover", "moving"):
        s = rtow.HostScene.cover(11, 1.5, kind == "moving")  # default mt19937 seed: 486 / 485 prims
        return s, f"RTOW cover scene ({s.c.n_prims} prims, {'moving' if kind == 'moving' else 'static'})"
    if kind == "suzanne":
        s = rtow.HostScene.obj(ROOT / "tests" / "golden" / "suzanne.obj", 16 / 9)
        return s, f"suzanne.obj triangle mesh ({s.c.n_prims} triangles)"
    tmp = Path(tempfile.gettempdir()) / f"rtow_mesh100k_{os.getpid()}.obj"
    subprocess.run([sys.executable, str(ROOT / "scripts" / "make_mesh.py"), str(tmp), "10"], check=True,
                   capture_output=True)
    try:
        s = rtow.HostScene.obj(tmp, 16 / 9)
    finally:
        tmp.unlink(missing_ok=True)
    return s, f"synthetic mesh, suzanne subdivided 10x10 ({s.c.n_prims} triangles; stands in for the absent dragon.obj)"


def prim_model(scene):
    c = scene.c
    n = {"sphere": c.n_spheres, "moving": c.n_moving, "triangle": c.n_triangles}
    bytes_all = sum(n[k] * BYTES_PER_PRIM[k] for k in n)
    flops_all = (n["sphere"] + n["moving"]) * FLOPS_SPHERE_TEST + n["triangle"] * FLOPS_TRI_TEST
    return n, bytes_all, flops_all


KERNEL_NAMES = {1: "stream (every lane tests every primitive, scalar-load broadcast)",
                2: "bvh (per-lane walk of the scene image in LDS, or in L2 when it does not fit)",
                3: "grid (per-lane 3D-DDA over the LDS scene image + large-primitive list)",
                4: "bvh4 (per-lane ordered walk of a 4-wide BVH, stack in LDS; image or its top levels in LDS)",
                5: "reftree (the reference's own median-split tree, left before right, f64 Aabb::hit; strict build)"}


def kernel_source_sha():
    """sha1 over the trace kernel's sources: a committed PMC profile belongs to the kernel it was taken from."""
    import hashlib

    h = hashlib.sha1()
    d = ROOT / "raytracing-one-weekend_amd" / "csrc"
    for f in sorted(list(d.glob("rtow_trace_*.h")) + [d / "rtow_device.h"]):
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load_pmc(workload, precision, kernel_used):
    """Derived PMC numbers of the committed profile of this workload/kernel (profiles/r02_pmc_*.json,
    written by scripts/pmc_summary.py from separate rocprofv3 --pmc passes), or None."""
    best = None
    sha = kernel_source_sha()
    for f in sorted((ROOT / "profiles").glob("r*_pmc_*.json")):
        try:
            pj = json.loads(f.read_text())
        except Exception:
            continue
        d = pj.get("derived")
        if not d or pj.get("workload") != workload or pj.get("precision") != precision or \
                pj.get("kernel_used") != kernel_used:
            continue
        cand = dict(d, pmc_file=f"profiles/{f.name}", kernel_source_sha=pj.get("kernel_source_sha"))
        # the profile of THIS kernel source wins over older ones of the same workload (kept for the record)
        if best is None or cand["kernel_source_sha"] == sha or best["kernel_source_sha"] != sha:
            best = cand
    return best


def rooflines(scene, st, kernel_ms, rows, W, workload, precision):
    """The two roofline objects of one launch (this rank's)."""
    n, bytes_all, flops_all = prim_model(scene)
    seg = float(st.segments)
    fb_bytes = rows * W * 24
    alg_bytes = math.ceil(seg / 64) * bytes_all + fb_bytes
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    pmc = load_pmc(workload, precision, st.kernel_used)
    # counters of ANOTHER build of the kernel are not this kernel's: null, with the file named
    pmc_stale = bool(pmc) and pmc.get("kernel_source_sha") != kernel_source_sha()
    traffic = pmc.get("hbm_bytes_per_launch") if (pmc and not pmc_stale) else None
    roof = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "kernel": "rtow_trace_" + precision, "kernel_ms": round(kernel_ms, 4),
        "binding_bound": "valu_issue (see roofline_valu): `achieved` here is the metric's equivalent-streaming figure, "
                         "not bytes the kernel moves; measured_hbm_GBps is",
        "traffic_source": None if not pmc else (pmc["pmc_file"] + (" (STALE: taken from another build of the kernel, "
                                                                   "not reported)" if pmc_stale else "")),
        "algorithmic_bytes_per_launch": alg_bytes,
        "model": "EQUIVALENT STREAMING bandwidth, SURVEY.md §8d: ceil(segments/64) * sum(N_class * record bytes) "
                 "+ rows*W*24 B; f64 records: sphere 32 B, moving sphere 56 B, triangle 72 B.  Not the bytes the "
                 "kernel moves: the scene image is read from LDS (L2 for the 96.8k mesh), see measured_*",
        "measured_hbm_GBps": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9, 2),
        "measured_frac": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
    }
    # what the kernel computes, from its own counters
    if st.kernel_used == rtow.KERNEL_BRUTE:
        flops = seg * flops_all
        fmodel = "segments * sum(N_class * flops per test): sphere 23 (to the discriminant reject), triangle 59"
        walk = None
    else:
        grid = st.kernel_used == rtow.KERNEL_GRID
        # per node test: grid = one 4-byte cell word and a DDA step; BVH = one 32-byte node and a slab test;
        # BVH4 = one 128-byte node (112 read) and four slab tests
        nb, nf = (4, 12) if grid else ((112, 68) if st.kernel_used == rtow.KERNEL_BVH4 else (32, 17))
        tri = n["triangle"] > 0
        pb, pf = (96 + 0, FLOPS_TRI_TEST) if tri else (32 + 4, FLOPS_SPHERE_TEST)
        flops = int(st.node_tests) * nf + int(st.prim_tests) * pf
        walk = {
            "image_bytes_read_per_launch": int(st.node_tests) * nb + int(st.prim_tests) * pb,
            "flops_per_launch": flops,
            "model": f"node_tests*{nb} B + prim_tests*{pb} B read from the scene image; node_tests*{nf} flop (f32 "
                     + ("DDA step" if grid else ("4 slab tests" if st.kernel_used == rtow.KERNEL_BVH4 else "slab test"))
                     + f") + prim_tests*{pf} flop (f64)",
        }
        walk["image_GBps"] = round(walk["image_bytes_read_per_launch"] / (kernel_ms * 1e-3) / 1e9, 1)
        fmodel = walk["model"]
    tf = flops / (kernel_ms * 1e-3) / 1e12
    valu = {
        "bound": "valu_issue", "achieved": round(tf, 3), "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s",
        "frac": round(tf / F64_VALU_PEAK_TF, 5),
        "model": "useful hit-test flops per second (counted by the kernel) vs the f64 vector peak; " + fmodel,
        "issue_utilisation": pmc.get("valu_issue_utilisation") if (pmc and not pmc_stale) else None,
        "lane_activity": pmc.get("lane_activity") if (pmc and not pmc_stale) else None,
        "valu_insts_per_launch": pmc.get("valu_insts_per_launch") if (pmc and not pmc_stale) else None,
        "pmc_file": pmc.get("pmc_file") if pmc else None,
        "pmc_stale": pmc_stale if pmc else None,
        "pmc_note": "issue_utilisation = SQ_ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); lane_activity = "
                    "SQ_THREAD_CYCLES_VALU / (SQ_INSTS_VALU*64); from separate rocprofv3 --pmc passes of this "
                    "workload (null: no committed profile matches this workload/kernel; the two counters come from "
                    "different passes, so a saturated kernel can read a few per cent above 1)",
    }
    # The line's `roofline` object is the bound that BINDS: VALU issue.  `frac` = the share of the chip's vector
    # lane-slots that carried an active lane's instruction while the kernel ran = issue utilisation x lane activity
    # (PMC, the committed profile of this very kernel source; null when the profile is of another build).  The
    # metric's own figure — SURVEY.md 8d's equivalent-streaming bandwidth against the HBM peak — stays beside it
    # under `hbm_equivalent_streaming`: it is nominal for kernels that cull and read LDS (it passes 1), which is
    # why it is not what `roofline.frac` says.
    iu, la = valu["issue_utilisation"], valu["lane_activity"]
    frac = round(min(iu, 1.0) * la, 4) if (iu is not None and la is not None) else None
    binding = {
        "bound": "valu_issue", "achieved": frac, "peak": 1.0,
        "unit": "fraction of VALU lane-slots carrying an active lane (issue utilisation x lane activity)",
        "frac": frac, "traffic": traffic,
        "kernel": roof["kernel"], "kernel_ms": roof["kernel_ms"],
        "issue_utilisation": iu, "lane_activity": la, "valu_insts_per_launch": valu["valu_insts_per_launch"],
        "pmc_file": valu["pmc_file"], "pmc_stale": valu["pmc_stale"],
        "measured_hbm_GBps": roof["measured_hbm_GBps"], "measured_hbm_frac_of_peak": roof["measured_frac"],
        "hbm_equivalent_streaming": roof,
        "note": "not MFMA (no dense contraction) and not HBM (scene in LDS; HBM traffic is the partial sums): the kernel "
                "is bound by vector-instruction issue.  hbm_equivalent_streaming is the metric's nominal figure.",
    }
    return binding, valu, walk


def timed_render_loop(ctx, cfg, d_ptr, stream, dev, steps, warmup):
    """Single-rank helper for the extra measurements: (ms per step, kernel ms, Stats)."""
    import ctypes as C

    st0 = ctx.render_device(cfg, d_ptr, stream, True)
    for _ in range(max(warmup - 1, 0)):
        ctx.render_device(cfg, d_ptr, stream, False)
    torch.cuda.synchronize(dev)
    L = rtow.lib()
    L.rtow_profile_collect(ctx._h, None, None)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.render_device(cfg, d_ptr, stream, False)
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t0) / steps
    kms, nl = C.c_double(), C.c_int32()
    rtow.check(L.rtow_profile_collect(ctx._h, C.byref(kms), C.byref(nl)), "profile_collect")
    return el * 1e3, kms.value / max(nl.value, 1), st0


def other_config(name, a, dev, precision, steps, stress=False):
    """One shortened line for another BASELINE config on this GPU (own context and scene).
    `stress`: configs[4]'s "SoA primitive streaming" reading — the STREAM kernel, every wave streaming ALL
    triangle records through scalar loads (960x540, 1 spp): the one line whose `roofline.achieved` is bytes
    the kernel really moves (L2 / Infinity Cache to the scalar unit), not the equivalent-streaming figure."""
    kind, W, aspect, spp, depth, spi, base = WORKLOADS[name]
    scene, label = make_scene(kind)
    if stress:
        W, spp, spi = 960, 1, 1
        base += " as a primitive-streaming stress (every ray tests every triangle)"
    H = rtow.image_height(W, aspect)
    cfg = rtow.make_config(W, H, spp, max(1, spp // spi), depth, seed=SEED, precision=precision,
                           kernel=rtow.KERNEL_BRUTE if stress else rtow.KERNEL_AUTO)
    ctx = rtow.Context(dev.index)
    ctx.upload(scene)
    bi = ctx.build_info()
    out = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
    ms, kms, st = timed_render_loop(ctx, cfg, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream, dev, steps, 1)
    roof, valu, walk = rooflines(scene, st, kms, H, W, name, a.precision)
    line = {
        "workload": f"{label} {W}x{H}, {spp} spp, {depth} bounces", "baseline_config": base,
        "value": round(st.samples / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 4),
        "steps": steps, "kernel_ms": round(kms, 4), "kernel": KERNEL_NAMES[st.kernel_used],
        "scene_image_bytes": bi.grid_image_bytes if st.kernel_used == rtow.KERNEL_GRID else
                             (bi.bvh4_image_bytes if st.kernel_used == rtow.KERNEL_BVH4 else bi.bvh_image_bytes),
        "build_ms": round(bi.bvh_build_ms + bi.grid_build_ms, 3),
        "segments_per_sample": round(st.segments / st.samples, 4),
        "node_tests_per_segment": round(st.node_tests / max(st.segments, 1), 3),
        "prim_tests_per_segment": round(st.prim_tests / max(st.segments, 1), 3),
        "roofline": roof, "roofline_valu": valu, "walk": walk,
    }
    if not stress and kind in ("suzanne", "mesh100k"):
        # What a caller waits for on a mesh (SURVEY.md §8d's region, the reference's own timer includes its tree build,
        # src/render.cpp:141-188): rtow_render_rgb8 = upload + acceleration build + trace + write_color + D2H of
        # the bytes, with the host SAH builder and with the device LBVH builder (both make the 4-wide image)
        import ctypes as C
        import numpy as np

        L = rtow.lib()
        host8 = np.zeros((H, W, 3), dtype=np.uint8)
        e2e = {}
        for bname, b in (("host_sah", rtow.BUILDER_HOST_SAH), ("device_lbvh", rtow.BUILDER_DEVICE_LBVH)):
            ctx.set_builder(b)
            n_calls = 2
            ts = []
            for i in range(n_calls + 1):
                t0 = time.perf_counter()
                rtow.check(L.rtow_render_rgb8(ctx._h, C.byref(scene.c), C.byref(cfg), host8.ctypes.data_as(C.c_void_p), None),
                           "rtow_render_rgb8")
                ts.append(time.perf_counter() - t0)
            dt = sum(ts[1:]) / n_calls
            bi2 = ctx.build_info()
            e2e[bname] = {"value": round(W * H * spp / dt / 1e6, 3), "ms_per_call": round(dt * 1e3, 3),
                          "build_ms": round(bi2.bvh_build_ms + bi2.grid_build_ms, 3), "upload_ms": round(bi2.upload_ms, 3),
                          "bvh4_nodes": bi2.bvh4_nodes, "bvh4_node_bytes": bi2.bvh4_node_bytes}
        best = max(e2e, key=lambda k: e2e[k]["value"])
        line["end_to_end_rgb8"] = dict(e2e, best_builder=best, value_e2e=e2e[best]["value"],
                                       region="rtow_render_rgb8(): upload + acceleration build + trace + reduce + "
                                              "write_color + D2H of W*H*3 bytes, per call")
        # the same call at a LOW sample count (32 spp; the reference's default is 20, src/render.h:15), where the host
        # build is a large share of the frame: host, device, and what RTOW_BUILDER_AUTO (the context's default) picks
        cfg_lo = rtow.make_config(W, H, 32, 2, depth, seed=SEED, precision=precision, kernel=rtow.KERNEL_AUTO)
        lo = {}
        for bname, b in (("host_sah", rtow.BUILDER_HOST_SAH), ("device_lbvh", rtow.BUILDER_DEVICE_LBVH), ("auto", rtow.BUILDER_AUTO)):
            ctx.set_builder(b)
            ts = []
            for i in range(4):
                t0 = time.perf_counter()
                rtow.check(L.rtow_render_rgb8(ctx._h, C.byref(scene.c), C.byref(cfg_lo), host8.ctypes.data_as(C.c_void_p), None),
                           "rtow_render_rgb8")
                ts.append(time.perf_counter() - t0)
            dt = sum(ts[1:]) / 3
            bi2 = ctx.build_info()
            lo[bname] = {"value": round(W * H * 32 / dt / 1e6, 3), "ms_per_call": round(dt * 1e3, 3),
                         "build_ms": round(bi2.bvh_build_ms + bi2.grid_build_ms, 3),
                         "builder_used": "device" if bi2.builder == rtow.BUILDER_DEVICE_LBVH else "host"}
        line["end_to_end_rgb8_32spp"] = lo
        ctx.set_builder(rtow.BUILDER_HOST_SAH)
    ctx.close()
    del out
    return line


def strip_height(H, world):
    """Rows per strip for `world` ranks: 8 (8x8-pixel tiles) when the strips deal out evenly, else 4 (16x4 tiles)."""
    if os.environ.get("RTOW_BENCH_TILE_ROWS"):  # experiment knob: strip height (also the tile shape: 64 / rows wide)
        return int(os.environ["RTOW_BENCH_TILE_ROWS"])
    return 8 if H % (8 * world) == 0 else (4 if H % (4 * world) == 0 else 8)


def scale_projection(ctx, W, H, depth, precision, kernel, dev, stream, base_value, launches=3, ranks_of=(2, 4, 8)):
    """configs[2] (cover, 500 spp) as every rank of an N-GPU run would trace it, timed ONE RANK AT A TIME on
    this GPU: for N in 2/4/8 and bench.py's own strip height, rank r's strips (rtow_config_t rank/nranks/
    tile_rows — the launch that rank runs, same scene, same seed) with HIP events around the trace kernel and
    the host clock around trace + reduce.  What it cannot hold: the gather (W*H*24 B / N per rank over xGMI,
    ~0.1 ms) and rank start-up skew.  Projection = W*H*500 / max_r(step ms of rank r)."""
    out = {"workload": f"configs[2]: {W}x{H}, 500 spp, {depth} bounces; each rank's strips traced alone on this GPU",
           "launches_per_rank": launches, "base_Msamples_per_s_1gpu": base_value, "by_n": []}
    spp = 500
    for N in ranks_of:
        tr = strip_height(H, N)
        per_rank = []
        for r in range(N):
            cfg = rtow.make_config(W, H, spp, spp // SAMPLES_PER_ITEM, depth, seed=SEED, precision=precision,
                                   kernel=kernel, rank=r, nranks=N, tile_rows=tr)
            rows = rtow.local_rows(cfg)
            buf = torch.zeros((len(rows), W, 3), dtype=torch.float64, device=dev)
            ms, kms, st = timed_render_loop(ctx, cfg, buf.data_ptr(), stream.cuda_stream, dev, launches, 1)
            per_rank.append({"rank": r, "rows": len(rows), "step_ms": round(ms, 4), "kernel_ms": round(kms, 4),
                             "segments_per_sample": round(st.segments / max(st.samples, 1), 4)})
            del buf
        step = [p["step_ms"] for p in per_rank]
        kern = [p["kernel_ms"] for p in per_rank]
        proj = W * H * spp / (max(step) * 1e-3) / 1e6
        out["by_n"].append({
            "n_gpus": N, "tile_rows": tr, "per_rank": per_rank,
            "step_ms_max": round(max(step), 4), "step_ms_mean": round(sum(step) / N, 4),
            "kernel_ms_max": round(max(kern), 4), "kernel_ms_mean": round(sum(kern) / N, 4),
            "imbalance_max_over_mean": round(max(kern) / (sum(kern) / N), 4),
            "projected_Msamples_per_s": round(proj, 1),
            "projected_efficiency": round(proj / (N * base_value), 4) if base_value else None,
        })
    out["note"] = ("efficiency = projected / (N x the same frame on this one GPU, scaling_base); excludes the one "
                   "gather per frame and start-up skew, which the driver's SCALE run adds")
    return out


def samples_per_item(ctx, cfg):
    """Length of a work item of this render (rtow_debug_schedule): spp / nstreams in the strict build, the
    divisor of the sample range nearest the aimed-at length in the fast builds (10; 16 for a mesh)."""
    import ctypes as C
    L = rtow.lib()
    if not hasattr(L, "rtow_debug_schedule"):
        return cfg.samples_per_pixel // cfg.nstreams
    pairs = (C.c_uint32 * 8)()
    n = L.rtow_debug_schedule(ctx._h, C.byref(cfg), pairs, 4)
    return int(pairs[1]) if n > 0 else 0


def write_details(out, where):
    """The full record of the run as a JSON file beside the compact line; returns the path written (repo-relative
    when inside the repo) or None."""
    cands = [Path(where)] if where else [ROOT / "gpurun_out" / "bench_details.json",
                                         Path(tempfile.gettempdir()) / "rtow_bench_details.json"]
    for p in cands:
        try:
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(json.dumps(out, indent=1) + "\n")
            try:
                return str(p.resolve().relative_to(ROOT))
            except ValueError:
                return str(p)
        except OSError:
            continue
    return None


def main():
    a = parse()
    if a.moving:
        a.workload = "moving"
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))  # nothing above touched the GPU
    # stdout carries ONE line, the JSON of rank 0: whatever the libraries underneath print there (gloo's "connected
    # to N peer ranks", an RCCL banner) goes to stderr instead.  The descriptor is swapped, not sys.stdout: those
    # prints come from C++.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    rehearsal = a.backend == "gloo"
    if rehearsal:
        local_rank = 0  # all ranks share the one GPU; collectives go through host memory
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: device {local_rank} not visible ({torch.cuda.device_count()} devices)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", timeout=__import__("datetime").timedelta(seconds=180))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=__import__("datetime").timedelta(seconds=180))

    kind, W, aspect, spp0, DEPTH, spi, base_cfg = WORKLOADS[a.workload]
    if a.width:
        W, base_cfg = a.width, "custom"
    H = rtow.image_height(W, aspect)
    spp = a.spp or (spp0 if (world == 1 or kind not in ("cover", "moving")) else 500)
    nstreams = max(1, spp // min(spi, spp))
    # strips of 8 rows (8x8-pixel tiles: +0.8 % over 16x4) when they deal out evenly, else 4
    tile_rows = strip_height(H, world)
    precision = {"fast": rtow.F64_FAST, "strict": rtow.F64_STRICT, "f32": rtow.F32}[a.precision]
    kernel = {"auto": rtow.KERNEL_AUTO, "brute": rtow.KERNEL_BRUTE, "bvh": rtow.KERNEL_BVH, "grid": rtow.KERNEL_GRID,
              "bvh4": rtow.KERNEL_BVH4, "reftree": rtow.KERNEL_REFTREE}[a.kernel]
    split_samples = a.split == "samples" and world > 1
    if split_samples:
        s_first, s_count = tiles.stream_range(nstreams, world, rank)
        cfg = rtow.make_config(W, H, spp, nstreams, DEPTH, seed=SEED, precision=precision, kernel=kernel,
                               stream_first=s_first, stream_count=s_count)
    else:
        cfg = rtow.make_config(W, H, spp, nstreams, DEPTH, seed=SEED, precision=precision,
                               kernel=kernel, rank=rank, nranks=world, tile_rows=tile_rows)

    scene, label = make_scene(kind)
    ctx = rtow.Context(local_rank)
    ctx.upload(scene)  # scene resident in HBM before the timed region

    rows = rtow.local_rows(cfg)
    if split_samples:
        sg = tiles.FrameSum(H, W, rank, world, torch.device("cpu") if rehearsal else dev)
        sg.gather = sg.reduce
    else:
        sg = tiles.StripGather(H, W, tile_rows, rank, world, torch.device("cpu") if rehearsal else dev)
        assert sg.rows == rows
    local = torch.zeros_like(sg.local, device=dev) if rehearsal else sg.local
    stream = torch.cuda.current_stream(dev)
    last = {}

    def step(want_stats=False):
        st = ctx.render_device(cfg, local.data_ptr(), stream.cuda_stream, want_stats)
        if world > 1:
            if rehearsal:
                sg.local.copy_(local)  # gloo has no device gather: stage through the host
            last["image"] = sg.gather()  # the one collective: framebuffer strips -> rank 0 (RCCL gather)
        return st

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    st0 = step(want_stats=True)  # also sizes the workspace (allocation outside the timed region)
    spi = samples_per_item(ctx, cfg)
    for _ in range(max(a.warmup - 1, 0)):
        step()
    fence()
    ctx_lib = rtow.lib()
    import ctypes as C
    ctx_lib.rtow_profile_collect(ctx._h, None, None)  # reset the event ring
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kms, nl = C.c_double(), C.c_int32()
    rtow.check(ctx_lib.rtow_profile_collect(ctx._h, C.byref(kms), C.byref(nl)), "profile_collect")
    kernel_ms = kms.value / max(nl.value, 1)

    # max over ranks of the step time; sums of per-rank work
    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    work = torch.tensor([float(st0.samples), float(st0.segments)], dtype=torch.float64, device=dev)
    if world > 1:
        if rehearsal:
            tt, work = tt.cpu(), work.cpu()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    elapsed, kernel_ms_max = float(tt[0]), float(tt[1])
    samples, segments = float(work[0]), float(work[1])

    if rank == 0:
        if a.dump_image:
            import numpy as np

            img = last["image"] if world > 1 else local[: len(rows)]
            np.save(a.dump_image, img.cpu().numpy())
        spp_eff = rtow.spp_effective(cfg)
        assert samples == W * H * spp_eff, (samples, W * H * spp_eff)
        ms_per_step = elapsed / a.steps * 1e3
        value = samples / (elapsed / a.steps) / 1e6
        # dominant kernel on THIS rank (rank 0): algorithmic bytes of its launch / its duration
        seg0 = float(st0.segments)
        roof, valu, walk = rooflines(scene, st0, kernel_ms, len(rows), W, a.workload, a.precision)
        roof["kernel_ms_max_over_ranks"] = round(kernel_ms_max, 4)
        roof["hbm_equivalent_streaming"]["kernel_ms_max_over_ranks"] = round(kernel_ms_max, 4)
        out = {
            "metric": "Msamples/sec (W×H×spp) on cover scene; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if a.precision != "f32" else "f32 (preview build: NOT the metric's binary64 arithmetic)",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {
                "workload": f"{label} {W}x{H}, {spp} spp, {DEPTH} bounces"
                            + ("" if world == 1 else
                               (f", sample-split over {world} GPUs + gather of full frames" if split_samples else
                                f", {tile_rows}-row strips over {world} GPUs + 1 RCCL gather")),
                "baseline_config": base_cfg if (world == 1 and spp == spp0 or base_cfg == "custom") else
                                   ("configs[2]" if (spp == 500 and kind == "cover") else "custom"),
                "spp_effective": spp_eff, "samples_per_item": spi, "nstreams": nstreams,
                "seed": SEED, "precision": a.precision,
                "kernel": KERNEL_NAMES[st0.kernel_used],
                "segments_per_sample": round(segments / samples, 4),
                "node_tests_per_segment": round(st0.node_tests / max(seg0, 1), 3),
                "prim_tests_per_segment": round(st0.prim_tests / max(seg0, 1), 3),
            },
            "roofline": roof,
            "walk": walk,
            "roofline_valu": valu,
        }
        if world == 1 and not a.no_end_to_end:
            # SURVEY.md §8d's region through the host-buffer entry point: scene upload (incl. the host
            # build of the acceleration structure), kernels, D2H of the f64 sums into caller memory
            import numpy as np

            import ctypes as C2

            host = np.zeros((H, W, 3), dtype=np.float64)
            host8 = np.zeros((H, W, 3), dtype=np.uint8)
            L = rtow.lib()
            L.rtow_render_rgb8.argtypes = [C2.c_void_p, C2.POINTER(rtow.Scene), C2.POINTER(rtow.Config), C2.c_void_p,
                                           C2.POINTER(rtow.Stats)]
            n_e2e = 10  # (≈ 90 ms per entry point: three calls scattered by ±1 % from run to run)

            def timed(fn):
                fn()
                t1 = time.perf_counter()
                for _ in range(n_e2e):
                    fn()
                return (time.perf_counter() - t1) / n_e2e

            e1 = timed(lambda: ctx.render(scene, cfg, into=host))
            e8 = timed(lambda: rtow.check(L.rtow_render_rgb8(ctx._h, C2.byref(scene.c), C2.byref(cfg),
                                                             host8.ctypes.data_as(C2.c_void_p), None), "rtow_render_rgb8"))
            bi = ctx.build_info()
            out["end_to_end"] = {
                "value": round(W * H * spp_eff / e1 / 1e6, 3), "unit": "Msamples/s", "ms_per_call": round(e1 * 1e3, 4),
                "calls": n_e2e, "upload_ms": round(bi.upload_ms, 3),
                "build_ms": round(bi.bvh_build_ms + bi.grid_build_ms, 3),
                "region": "rtow_render(): scene upload + acceleration build (only what this config's kernel reads) + trace + "
                          "reduce + D2H of W*H*3 f64 sums into caller memory (SURVEY.md §8d)",
                "rgb8": {"value": round(W * H * spp_eff / e8 / 1e6, 3), "ms_per_call": round(e8 * 1e3, 4),
                         "region": "rtow_render_rgb8(): the same with write_color on the device and W*H*3 BYTES to the "
                                   "host — the values the reference prints into its PPM (src/render.cpp:11-20,182-186)"},
            }
            # SURVEY §8d's timed region as a co-equal headline: `value` is the device-resident rate the contract
            # asks for (inputs in HBM when the clock starts), `value_e2e` what one call of the boundary delivers
            out["value_e2e"] = out["end_to_end"]["rgb8"]["value"]
            out["ms_per_step_e2e"] = out["end_to_end"]["rgb8"]["ms_per_call"]
            out["e2e_over_device_resident"] = round(out["value_e2e"] / value, 4)
            ctx.upload(scene)  # (rtow_render uploads only what its kernel reads: the full scene again for what follows)
            # The multi-GPU product path (csrc/rtow_multi.cpp) with a ONE-device handle, RCCL on: what a frame costs
            # beyond the kernel — worker hand-off, one ncclGather, the row-placement kernel, one D2H into caller memory.
            # (rtow_render_rgb8 above also uploads the scene per call; the handle renders a resident scene.)
            try:
                mh = rtow.MultiContext([dev.index], use_rccl=True)
                mh.upload(scene)
                cfg_m = rtow.make_config(W, H, spp, nstreams, DEPTH, seed=SEED, precision=precision, kernel=kernel,
                                         tile_rows=tile_rows)
                def timed_frames(fn):
                    """mean seconds per frame and the mean of the handle's own breakdown over the same frames"""
                    fn()
                    acc = {}
                    t1 = time.perf_counter()
                    for _ in range(n_e2e):
                        fn()
                        for k, v in mh.frame_breakdown().items():  # (a dozen doubles: ~10 us, inside the timed loop)
                            acc[k] = acc.get(k, 0.0) + v / n_e2e
                    return (time.perf_counter() - t1) / n_e2e, {k: round(v, 4) for k, v in acc.items()}

                m8, b8 = timed_frames(lambda: rtow.check(L.rtow_multi_render_rgb8(mh._h, C2.byref(cfg_m), host8.ctypes.data_as(C2.c_void_p),
                                                                                  None), "rtow_multi_render_rgb8"))
                m64, b64 = timed_frames(lambda: rtow.check(L.rtow_multi_render(mh._h, C2.byref(cfg_m), host.ctypes.data_as(C2.POINTER(C2.c_double)),
                                                                               None), "rtow_multi_render"))
                mh.close()
                out["multi_handle"] = {
                    "devices": [dev.index], "use_rccl": True, "calls": n_e2e,
                    "rgb8": {"value": round(W * H * spp_eff / m8 / 1e6, 3), "ms_per_frame": round(m8 * 1e3, 4),
                             "over_rtow_render_rgb8": round(e8 / m8, 4), "breakdown_ms": b8,
                             # what the handle adds to the device's own trace + reduce time, per frame
                             "over_kernel_ms": round(m8 * 1e3 - b8.get("dev_trace", float("nan")), 4)},
                    "f64": {"value": round(W * H * spp_eff / m64 / 1e6, 3), "ms_per_frame": round(m64 * 1e3, 4),
                            "over_rtow_render": round(e1 / m64, 4), "breakdown_ms": b64,
                            "over_kernel_ms": round(m64 * 1e3 - b64.get("dev_trace", float("nan")), 4)},
                    "region": "rtow_multi_render{_rgb8,}(): resident scene; per frame one hand-off to the workers, one trace "
                              "launch, ONE ncclGather (one-rank communicator), one D2H into caller memory, one wait",
                }
            except rtow.RtowError as e:  # (no librccl.so on this machine: the line says so instead of failing the bench)
                out["multi_handle"] = {"error": str(e)}
        if world == 1 and a.workload == "cover" and not a.no_reference_boundary:
            # What a drop-in caller of render(scene, cfg) gets: Config::nthreads = 4 (src/render.h:18) -> nstreams = 4,
            # at the two sample counts of BASELINE configs[1] / [2], device-resident and through the boundary call
            rb = {"nstreams": 4, "note": "the reference's default thread count; fast build: the work items follow the "
                                         "sample schedule whatever nstreams is (include/rtow.h, rtow_debug_schedule)"}
            import numpy as np

            import ctypes as C3

            host8b = np.zeros((H, W, 3), dtype=np.uint8)
            Lb = rtow.lib()
            Lb.rtow_render_rgb8.argtypes = [C3.c_void_p, C3.POINTER(rtow.Scene), C3.POINTER(rtow.Config), C3.c_void_p,
                                            C3.POINTER(rtow.Stats)]
            for spp_rb, nsteps in ((100, 5), (500, 2)):
                cfg4 = rtow.make_config(W, H, spp_rb, 4, DEPTH, seed=SEED, precision=precision, kernel=kernel)
                ms4, kms4, _ = timed_render_loop(ctx, cfg4, local.data_ptr(), stream.cuda_stream, dev, nsteps, 1)

                def one():
                    rtow.check(Lb.rtow_render_rgb8(ctx._h, C3.byref(scene.c), C3.byref(cfg4),
                                                   host8b.ctypes.data_as(C3.c_void_p), None), "rtow_render_rgb8")

                one()
                tb = time.perf_counter()
                for _ in range(nsteps):
                    one()
                eb = (time.perf_counter() - tb) / nsteps
                ctx.upload(scene)
                rb[f"spp{spp_rb}"] = {"value": round(W * H * spp_rb / (ms4 * 1e-3) / 1e6, 3), "ms_per_step": round(ms4, 4),
                                      "kernel_ms": round(kms4, 4), "samples_per_item_strict_would_be": spp_rb // 4,
                                      "value_e2e_rgb8": round(W * H * spp_rb / eb / 1e6, 3), "ms_per_call_e2e": round(eb * 1e3, 4)}
            out["reference_boundary"] = rb
        if world == 1 and a.workload == "cover" and spp == 100 and not a.no_scaling_base:
            # like-for-like base of the N>1 lines (configs[2], 500 spp): the same frame on this one
            # GPU, measured after the timed region (a longer launch amortises the end-of-launch tail)
            cfg5 = rtow.make_config(W, H, 500, 500 // SAMPLES_PER_ITEM, DEPTH, seed=SEED, precision=precision, kernel=kernel)
            ctx.render_device(cfg5, local.data_ptr(), stream.cuda_stream, True)
            torch.cuda.synchronize(dev)
            t5 = time.perf_counter()
            for _ in range(3):
                ctx.render_device(cfg5, local.data_ptr(), stream.cuda_stream, False)
            torch.cuda.synchronize(dev)
            e5 = (time.perf_counter() - t5) / 3
            out["scaling_base"] = {
                "workload": f"configs[2] on one GPU: {W}x{H}, 500 spp, {DEPTH} bounces (no gather)",
                "value": round(W * H * 500 / e5 / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(e5 * 1e3, 4),
                "steps": 3,
            }
            if not a.no_scale_projection:
                out["scale_projection"] = scale_projection(ctx, W, H, DEPTH, precision, kernel, dev, stream,
                                                           out["scaling_base"]["value"])
                # second column: the C++ product path's own per-frame cost on top of every rank's launch (measured on
                # the one-device handle above: hand-off, gather of one rank, D2H, wait; an N-rank frame adds the
                # placement kernel and the gather's xGMI time, which one device cannot show)
                over = (out.get("multi_handle") or {}).get("rgb8", {}).get("over_kernel_ms")
                if over is not None and over == over:
                    out["scale_projection"]["handle_over_kernel_ms"] = over
                    for b in out["scale_projection"]["by_n"]:
                        proj = W * H * 500 / ((b["step_ms_max"] + over) * 1e-3) / 1e6
                        b["projected_Msamples_per_s_with_handle"] = round(proj, 1)
                        b["projected_efficiency_with_handle"] = round(proj / (b["n_gpus"] * out["scaling_base"]["value"]), 4)
        if world == 1 and a.workload == "cover" and spp == 100 and not a.no_scaling_base:
            # A SEQUENCE of frames, two in flight (never `value`, which is one render at a time): two contexts on this
            # device, each with its own stream, workspace and output, frames alternating between them, so that the next
            # frame's workgroups are dispatched onto the CUs the draining frame leaves.  Each
            # frame is a complete, separate render; the two outputs are compared bit for bit afterwards.
            ctx_b = rtow.Context(local_rank)
            ctx_b.upload(scene)
            out_b = torch.zeros_like(local)
            st_b = torch.cuda.Stream(dev)
            pairs = ((ctx, local, stream), (ctx_b, out_b, st_b))
            n_fif = max(a.steps, 2)
            for k in range(4):
                c2, o2, s2 = pairs[k % 2]
                c2.render_device(cfg, o2.data_ptr(), s2.cuda_stream, False)
            torch.cuda.synchronize(dev)
            tf = time.perf_counter()
            for k in range(n_fif):
                c2, o2, s2 = pairs[k % 2]
                c2.render_device(cfg, o2.data_ptr(), s2.cuda_stream, False)
            torch.cuda.synchronize(dev)
            ef = (time.perf_counter() - tf) / n_fif
            out["two_frames_in_flight"] = {
                "value": round(samples / ef / 1e6, 3), "unit": "Msamples/s", "ms_per_frame": round(ef * 1e3, 4), "frames": n_fif,
                "frames_identical": bool(torch.equal(local, out_b)),
                "note": "sustained rate of a sequence of frames alternating between two contexts / streams on one device; "
                        "`value` is one render at a time",
            }
            ctx_b.close()
        if world == 1 and a.workload == "cover" and not a.spp and not a.no_other_configs:
            out["other_configs"] = [other_config(n, a, dev, precision, s)
                                    for n, s in (("moving", 4), ("suzanne", 3), ("mesh100k", 2))]
            out["other_configs"].append(other_config("mesh100k", a, dev, precision, 1, stress=True))
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, label, W, H, DEPTH, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        details_path = write_details(out, a.details_out)
        sys.stderr.write("bench.py details: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        os.write(json_fd, (benchline.dumps(benchline.compact_line(out, details_path)) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
