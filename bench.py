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

One JSON line on rank 0, with
  roofline     — SURVEY.md §8d's streaming model: algorithmic bytes per launch =
                 ceil(segments/64) * N_prim * 32 B + W*H*24 B, divided by the trace
                 kernel's mean duration measured with HIP events on the launch
                 stream inside the timed region; peak = 8 TB/s HBM; `traffic` = HBM
                 bytes per launch from the committed PMC passes (profiles/).  The
                 scene is LDS resident, so the kernel is VALU-bound, not HBM-bound:
                 `walk` gives what the kernel really reads (LDS) and computes.
  cpu_baseline — oracle/ (the CPU restatement of the reference's sample loop, same
                 Philox stream) timed on this host's cores on a bounded sample of
                 the same workload (same scene and resolution, fewer spp).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import rtow  # noqa: E402
import tiles  # noqa: E402

W, ASPECT, DEPTH, SEED = 1200, 1.5, 50, 1
# Samples per work item (= spp / nstreams).  Short items shorten the end-of-launch tail, long items
# save partial-sum traffic and item bookkeeping.  With the tail measures of the end of round 1
# (queue ending on cheap rows, sample donation, no polling of the empty queue) the optimum moved
# from 4 to 10: C2 (100 spp) 2 -> 8.44, 4 -> 9.38, 5 -> 9.46, 10 -> 9.54, 20 -> 8.96 Gsamples/s;
# 500 spp on one GPU 4 -> 9.66, 10 -> 10.11, 20 -> 10.19 (scripts/spi_sweep_500.sh).
SAMPLES_PER_ITEM = int(os.environ.get("RTOW_BENCH_SPI", "10"))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_VALU_PEAK_TF = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
BYTES_PER_SPHERE = 32    # cx cy cz r^2 as f64 (16 B in f32, doubled for f64)
FLOPS_PER_SPHERE_TEST = 23  # to the discriminant reject (src/common-model.cpp:70-75)
BYTES_PER_NODE = 32      # f32 box + skip link + leaf word
FLOPS_PER_BOX_TEST = 17  # 6 fma-slabs + min/max network


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--no-scaling-base", action="store_true",
                    help="N=1: skip the extra configs[2] (500 spp) measurement on this GPU")
    ap.add_argument("--precision", choices=["fast", "strict", "f32"], default="fast",
                    help="fast/strict: binary64 (the metric's arithmetic); f32: the preview build, never the headline")
    ap.add_argument("--kernel", choices=["auto", "brute", "bvh", "grid"], default="auto")
    ap.add_argument("--moving", action="store_true", help="moving-sphere variant of the cover scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--split", choices=["tiles", "samples"], default="tiles",
                    help="N>1 decomposition: strips of rows + one gather (default, bit-identical for any N) "
                         "or the reference's own: full frame per rank with spp/N of the samples, frames summed")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the N>1 path with every rank on cuda:0 (1-GPU box)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


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


def cpu_baseline(scene, height, seconds):
    """Time the oracle (kind 'port') on this host: same scene/resolution, reduced spp."""
    sys.path.insert(0, str(ROOT / "tests"))
    import orc

    cores = host_cores()
    cal = rtow.make_config(W, height, 2, 1, DEPTH, seed=SEED)
    orc.render(scene, rtow.make_config(64, 48, 1, 1, DEPTH, seed=SEED), orc.RNG_PHILOX, nthreads=cores)  # warm up
    t0 = time.perf_counter()
    _, cst = orc.render(scene, cal, orc.RNG_PHILOX, nthreads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    rate = cst.samples / dt  # samples per second of the oracle on this host
    spp = int(max(1, min(256, seconds * rate / (W * height))))
    cfg = rtow.make_config(W, height, spp, 1, DEPTH, seed=SEED)
    t0 = time.perf_counter()
    _, st = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": round(st.samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores,
        "kind": "port",
        "sample": f"cover scene {W}x{height}, {spp} spp of the workload's spp, 50 bounces, "
                  f"oracle/ (Philox stream) on {cores} threads, {dt:.1f} s",
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    rehearsal = a.backend == "gloo"
    if rehearsal:
        local_rank = 0  # all ranks share the one GPU; collectives go through host memory
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    H = rtow.image_height(W, ASPECT)
    spp = a.spp or (100 if world == 1 else 500)
    nstreams = max(1, spp // SAMPLES_PER_ITEM)
    # strips of 8 rows (8x8-pixel tiles: +0.8 % over 16x4) when they deal out evenly, else 4
    tile_rows = 8 if H % (8 * world) == 0 else (4 if H % (4 * world) == 0 else 8)
    if os.environ.get("RTOW_BENCH_TILE_ROWS"):  # experiment knob: strip height (also the tile shape: 64 / rows wide)
        tile_rows = int(os.environ["RTOW_BENCH_TILE_ROWS"])
    precision = {"fast": rtow.F64_FAST, "strict": rtow.F64_STRICT, "f32": rtow.F32}[a.precision]
    kernel = {"auto": rtow.KERNEL_AUTO, "brute": rtow.KERNEL_BRUTE, "bvh": rtow.KERNEL_BVH, "grid": rtow.KERNEL_GRID}[a.kernel]
    split_samples = a.split == "samples" and world > 1
    if split_samples:
        s_first, s_count = tiles.stream_range(nstreams, world, rank)
        cfg = rtow.make_config(W, H, spp, nstreams, DEPTH, seed=SEED, precision=precision, kernel=kernel,
                               stream_first=s_first, stream_count=s_count)
    else:
        cfg = rtow.make_config(W, H, spp, nstreams, DEPTH, seed=SEED, precision=precision,
                               kernel=kernel, rank=rank, nranks=world, tile_rows=tile_rows)

    scene = rtow.HostScene.cover(11, ASPECT, a.moving)  # default mt19937 seed: 486 / 485 prims
    n_prims = scene.c.n_prims
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

    def step(want_stats=False):
        st = ctx.render_device(cfg, local.data_ptr(), stream.cuda_stream, want_stats)
        if world > 1:
            if rehearsal:
                sg.local.copy_(local)  # gloo has no device gather: stage through the host
            sg.gather()  # the one collective: framebuffer strips -> rank 0 (RCCL gather)
        return st

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    st0 = step(want_stats=True)  # also sizes the workspace (allocation outside the timed region)
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
        spp_eff = rtow.spp_effective(cfg)
        assert samples == W * H * spp_eff, (samples, W * H * spp_eff)
        ms_per_step = elapsed / a.steps * 1e3
        value = samples / (elapsed / a.steps) / 1e6
        # dominant kernel on THIS rank (rank 0): algorithmic bytes of its launch / its duration
        seg0 = float(st0.segments)
        fb_bytes = len(rows) * W * 24
        # SURVEY.md §8d, the figure the metric is defined on: one unit = a wave of 64 segments
        # streaming the primitive array once (f64 records: 32 B per sphere) + the framebuffer.
        alg_bytes = math.ceil(seg0 / 64) * n_prims * BYTES_PER_SPHERE + fb_bytes
        alg_flops = seg0 * n_prims * FLOPS_PER_SPHERE_TEST
        model = "ceil(segments/64)*N_prim*32B + rows*W*24B (SURVEY.md §8d streaming model, f64 records)"
        fmodel = "segments*N_prim*23 flop (sphere test to the discriminant reject, streaming model)"
        walked = None
        if st0.kernel_used in (rtow.KERNEL_BVH, rtow.KERNEL_GRID):
            # what the walking kernels really read and compute per launch (their own counters),
            # all from the LDS scene image: BVH = a 32 B node per box test, GRID = a 4 B cell word
            # per DDA step; both a 32 B record + 4 B id per primitive test
            grid = st0.kernel_used == rtow.KERNEL_GRID
            nb, nf = (4, 12) if grid else (BYTES_PER_NODE, FLOPS_PER_BOX_TEST)
            walked = {
                "lds_bytes_per_launch": int(st0.node_tests) * nb + int(st0.prim_tests) * (BYTES_PER_SPHERE + 4),
                "flops_per_launch": int(st0.node_tests) * nf + int(st0.prim_tests) * FLOPS_PER_SPHERE_TEST,
                "model": (f"node_tests*{nb}B + prim_tests*36B ; node_tests*{nf} flop (f32 "
                          + ("DDA step" if grid else "slab") + ") + prim_tests*23 flop (f64)"),
            }
            walked["lds_GBps"] = round(walked["lds_bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9, 1)
            walked["TFLOPs"] = round(walked["flops_per_launch"] / (kernel_ms * 1e-3) / 1e12, 3)
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        prof = ROOT / "profiles" / "r01_hbm_traffic.json"
        if prof.exists():
            try:
                pj = json.loads(prof.read_text())
                if pj.get("workload_spp") == spp and pj.get("n_gpus") == world:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec (W×H×spp) on cover scene; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if a.precision != "f32" else "f32 (preview build: NOT the metric's binary64 arithmetic)",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {
                "workload": f"RTOW cover scene ({n_prims} prims, {'moving' if a.moving else 'static'}) "
                            f"{W}x{H}, {spp} spp, {DEPTH} bounces"
                            + ("" if world == 1 else
                               (f", sample-split over {world} GPUs + gather of full frames" if split_samples else
                                f", {tile_rows}-row strips over {world} GPUs + 1 RCCL gather")),
                "baseline_config": "configs[1]" if (world == 1 and spp == 100) else
                                   ("configs[2]" if spp == 500 else "custom"),
                "spp_effective": spp_eff, "samples_per_item": spp // nstreams, "nstreams": nstreams,
                "seed": SEED, "precision": a.precision,
                "kernel": {1: "stream (every lane tests every primitive, scalar-load broadcast)",
                           2: "bvh (per-lane threaded walk of the LDS scene image)",
                           3: "grid (per-lane 3D-DDA over the LDS scene image + large-primitive list)"}[st0.kernel_used],
                "segments_per_sample": round(segments / samples, 4),
                "node_tests_per_segment": round(st0.node_tests / max(seg0, 1), 3),
                "prim_tests_per_segment": round(st0.prim_tests / max(seg0, 1), 3),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "rtow_trace_" + a.precision, "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
                "algorithmic_bytes_per_launch": alg_bytes,
                "model": model,
            },
            "walk": walked,
            "roofline_valu": None if walked else {
                "bound": "valu_f64", "achieved": round(alg_flops / (kernel_ms * 1e-3) / 1e12, 3),
                "peak": F64_VALU_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(alg_flops / (kernel_ms * 1e-3) / 1e12 / F64_VALU_PEAK_TF, 5),
                "model": fmodel,
            },
        }
        if world == 1 and spp == 100 and not a.no_scaling_base:
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
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, H, a.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
