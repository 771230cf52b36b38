#!/usr/bin/env python3
"""Knob sweep at the launch shape one rank of an N-GPU run traces (configs[2]: cover, 1200x800, 500 spp).

usage: scale_sweep.py [--n 8] [--launches 3] [--lib path.so] "ENV1=a ENV2=b" "ENV1=c" ...
Each argument is one setting (space-separated RTOW_* assignments; "" = defaults; BENCH_TILE_ROWS=k sets the strip
height, BENCH_SPI=k the samples per item).  For every setting: a fresh context (knobs are read at rtow_ctx_create),
every rank's strips traced alone on this GPU, HIP-event kernel time; also the N = 1 frame at 100 spp and 500 spp.
Prints one line per setting: max / mean per-rank kernel ms, projected Gsamples/s, efficiency against the same
setting's own 500-spp frame on one GPU.
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8)
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--lib", default="")
ap.add_argument("--moving", action="store_true")
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
if a.lib:
    os.environ["RTOW_LIB"] = a.lib
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import rtow  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = torch.cuda.current_stream(dev)
W, H, DEPTH = 1200, 800, 50
scene = rtow.HostScene.cover(11, 1.5, a.moving)
for setting in a.settings:
    saved = dict(os.environ)
    spi, tile_rows = 10, None
    for kv in setting.split():
        k, v = kv.split("=", 1)
        if k == "BENCH_SPI":
            spi = int(v)
        elif k == "BENCH_TILE_ROWS":
            tile_rows = int(v)
        else:
            os.environ[k] = v
    ctx = rtow.Context(0)
    ctx.upload(scene)
    res = {}
    for spp in (100, 500):
        cfg = rtow.make_config(W, H, spp, spp // spi, DEPTH, seed=1, precision=rtow.F64_FAST)
        buf = torch.zeros((H, W, 3), dtype=torch.float64, device=dev)
        ms, kms, st = bench.timed_render_loop(ctx, cfg, buf.data_ptr(), stream.cuda_stream, dev, a.launches, 1)
        res[spp] = (ms, kms)
    N = a.n
    tr = tile_rows or bench.strip_height(H, N)
    ks, ss = [], []
    for r in range(N):
        cfg = rtow.make_config(W, H, 500, 500 // spi, DEPTH, seed=1, precision=rtow.F64_FAST, rank=r, nranks=N, tile_rows=tr)
        rows = rtow.local_rows(cfg)
        buf = torch.zeros((len(rows), W, 3), dtype=torch.float64, device=dev)
        ms, kms, st = bench.timed_render_loop(ctx, cfg, buf.data_ptr(), stream.cuda_stream, dev, a.launches, 1)
        ks.append(kms)
        ss.append(ms)
    proj = W * H * 500 / (max(ss) * 1e-3) / 1e9
    base = W * H * 500 / (res[500][0] * 1e-3) / 1e9
    print(f"[{setting or 'defaults'}] N1: 100spp {res[100][1]:.3f} ms ({W*H*100/res[100][0]/1e6:.2f} G/s)  500spp {res[500][1]:.3f} ms ({base:.2f} G/s) | "
          f"N={N} strips {tr}: kernel max {max(ks):.3f} mean {sum(ks)/N:.3f}  step max {max(ss):.3f}  -> {proj:.1f} G/s, "
          f"eff {proj/(N*base):.3f}, fixed part {max(ks) - res[500][1]/N:.3f} ms", flush=True)
    ctx.close()
    os.environ.clear()
    os.environ.update(saved)
