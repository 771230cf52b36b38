#!/usr/bin/env python3
"""What the end of a launch costs in SUSTAINED throughput: K frames of the cover scene rendered back to back
   (a) on one context and one stream, as bench.py's `value` does, and
   (b) alternating between F contexts, each with its own stream, workspace and output buffer, so that frame
       k+1's workgroups are dispatched onto the CUs frame k's draining waves leave idle.
Each frame is a complete, separate render (its own queue counter, partial sums and output); nothing is shared
but the GPU.  Prints one JSON line per shape.

   python scripts/frames_in_flight.py [--frames 20] [--inflight 2] [--spp 100] [--ranks-of 1]
`--ranks-of N`: the launch rank 0 of N runs (strips of the frame), the shape whose fixed end-of-launch
part weighs most."""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import torch
import rtow

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--inflight", type=int, default=2)
ap.add_argument("--spp", type=int, default=100)
ap.add_argument("--ranks-of", type=int, default=1)
ap.add_argument("--moving", action="store_true")
a = ap.parse_args()

W, aspect, depth = 1200, 1.5, 50
H = rtow.image_height(W, aspect)
scene = rtow.HostScene.cover(11, aspect, a.moving)
nranks = a.ranks_of
tile_rows = 8 if (H // 8) % nranks == 0 else 4
cfg = rtow.make_config(W, H, a.spp, max(1, a.spp // 10), depth, seed=1, precision=rtow.F64_FAST,
                       rank=0, nranks=nranks, tile_rows=tile_rows)
rows = len(rtow.local_rows(cfg))
samples = rows * W * rtow.spp_effective(cfg)


def run(nctx):
    ctxs, streams, outs = [], [], []
    for _ in range(nctx):
        c = rtow.Context(0)
        c.upload(scene)
        ctxs.append(c)
        streams.append(torch.cuda.Stream())
        outs.append(torch.zeros((rows, W, 3), dtype=torch.float64, device="cuda"))
    for i in range(2 * nctx):  # warm-up: sizes every workspace
        ctxs[i % nctx].render_device(cfg, outs[i % nctx].data_ptr(), streams[i % nctx].cuda_stream, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.frames):
        k = i % nctx
        ctxs[k].render_device(cfg, outs[k].data_ptr(), streams[k].cuda_stream, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    for c in ctxs:
        c.close()
    return dt, same


res = {"spp": a.spp, "ranks_of": nranks, "rows": rows, "frames": a.frames, "moving": a.moving}
for n in (1, a.inflight, 1, a.inflight):
    dt, same = run(n)
    res.setdefault(f"inflight{n}_ms_per_frame", []).append(round(dt * 1e3, 4))
    res.setdefault(f"inflight{n}_Gsamples_per_s", []).append(round(samples / dt / 1e9, 3))
    res[f"inflight{n}_frames_identical"] = same
print(json.dumps(res))
