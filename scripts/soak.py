#!/usr/bin/env python3
"""Soak: many seeds and shapes through the fast build of every workload, to catch a hang or a non-finite pixel that
the fixed-seed tests would miss (resumable walks, quorums).  Prints a progress line per workload."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import numpy as np
import torch  # noqa: F401  (load order, see rtow.lib)
import rtow

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = rtow.Context(0)
for name, scene, depth in (("cover", rtow.HostScene.cover(11, 1.5, False), 50), ("moving", rtow.HostScene.cover(11, 1.5, True), 50),
                           ("suzanne", rtow.HostScene.obj(ROOT / "tests/golden/suzanne.obj", 16 / 9), 20)):
    ctx.upload(scene)
    t0 = time.time()
    segs = 0
    for seed in range(1, n + 1):
        w = 320 + 64 * (seed % 7)
        h = 200 + 8 * (seed % 11)
        cfg = rtow.make_config(w, h, 20 + seed % 13, 2 + seed % 3, depth, seed=seed * 7919, precision=rtow.F64_FAST,
                               tile_rows=[8, 4, 2, 1][seed % 4])
        img, st = ctx.render(scene, cfg)
        assert np.isfinite(img).all() and (img >= 0).all(), (name, seed)
        assert st.samples == w * h * rtow.spp_effective(cfg)
        segs += st.segments
    print(f"{name}: {n} renders, {segs/1e6:.0f} M segments, {time.time()-t0:.1f} s", flush=True)
ctx.close()
print("soak ok")
