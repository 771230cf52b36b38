#!/usr/bin/env python3
"""Trace throughput on a 3-D cloud of random spheres (not a BASELINE config: a check that grid heuristics
tuned on the one-layer cover scene do not hurt volumes).  usage: bench_cloud.py N_SPHERES [N_MOVING]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
sys.path.insert(0, str(ROOT / "tests"))
import torch
import rtow
from test_gpu_fuzz import random_scene

n = int(sys.argv[1]); nm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keep = []
sc = random_scene(7, n, nm, 0, keep)
W, H, spp = 1200, 800, 20
cfg = rtow.make_config(W, H, spp, 2, 50, seed=1, precision=rtow.F64_FAST)
ctx = rtow.Context(0)
ctx.upload(sc)
out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
best = 1e30
for _ in range(4):
    st = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
    best = min(best, st.kernel_ms)
print(json.dumps({"spheres": n, "moving": nm, "kernel": st.kernel_used, "kernel_ms": round(best, 3),
                  "Msamples_per_s": round(st.samples / best / 1e3, 1),
                  "node_tests_per_segment": round(st.node_tests / st.segments, 2),
                  "prim_tests_per_segment": round(st.prim_tests / st.segments, 2)}))
