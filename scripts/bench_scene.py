#!/usr/bin/env python3
"""Throughput of the trace kernel on the other BASELINE configs (not the bench.py headline):
   python scripts/bench_scene.py suzanne|mesh100k|cover|moving [--width W --spp S --steps K ...]
C4 = suzanne 1920x1080x256 spp (20 bounces, the reference default); C5 = synthetic
96,800-triangle mesh 1920x1080x1024 spp (scripts/make_mesh.py)."""
import argparse, json, subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import torch
import rtow

ap = argparse.ArgumentParser()
ap.add_argument("scene")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--precision", default="fast", choices=["fast", "strict", "f32"])
ap.add_argument("--kernel", default="auto")
ap.add_argument("--subdiv", type=int, default=10)
ap.add_argument("--builder", default="host", choices=["host", "device"])
a = ap.parse_args()
aspect = 16 / 9
if a.scene == "suzanne":
    scene = rtow.HostScene.obj(ROOT / "tests/golden/suzanne.obj", aspect); W, spp, depth = 1920, 256, 20
elif a.scene == "mesh100k":
    tmp = Path(tempfile.gettempdir()) / f"suz{a.subdiv}.obj"
    subprocess.run([sys.executable, str(ROOT / "scripts/make_mesh.py"), str(tmp), str(a.subdiv)], check=True, capture_output=True)
    scene = rtow.HostScene.obj(tmp, aspect); W, spp, depth = 1920, 1024, 20
else:
    aspect = 1.5
    scene = rtow.HostScene.cover(11, aspect, a.scene == "moving"); W, spp, depth = 1200, 100, 50
W = a.width or W; spp = a.spp or spp; depth = a.depth or depth
H = rtow.image_height(W, aspect)
cfg = rtow.make_config(W, H, spp, max(1, spp // 8 if a.scene in ("suzanne", "mesh100k") else spp // 10), depth, seed=1,
                       precision={"fast": rtow.F64_FAST, "strict": rtow.F64_STRICT, "f32": rtow.F32}[a.precision],
                       kernel={"auto": 0, "brute": 1, "bvh": 2, "grid": 3}[a.kernel])
ctx = rtow.Context(0)
ctx.set_builder(rtow.BUILDER_DEVICE_LBVH if a.builder == "device" else rtow.BUILDER_HOST_SAH)
ctx.upload(scene)  # first upload pays one-time costs (module load, allocations)
t0 = time.perf_counter(); ctx.upload(scene); t_up = time.perf_counter() - t0
bi = ctx.build_info()
out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
st = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
ms = []
for _ in range(a.steps):
    s2 = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
    ms.append(s2.kernel_ms)
best = min(ms)
print(json.dumps({"scene": a.scene, "prims": scene.c.n_prims, "W": W, "H": H, "spp": rtow.spp_effective(cfg), "depth": depth,
                  "kernel": st.kernel_used, "builder": a.builder, "bvh_nodes": bi.bvh_nodes, "bvh_build_ms": round(bi.bvh_build_ms, 3),
                  "grid_build_ms": round(bi.grid_build_ms, 3), "upload_s": round(t_up, 4), "kernel_ms": round(best, 3),
                  "Msamples_per_s": round(st.samples / best / 1e3, 1), "segments_per_sample": round(st.segments / st.samples, 3),
                  "node_tests_per_segment": round(st.node_tests / max(st.segments, 1), 2),
                  "prim_tests_per_segment": round(st.prim_tests / max(st.segments, 1), 2),
                  "mean_rgb": [round(float(x), 4) for x in (out.mean(dim=(0, 1)) / rtow.spp_effective(cfg)).tolist()]}))
