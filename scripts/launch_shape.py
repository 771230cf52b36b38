#!/usr/bin/env python3
"""Does the launch shape matter for an image that fits LDS twice?  Cover scene with `--n` balls per side (n = 10: a
73 KB grid image, which AUTO runs as two 512-lane workgroups per CU), kernel ms at 100 and 500 spp for the settings
given as arguments ("" = defaults, "RTOW_BVH_BLOCK=1024" = one 1024-lane workgroup per CU, the cover scene's shape).
   python scripts/launch_shape.py --n 10 "" "RTOW_BVH_BLOCK=1024" """
import argparse, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("settings", nargs="*", default=[""])
a = ap.parse_args()
import torch
import rtow
W, aspect, depth = 1200, 1.5, 50
H = rtow.image_height(W, aspect)
scene = rtow.HostScene.cover(a.n, aspect, False)
out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for r in range(a.rounds):
    for setting in a.settings:
        saved = dict(os.environ)
        for kv in setting.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
        ctx = rtow.Context(0)  # knobs are read at context creation
        ctx.upload(scene)
        res = {"setting": setting, "prims": scene.c.n_prims, "grid_image_bytes": ctx.build_info().grid_image_bytes}
        for spp in (100, 500):
            cfg = rtow.make_config(W, H, spp, spp // 10, depth, seed=1, precision=rtow.F64_FAST)
            ctx.render_device(cfg, out.data_ptr(), st, True)
            ms = min(ctx.render_device(cfg, out.data_ptr(), st, True).kernel_ms for _ in range(3))
            res[f"spp{spp}_ms"] = round(ms, 4)
            res[f"spp{spp}_Gsps"] = round(W * H * spp / ms / 1e6, 3)
        ctx.close()
        os.environ.clear(); os.environ.update(saved)
        print(json.dumps(res), flush=True)
