#!/usr/bin/env python3
"""Host vs device build of the grid image: build time and byte equality, on the cover scene and on
N random small spheres over a ground sphere (python scripts/grid_build_compare.py [N])."""
import ctypes as C, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import rtow

def sphere_scene(n, keep):
    rng = np.random.default_rng(7)
    geom = np.zeros((n + 1, 4))
    geom[0] = [0, -1000, 0, 1000]
    side = max(4.0, (n / 2.0) ** 0.5)
    geom[1:, 0] = rng.uniform(-side, side, n); geom[1:, 2] = rng.uniform(-side, side, n)
    geom[1:, 1] = rng.uniform(0.1, 0.5, n); geom[1:, 3] = rng.uniform(0.05, 0.2, n)
    base = rtow.HostScene.cover(0, 1.5, False)
    mats = (rtow.Material * 1)(); mats[0].kind = rtow.MAT_LAMBERTIAN; mats[0].albedo = (C.c_double * 3)(.5, .5, .5); mats[0].ir = 1.5
    g = np.ascontiguousarray(geom); mi = np.zeros(n + 1, dtype=np.int32); idx = np.arange(n + 1, dtype=np.int32)
    sc = rtow.Scene(); sc.camera = base.c.camera; sc.n_spheres = n + 1
    sc.sphere_geom = g.ctypes.data_as(C.POINTER(C.c_double)); sc.sphere_mat = mi.ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = 1; sc.materials = mats; sc.n_prims = n + 1
    sc.prim_kind = mi.ctypes.data_as(C.POINTER(C.c_int32)); sc.prim_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    keep.extend([g, mi, idx, mats, base]); return sc

keep = []
scenes = {"cover (486 spheres)": rtow.HostScene.cover(11, 1.5, False)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
scenes[f"{n} random spheres"] = sphere_scene(n, keep)
for name, sc in scenes.items():
    row = {"scene": name}
    imgs = {}
    for b, tag in ((rtow.BUILDER_HOST_SAH, "host"), (rtow.BUILDER_DEVICE_LBVH, "device")):
        ctx = rtow.Context(0); ctx.set_builder(b)
        ctx.upload(sc); ctx.upload(sc)  # second upload: buffers and code objects warm
        bi = ctx.build_info()
        row[tag] = {"grid_build_ms": round(bi.grid_build_ms, 3), "bvh_build_ms": round(bi.bvh_build_ms, 3),
                    "upload_ms": round(bi.upload_ms, 3), "grid_image_bytes": bi.grid_image_bytes}
        imgs[tag] = ctx.debug_image(1); ctx.close()
    row["grid_images_identical"] = imgs["host"] == imgs["device"]
    print(json.dumps(row))
