#!/usr/bin/env python3
"""Fast build against strict build of the cover scene moved away from the origin (advisor, round 4: the k-form of the
GRID walk's sphere test cancels at the size of |o|^2 when written in world coordinates).  Prints, per offset:
mean |fast - strict| per sample, share of pixels equal to 1e-6 relative, segment counts.
  RTOW_LIB=raytracing-one-weekend_amd/variants/base.so python scripts/far_origin_check.py   (another build)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import rtow

def moved(offset, moving=False):
    scene = rtow.HostScene.cover(11, 1.5, moving)
    sc = scene.c
    for i in range(sc.n_spheres):
        for k in range(3):
            sc.sphere_geom[4 * i + k] += offset[k]
    for k in range(3):
        sc.camera.origin[k] += offset[k]
        sc.camera.lower_left_corner[k] += offset[k]
    return scene

ctx = rtow.Context(0)
w, h, spp = 480, 320, 16
for off in [(0, 0, 0), (1e3, 1e3, 1e3), (1e4, 1e4, 1e4), (1e5, -1e5, 1e5), (1e6, 1e6, 1e6)]:
    scene = moved([float(x) for x in off])
    s, st = ctx.render(scene, rtow.make_config(w, h, spp, 2, 50, seed=9, precision=rtow.F64_STRICT))
    f, sf = ctx.render(scene, rtow.make_config(w, h, spp, 2, 50, seed=9, precision=rtow.F64_FAST))
    d = np.abs(f - s)
    close6 = np.isclose(f, s, rtol=1e-6, atol=1e-9).all(axis=-1).mean()
    close9 = np.isclose(f, s, rtol=1e-9, atol=1e-12).all(axis=-1).mean()
    print(f"offset {off[0]:>9.0e}: mean|d|/spp {d.mean() / spp:.3e}  median|d| {np.median(d):.3e}  pixels equal to 1e-6 {close6:.4f}  to 1e-9 {close9:.4f}  "
          f"segments strict {st.segments} fast {sf.segments}  kernel {sf.kernel_used}")
