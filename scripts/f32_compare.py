import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import rtow
ctx = rtow.Context(0)
def cmp(name, scene, w, h, spp, depth, kernel):
    out = {}
    for prec in (rtow.F64_FAST, rtow.F32):
        cfg = rtow.make_config(w, h, spp, max(1, spp // 4), depth, seed=5, precision=prec, kernel=kernel)
        img, st = ctx.render(scene, cfg)
        img2, st2 = ctx.render(scene, cfg)
        out[prec] = (img / spp, st2)
    a, sa = out[rtow.F64_FAST]; b, sb = out[rtow.F32]
    d = np.abs(np.sqrt(np.clip(a,0,1)) - np.sqrt(np.clip(b,0,1)))
    print(f"{name:14s} kernel {sa.kernel_used} f64 {sa.kernel_ms:8.3f} ms  f32 {sb.kernel_ms:8.3f} ms  x{sa.kernel_ms/sb.kernel_ms:.2f}  "
          f"mean|d sqrt| per channel {d.mean(axis=(0,1))}  means f64 {a.mean(axis=(0,1))} f32 {b.mean(axis=(0,1))} finite {np.isfinite(b).all()} seg/smp {sa.segments/sa.samples:.3f} {sb.segments/sb.samples:.3f}")
cover = rtow.HostScene.cover(11, 1.5, False)
cmp("cover grid", cover, 1200, 800, 100, 50, rtow.KERNEL_AUTO)
cmp("cover bvh", cover, 600, 400, 64, 50, rtow.KERNEL_BVH)
mov = rtow.HostScene.cover(11, 1.5, True)
cmp("moving grid", mov, 600, 400, 64, 50, rtow.KERNEL_AUTO)
c1 = rtow.HostScene.cover(0, 16/9, True)
cmp("c1 stream", c1, 400, 225, 64, 10, rtow.KERNEL_AUTO)
suz = rtow.HostScene.obj(ROOT / "tests/golden/suzanne.obj", 16/9)
cmp("suzanne bvh", suz, 960, 540, 64, 20, rtow.KERNEL_AUTO)
cmp("suzanne grid", suz, 480, 270, 32, 20, rtow.KERNEL_GRID)
