import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
import rtow
out = sys.argv[1]
scene = rtow.HostScene.cover(0, 16 / 9, True)
cfg = rtow.make_config(64, 36, 8, 2, 10, seed=7, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
ctx = rtow.Context(0)
img, st = ctx.render(scene, cfg)
np.save(out, img)
print(out, "segments", st.segments, "samples", st.samples, "kernel_ms", st.kernel_ms)

import ctypes as C
buf = (C.c_ulonglong * 48)()
rtow.lib().rtow_debug_counters(ctx._h, buf)
print("take", buf[9], "release", buf[10], "gives", buf[11], "helps", buf[12], "hold", buf[13], "guard lanes", buf[14], "guard owners", buf[15])
