#!/usr/bin/env python3
"""Region shares of the BVH trace kernel (diagnostic RTOW_STAMPS build)."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "raytracing-one-weekend_amd"))
os.environ["RTOW_STAMPS"] = "1"
import rtow
which = sys.argv[1] if len(sys.argv) > 1 else "cover"
if which == "suzanne":
    scene = rtow.HostScene.obj(ROOT / "tests/golden/suzanne.obj")
    cfg = rtow.make_config(1920, 1080, 16, 2, 20, seed=1, precision=rtow.F64_FAST, kernel={'auto': 0, 'bvh': 2, 'grid': 3, 'bvh4': 4}[os.environ.get('RTOW_STAMP_KERNEL', 'auto')])
else:
    scene = rtow.HostScene.cover(11, 1.5, which == "moving")
    cfg = rtow.make_config(1200, 800, 100, int(os.environ.get('RTOW_NSTREAMS', '10')), 50, seed=1, precision=rtow.F64_FAST, kernel={'auto': 0, 'bvh': 2, 'grid': 3, 'bvh4': 4}[os.environ.get('RTOW_STAMP_KERNEL', 'auto')])
ctx = rtow.Context(0)
img, st = ctx.render(scene, cfg)
out = (C.c_ulonglong * 48)()
L = rtow.lib(); L.rtow_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
rtow.check(L.rtow_debug_counters(ctx._h, out))
names = ["fetch", "regen", "walk-steps", "shade", "walk-leaves"]
tot = (sum(out[8:13]) + sum(out[29:34])) or 1
print(which, "kernel_ms(with stamps)", round(st.kernel_ms, 3), "Msamples/s", round(st.samples / st.kernel_ms / 1e3, 1),
      "segments", st.segments, "node/seg", round(st.node_tests / st.segments, 2), "prim/seg", round(st.prim_tests / st.segments, 2))
for i, n in enumerate(names):
    print(f"{n:12s} {out[8+i]/tot*100:5.1f}%")
ttot = sum(out[29:34]) or 1
print("after a wave found the queue empty: shader-clock cycles per trip by region (s_memtime), and the same in the whole launch:")
for i, n in enumerate(names):
    print(f"  {n:12s} {out[29+i]/max(out[37],1):9.0f} cycles/trip ({out[29+i]/ttot*100:5.1f}%)   whole launch {out[8+i]/max(trips,1):9.0f} cycles/trip" if False else "", end="")
trips_all = out[14] or 1
for i, n in enumerate(names):
    print(f"  {n:12s} tail {out[29+i]/max(out[37],1):9.0f} cycles/trip ({out[29+i]/ttot*100:5.1f}%)   launch {out[8+i]/trips_all:9.0f} cycles/trip")
print(f"  trips after queue-empty, all waves: {int(out[37])}; wall time after queue-empty per such trip: {out[38]/max(out[37],1)/100:.2f} us")
print("lanes a region worked for (of 64; entry masks, so an upper bound on what its instructions saw):")
for i, n in enumerate(names):
    if out[8 + i]:
        print(f"{n:12s} {out[23+i]/out[8+i]:5.1f}")
iters, trips, phases = out[13], out[14], out[15]
print(f"lanes stepping per step-loop iteration {out[34]/max(iters,1):.1f}; lanes testing per leaf phase {out[35]/max(phases,1):.1f}")
print(f"wave trips {trips}, step-loop iterations per trip {iters/max(trips,1):.1f}, leaf phases per trip {phases/max(trips,1):.2f}, "
      f"lane-segments per trip {st.segments/max(trips,1):.1f} of 64")
if out[45] or out[46]:
    print(f"GRID: step-loop iterations that only camera rays needed {out[45]/max(iters,1)*100:.1f} % of all; leaf phases only camera rays "
          f"needed {out[46]/max(phases,1)*100:.1f} % (the most a separate treatment of primary rays could take out of the walk)")
print(f"Philox block evaluations per trip (wave level, new-ray stage): {out[44]/max(trips,1):.2f} (one per request since the direct samplers of round 5; "
      f"with the rejection loops of rounds 1-5a: 3.35, of which 2.35 for 7.4 lanes each, profiles/r05_stamps.log)")
nw = 4096.0
print(f"wave end times (ms after first wave start): mean {out[4]/nw/1e5:.3f}  min {out[5]/1e5:.3f}  max {out[6]/1e5:.3f};  queue seen empty (mean over waves) {out[7]/nw/1e5:.3f}")

print("waves by (end - own queue-empty time), 0.2 ms bins [0-0.2, .., 0.8-1.0, >=1.0]:", [int(out[17 + i]) for i in range(6)])
print(f"max trips after queue-empty {int(out[40])}; stragglers (>= 0.4 ms): {int(out[43])} waves, mean trips {out[41]/max(out[43],1):.1f}, mean time {out[42]/max(out[43],1)/1e5:.3f} ms -> {out[42]/max(out[41],1)/100:.2f} us per trip")
