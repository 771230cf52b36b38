"""Kernel time vs samples per pixel at a fixed samples-per-item (env SPI): fits the per-launch
fixed cost (end-of-launch tail) and the steady-state rate.  DESIGN.md, End-of-launch tail."""
import sys, os
sys.path.insert(0, "raytracing-one-weekend_amd")
import rtow, torch
scene = rtow.HostScene.cover(11, 1.5, False)
ctx = rtow.Context(0); ctx.upload(scene)
out = torch.zeros((800, 1200, 3), dtype=torch.float64, device="cuda")
spi = int(os.environ.get("SPI", "10"))
for spp in [10, 20, 50, 100, 200, 400, 800]:
    ns = max(1, spp // spi)
    cfg = rtow.make_config(1200, 800, spp, ns, 50, seed=1, precision=rtow.F64_FAST)
    ms = []
    for _ in range(3):
        st = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
        ms.append(st.kernel_ms)
    print(f"spp {spp} nstreams {ns} samples/item {spp//ns}: {min(ms):.3f} ms  {st.samples/min(ms)/1e3:.0f} Msamples/s  segs/sample {st.segments/st.samples:.3f}")
