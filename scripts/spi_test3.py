import sys, os
sys.path.insert(0, "raytracing-one-weekend_amd")
import rtow, torch
scene = rtow.HostScene.cover(11, 1.5, False)
ctx = rtow.Context(0); ctx.upload(scene)
for (W, H, spp) in [(300, 200, 1600), (600, 400, 400), (1200, 800, 100), (2400, 1600, 25), (4800, 3200, 6), (300, 200, 160), (600, 400, 40), (1200, 800, 10), (2400, 1600, 3)]:
    out = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    ns = max(1, spp // 10)
    cfg = rtow.make_config(W, H, spp, ns, 50, seed=1, precision=rtow.F64_FAST)
    ms = []
    for _ in range(3):
        st = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
        ms.append(st.kernel_ms)
    print(f"{W}x{H} spp {spp} ns {ns} samples {st.samples/1e6:.1f}M: {min(ms):.3f} ms  {st.samples/min(ms)/1e3:.0f} Msamples/s")
