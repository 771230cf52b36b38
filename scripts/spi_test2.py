import sys, os
sys.path.insert(0, "raytracing-one-weekend_amd")
import rtow, torch
scene = rtow.HostScene.cover(11, 1.5, False)
ctx = rtow.Context(0); ctx.upload(scene)
out = torch.zeros((800, 1200, 3), dtype=torch.float64, device="cuda")
for spp, ns in [(1, 1), (2, 1), (4, 1), (8, 1), (8, 2), (16, 2), (16, 4), (32, 4)]:
    cfg = rtow.make_config(1200, 800, spp, ns, 50, seed=1, precision=rtow.F64_FAST)
    ms = []
    for _ in range(3):
        st = ctx.render_device(cfg, out.data_ptr(), torch.cuda.current_stream().cuda_stream, True)
        ms.append(st.kernel_ms)
    print(f"spp {spp} nstreams {ns}: {min(ms):.3f} ms  {st.samples/min(ms)/1e3:.0f} Msamples/s total_ms {st.total_ms:.3f}")
