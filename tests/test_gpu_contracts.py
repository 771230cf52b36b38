"""Contracts of the boundary that are about WHEN and WHETHER, not about pixel values (round 4):
  * samples given up at the trace kernel's end-of-launch bound are an error on every entry point, with or
    without `stats` (include/rtow.h, rtow_render_device);
  * a render on a hipStreamNonBlocking stream straight after rtow_scene_upload sees the whole scene (the upload's
    copies are queued on the null stream without a host wait);
  * a sample count with no divisor near the item length (101, 127: primes) is cut into bounded levels, the image
    stays the fast build's image.
"""
import numpy as np
import pytest

import orc
import rtow

pytestmark = pytest.mark.gpu


def test_dropped_samples_are_an_error_with_and_without_stats(monkeypatch):
    """RTOW_TAIL_BOUND (read at rtow_ctx_create; tests only) shrinks the structural trip bound of the end-of-launch
    protocol to a handful, so waves give up samples: every synchronising entry point must return RTOW_EHIP — never
    RTOW_OK with a darker image — and the context must work again afterwards (the bound is a per-context knob)."""
    import torch

    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(240, 160, 20, 2, 50, seed=3, precision=rtow.F64_FAST)
    monkeypatch.setenv("RTOW_TAIL_BOUND", "2")
    bad = rtow.Context(0)
    monkeypatch.delenv("RTOW_TAIL_BOUND")
    good = rtow.Context(0)
    try:
        want, _ = good.render(scene, cfg)
        for call in (lambda: bad.render(scene, cfg),                       # rtow_render, stats
                     lambda: bad.render_rgb8(scene, cfg, want_stats=True),
                     lambda: bad.render_rgb8(scene, cfg, want_stats=False)):  # stats == NULL: still an error
            with pytest.raises(rtow.RtowError, match="end-of-launch bound"):
                call()
        # rtow_render with stats == NULL
        out = np.zeros((160, 240, 3))
        L = rtow.lib()
        import ctypes as C
        rc = L.rtow_render(bad._h, C.byref(scene.c), C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_double)), None)
        assert rc == rtow.RTOW_EHIP and b"end-of-launch bound" in L.rtow_last_error()
        # the asynchronous entry point reports at rtow_profile_collect
        bad.upload(scene)
        buf = torch.zeros((160, 240, 3), dtype=torch.float64, device="cuda:0")
        bad.render_device(cfg, buf.data_ptr(), 0, False)
        with pytest.raises(rtow.RtowError, match="end-of-launch bound"):
            bad.profile_collect()
        bad.profile_collect()  # the word was cleared by the report: nothing pending
        # strict build: the same contract
        scfg = rtow.make_config(120, 80, 8, 2, 50, seed=3, precision=rtow.F64_STRICT)
        with pytest.raises(rtow.RtowError, match="end-of-launch bound"):
            bad.render(scene, scfg)
        # the multi-device handle
        monkeypatch.setenv("RTOW_TAIL_BOUND", "2")
        m = rtow.MultiContext([0, 0], use_rccl=False)
        monkeypatch.delenv("RTOW_TAIL_BOUND")
        try:
            m.upload(scene)
            for call in (lambda: m.render(cfg, want_stats=False), lambda: m.render_rgb8(cfg, want_stats=False)):
                with pytest.raises(rtow.RtowError, match="end-of-launch bound"):
                    call()
        finally:
            m.close()
        again, _ = good.render(scene, cfg)
        assert np.array_equal(again, want)
    finally:
        bad.close()
        good.close()


@pytest.mark.parametrize("kind", ["cover", "suzanne"])
def test_render_on_a_nonblocking_stream_right_after_upload(kind):
    """rtow_scene_upload queues its copies (and, for the device builder, its kernels) on the null stream and returns;
    a torch side stream is hipStreamNonBlocking, i.e. NOT ordered behind the null stream by the runtime.  The library
    orders it (an event behind the upload): the strict image equals the oracle's, every time, also with the device
    builder and when uploads and renders alternate."""
    import torch
    from conftest import GOLDEN

    if kind == "cover":
        scenes = [rtow.HostScene.cover(11, 1.5, False), rtow.HostScene.cover(11, 1.5, True)]
        cfg = rtow.make_config(120, 80, 6, 2, 50, seed=11, precision=rtow.F64_STRICT)
    else:
        scenes = [rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), rtow.HostScene.cover(11, 1.5, False)]
        cfg = rtow.make_config(96, 54, 4, 2, 20, seed=11, precision=rtow.F64_STRICT)
    refs = [orc.render(s, cfg, orc.RNG_PHILOX, nthreads=4)[0] for s in scenes]
    side = torch.cuda.Stream(device="cuda:0")
    buf = torch.zeros((cfg.image_height, cfg.image_width, 3), dtype=torch.float64, device="cuda:0")
    for builder in (rtow.BUILDER_HOST_SAH, rtow.BUILDER_DEVICE_LBVH):
        c = rtow.Context(0)
        try:
            c.set_builder(builder)
            for rep in range(3):
                for s, ref in zip(scenes, refs):
                    c.upload(s)  # no wait in between: the render must find the scene complete
                    c.render_device(cfg, buf.data_ptr(), side.cuda_stream, False)
                    side.synchronize()
                    got = buf.cpu().numpy()
                    assert np.array_equal(got, ref), (builder, rep, int((got != ref).sum()))
        finally:
            c.close()


@pytest.mark.parametrize("spp", [101, 127, 23])
def test_prime_sample_counts_get_bounded_levels_and_the_same_picture(ctx, spp):
    """level_plan's ragged schedule (no divisor of the sample count within a factor of two of the item length):
    levels of 10 with the remainder on the last one — not `spp` levels of one sample.  All samples are traced
    (stats), the picture is the strict build's within the fast build's tolerance, repeatable, and the partial-sum
    workspace is the bounded one (a second call with a small RTOW_PARTIALS cap is not needed: count the levels)."""
    import ctypes as C

    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(120, 80, spp, 1, 50, seed=9, precision=rtow.F64_FAST)
    ctx.upload(scene)  # (the item length aimed at depends on the resident scene: 10 for spheres, 16 for a mesh)
    pairs = (C.c_uint32 * 512)()
    n = rtow.lib().rtow_debug_schedule(ctx._h, C.byref(cfg), pairs, 256)
    sched = [(pairs[2 * i], pairs[2 * i + 1]) for i in range(n)]
    assert n == max(spp // 10, 1) and sum(c for _, c in sched) == spp and all(c == 10 for _, c in sched[:-1])
    img, st = ctx.render(scene, cfg)
    assert st.samples == 120 * 80 * spp
    strict, sst = ctx.render(scene, rtow.make_config(120, 80, spp, 1, 50, seed=9, precision=rtow.F64_STRICT))
    assert sst.samples == st.samples
    assert np.abs(img - strict).mean() / spp <= 2e-4
    again, _ = ctx.render(scene, cfg)
    assert np.array_equal(again, img)
    # split in two stream ranges (accumulated): every sample still traced exactly once
    if spp % 2 == 0 or True:
        c2 = rtow.make_config(120, 80, spp - spp % 2, 2, 50, seed=9, precision=rtow.F64_FAST)
        whole, wst = ctx.render(scene, c2)
        a = rtow.make_config(120, 80, spp - spp % 2, 2, 50, seed=9, precision=rtow.F64_FAST, stream_first=0, stream_count=1)
        b = rtow.make_config(120, 80, spp - spp % 2, 2, 50, seed=9, precision=rtow.F64_FAST, stream_first=1, stream_count=1,
                             accumulate=1)
        part, pst = ctx.render(scene, a)
        part, qst = ctx.render(scene, b, into=part)
        assert pst.samples + qst.samples == wst.samples
        assert np.abs(part - whole).mean() / spp <= 2e-4
