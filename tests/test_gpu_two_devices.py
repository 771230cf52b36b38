"""The N > 1 paths over RCCL on REAL distinct devices.  The GPU boxes this project is built on show one
device, so these tests enable themselves: with fewer than two visible devices they skip (and say so);
on the first machine with two or more they check, bit for bit against the one-device frame,
  * `python bench.py --gpus 2 --backend nccl` — one process per GPU, torch.distributed gather = RCCL;
  * the persistent single-process handle (rtow_multi_*: ncclCommInitAll + one ncclGather per frame), several
    frames on one handle, and the one-shot rtow_render_multi;
  * `rtweekend --gpus 2`.
Nothing here needs to be asked for: the driver's `pytest -m gpu` on a multi-GPU node runs them."""
import json
import subprocess
import sys

import numpy as np
import pytest

import rtow
from conftest import REPO

pytestmark = pytest.mark.gpu


def _n_devices():
    import torch

    return torch.cuda.device_count()  # (does not initialise the GPU on this image)


needs_two = pytest.mark.skipif(_n_devices() < 2, reason="RCCL with N > 1 needs two visible HIP devices (this box has fewer)")


def test_persistent_handle_on_one_device_many_frames(ctx):
    """Always runs: the handle with a one-rank communicator renders several frames (different configs, a new
    scene in between) without re-creating anything, each equal to the one-context frame."""
    scene = rtow.HostScene.cover(11, 1.5, True)
    m = rtow.MultiContext([0], use_rccl=True)
    try:
        m.upload(scene)
        assert m.build_info().grid_image_bytes > 0
        for (w, h, spp, ns, seed) in ((150, 100, 12, 3, 9), (96, 64, 8, 2, 3), (150, 100, 12, 3, 9)):
            cfg = rtow.make_config(w, h, spp, ns, 30, seed=seed, precision=rtow.F64_STRICT, tile_rows=8)
            whole, st = ctx.render(scene, cfg)
            img, s = m.render(cfg)
            assert np.array_equal(img, whole) and s.segments == st.segments and s.samples == st.samples
            img2, none = m.render(cfg, want_stats=False)  # asynchronous launches, one wait at the end
            assert none is None and np.array_equal(img2, whole)
        mesh = rtow.HostScene.obj(REPO / "tests" / "golden" / "suzanne.obj", 16 / 9)
        m.upload(mesh)
        cfg = rtow.make_config(64, 36, 4, 2, 20, seed=5, precision=rtow.F64_STRICT)
        assert np.array_equal(m.render(cfg)[0], ctx.render(mesh, cfg)[0])
    finally:
        m.close()
    three = rtow.MultiContext([0, 0, 0], use_rccl=False)  # the partition without the collective
    try:
        three.upload(scene)
        cfg = rtow.make_config(150, 100, 12, 3, 30, seed=9, precision=rtow.F64_STRICT, tile_rows=8)
        assert np.array_equal(three.render(cfg)[0], ctx.render(scene, cfg)[0])
    finally:
        three.close()
    with pytest.raises(rtow.RtowError):
        rtow.MultiContext([0, 0], use_rccl=True)  # RCCL refuses one device twice: an error, not a wrong image


@pytest.mark.parametrize("devices,use_rccl", [([0], True), ([0, 0, 0], False)])
def test_multi_handle_rgb8_bytes_equal_the_one_device_bytes(ctx, devices, use_rccl):
    """rtow_multi_render_rgb8 (write_color by the rank that owns the pixel, bytes gathered, rows placed by a kernel
    on the first device, one copy into the caller's buffer) against rtow_render_rgb8 of one context: the same bytes,
    for a one-rank RCCL communicator and for three ranks without the collective; strip heights that tile evenly,
    unevenly (ranks with fewer rows: zero-padded strip buffers) and rows whose byte count is not a multiple of 16."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    m = rtow.MultiContext(devices, use_rccl=use_rccl)
    try:
        m.upload(scene)
        for (w, h, spp, ns, tr, prec) in ((150, 100, 12, 3, 8, rtow.F64_STRICT), (96, 64, 8, 2, 4, rtow.F64_FAST),
                                          (150, 100, 20, 1, 16, rtow.F64_FAST), (67, 45, 6, 2, 2, rtow.F64_STRICT)):
            cfg = rtow.make_config(w, h, spp, ns, 30, seed=7, precision=prec, tile_rows=tr)
            one8, st = ctx.render_rgb8(scene, cfg)
            got8, s = m.render_rgb8(cfg)
            assert got8.shape == one8.shape and np.array_equal(got8, one8), int((got8 != one8).sum())
            assert s.samples == st.samples and s.segments == st.segments
            again, none = m.render_rgb8(cfg, want_stats=False)
            assert none is None and np.array_equal(again, one8)
            sums, _ = m.render(cfg)  # the f64 form goes through the same placement kernel
            assert np.array_equal(sums, ctx.render(scene, cfg)[0])
            b = m.frame_breakdown()  # where that frame's time went: every stage non-negative, the parts within the total
            assert set(b) == set(rtow.MULTI_BREAKDOWN) and all(v >= 0 for v in b.values()), b
            assert b["handoff_enqueue"] + b["place_enqueue"] + b["wait_and_copy"] <= b["total"] + 1e-6
            assert b["dev_trace"] > 0 and b["dev_trace"] <= b["total"]
    finally:
        m.close()


def test_failed_gather_enqueue_is_an_error_code_and_the_handle_refuses_further_frames(ctx, monkeypatch):
    """A rank whose side of the gather cannot be enqueued (forced with RTOW_MULTI_FAIL_GATHER, read at create):
    every communicator is aborted before anything is waited for, the call returns RTOW_EHIP, later frames on the
    handle are refused — never a hang."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    monkeypatch.setenv("RTOW_MULTI_FAIL_GATHER", "0")
    m = rtow.MultiContext([0], use_rccl=True)
    monkeypatch.delenv("RTOW_MULTI_FAIL_GATHER")
    try:
        m.upload(scene)
        cfg = rtow.make_config(96, 64, 8, 2, 30, seed=7, precision=rtow.F64_FAST)
        with pytest.raises(rtow.RtowError, match="ncclGather"):
            m.render(cfg)
        with pytest.raises(rtow.RtowError, match="aborted"):
            m.render_rgb8(cfg)
    finally:
        m.close()
    ok = rtow.MultiContext([0], use_rccl=True)  # a new handle works
    try:
        ok.upload(scene)
        assert np.array_equal(ok.render(cfg)[0], ctx.render(scene, cfg)[0])
    finally:
        ok.close()


@needs_two
def test_gather_that_only_one_of_two_ranks_enqueued_returns_an_error_and_close_returns(monkeypatch):
    """RTOW_MULTI_FAIL_GATHER=1 on two devices: rank 0 enqueues its side of the collective, rank 1 does not — the
    half-issued case the abort exists for.  The frame returns RTOW_EHIP without waiting for the stream that can never
    finish, the handle refuses further frames, and close() (which waits for the streams after the abort) returns."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    monkeypatch.setenv("RTOW_MULTI_FAIL_GATHER", "1")
    m = rtow.MultiContext([0, 1], use_rccl=True)
    monkeypatch.delenv("RTOW_MULTI_FAIL_GATHER")
    try:
        m.upload(scene)
        cfg = rtow.make_config(96, 64, 8, 2, 30, seed=7, precision=rtow.F64_FAST)
        with pytest.raises(rtow.RtowError, match="ncclGather"):
            m.render_rgb8(cfg)
        with pytest.raises(rtow.RtowError, match="aborted"):
            m.render(cfg)
    finally:
        m.close()


@needs_two
def test_rccl_gather_over_two_devices_single_process(ctx):
    scene = rtow.HostScene.cover(11, 1.5, False)
    n = min(_n_devices(), 4)
    m = rtow.MultiContext(list(range(n)), use_rccl=True)
    try:
        m.upload(scene)
        for prec in (rtow.F64_STRICT, rtow.F64_FAST):
            cfg = rtow.make_config(600, 400, 20, 4, 50, seed=1, precision=prec, tile_rows=8)
            whole, st = ctx.render(scene, cfg)
            for _ in range(3):
                img, s = m.render(cfg)
                assert np.array_equal(img, whole), int((img != whole).sum())
                assert s.segments == st.segments
            one8, _ = ctx.render_rgb8(scene, cfg)
            got8, _ = m.render_rgb8(cfg)  # bytes over RCCL (ncclUint8), placed on device 0
            assert np.array_equal(got8, one8), int((got8 != one8).sum())
    finally:
        m.close()
    one_shot, _ = rtow.render_multi([0, 1], scene, rtow.make_config(150, 100, 12, 3, 30, seed=9, precision=rtow.F64_STRICT))
    assert np.array_equal(one_shot, ctx.render(scene, rtow.make_config(150, 100, 12, 3, 30, seed=9,
                                                                      precision=rtow.F64_STRICT))[0])


@needs_two
def test_bench_nccl_two_ranks_equals_one_rank_frame(ctx, tmp_path):
    out = tmp_path / "frame.npy"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--backend", "nccl", "--spp", "100",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--dump-image", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and "REHEARSAL" not in line["data"]
    got = np.load(out)
    scene = rtow.HostScene.cover(11, 1.5, False)
    whole, _ = ctx.render(scene, rtow.make_config(1200, 800, 100, 10, 50, seed=1, precision=rtow.F64_FAST))
    assert np.array_equal(got, whole), int((got != whole).sum())


@needs_two
def test_rtweekend_two_gpus_prints_the_one_gpu_image():
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    base = [str(exe), "-w", "96", "-a", "1.5", "-s", "8", "-c", "20", "-t", "2", "--precision", "strict"]
    plain = subprocess.run(base, capture_output=True, check=True)
    two = subprocess.run(base + ["--gpus", "2"], capture_output=True, check=True)
    assert two.stdout == plain.stdout
    p6 = subprocess.run(base + ["--p6"], capture_output=True, check=True)
    two6 = subprocess.run(base + ["--gpus", "2", "--p6"], capture_output=True, check=True)
    assert two6.stdout == p6.stdout and two6.stdout.startswith(b"P6")
