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
