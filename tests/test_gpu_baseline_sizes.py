"""GPU parity AT THE BENCHMARKED SIZES AND BUILD (BASELINE.json configs[1..4]).

The frame is rendered at the config's real W, H, spp and stream count; full-width strips of it —
top, horizon / silhouette, bottom — are selected through rank / nranks / tile_rows (one strip per
"rank": nranks = H / tile_rows), traced on the device and by the oracle with the same settings:

  strict build  ⇒ np.array_equal (bit-identical f64 sums, equal segment counts);
  fast build    ⇒ the bench's build.  FMA contraction and Newton-refined rcp/rsq flip rare hit/miss
                  decisions, so: mean |Δ| per channel and sample ≤ 1e-4 (linear radiance) and at most
                  0.1 % of the displayed 8-bit channels off by more than one level.

The oracle finds closest hits through its checker tree here (orc.render(..., accel=True)): the
reference's own tree needs ~15,000 tests per segment on the 96,800-triangle mesh.  The checker tree
gives the reference-tree image bit for bit (tests/test_oracle_units.py, and re-checked below on a
low-spp strip of every config).  configs[2] (500 spp over 8 GPUs + gather) is covered by the 500-spp
strips here plus tests/test_gpu_multi_rank_bench.py (bench.py's own N-rank path, kernel behind each rank).
"""
import subprocess
import sys

import numpy as np
import pytest

import orc
import rtow
from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def to8(img, spp):
    return (256 * np.clip(np.sqrt(img / spp), 0.0, 0.999)).astype(np.int32)


def check_fast(img, ref, spp, what):
    assert np.isfinite(img).all(), what
    mad = np.abs(img - ref).mean() / spp
    off = (np.abs(to8(img, spp) - to8(ref, spp)) > 1).mean()
    assert mad <= 1e-4, (what, mad)
    assert off <= 1e-3, (what, off)
    return mad, off


@pytest.fixture(scope="module")
def mesh_obj(tmp_path_factory):
    obj = tmp_path_factory.mktemp("mesh") / "mesh100k.obj"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(obj), "10"], check=True,
                   capture_output=True)
    return obj


# config: (W, H, spp, nstreams, bounces, strip height, strips to check)
C2 = (1200, 800, 100, 10, 50, 8, (0, 30, 47, 99))       # top (sky), horizon, big spheres, bottom rows
C3 = (1200, 800, 500, 50, 50, 8, (45,))                 # configs[2]'s per-pixel work, one strip
C4 = (1920, 1080, 256, 16, 20, 8, (4, 67, 130))          # suzanne fills the frame
C5 = (1920, 1080, 1024, 64, 20, 4, (20, 135, 250))      # 4-row strips: 7.9 M samples each


def strips(ctx, scene, conf, seed=1, fast_on=None):
    """For every strip of `conf`: (strip, strict image + stats, fast image or None, oracle image + stats).
    One oracle render serves both device builds (`fast_on`: strips that also get a fast render; default all)."""
    W, H, spp, ns, depth, tile, which = conf
    assert H % tile == 0
    nranks = H // tile
    ctx.upload(scene)
    for r in which:
        cfg = rtow.make_config(W, H, spp, ns, depth, seed=seed, precision=rtow.F64_STRICT, rank=r, nranks=nranks,
                               tile_rows=tile)
        img, st = ctx.render(scene, cfg)
        ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
        assert img.shape == ref.shape == (tile, W, 3) and st.samples == ost.samples == tile * W * spp
        fast = None
        if fast_on is None or r in fast_on:
            cfg.precision = rtow.F64_FAST
            fast, fst = ctx.render(scene, cfg)
            assert fst.kernel_used == st.kernel_used
        yield r, img, st, fast, ref, ost


def recheck_checker_tree(scene, conf, strip):
    """The checker tree against the reference tree on the same strip geometry at 1 spp per stream."""
    W, H, spp, ns, depth, tile, _ = conf
    cfg = rtow.make_config(W, H, 2, 2, depth, seed=5, rank=strip, nranks=H // tile, tile_rows=tile)
    a, sa = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16)
    b, sb = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
    assert np.array_equal(a, b) and sa.segments == sb.segments


@pytest.mark.parametrize("moving", [False, True])
def test_c2_cover_strips_strict_bitwise_and_fast_within_tolerance(ctx, moving):
    """configs[1]: 1200x800, 100 spp, 50 bounces, 10 streams, 8-row strips (8x8 tiles) — the bench's shape."""
    scene = rtow.HostScene.cover(11, 1.5, moving)
    recheck_checker_tree(scene, C2, 47)
    for r, img, st, fast, ref, ost in strips(ctx, scene, C2):
        assert st.kernel_used == rtow.KERNEL_GRID
        assert st.segments == ost.segments, r
        assert np.array_equal(img, ref), (r, int((img != ref).sum()))
        check_fast(fast, ref, 100, ("C2 fast", moving, r))


def test_c3_500spp_strip_strict_and_fast(ctx):
    """configs[2]'s per-pixel work (500 spp, 50 streams) on one horizon strip."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    for r, img, st, fast, ref, ost in strips(ctx, scene, C3):
        assert st.segments == ost.segments and np.array_equal(img, ref)
        check_fast(fast, ref, 500, ("C3 fast", r))


def test_c2_whole_frame_fast_vs_strict_on_the_device(ctx):
    """The whole configs[1] frame, fast build against strict build (which is the oracle bit for bit
    on the strips above): same tolerance as fast-vs-oracle, plus equal sample accounting."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    W, H, spp, ns, depth, tile, _ = C2
    a, sa = ctx.render(scene, rtow.make_config(W, H, spp, ns, depth, seed=1, precision=rtow.F64_STRICT))
    b, sb = ctx.render(scene, rtow.make_config(W, H, spp, ns, depth, seed=1, precision=rtow.F64_FAST))
    assert sa.samples == sb.samples == W * H * spp
    assert abs(int(sa.segments) - int(sb.segments)) <= 1e-5 * sa.segments
    mad, off = check_fast(b, a, spp, "C2 whole frame fast vs strict")
    # strips of the whole-frame render are the strip renders (partition independence at full size)
    part, _ = ctx.render(scene, rtow.make_config(W, H, spp, ns, depth, seed=1, precision=rtow.F64_STRICT, rank=30,
                                                 nranks=100, tile_rows=8))
    assert np.array_equal(part, a[240:248])


@pytest.mark.parametrize("moving", [False, True])
def test_c2_whole_frame_strict_equals_the_oracle_bit_for_bit(ctx, moving):
    """configs[1] WHOLE: every one of the 960,000 pixels of the 1200x800 frame at the config's 100 spp, 50 bounces and
    the bench's 10 streams — 96 M samples, 234 M ray segments — strict build against the oracle, f64 sums bit for bit
    and the same segment count; static spheres (the bench's scene) and moving ones (the reference's default).  (The
    strips above cover the same frame in pieces next to a fast render each; this is the statement without a selection:
    some ten seconds of the box's 16 host threads per scene.)"""
    scene = rtow.HostScene.cover(11, 1.5, moving)
    W, H, spp, ns, depth, _, _ = C2
    cfg = rtow.make_config(W, H, spp, ns, depth, seed=1, precision=rtow.F64_STRICT)
    img, st = ctx.render(scene, cfg)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
    assert st.kernel_used == rtow.KERNEL_GRID
    assert st.samples == ost.samples == W * H * spp
    assert st.segments == ost.segments
    assert np.array_equal(img, ref), int((img != ref).sum())


def test_c4_suzanne_whole_frame_strict_equals_the_oracle_bit_for_bit(ctx):
    """configs[3]'s WHOLE 1920x1080 frame (every pixel, 20 bounces), at 32 of its 256 spp so that the oracle's share
    stays within seconds: strict build against the oracle bit for bit.  (The full 256 spp are checked on the strips
    below.)"""
    scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    W, H, _, _, depth, _, _ = C4
    cfg = rtow.make_config(W, H, 32, 2, depth, seed=1, precision=rtow.F64_STRICT)
    img, st = ctx.render(scene, cfg)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
    assert st.kernel_used == rtow.KERNEL_BVH4
    assert st.samples == ost.samples == W * H * 32 and st.segments == ost.segments
    assert np.array_equal(img, ref), int((img != ref).sum())


def test_c4_suzanne_strips_strict_bitwise_and_fast_within_tolerance(ctx):
    """configs[3]: suzanne.obj, 1920x1080, 256 spp, 20 bounces (the reference's default max_child_rays)."""
    scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    assert scene.c.n_triangles == 968
    recheck_checker_tree(scene, C4, 67)
    for r, img, st, fast, ref, ost in strips(ctx, scene, C4):
        assert st.kernel_used == rtow.KERNEL_BVH4
        assert st.segments == ost.segments, r
        assert np.array_equal(img, ref), (r, int((img != ref).sum()))
        check_fast(fast, ref, 256, ("C4 fast", r))


def test_c5_mesh100k_strips_strict_bitwise_and_fast_within_tolerance(ctx, mesh_obj):
    """configs[4]: the 96,800-triangle mesh (stand-in for the absent dragon.obj), 1920x1080, 1024 spp:
    the scene image does not fit LDS.  4-row strips (7.9 M samples, ~17 M segments each)."""
    scene = rtow.HostScene.obj(mesh_obj, 16 / 9)
    assert scene.c.n_triangles == 96800
    recheck_checker_tree(scene, C5, 135)
    for r, img, st, fast, ref, ost in strips(ctx, scene, C5):
        assert st.kernel_used == rtow.KERNEL_BVH4
        assert st.segments == ost.segments, r
        assert np.array_equal(img, ref), (r, int((img != ref).sum()))
        check_fast(fast, ref, 1024, ("C5 fast", r))


def test_c5_mesh100k_whole_frame_strict_equals_the_oracle_with_both_builders(ctx, mesh_obj):
    """configs[4]'s WHOLE 1920x1080 frame on the 96,800-triangle mesh, at 8 of its 1,024 spp (16.6 M samples: what the
    oracle traces in seconds), strict build against the oracle bit for bit — with the tree AUTO takes for a mesh of this
    size (built on the device: binned SAH, csrc/rtow_build.hip pass 3c) and with the host builder's."""
    scene = rtow.HostScene.obj(mesh_obj, 16 / 9)
    W, H, _, _, depth, _, _ = C5
    cfg = rtow.make_config(W, H, 8, 2, depth, seed=3, precision=rtow.F64_STRICT)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
    used = []
    for builder in (rtow.BUILDER_AUTO, rtow.BUILDER_HOST_SAH):
        ctx.set_builder(builder)
        try:
            img, st = ctx.render(scene, cfg)
            used.append(ctx.build_info().builder)
        finally:
            ctx.set_builder(rtow.BUILDER_AUTO)
        assert st.kernel_used == rtow.KERNEL_BVH4
        assert st.samples == ost.samples == W * H * 8 and st.segments == ost.segments
        assert np.array_equal(img, ref), (builder, int((img != ref).sum()))
    assert used == [rtow.BUILDER_DEVICE_LBVH, rtow.BUILDER_HOST_SAH]
