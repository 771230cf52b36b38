"""Unit checks of the oracle's building blocks: RNG known answers, hit tests on
hand-computable cases, and properties of the Philox render mode."""
import ctypes as C

import numpy as np
import pytest

import orc
import rtow

_pd = C.POINTER(C.c_double)


def _d3(*v):
    return (C.c_double * 3)(*v)


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    L = orc.lib()
    for ctr, key, want in kat:
        out = (C.c_uint32 * 4)()
        L.orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert tuple(out) == want


def test_mt19937_stream_and_double_mapping():
    L = orc.lib()
    L.orc_mt_reset()
    # std::mt19937 default seed: first two outputs are 3499211612, 581869302
    want = (3499211612 + 581869302 * 2.0**32) / 2.0**64
    assert L.orc_mt_random_double(0.0, 1.0) == want
    L.orc_mt_reset()
    assert L.orc_mt_random_double(-1.0, 1.0) == want * 2.0 + -1.0


def test_philox_requests():
    """A request is one Philox4x32-R block, a pure function of (seed, pixel, sample, request)."""
    L = orc.lib()
    R = L.orc_philox_rounds()
    assert R == 7
    a = L.orc_philox_request(42, 1000, 7, 3, 1, 0)
    assert a == L.orc_philox_request(42, 1000, 7, 3, 1, 0) and 0.0 <= a < 1.0
    vals = {L.orc_philox_request(42, p, s, r, 1, k)
            for p in range(4) for s in range(4) for r in range(3) for k in range(2)}
    assert len(vals) == 96
    assert L.orc_philox_request(43, 1000, 7, 3, 1, 0) != a
    # the words of the block, by hand: counter (request, sample, pixel, 0), key = seed
    out = (C.c_uint32 * 4)()
    L.orc_philox4x32((C.c_uint32 * 4)(3, 7, 1000, 0), (C.c_uint32 * 2)(42, 0), out, R)
    w = list(out)
    # a lens-disk block after the first candidate: two candidates, 32 bits per coordinate
    for k in range(4):
        assert L.orc_philox_request(42, 1000, 7, 3, 1, k) == w[k] / 2.0**32
    # first block of a sample: jitter + time, the top 21 bits of words 0..2 ...
    for k in range(3):
        want = (w[k] >> 11) / 2.0**21
        assert L.orc_philox_request(42, 1000, 7, 3, 0, k) == want and want < 1.0
    # ... and the first lens-disk candidate: word 3, and the 11+11+10 low bits left in words 0..2
    assert L.orc_philox_request(42, 1000, 7, 3, 0, 3) == w[3] / 2.0**32
    low = (w[0] & 0x7ff) | ((w[1] & 0x7ff) << 11) | ((w[2] & 0x3ff) << 22)
    assert L.orc_philox_request(42, 1000, 7, 3, 0, 4) == low / 2.0**32
    # unit-ball candidates: 21 bits per coordinate, one candidate per pair of words; the first block
    # of a bounce = one candidate (words 0, 1) + EITHER the coin (word 2; a bounce that draws one) OR a second
    # candidate (words 2, 3; every other bounce); a later block = two candidates
    def ball(lo, hi):
        return [(lo >> 11) / 2.0**21, (hi >> 11) / 2.0**21, ((lo & 0x7ff) | ((hi & 0x3ff) << 11)) / 2.0**21]
    first = ball(w[0], w[1])
    for k in range(3):
        assert L.orc_philox_request(42, 1000, 7, 3, 2, k) == first[k]
    assert L.orc_philox_request(42, 1000, 7, 3, 2, 3) == w[2] / 2.0**32
    second = ball(w[2], w[3])
    for k in range(3):
        assert L.orc_philox_request(42, 1000, 7, 3, 2, 4 + k) == second[k]
    later = ball(w[0], w[1]) + ball(w[2], w[3])
    for k in range(6):
        assert L.orc_philox_request(42, 1000, 7, 3, 3, k) == later[k]
    # reduced-round blocks differ from the 10-round ones (the KAT test pins the round function)
    out10 = (C.c_uint32 * 4)()
    L.orc_philox4x32((C.c_uint32 * 4)(3, 7, 1000, 0), (C.c_uint32 * 2)(42, 0), out10, 10)
    assert list(out10) != w


def test_sphere_hit_cases():
    L = orc.lib()
    t, p, n = C.c_double(), _d3(0, 0, 0), _d3(0, 0, 0)
    front = C.c_int()
    # ray along -z from (0,0,5) at unit sphere: roots 4 and 6
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 0, 5), _d3(0, 0, -1), 0.001, np.inf,
                            C.byref(t), p, n, C.byref(front)) == 1
    assert t.value == 4.0 and tuple(p) == (0, 0, 1) and tuple(n) == (0, 0, 1) and front.value == 1
    # from inside: first root negative -> second root, back face, flipped normal
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 0, 0), _d3(0, 0, -1), 0.001, np.inf,
                            C.byref(t), p, n, C.byref(front)) == 1
    assert t.value == 1.0 and front.value == 0 and tuple(n) == (0, 0, 1)
    # negative radius flips front_facing (src/common-model.cpp:88)
    assert L.orc_sphere_hit(_d3(0, 0, 0), -1.0, _d3(0, 0, 5), _d3(0, 0, -1), 0.001, np.inf,
                            C.byref(t), p, n, C.byref(front)) == 1
    assert front.value == 0
    # tmax is inclusive (root > tmax rejects), tmin is inclusive too
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 0, 5), _d3(0, 0, -1), 0.001, 4.0,
                            C.byref(t), p, n, C.byref(front)) == 1
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 0, 5), _d3(0, 0, -1), 0.001, 3.999,
                            C.byref(t), p, n, C.byref(front)) == 0
    # un-normalised direction scales t
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 0, 5), _d3(0, 0, -2), 0.001, np.inf,
                            C.byref(t), p, n, C.byref(front)) == 1
    assert t.value == 2.0
    # miss
    assert L.orc_sphere_hit(_d3(0, 0, 0), 1.0, _d3(0, 2, 5), _d3(0, 0, -1), 0.001, np.inf,
                            C.byref(t), p, n, C.byref(front)) == 0


def test_triangle_hit_cases():
    L = orc.lib()
    t, p, n = C.c_double(), _d3(0, 0, 0), _d3(0, 0, 0)
    a, b, c = _d3(0, 0, 0), _d3(1, 0, 0), _d3(0, 1, 0)  # normal +z, un-normalised (0,0,1)
    assert L.orc_triangle_hit(a, b, c, _d3(0.25, 0.25, 1), _d3(0, 0, -1), 0.001, np.inf,
                              C.byref(t), p, n) == 1
    assert t.value == 1.0 and tuple(n) == (0, 0, 1) and tuple(p) == (0.25, 0.25, 0)
    # back face is culled (det >= 1e-6)
    assert L.orc_triangle_hit(a, b, c, _d3(0.25, 0.25, -1), _d3(0, 0, 1), 0.001, np.inf,
                              C.byref(t), p, n) == 0
    # outside the edge u+v<=1
    assert L.orc_triangle_hit(a, b, c, _d3(0.75, 0.75, 1), _d3(0, 0, -1), 0.001, np.inf,
                              C.byref(t), p, n) == 0
    # the normal is the raw cross product: scale the triangle by 2 -> |n| = 4
    assert L.orc_triangle_hit(a, _d3(2, 0, 0), _d3(0, 2, 0), _d3(0.5, 0.5, 1), _d3(0, 0, -1),
                              0.001, np.inf, C.byref(t), p, n) == 1
    assert tuple(n) == (0, 0, 4)


def test_aabb_hit_quirks():
    L = orc.lib()
    lo, hi = _d3(-1, -1, -1), _d3(1, 1, 1)
    assert L.orc_aabb_hit(lo, hi, _d3(0, 0, 5), _d3(0, 0, -1), 0.001, np.inf) == 1
    assert L.orc_aabb_hit(lo, hi, _d3(0, 2, 5), _d3(0, 0, -1), 0.001, np.inf) == 0
    # zero-thickness box is never hit (t_max <= t_min), src/common-model.h:80
    assert L.orc_aabb_hit(_d3(-1, -1, 0), _d3(1, 1, 0), _d3(0, 0, 5), _d3(0, 0, -1), 0.001,
                          np.inf) == 0


def test_philox_render_is_thread_and_partition_invariant():
    scene = orc.OrcScene.cover(3, 1.5, True)
    cfg = rtow.make_config(30, 20, 6, nstreams=3, max_child_rays=8, seed=5)
    a, sa = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=1)
    b, sb = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=4)
    assert np.array_equal(a, b) and sa.segments == sb.segments
    # two "ranks" with 3-row strips reassemble to the same image
    full = np.zeros_like(a)
    for r in range(2):
        cr = rtow.make_config(30, 20, 6, 3, 8, seed=5, rank=r, nranks=2, tile_rows=3)
        part, _ = orc.render(scene, cr, orc.RNG_PHILOX)
        rows = [i for i in range(20) if (i // 3) % 2 == r]
        full[rows] = part
    assert np.array_equal(full, a)


def test_effective_spp_rounds_down_like_the_reference():
    scene = orc.OrcScene.cover(0, 1.5, False)
    cfg = rtow.make_config(8, 5, 7, nstreams=3, max_child_rays=4, seed=2)  # 7/3*3 = 6 samples
    _, st = orc.render(scene, cfg, orc.RNG_PHILOX)
    assert st.samples == 8 * 5 * 6
    assert rtow.spp_effective(cfg) == 6


def test_philox_and_mt_images_agree_statistically():
    """Different streams, same estimator: image means agree within Monte-Carlo noise."""
    scene = orc.OrcScene.cover(11, 1.5, False)
    cfg = rtow.make_config(60, 40, 16, 1, 50, seed=9)
    a, _ = orc.render(scene, cfg, orc.RNG_MT19937)
    b, _ = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=4)
    ma, mb = a.mean(axis=(0, 1)) / 16, b.mean(axis=(0, 1)) / 16
    assert np.all(np.abs(ma - mb) < 0.01), (ma, mb)


def _scene(kind):
    from conftest import GOLDEN

    if kind == "suzanne":
        return orc.OrcScene.obj(GOLDEN / "suzanne.obj", 1.5), 20
    return orc.OrcScene.cover(11, 1.5, kind == "moving"), 50


@pytest.mark.parametrize("kind", ["static", "moving", "suzanne"])
def test_checker_tree_gives_the_reference_tree_image_bit_for_bit(kind):
    """orc_render_ex(accel=1) finds the closest hit through the checker's own SAH tree instead of
    the reference's median-split tree (oracle/rtow_oracle.cpp, FastTree): same primitive tests, so
    the same image, segment count and RNG consumption, in both RNG modes.  The GPU parity tests at
    BASELINE sizes rely on this (the reference tree needs ~15,000 tests per segment on the
    96,800-triangle mesh).  The mesh itself is compared the same way in tests/test_gpu_baseline_sizes.py
    and below at a small size."""
    scene, depth = _scene(kind)
    for mode, nthreads in ((orc.RNG_PHILOX, 4), (orc.RNG_MT19937, 1)):
        cfg = rtow.make_config(120, 80, 6, 2 if mode == orc.RNG_PHILOX else 1, depth, seed=3)
        scene2, _ = _scene(kind)  # mt19937 mode: same generator state for both renders
        a, sa = orc.render(scene2, cfg, mode, nthreads=nthreads)
        scene3, _ = _scene(kind)
        b, sb = orc.render(scene3, cfg, mode, nthreads=nthreads, accel=True)
        assert np.array_equal(a, b), (kind, mode, int((a != b).sum()))
        assert sa.segments == sb.segments and sa.rng_doubles == sb.rng_doubles
        assert sb.prim_tests < sa.prim_tests  # it is a better tree


def test_checker_tree_on_the_subdivided_mesh(tmp_path):
    import subprocess
    import sys

    from conftest import REPO

    obj = tmp_path / "mesh.obj"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(obj), "10"], check=True,
                   capture_output=True)
    scene = orc.OrcScene.obj(obj, 16 / 9)
    assert scene.c.n_triangles == 96800
    cfg = rtow.make_config(1920, 1080, 1, 1, 20, seed=2, rank=137, nranks=540, tile_rows=2)
    a, sa = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    b, sb = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8, accel=True)
    assert np.array_equal(a, b) and sa.segments == sb.segments
    assert sb.prim_tests * 100 < sa.prim_tests


@pytest.mark.parametrize("kind", ["static", "moving", "suzanne"])
def test_t3_philox_vs_mt19937_rmse_against_two_independent_mt_runs(kind):
    """SURVEY.md §8c T3 at the survey's bar: the counter-based stream (Philox4x32-7, 21-bit
    quantised jitter and unit-ball candidates) against the reference's mt19937 stream
    (src/random-utils.cpp:6-41) on 300x200x64 spp renders — per-channel means of the displayed
    8-bit image within 0.5/255, and RMSE <= 1.5x the RMSE between two INDEPENDENT mt19937 renders
    (the second one made independent by burning draws before render).  The device path is
    bit-identical to the Philox oracle (GPU tests), so this carries over to it.  (Closest hits go
    through the checker's tree, which the test above shows changes nothing.)"""
    spp, W, H = 64, 300, 200

    def to8(img):
        return 256 * np.clip(np.sqrt(img / spp), 0.0, 0.999)

    cfg = rtow.make_config(W, H, spp, 1, 50 if kind != "suzanne" else 20, seed=31)
    s0, _ = _scene(kind)
    a, _ = orc.render(s0, cfg, orc.RNG_MT19937, accel=True)
    s1, _ = _scene(kind)                            # same scene (generator reset) ...
    orc.lib().orc_mt_burn(12345)                   # ... different render stream
    b, _ = orc.render(s1, cfg, orc.RNG_MT19937, accel=True)
    c, _ = orc.render(s0, cfg, orc.RNG_PHILOX, nthreads=8, accel=True)
    for key in ("sphere_geom", "moving_geom", "triangle_geom"):
        assert np.array_equal(orc.scene_arrays(s0.c)[key], orc.scene_arrays(s1.c)[key])
    rmse_ref = np.sqrt(np.mean((to8(a) - to8(b)) ** 2))
    rmse_phi = np.sqrt(np.mean((to8(a) - to8(c)) ** 2))
    assert rmse_phi <= 1.5 * rmse_ref, (rmse_phi, rmse_ref)
    d_ref = np.abs(to8(a).mean(axis=(0, 1)) - to8(b).mean(axis=(0, 1)))
    d_phi = np.abs(to8(a).mean(axis=(0, 1)) - to8(c).mean(axis=(0, 1)))
    assert np.all(d_phi < 0.5), (d_phi, d_ref)


def test_unit_ball_candidates_are_uniform_per_coordinate():
    """The 21-bit unit-ball mapping (csrc/rtow_trace_rng.h ball_from_pair, oracle PhiloxDraw): x and
    y are the top 21 bits of two Philox words, z is assembled from the 11 + 10 LOW bits left over.
    Each coordinate must be uniform on [0, 1) at every bit, z's low-order assembly included, and the
    three must be uncorrelated — checked on 200,000 candidates of the first scatter block (kind 2)
    and of the later blocks (kind 3, two candidates per block)."""
    L = orc.lib()
    n = 50_000
    cols = []
    for kind, ks in ((2, (0, 1, 2)), (3, (0, 1, 2)), (3, (3, 4, 5))):
        v = np.array([[L.orc_philox_request(7, p, p % 13, 1 + p % 3, kind, k) for k in ks] for p in range(n)])
        cols.append(v)
    v = np.concatenate(cols)  # 150,000 x 3
    assert v.min() >= 0.0 and v.max() < 1.0
    q = np.round(v * 2**21).astype(np.int64)
    assert np.array_equal(q / 2**21, v)  # exactly 21 bits per coordinate
    m = len(v)
    for c in range(3):
        # mean and variance of U[0,1): 1/2 +- 5 sigma, 1/12
        assert abs(v[:, c].mean() - 0.5) < 5 * np.sqrt(1 / 12 / m), (c, v[:, c].mean())
        assert abs(v[:, c].var() - 1 / 12) < 0.002
        # every one of the 21 bits is a fair coin (5 sigma)
        for bit in range(21):
            ones = ((q[:, c] >> bit) & 1).mean()
            assert abs(ones - 0.5) < 5 * 0.5 / np.sqrt(m), (c, bit, ones)
        # 64-bin chi-square (63 dof: mean 63, sd 11.2)
        hist = np.bincount((v[:, c] * 64).astype(int), minlength=64)
        chi = ((hist - m / 64) ** 2 / (m / 64)).sum()
        assert chi < 63 + 6 * 11.3, (c, chi)
    cc = np.corrcoef(v.T)
    assert np.abs(cc[np.triu_indices(3, 1)]).max() < 5 / np.sqrt(m)
    # acceptance rate of the rejection loop: volume of the unit ball's positive octant, pi/6
    acc = (np.sum(v * v, axis=1) < 1.0).mean()
    assert abs(acc - np.pi / 6) < 5 * np.sqrt(0.25 / m)


def test_stream_ranges_and_accumulation_are_bit_identical_to_one_call():
    """include/rtow.h stream_first / stream_count / accumulate: [0,a) then [a,n) accumulated ==
    one call over [0,n), bit for bit (same samples, same summation order)."""
    scene = orc.OrcScene.cover(5, 1.5, True)
    full_cfg = rtow.make_config(40, 26, 12, 4, 12, seed=17)
    full, fst = orc.render(scene, full_cfg, orc.RNG_PHILOX, nthreads=2)
    a_cfg = rtow.make_config(40, 26, 12, 4, 12, seed=17, stream_first=0, stream_count=1)
    b_cfg = rtow.make_config(40, 26, 12, 4, 12, seed=17, stream_first=1, stream_count=3, accumulate=1)
    acc, sa = orc.render(scene, a_cfg, orc.RNG_PHILOX)
    acc, sb = orc.render(scene, b_cfg, orc.RNG_PHILOX, nthreads=2, into=acc)
    assert np.array_equal(acc, full)
    assert sa.samples + sb.samples == fst.samples and sa.segments + sb.segments == fst.segments
    # a range alone is a different (smaller) estimate
    part, _ = orc.render(scene, rtow.make_config(40, 26, 12, 4, 12, seed=17, stream_first=2, stream_count=2),
                         orc.RNG_PHILOX)
    assert not np.array_equal(part, full) and part.sum() < full.sum()
