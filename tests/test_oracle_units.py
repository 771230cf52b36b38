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


SIN_C = [float.fromhex(h) for h in ("0x1.ffffffdb33084p-1", "-0x1.555549a9260fdp-3", "0x1.110eb1f04c8ffp-7",
                                     "-0x1.9f6d0201a288bp-13", "0x1.5da8d4e70fe23p-19")]
HALF_PI = 1.5707963267948966


def sin_quarter(x):
    """The samplers' sine polynomial, operation for operation (oracle sin_quarter, csrc/rtow_trace_rng.h): Python floats
    are binary64 and every operation below rounds once, like the strict build's."""
    x2 = x * x
    return x * (SIN_C[0] + x2 * (SIN_C[1] + x2 * (SIN_C[2] + x2 * (SIN_C[3] + x2 * SIN_C[4]))))


def test_philox_requests():
    """A request is ONE Philox4x32-R block, a pure function of (seed, pixel, sample, request), and carries everything the
    request needs (round 5: no request draws a second block).  The layouts, word by word, recomputed by hand."""
    import math

    L = orc.lib()
    R = L.orc_philox_rounds()
    assert R == 7
    a = L.orc_philox_request(42, 1000, 7, 3, 2, 0)
    assert a == L.orc_philox_request(42, 1000, 7, 3, 2, 0) and 0.0 <= a < 1.0
    vals = {L.orc_philox_request(42, p, s, r, 2, k)
            for p in range(4) for s in range(4) for r in range(3) for k in range(2)}
    assert len(vals) == 96
    assert L.orc_philox_request(43, 1000, 7, 3, 2, 0) != a
    # the words of the block, by hand: counter (request, sample, pixel, 0), key = seed
    out = (C.c_uint32 * 4)()
    L.orc_philox4x32((C.c_uint32 * 4)(3, 7, 1000, 0), (C.c_uint32 * 2)(42, 0), out, R)
    w = list(out)
    # the block of a new sample: jitter + time, the top 21 bits of words 0..2 ...
    for k in range(3):
        want = (w[k] >> 11) / 2.0**21
        assert L.orc_philox_request(42, 1000, 7, 3, 0, k) == want and want < 1.0
    # ... and the lens point from two 32-bit uniforms — word 3, and the 11+11+10 low bits left in words 0..2 — in polar
    # coordinates: radius sqrt(U1), angle 2 pi U2 (the top two bits of U2: the quadrant)
    low = (w[0] & 0x7ff) | ((w[1] & 0x7ff) << 11) | ((w[2] & 0x3ff) << 22)
    rho = math.sqrt(w[3] / 2.0**32)
    th = (low & 0x3fffffff) * (2.0**-30 * HALF_PI)
    sn, cs = sin_quarter(th), sin_quarter(HALF_PI - th)
    q = low >> 30
    cx, sy = (sn, cs) if q & 1 else (cs, sn)
    assert L.orc_philox_request(42, 1000, 7, 3, 0, 3) == rho * (-cx if q in (1, 2) else cx)
    assert L.orc_philox_request(42, 1000, 7, 3, 0, 4) == rho * (-sy if q >= 2 else sy)
    # the block of a bounce: z = top 24 bits of word 0, azimuth = top 24 bits of word 1, radius = the largest of three
    # 16-bit uniforms (the halves of word 2, the low half of word 3); the coin = high half of word 3 + low bytes of 0, 1
    z = (w[0] >> 8) / 2.0**24
    phi = (w[1] >> 8) * (2.0**-24 * HALF_PI)
    r = max(w[2] & 0xffff, w[2] >> 16, w[3] & 0xffff) / 2.0**16
    sn, cs = sin_quarter(phi), sin_quarter(HALF_PI - phi)
    rs = r * math.sqrt(1.0 - z * z)
    for k, want in enumerate((rs * cs, rs * sn, r * z)):
        assert L.orc_philox_request(42, 1000, 7, 3, 2, k) == want
    coin = (((w[3] >> 16) << 16) | ((w[0] & 0xff) << 8) | (w[1] & 0xff)) / 2.0**32
    assert L.orc_philox_request(42, 1000, 7, 3, 2, 3) == coin and 0.0 <= coin < 1.0
    # the polynomial is a sine to 7e-9 on the whole quarter turn, and sin^2 + cos^2 stays 1 to 2e-8
    xs = np.linspace(0.0, HALF_PI, 20001)
    sp = np.array([sin_quarter(float(x)) for x in xs])
    assert np.abs(sp - np.sin(xs)).max() < 7e-9
    assert np.abs(sp**2 + sp[::-1] ** 2 - 1.0).max() < 2e-8
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


def _ks_two_sample(a, b):
    """Kolmogorov-Smirnov distance of two samples and its 0.1 % critical value."""
    a, b = np.sort(a), np.sort(b)
    allv = np.concatenate([a, b])
    d = np.abs(np.searchsorted(a, allv, side="right") / len(a) - np.searchsorted(b, allv, side="right") / len(b)).max()
    return d, 1.95 * np.sqrt((len(a) + len(b)) / (len(a) * len(b)))


def test_direct_samplers_have_the_distributions_of_the_reference_rejection_loops():
    """The Philox policy draws random_in_unit_sphere() and random_in_unit_disk() DIRECTLY, one block per request (oracle
    PhiloxDraw overloads, csrc/rtow_trace_rng.h); the reference draws them by rejection (src/random-utils.cpp:23-41: a
    point of [0,1)^3 inside the unit ball — the positive octant, returned un-normalised as the "unit vector" — and a
    point of [-1,1)^2 inside the unit disk).  Same distributions: checked here against the reference's own loops run on
    numpy's generator — every coordinate's marginal, the radius, and products that would show a dependence — by
    two-sample Kolmogorov-Smirnov tests at the 0.1 % level on 100,000 points each, plus the moments that have closed
    forms and the independence of the coin from the point it comes with."""
    L = orc.lib()
    n = 100_000
    ball = np.array([[L.orc_philox_request(7, p, p % 13, 1 + p % 3, 2, k) for k in range(4)] for p in range(n)])
    coin, ball = ball[:, 3], ball[:, :3]
    disk = np.array([[L.orc_philox_request(7, p, p % 13, 0, 0, k) for k in (3, 4)] for p in range(n)])
    rng = np.random.default_rng(12345)
    c = rng.random((3 * n, 3))
    ref_ball = c[(c * c).sum(axis=1) < 1.0][:n]  # src/random-utils.cpp:23-29
    c = rng.random((3 * n, 2)) * 2.0 - 1.0
    ref_disk = c[(c * c).sum(axis=1) < 1.0][:n]  # :34-41
    assert len(ref_ball) == n and len(ref_disk) == n
    # inside the ball / the disk, in the positive octant
    assert (ball >= 0.0).all() and ((ball * ball).sum(axis=1) < 1.0).all()
    assert ((disk * disk).sum(axis=1) < 1.0).all() and (disk.min(axis=0) < -0.99).all() and (disk.max(axis=0) > 0.99).all()
    for k in range(3):
        d, crit = _ks_two_sample(ball[:, k], ref_ball[:, k])
        assert d < crit, ("ball coordinate", k, d, crit)
    for k in range(2):
        d, crit = _ks_two_sample(disk[:, k], ref_disk[:, k])
        assert d < crit, ("disk coordinate", k, d, crit)
    for name, f in (("radius", lambda v: np.sqrt((v * v).sum(axis=1))), ("xy", lambda v: v[:, 0] * v[:, 1]),
                    ("yz", lambda v: v[:, 1] * v[:, -1]), ("x+y", lambda v: v[:, 0] + v[:, 1])):
        d, crit = _ks_two_sample(f(ball), f(ref_ball))
        assert d < crit, ("ball", name, d, crit)
        d, crit = _ks_two_sample(f(disk), f(ref_disk))
        assert d < crit, ("disk", name, d, crit)
    # closed forms: E[x] = 3/8 per coordinate and E[r] = 3/4 in the octant ball; E[r] = 2/3 and E[x] = 0 on the disk
    m = n
    assert np.abs(ball.mean(axis=0) - 0.375).max() < 5 * 0.25 / np.sqrt(m)
    assert abs(np.sqrt((ball * ball).sum(axis=1)).mean() - 0.75) < 5 * 0.2 / np.sqrt(m)
    assert np.abs(disk.mean(axis=0)).max() < 5 * 0.5 / np.sqrt(m)
    assert abs(np.sqrt((disk * disk).sum(axis=1)).mean() - 2.0 / 3.0) < 5 * 0.24 / np.sqrt(m)
    # the dielectric coin: uniform, and uncorrelated with the point that shares its block
    assert 0.0 <= coin.min() and coin.max() < 1.0 and abs(coin.mean() - 0.5) < 5 * np.sqrt(1 / 12 / m)
    hist = np.bincount((coin * 64).astype(int), minlength=64)
    assert ((hist - m / 64) ** 2 / (m / 64)).sum() < 63 + 6 * 11.3
    for k in range(3):
        assert abs(np.corrcoef(coin, ball[:, k])[0, 1]) < 5 / np.sqrt(m)


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
