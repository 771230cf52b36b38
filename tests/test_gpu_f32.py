"""GPU: the binary32 build (RTOW_F32) against the binary64 builds — SURVEY.md §8c, step T2.

RTOW_F32 is NOT the reference's arithmetic (the reference is all binary64) and never the bench
line; it is the preview mode SURVEY.md §2/§7 plans ("f64 mode first, then f32").  What runs in
binary32: rays, hit tests on grid-cell spheres and on triangles (binary32 records), shading.
What stays binary64: the always-test large primitives of the grid kernel and every sphere met
by the STREAM and BVH kernels (an r = 1000 sphere cancels catastrophically in binary32), and
the pixel sums.  Same Philox blocks, so the two builds trace the same paths until a rounding
difference flips a decision.

Tolerances (stated here, checked below; SURVEY's T2 asks for <= 1/255 mean |Δ| per channel at
>= 100 spp — because both builds consume the same random blocks the measured differences are
far smaller, 3e-8 .. 6e-5, and the tests hold the build to a tenth of the survey's bar):
  T2a  mean |Δ| of the displayed value sqrt(radiance) per channel <= 0.1/255 at 128 spp;
  T2b  channel means within 0.05/255;
  T2c  no structured artefacts: 16x16-pixel block means differ by <= 2/255 everywhere
       (a flipped decision moves a dark pixel by up to sqrt(1/spp) = 0.09, a block by 3e-4;
       banding or acne on the ground would move whole blocks).
"""
import subprocess

import numpy as np
import pytest

import rtow
from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu


def shown(img, spp):
    return np.sqrt(np.clip(img / spp, 0.0, 1.0))


def block_means(a, b=16):
    h, w, _ = a.shape
    h, w = h // b * b, w // b * b
    return a[:h, :w].reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


def pair(ctx, scene, w, h, spp, depth, kernel, seed=5):
    out = []
    for prec in (rtow.F64_FAST, rtow.F32):
        cfg = rtow.make_config(w, h, spp, max(1, spp // 4), depth, seed=seed, precision=prec, kernel=kernel)
        img, st = ctx.render(scene, cfg)
        out.append((shown(img, rtow.spp_effective(cfg)), st))
    return out


CASES = [
    ("cover_grid", lambda: rtow.HostScene.cover(11, 1.5, False), 600, 400, 128, 50, rtow.KERNEL_AUTO),
    ("cover_bvh", lambda: rtow.HostScene.cover(11, 1.5, False), 300, 200, 128, 50, rtow.KERNEL_BVH),
    ("cover_moving_grid", lambda: rtow.HostScene.cover(11, 1.5, True), 600, 400, 128, 50, rtow.KERNEL_AUTO),
    ("c1_stream", lambda: rtow.HostScene.cover(0, 16 / 9, True), 400, 225, 128, 10, rtow.KERNEL_AUTO),
    # (explicit BVH: AUTO gives the binary64 builds the 4-wide walk, which the f32 preview build does not have)
    ("suzanne_bvh", lambda: rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), 480, 270, 128, 20, rtow.KERNEL_BVH),
    ("suzanne_grid", lambda: rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), 320, 180, 128, 20, rtow.KERNEL_GRID),
]


@pytest.mark.parametrize("name,make,w,h,spp,depth,kernel", CASES, ids=[c[0] for c in CASES])
def test_f32_build_matches_f64_within_t2(ctx, name, make, w, h, spp, depth, kernel):
    scene = make()
    (a, sa), (b, sb) = pair(ctx, scene, w, h, spp, depth, kernel)
    assert sa.kernel_used == sb.kernel_used
    assert np.isfinite(b).all()
    assert sa.samples == sb.samples
    # the same paths but for rare flipped decisions: segment counts within 0.5 %
    assert abs(int(sa.segments) - int(sb.segments)) <= 0.005 * sa.segments
    d = np.abs(a - b)
    assert d.mean(axis=(0, 1)).max() <= 0.1 / 255.0, d.mean(axis=(0, 1))                      # T2a
    assert np.abs(a.mean(axis=(0, 1)) - b.mean(axis=(0, 1))).max() <= 0.05 / 255.0            # T2b
    assert np.abs(block_means(a) - block_means(b)).max() <= 2.0 / 255.0                       # T2c


def test_f32_ground_sphere_has_no_acne(ctx):
    """Only the r = 1000 ground sphere, lit by the sky: every camera ray that hits it must see the
    same smooth shading as the binary64 build (binary32 self-intersections would darken it)."""
    import ctypes as C

    base = rtow.HostScene.cover(0, 2.0, False)
    geom = np.array([[0.0, -1000.0, 0.0, 1000.0]])
    mats = (rtow.Material * 1)()
    mats[0].kind = rtow.MAT_LAMBERTIAN
    mats[0].albedo = (C.c_double * 3)(0.5, 0.5, 0.5)
    mats[0].ir = 1.5
    zero = np.zeros(1, dtype=np.int32)
    sc = rtow.Scene()
    sc.camera = base.c.camera
    sc.n_spheres = 1
    sc.sphere_geom = geom.ctypes.data_as(C.POINTER(C.c_double))
    sc.sphere_mat = zero.ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = 1
    sc.materials = mats
    sc.n_prims = 1
    sc.prim_kind = zero.ctypes.data_as(C.POINTER(C.c_int32))
    sc.prim_index = zero.ctypes.data_as(C.POINTER(C.c_int32))
    (a, sa), (b, sb) = pair(ctx, sc, 256, 128, 256, 50, rtow.KERNEL_AUTO)
    ground = a[-32:]  # bottom rows: ground only
    assert np.abs(ground - b[-32:]).mean() <= 0.1 / 255.0
    assert np.abs(block_means(a) - block_means(b)).max() <= 2.0 / 255.0
    assert abs(int(sa.segments) - int(sb.segments)) <= 0.002 * sa.segments


def test_f32_is_deterministic_and_partition_independent(ctx):
    scene = rtow.HostScene.cover(11, 1.5, False)
    w, h = 240, 160
    full = rtow.make_config(w, h, 8, 2, 50, seed=3, precision=rtow.F32)
    a, _ = ctx.render(scene, full)
    b, _ = ctx.render(scene, full)
    assert np.array_equal(a, b)
    out = np.zeros_like(a)
    for rank in range(4):
        cfg = rtow.make_config(w, h, 8, 2, 50, seed=3, precision=rtow.F32, rank=rank, nranks=4, tile_rows=8)
        part, _ = ctx.render(scene, cfg)
        out[rtow.local_rows(cfg)] = part
    assert np.array_equal(out, a)


def test_f32_with_device_built_tree(ctx):
    scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    cfg = rtow.make_config(160, 90, 8, 2, 20, seed=4, precision=rtow.F32, kernel=rtow.KERNEL_BVH)
    a, sa = ctx.render(scene, cfg)
    d = rtow.Context(0)
    d.set_builder(rtow.BUILDER_DEVICE_LBVH)
    b, sb = d.render(scene, cfg)
    d.close()
    # same arithmetic, another (conservative) tree: the same image
    assert sa.segments == sb.segments and np.array_equal(a, b)


def test_rtweekend_precision_f32_flag(ctx):
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    args = [str(exe), "-w", "96", "-a", "1.5", "-s", "64", "-c", "20", "-t", "4", "-n", "4"]
    a = subprocess.run(args + ["--precision", "fast"], capture_output=True, check=True)
    b = subprocess.run(args + ["--precision", "f32"], capture_output=True, check=True)
    va = np.array(a.stdout.split()[4:], dtype=np.int32)
    vb = np.array(b.stdout.split()[4:], dtype=np.int32)
    assert va.shape == vb.shape == (96 * 64 * 3,)
    assert np.abs(va - vb).mean() <= 1.5 and abs(va.mean() - vb.mean()) <= 0.5


def test_unknown_precision_is_rejected(ctx):
    scene = rtow.HostScene.cover(0, 2.0, False)
    cfg = rtow.make_config(32, 16, 2, 1, 5, precision=3)
    with pytest.raises(rtow.RtowError):
        ctx.render(scene, cfg)
