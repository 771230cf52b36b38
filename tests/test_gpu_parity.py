"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle.

Bars:
  strict build  — bit-identical f64 radiance sums to oracle/ with the same Philox seed;
  fast build    — FMA contraction flips rare hit/miss decisions, so parity is by
                  tolerance: mean |Δ| per channel ≤ 2e-3 (linear radiance, per sample
                  average) at the small test sizes and ≤ 0.1 % of 8-bit channels off by
                  more than one level at ≥ 64 spp.
"""
import ctypes as C

import numpy as np
import pytest

import orc
import rtow
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = {
    # name: (scene kind, args, width, aspect, spp, nstreams, depth, seed) — same as make_fixtures.py
    "c1_3spheres": ("cover", (0, 16 / 9, True), 64, 16 / 9, 8, 2, 10, 7),
    "cover_static": ("cover", (11, 1.5, False), 60, 1.5, 4, 2, 50, 1),
    "cover_moving": ("cover", (11, 1.5, True), 60, 1.5, 4, 1, 50, 3),
    "suzanne": ("obj", (16 / 9,), 64, 16 / 9, 4, 2, 20, 5),
}


def make_scene(kind, args):
    if kind == "cover":
        return rtow.HostScene.cover(*args)
    return rtow.HostScene.obj(GOLDEN / "suzanne.obj", *args)


def to8(img, spp):
    return (256 * np.clip(np.sqrt(img / spp), 0.0, 0.999)).astype(np.int32)


KERNELS = {"stream": rtow.KERNEL_BRUTE, "bvh": rtow.KERNEL_BVH, "grid": rtow.KERNEL_GRID, "bvh4": rtow.KERNEL_BVH4}


def expected_kernel(name, kernel):
    """BVH4 is for triangle meshes; on scenes with spheres a BVH4 request runs the binary walk."""
    if kernel == "bvh4" and name != "suzanne":
        return rtow.KERNEL_BVH
    return KERNELS[kernel]


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("name", list(CASES))
def test_strict_is_bit_identical_to_oracle_and_golden(ctx, name, kernel):
    kind, args, w, aspect, spp, ns, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    cfg = rtow.make_config(w, rtow.image_height(w, aspect), spp, ns, depth, seed=seed,
                           precision=rtow.F64_STRICT, kernel=KERNELS[kernel])
    img, st = ctx.render(scene, cfg)
    assert st.kernel_used == expected_kernel(name, kernel)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), f"{int((img != ref).sum())} of {img.size} values differ"
    assert st.segments == ost.segments and st.samples == ost.samples
    gold = np.load(GOLDEN / "oracle_philox.npz")
    assert np.array_equal(img, gold[name + "_img"])
    assert st.segments == int(gold[name + "_segments"][0])


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("name", list(CASES))
def test_fast_build_within_tolerance(ctx, name, kernel):
    kind, args, w, aspect, spp, ns, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    cfg = rtow.make_config(w, rtow.image_height(w, aspect), spp, ns, depth, seed=seed,
                           precision=rtow.F64_FAST, kernel=KERNELS[kernel])
    img, st = ctx.render(scene, cfg)
    ref = np.load(GOLDEN / "oracle_philox.npz")[name + "_img"]
    assert np.isfinite(img).all()
    assert np.abs(img - ref).mean() / spp <= 2e-3
    # almost every pixel is still identical to rounding
    close = np.isclose(img, ref, rtol=1e-9, atol=1e-12).all(axis=-1).mean()
    assert close > 0.97, close


def _translated_cover(offset, moving=False):
    """The cover scene moved by `offset` (camera and primitives; directions unchanged), in place on the host scene."""
    scene = rtow.HostScene.cover(11, 1.5, moving)
    sc = scene.c
    t = [float(x) for x in offset]
    for i in range(sc.n_spheres):
        for k in range(3):
            sc.sphere_geom[4 * i + k] += t[k]
    for i in range(sc.n_moving):
        for k in range(3):
            sc.moving_geom[8 * i + k] += t[k]      # centre at time 0
            sc.moving_geom[8 * i + 3 + k] += t[k]  # centre at time 1
    for k in range(3):
        sc.camera.origin[k] += t[k]
        sc.camera.lower_left_corner[k] += t[k]
    return scene


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (1e4, 1e4, 1e4), (1e5, -1e5, 1e5)])
def test_fast_build_far_from_the_origin_stays_within_its_tolerance(ctx, offset):
    """The fast GRID walk tests cell-list spheres in 8 operations with k = |c|^2 - r^2 stored beside the centre
    (48-byte fat entries).  In world coordinates the cancellation in c = |o|^2 - 2 c.o + k grows with the scene's
    distance from the origin.  Measured with round 4's kernel on this scene (scripts/far_origin_check.py,
    profiles/r05_far_origin.log): at coordinates of 1e3 / 1e4 the fast image differs from the strict one beyond 1e-9
    in 9 % / 45 % of the pixels (2e-9 / 5e-7 per sample: invisible, but not the build's usual agreement), at the origin
    in none; at 1e5 the padded lists no longer fit LDS and the plain test runs anyway.  Every parity test used the
    origin-centred scene.  Round 5: a scene further out than 1,000 times its smallest radius takes the 80-byte entries
    and the 12-operation test on o - c (rtow_grid.h grid_wants_fat_lists).  Fast against strict ON THE SAME translated
    scene: no more than the origin's handful of flipped decisions at any offset."""
    import struct

    scene = _translated_cover(offset)
    w, h, spp = 240, 160, 16
    strict, st = ctx.render(scene, rtow.make_config(w, h, spp, 2, 50, seed=9, precision=rtow.F64_STRICT))
    fast, sf = ctx.render(scene, rtow.make_config(w, h, spp, 2, 50, seed=9, precision=rtow.F64_FAST))
    assert st.kernel_used == rtow.KERNEL_GRID and sf.kernel_used == rtow.KERNEL_GRID
    assert np.isfinite(fast).all()
    assert np.abs(fast - strict).mean() / spp <= 1e-6  # (2e-3 is the build's tolerance; 2e-10 is what it does here)
    close = np.isclose(fast, strict, rtol=1e-9, atol=1e-12).all(axis=-1).mean()
    assert close > 0.995, close
    assert abs(sf.segments - st.segments) <= 1e-5 * st.segments
    stride = struct.unpack_from("<I", ctx.debug_image(1), 60)[0]
    assert stride == (48 if offset[0] == 0.0 else (80 if abs(offset[0]) < 1e5 else stride))  # (1e5: 80 or no fat lists)


def test_fast_build_8bit_agreement_at_64spp(ctx):
    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(96, 64, 64, 4, 50, seed=21, precision=rtow.F64_FAST)
    img, _ = ctx.render(scene, cfg)
    ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    off = np.abs(to8(img, 64) - to8(ref, 64)) > 1
    assert off.mean() <= 1e-3, off.mean()
    assert np.abs(img - ref).mean() / 64 <= 1e-4


@pytest.mark.parametrize("name", ["cover_static", "cover_moving", "suzanne"])
def test_bvh_and_stream_kernels_agree_bitwise(ctx, name):
    """Conservative culling: walking the BVH must accept exactly the hits streaming does.
    ~1.5 M segments per case, strict arithmetic, compared bit for bit."""
    kind, args, _, aspect, _, _, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    w = 320
    h = rtow.image_height(w, aspect)
    imgs = []
    kernels = [rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_GRID] + ([rtow.KERNEL_BVH4] if name == "suzanne" else [])
    for k in kernels:
        cfg = rtow.make_config(w, h, 8, 2, depth, seed=seed + 100, precision=rtow.F64_STRICT, kernel=k)
        img, st = ctx.render(scene, cfg)
        assert st.kernel_used == k
        imgs.append((img, st.segments))
    for img, segs in imgs[1:]:
        assert segs == imgs[0][1] and np.array_equal(img, imgs[0][0])


def test_mesh100k_global_image_path_is_bit_identical(ctx, tmp_path):
    """BASELINE config C5 shape: the synthetic 96,800-triangle mesh (scripts/make_mesh.py; the
    reference's dragon.obj is absent) does not fit LDS, so the BVH kernel walks the scene image
    in global memory.  Strict build vs oracle (which walks the REFERENCE's BVH), bit for bit."""
    import subprocess
    import sys

    from conftest import REPO

    obj = tmp_path / "mesh.obj"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(obj), "10"], check=True,
                   capture_output=True)
    scene = rtow.HostScene.obj(obj, 16 / 9)
    assert scene.c.n_triangles == 96800
    ref = None
    for kernel in (rtow.KERNEL_BVH, rtow.KERNEL_BVH4):  # binary walk over the global image; 4-wide walk, top of the tree in LDS
        cfg = rtow.make_config(96, 54, 2, 1, 20, seed=13, precision=rtow.F64_STRICT, kernel=kernel)
        img, st = ctx.render(scene, cfg)
        assert st.kernel_used == kernel
        if ref is None:
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
        assert st.segments == ost.segments
        assert np.array_equal(img, ref), f"{int((img != ref).sum())} values differ"
    # the same image from a subdivided mesh and from the original (same surface): statistically equal
    base = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    cfgf = rtow.make_config(96, 54, 16, 2, 20, seed=13, precision=rtow.F64_FAST)
    a, _ = ctx.render(scene, cfgf)
    b, _ = ctx.render(base, cfgf)
    assert np.abs(a.mean(axis=(0, 1)) - b.mean(axis=(0, 1))).max() / 16 < 0.01


def test_rtweekend_binary_ppm_is_byte_identical_to_oracle(ctx):
    """End to end through the product's own CLI, host API, C-ABI and kernels: the P3 text the
    `rtweekend` binary prints in strict mode equals the oracle's PPM for the same seed."""
    import subprocess

    from conftest import REPO

    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    args = ["-w", "90", "-a", "1.5", "-s", "9", "-t", "3", "-c", "12", "-n", "4", "--seed", "77",
            "--precision", "strict"]
    for extra, moving in ((["-m"], True),):  # the reference's default is moving_spheres = true
        r = subprocess.run([str(exe)] + args + extra, capture_output=True, check=True)
        scene = orc.OrcScene.cover(4, 1.5, moving)
        cfg = rtow.make_config(90, 60, 9, 3, 12, seed=77)
        ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=4)
        assert r.stdout == orc.ppm_text(ref, 90, 60, 9)
        # the reference's stderr lines: the scanline countdown (its last state) right before "Done in" (src/render.cpp:154,188)
        assert b"\rScanlines remaining: 0 " in r.stderr and r.stderr.index(b"Scanlines remaining") < r.stderr.index(b"\nDone in")
    # OBJ path (-l): suzanne
    r = subprocess.run([str(exe), "-l", str(GOLDEN / "suzanne.obj"), "-w", "64", "-a", str(16 / 9), "-s", "4",
                        "-t", "2", "-c", "20", "--seed", "5", "--precision", "strict"], capture_output=True, check=True)
    scene = orc.OrcScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    cfg = rtow.make_config(64, 36, 4, 2, 20, seed=5)
    ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=4)
    assert r.stdout == orc.ppm_text(ref, 64, 36, 4)
    assert b"Scene has 968 triangles" in r.stderr
    # --p6: binary PPM, write_color on the device: same numbers as the P3 text
    r6 = subprocess.run([str(exe), "-l", str(GOLDEN / "suzanne.obj"), "-w", "64", "-a", str(16 / 9), "-s", "4",
                         "-t", "2", "-c", "20", "--seed", "5", "--precision", "strict", "--p6"],
                        capture_output=True, check=True)
    head = b"P6\n64 36\n255\n"
    assert r6.stdout.startswith(head) and len(r6.stdout) == len(head) + 64 * 36 * 3
    assert list(r6.stdout[len(head):]) == [int(x) for x in r.stdout.split()[4:]]


def _random_sphere_scene(hollow):
    import ctypes as C

    n = 40
    rng = np.random.default_rng(3)
    geom = np.zeros((n + 3, 4))
    geom[0] = [0, -100.5, -1, 100]
    geom[1] = [0, 0, -1, 0.5]
    geom[2] = [0, 0, -1, -0.45 if hollow else 0.2]  # hollow glass: negative inner radius
    geom[3:, :3] = rng.uniform(-2, 2, size=(n, 3)) + [0, 0.5, -3]
    geom[3:, 3] = rng.uniform(0.05, 0.3, size=n)
    mats = (rtow.Material * (n + 3))()
    for i in range(n + 3):
        kind = [rtow.MAT_LAMBERTIAN, rtow.MAT_DIELECTRIC, rtow.MAT_DIELECTRIC][i] if i < 3 else int(rng.integers(0, 3))
        mats[i].kind = kind
        mats[i].albedo = (C.c_double * 3)(*rng.uniform(0.2, 0.9, 3))
        mats[i].fuzz = float(rng.uniform(0, 0.4)) if kind == rtow.MAT_METAL else 0.0
        mats[i].ir = 1.5
    base = rtow.HostScene.cover(0, 2.0, False)
    sc = rtow.Scene()
    sc.camera = base.c.camera
    keep = dict(g=np.ascontiguousarray(geom), mi=np.arange(n + 3, dtype=np.int32),
                kinds=np.zeros(n + 3, dtype=np.int32), mats=mats)
    sc.n_spheres = n + 3
    sc.sphere_geom = keep["g"].ctypes.data_as(C.POINTER(C.c_double))
    sc.sphere_mat = keep["mi"].ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = n + 3
    sc.materials = mats
    sc.n_prims = n + 3
    sc.prim_kind = keep["kinds"].ctypes.data_as(C.POINTER(C.c_int32))
    sc.prim_index = keep["mi"].ctypes.data_as(C.POINTER(C.c_int32))
    return sc, keep


def test_overlapping_spheres_one_material_each(ctx):
    """43 overlapping spheres, a material per primitive (all three kinds); strict vs oracle."""
    sc, keep = _random_sphere_scene(hollow=False)
    for kernel in (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_GRID):
        cfg = rtow.make_config(80, 40, 6, 2, 30, seed=9, precision=rtow.F64_STRICT, kernel=kernel)
        img, st = ctx.render(sc, cfg)
        ref, ost = orc.render(sc, cfg, orc.RNG_PHILOX, nthreads=4)
        assert st.segments == ost.segments and np.array_equal(img, ref)


def test_negative_radius_hollow_sphere(ctx):
    """Negative radius (hollow glass, src/common-model.cpp:88).  The REFERENCE's BVH is not
    conservative here: Sphere::bounding_box (src/common-model.cpp:168-171) builds center -/+
    radius without |.|, an inverted box, so its own traversal can miss the sphere depending on
    leaf membership — the oracle restates that faithfully.  The device bounds use |radius|, so it
    intersects the sphere like the reference's hit test would without a BVH.  (None of the
    reference's scenes has a negative radius.)  Checked here: both device kernels agree bit for
    bit, and the hollow shell changes the image."""
    sc, keep = _random_sphere_scene(hollow=True)
    plain, keep2 = _random_sphere_scene(hollow=False)
    imgs = []
    for kernel in (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH):
        cfg = rtow.make_config(80, 40, 6, 2, 30, seed=9, precision=rtow.F64_STRICT, kernel=kernel)
        imgs.append(ctx.render(sc, cfg)[0])
    assert np.array_equal(imgs[0], imgs[1])
    other = ctx.render(plain, rtow.make_config(80, 40, 6, 2, 30, seed=9, precision=rtow.F64_STRICT))[0]
    assert not np.array_equal(imgs[0], other) and np.isfinite(imgs[0]).all()


def test_context_reuse_across_scenes_and_shapes(ctx):
    """One context, several uploads and render shapes in a row (workspace regrowth, kernel
    switches STREAM/GRID/BVH, LDS and global scene images) — every result still bit-identical."""
    scenes = [
        (rtow.HostScene.cover(0, 1.5, False), 10, rtow.KERNEL_BRUTE),     # 4 primitives -> STREAM
        (rtow.HostScene.cover(11, 1.5, True), 50, rtow.KERNEL_GRID),      # spheres -> GRID
        (rtow.HostScene.obj(GOLDEN / "suzanne.obj", 1.5), 20, rtow.KERNEL_BVH4),  # mesh -> 4-wide BVH
        (rtow.HostScene.cover(5, 1.5, False), 30, rtow.KERNEL_GRID),
    ]
    for (scene, depth, expect), (w, h, spp, ns) in zip(scenes, [(64, 40, 6, 3), (31, 17, 5, 1), (96, 64, 4, 2),
                                                                (200, 120, 2, 2)]):
        cfg = rtow.make_config(w, h, spp, ns, depth, seed=w, precision=rtow.F64_STRICT)
        img, st = ctx.render(scene, cfg)
        assert st.kernel_used == expect
        ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
        assert st.segments == ost.segments and np.array_equal(img, ref)


def test_degenerate_image_shapes(ctx):
    scene = rtow.HostScene.cover(3, 1.5, False)
    for w, h in ((2, 2), (64, 1), (1, 5), (65, 3)):   # W-1 or H-1 may be 0: the reference divides by them too
        cfg = rtow.make_config(w, h, 3, 1, 8, seed=5, precision=rtow.F64_STRICT)
        img, st = ctx.render(scene, cfg)
        ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX)
        assert img.shape == (h, w, 3)
        assert np.array_equal(img, ref, equal_nan=True)


def test_device_tonemap_equals_reference_write_color(ctx):
    """write_color as a device epilogue; bytes == the ints of the oracle's P3."""
    import torch

    scene = rtow.HostScene.cover(11, 1.5, False)
    W, H, spp = 120, 80, 12
    cfg = rtow.make_config(W, H, spp, 3, 50, seed=6, precision=rtow.F64_STRICT)
    sums = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    rgb8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    ctx.upload(scene)
    st = torch.cuda.current_stream().cuda_stream
    ctx.render_device(cfg, sums.data_ptr(), st, True)
    # poke the clamp edges too
    sums[0, 0] = torch.tensor([0.0, 1e9, spp * 0.999**2], dtype=torch.float64)
    ctx.tonemap_device(sums.data_ptr(), sums.numel(), spp, rgb8.data_ptr(), st)
    torch.cuda.synchronize()
    txt = orc.ppm_text(sums.cpu().numpy(), W, H, spp)
    want = np.array(txt.split()[4:], dtype=np.int64).reshape(H, W, 3)
    assert np.array_equal(rgb8.cpu().numpy().astype(np.int64), want)
    assert tuple(want[0, 0]) == (0, 255, 255)


@pytest.mark.parametrize("name,w,spp", [("cover_static", 960, 8), ("cover_moving", 960, 8), ("suzanne", 800, 6)])
def test_large_strict_parity_all_kernels(ctx, name, w, spp):
    """Millions of samples per scene, strict build, every closest-hit strategy against the oracle,
    bit for bit: the f32 culling of the BVH boxes and of the grid cells must never drop a hit the
    exact test accepts (≈5 M samples / 12 M segments per case; the oracle walks the reference's BVH)."""
    kind, args, _, aspect, _, _, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    h = rtow.image_height(w, aspect)
    cfg0 = rtow.make_config(w, h, spp, 2, depth, seed=seed + 1000, precision=rtow.F64_STRICT)
    ref, ost = orc.render(scene, cfg0, orc.RNG_PHILOX, nthreads=16)
    kernels = [rtow.KERNEL_BVH, rtow.KERNEL_GRID] + ([rtow.KERNEL_BRUTE] if name != "suzanne" else [rtow.KERNEL_BVH4])
    for k in kernels:
        cfg = rtow.make_config(w, h, spp, 2, depth, seed=seed + 1000, precision=rtow.F64_STRICT, kernel=k)
        img, st = ctx.render(scene, cfg)
        assert st.segments == ost.segments, (k, st.segments, ost.segments)
        assert np.array_equal(img, ref), (k, int((img != ref).sum()))


def test_grid_global_image_path_many_spheres(ctx):
    """30,000 small spheres + a ground sphere: the grid scene image (~1.3 MB) does not fit LDS, so
    the GRID kernel walks it in global memory.  Strict build vs oracle, bit for bit; BVH likewise."""
    import ctypes as C

    n = 30000
    rng = np.random.default_rng(11)
    geom = np.zeros((n + 1, 4))
    geom[0] = [0, -1000, 0, 1000]
    geom[1:, 0] = rng.uniform(-30, 30, n)
    geom[1:, 2] = rng.uniform(-30, 30, n)
    geom[1:, 3] = rng.uniform(0.03, 0.12, n)
    geom[1:, 1] = geom[1:, 3] + rng.uniform(0, 1.5, n)
    mats = (rtow.Material * 3)()
    for i, (kind, alb) in enumerate([(rtow.MAT_LAMBERTIAN, (0.5, 0.5, 0.5)), (rtow.MAT_METAL, (0.8, 0.7, 0.6)),
                                     (rtow.MAT_DIELECTRIC, (0, 0, 0))]):
        mats[i].kind = kind
        mats[i].albedo = (C.c_double * 3)(*alb)
        mats[i].fuzz = 0.1 if kind == rtow.MAT_METAL else 0.0
        mats[i].ir = 1.5
    base = rtow.HostScene.cover(0, 1.5, False)
    sc = rtow.Scene()
    sc.camera = base.c.camera
    g = np.ascontiguousarray(geom)
    mi = np.concatenate([[0], rng.integers(0, 3, n)]).astype(np.int32)
    kinds = np.zeros(n + 1, dtype=np.int32)
    idx = np.arange(n + 1, dtype=np.int32)
    sc.n_spheres = n + 1
    sc.sphere_geom = g.ctypes.data_as(C.POINTER(C.c_double))
    sc.sphere_mat = mi.ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = 3
    sc.materials = mats
    sc.n_prims = n + 1
    sc.prim_kind = kinds.ctypes.data_as(C.POINTER(C.c_int32))
    sc.prim_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    cfg0 = rtow.make_config(240, 160, 4, 2, 30, seed=21, precision=rtow.F64_STRICT)
    ref, ost = orc.render(sc, cfg0, orc.RNG_PHILOX, nthreads=16)
    for k in (rtow.KERNEL_GRID, rtow.KERNEL_BVH):
        cfg = rtow.make_config(240, 160, 4, 2, 30, seed=21, precision=rtow.F64_STRICT, kernel=k)
        img, st = ctx.render(sc, cfg)
        assert st.kernel_used == k
        assert st.segments == ost.segments and np.array_equal(img, ref), (k, int((img != ref).sum()))


def test_progressive_accumulation_is_bit_identical(ctx):
    """SURVEY §8 row f3 mechanism: stream ranges + accumulate.  Three calls [0,2) [2,3) [3,6) that
    accumulate give the sums of one call (and of the oracle) bit for bit, on both walking kernels."""
    scene = rtow.HostScene.cover(11, 1.5, True)
    for kernel in (rtow.KERNEL_GRID, rtow.KERNEL_BVH):
        one = rtow.make_config(70, 46, 18, 6, 25, seed=12, precision=rtow.F64_STRICT, kernel=kernel)
        full, fst = ctx.render(scene, one)
        acc, tot = None, 0
        for first, count in ((0, 2), (2, 1), (3, 3)):
            cfg = rtow.make_config(70, 46, 18, 6, 25, seed=12, precision=rtow.F64_STRICT, kernel=kernel,
                                   stream_first=first, stream_count=count, accumulate=int(acc is not None))
            acc, st = ctx.render(scene, cfg, into=acc)
            tot += st.segments
        assert np.array_equal(acc, full) and tot == fst.segments
    ref, _ = orc.render(scene, rtow.make_config(70, 46, 18, 6, 25, seed=12), orc.RNG_PHILOX, nthreads=4)
    assert np.array_equal(full, ref)
    L = rtow.lib()
    bad = rtow.make_config(8, 8, 4, 2, 5, stream_first=1, stream_count=2)
    assert L.rtow_local_rows(C.byref(bad)) == rtow.RTOW_EINVAL and b"stream range" in L.rtow_last_error()


def test_image_does_not_depend_on_the_partition(ctx):
    scene = rtow.HostScene.cover(11, 1.5, True)
    W, H = 50, 37
    base = rtow.make_config(W, H, 6, 3, 20, seed=4, precision=rtow.F64_STRICT)
    whole, _ = ctx.render(scene, base)
    for nranks, tile in ((2, 4), (3, 5), (8, 1), (8, 8)):
        full = np.full_like(whole, np.nan)
        for r in range(nranks):
            cfg = rtow.make_config(W, H, 6, 3, 20, seed=4, precision=rtow.F64_STRICT, rank=r,
                                   nranks=nranks, tile_rows=tile)
            part, st = ctx.render(scene, cfg)
            rows = rtow.local_rows(cfg)
            assert part.shape[0] == len(rows) == st.local_rows
            if rows:
                full[rows] = part
        assert np.array_equal(full, whole), (nranks, tile)


def test_repeatable_and_independent_of_stream_count_in_expectation(ctx):
    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(48, 32, 12, 3, 50, seed=8, precision=rtow.F64_FAST)
    a, _ = ctx.render(scene, cfg)
    b, _ = ctx.render(scene, cfg)
    assert np.array_equal(a, b)  # dynamic work distribution must not leak into the image
    # same samples through another stream count.  Fast build: the same work items (a schedule over the sample
    # range, independent of nstreams), the same image.  Strict build: one item per stream like the reference's
    # threads, i.e. another summation tree: equal to rounding only
    cfg1 = rtow.make_config(48, 32, 12, 1, 50, seed=8, precision=rtow.F64_FAST)
    c, _ = ctx.render(scene, cfg1)
    assert np.array_equal(a, c)
    s3, _ = ctx.render(scene, rtow.make_config(48, 32, 12, 3, 50, seed=8, precision=rtow.F64_STRICT))
    s1, _ = ctx.render(scene, rtow.make_config(48, 32, 12, 1, 50, seed=8, precision=rtow.F64_STRICT))
    assert np.allclose(s3, s1, rtol=1e-12, atol=1e-12) and not np.array_equal(s3, s1)
    assert np.allclose(a, s1, rtol=1e-9, atol=1e-9)
    # another seed: a different image
    cfg2 = rtow.make_config(48, 32, 12, 3, 50, seed=9, precision=rtow.F64_FAST)
    d, _ = ctx.render(scene, cfg2)
    assert not np.array_equal(a, d)


def test_depth_zero_and_rounded_down_spp(ctx):
    scene = rtow.HostScene.cover(0, 1.5, False)
    # max_child_rays = 0: every hit is black, every miss is sky (src/render.cpp:113-115)
    cfg = rtow.make_config(32, 20, 4, 2, 0, seed=3, precision=rtow.F64_STRICT)
    img, st = ctx.render(scene, cfg)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX)
    assert np.array_equal(img, ref) and st.segments == st.samples == ost.samples
    # spp 7 over 3 streams -> 6 effective samples (src/render.cpp:174,185)
    cfg = rtow.make_config(32, 20, 7, 3, 5, seed=3, precision=rtow.F64_STRICT)
    img, st = ctx.render(scene, cfg)
    ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX)
    assert st.samples == 32 * 20 * 6 and np.array_equal(img, ref)
    # fewer samples than streams -> zero samples, black image, no hang
    cfg = rtow.make_config(32, 20, 2, 3, 5, seed=3, precision=rtow.F64_STRICT)
    img, st = ctx.render(scene, cfg)
    assert st.samples == 0 and not img.any()


@pytest.mark.parametrize("sm", [False, True])
def test_bvh4_edge_cases(monkeypatch, sm):
    """The 4-wide BVH kernel (trip form and state-machine form) on degenerate launches: images of one or two
    pixels, a width that is not a multiple of the tile, zero bounces (every hit is black, src/render.cpp:113-115),
    more streams than samples (zero effective samples), one sample per stream, a strip partition with empty
    ranks — strict build against the oracle, bit for bit."""
    if sm:
        monkeypatch.setenv("RTOW_BVH4_SM", "1")
    c = rtow.Context(0)
    try:
        scene = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 1.5)
        for (w, h, spp, ns, depth), part in (((2, 2, 3, 1, 20), (0, 1, 8)), ((65, 3, 4, 2, 20), (0, 1, 8)), ((1, 5, 2, 2, 20), (0, 1, 8)),
                                             ((40, 24, 4, 4, 0), (0, 1, 8)), ((40, 24, 5, 5, 3), (0, 1, 8)),
                                             ((33, 17, 4, 2, 20), (2, 3, 5)), ((16, 8, 4, 2, 20), (7, 8, 1)),
                                             ((16, 8, 4, 2, 20), (5, 6, 4))):
            rank, nranks, tile = part
            cfg = rtow.make_config(w, h, spp, ns, depth, seed=w + h, precision=rtow.F64_STRICT, rank=rank, nranks=nranks,
                                   tile_rows=tile)
            img, st = c.render(scene, cfg)
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=4, accel=True)
            assert img.shape == ref.shape
            if img.size:
                assert st.kernel_used == rtow.KERNEL_BVH4
                assert st.segments == ost.segments and np.array_equal(img, ref, equal_nan=True), (w, h, spp, ns, depth, part)
        cfg = rtow.make_config(16, 8, 2, 3, 5, seed=1, precision=rtow.F64_STRICT)  # fewer samples than streams
        img, st = c.render(scene, cfg)
        assert st.samples == 0 and not img.any()
    finally:
        c.close()


@pytest.mark.parametrize("sm", [False, True])
def test_half_node_walk_is_bit_identical_to_the_oracle(monkeypatch, tmp_path, sm):
    """A mesh whose 4-wide image does not fit LDS whole gets 64-byte nodes with binary16 planes in the mesh's own
    frame (rtow_bvh4.h): boxes a little bigger, never smaller, so the closest hit — and the strict image — is what
    the oracle computes with the reference's tree.  suzanne subdivided 2x2 (3,872 triangles), as it is and shrunk and
    moved off the origin (the frame's centre and scale are then not 0 and 1), trip form and state machine."""
    import sys

    from conftest import REPO
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    if sm:
        monkeypatch.setenv("RTOW_BVH4_SM", "1")
    v, f = make_mesh.load(GOLDEN / "suzanne.obj")
    tris = make_mesh.subdivide(v, f, 2)
    c = rtow.Context(0)
    try:
        for k, (scale, shift) in enumerate(((1.0, (0.0, 0.0, 0.0)), (0.55, (0.3125, -0.11, 0.27)))):
            obj = tmp_path / f"m{k}.obj"
            t = tris * scale + np.array(shift)
            with open(obj, "w") as fh:
                for tri in t:
                    for p in tri:
                        fh.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
                for i in range(len(t)):
                    fh.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")
            scene = rtow.HostScene.obj(obj, 16 / 9)
            assert scene.c.n_triangles == 3872
            cfg = rtow.make_config(96, 54, 4, 2, 20, seed=5 + k, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH4)
            img, st = c.render(scene, cfg)
            bi = c.build_info()
            assert bi.bvh4_node_bytes == 64 and bi.bvh4_image_bytes > 160 * 1024
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
            assert st.kernel_used == rtow.KERNEL_BVH4 and st.segments == ost.segments
            assert np.array_equal(img, ref), f"{int((img != ref).sum())} values differ"
        small = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)  # fits LDS whole: binary32 planes
        c.render(small, rtow.make_config(16, 9, 2, 1, 20, seed=1, precision=rtow.F64_STRICT))
        assert c.build_info().bvh4_node_bytes == 128
    finally:
        c.close()


def test_meshes_around_the_lds_limit_switch_node_format_and_stay_bit_identical(tmp_path):
    """The node format of the 4-wide image is decided by its size (rtow_capi.cpp): meshes of 850 … 1,400 triangles —
    the first N of suzanne subdivided 2x2 — straddle the point where the binary32 image plus six stack entries per
    lane (round 5; eight before) stops fitting the 160 KB of LDS.  Each renders bit for bit like the oracle, whichever side it falls on, and
    both formats occur."""
    import sys

    from conftest import REPO
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    v, f = make_mesh.load(GOLDEN / "suzanne.obj")
    tris = make_mesh.subdivide(v, f, 2)
    formats = set()
    c = rtow.Context(0)
    try:
        for n in (850, 1000, 1100, 1400):
            obj = tmp_path / f"first{n}.obj"
            with open(obj, "w") as fh:
                for tri in tris[:n]:
                    for p in tri:
                        fh.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
                for i in range(n):
                    fh.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")
            scene = rtow.HostScene.obj(obj, 16 / 9)
            cfg = rtow.make_config(64, 36, 4, 2, 20, seed=n, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH4)
            img, st = c.render(scene, cfg)
            bi = c.build_info()
            formats.add(bi.bvh4_node_bytes)
            wide = bi.bvh4_image_bytes + (64 * bi.bvh4_nodes if bi.bvh4_node_bytes == 64 else 0)  # with 128-byte nodes
            assert (bi.bvh4_node_bytes == 128) == (wide + 6 * 4096 <= 160 * 1024), (n, bi.bvh4_image_bytes, bi.bvh4_nodes)
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
            assert st.kernel_used == rtow.KERNEL_BVH4 and st.segments == ost.segments, n
            assert np.array_equal(img, ref), (n, int((img != ref).sum()))
    finally:
        c.close()
    assert formats == {64, 128}


def test_mesh_beyond_the_bvh4_limits_takes_the_binary_walk(ctx, tmp_path):
    """The 4-wide image addresses triangles with 18 bits (rtow_bvh4.h): a mesh of 279,752 triangles (suzanne
    subdivided 17x17) is rendered by the binary threaded walk instead — same surface, so the image agrees with the
    968-triangle original within Monte-Carlo noise; an explicit BVH4 request falls back, it does not fail."""
    import subprocess
    import sys

    from conftest import REPO

    obj = tmp_path / "mesh17.obj"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(obj), "17"], check=True, capture_output=True)
    big = rtow.HostScene.obj(obj, 16 / 9)
    assert big.c.n_triangles == 968 * 17 * 17
    cfg = rtow.make_config(96, 54, 16, 2, 20, seed=13, precision=rtow.F64_FAST, kernel=rtow.KERNEL_BVH4)
    a, st = ctx.render(big, cfg)
    assert st.kernel_used == rtow.KERNEL_BVH and np.isfinite(a).all()
    b, sb = ctx.render(rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), cfg)
    assert sb.kernel_used == rtow.KERNEL_BVH4
    assert np.abs(a.mean(axis=(0, 1)) - b.mean(axis=(0, 1))).max() / 16 < 0.01


def test_error_paths(ctx):
    L = rtow.lib()
    fresh = rtow.Context(0)
    cfg = rtow.make_config(8, 8, 1)
    assert L.rtow_render_device(fresh._h, C.byref(cfg), C.c_void_p(8), None, None) == rtow.RTOW_ENOSCENE
    scene = rtow.HostScene.cover(0, 1.5, False)
    empty = rtow.Scene()
    empty.camera = scene.c.camera
    assert L.rtow_scene_upload(fresh._h, C.byref(empty)) == rtow.RTOW_EEMPTY  # ref.: UB
    s = scene.c
    bad = rtow.Scene.from_buffer_copy(s)
    bad_mats = (C.c_int32 * s.n_spheres)(*([s.n_materials] * s.n_spheres))
    bad.sphere_mat = C.cast(bad_mats, C.POINTER(C.c_int32))
    assert L.rtow_scene_upload(fresh._h, C.byref(bad)) == rtow.RTOW_EINVAL
    assert b"material index" in L.rtow_last_error()
    fresh.upload(scene)
    assert L.rtow_render_device(fresh._h, C.byref(cfg), None, None, None) == rtow.RTOW_EINVAL
    fresh.close()


def test_full_size_cover_properties(ctx):
    """BASELINE config C2 (1200x800, 100 spp, 50 bounces) through size-independent
    properties: sample/segment accounting, sky pixels, statistics vs a low-spp oracle
    render of the same scene, and 8-way strip reassembly at full size."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    W, H, spp, ns = 1200, 800, 100, 10
    cfg = rtow.make_config(W, H, spp, ns, 50, seed=1, precision=rtow.F64_FAST)
    img, st = ctx.render(scene, cfg)
    assert st.samples == W * H * spp
    assert abs(st.segments / st.samples - 2.43) < 0.02
    assert np.isfinite(img).all() and (img >= 0).all()
    # the top-left pixel only sees sky: every sample is (1-t)*white + t*(0.5,0.7,1.0)
    px = img[0, 0] / spp
    assert 0.5 <= px[0] <= 1.0 and px[0] <= px[1] <= px[2] == 1.0
    # oracle at reduced resolution (same camera): channel means agree within MC noise
    small = rtow.make_config(150, 100, 16, 1, 50, seed=2)
    ref, _ = orc.render(scene, small, orc.RNG_PHILOX, nthreads=8)
    m_gpu, m_ref = img.mean(axis=(0, 1)) / spp, ref.mean(axis=(0, 1)) / 16
    assert np.all(np.abs(m_gpu - m_ref) < 0.01), (m_gpu, m_ref)
    # strips of 8 rows over 8 ranks: rank 3's rows equal the same rows of the whole image
    part_cfg = rtow.make_config(W, H, spp, ns, 50, seed=1, precision=rtow.F64_FAST, rank=3,
                                nranks=8, tile_rows=8)
    part, pst = ctx.render(scene, part_cfg)
    rows = rtow.local_rows(part_cfg)
    assert len(rows) == 104 and np.array_equal(part, img[rows])


def test_stream_ranges_are_chunked_automatically_and_bit_identical(ctx, monkeypatch):
    """Large sample counts: the per-stream partial sums are bounded by tracing the streams in
    ranges (RTOW_PARTIALS_MAX_MB, default 8 GiB); the image is the one-launch image bit for bit."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(120, 80, 24, 12, 50, seed=2, precision=rtow.F64_STRICT)
    whole, sw = ctx.render(scene, cfg)
    monkeypatch.setenv("RTOW_PARTIALS_MAX_MB", "1")  # 120*80*24 B = 230 kB per stream -> 4 streams per launch
    small = rtow.Context(0)  # the knobs are read once, when a context is created
    parts, sp = small.render(scene, cfg)
    small.close()
    assert np.array_equal(whole, parts)
    assert sp.samples == sw.samples and sp.segments == sw.segments
    ref, _ = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    assert np.array_equal(parts, ref)
    # the fast build's levels (here 120 samples in 12 levels of 10, whatever nstreams is) in ranges of 4: the same
    fcfg = rtow.make_config(120, 80, 120, 3, 50, seed=2, precision=rtow.F64_FAST)
    fwhole, fw = ctx.render(scene, fcfg)
    small = rtow.Context(0)
    fparts, fp = small.render(scene, fcfg)
    rgb8a, rgb8b = np.zeros((80, 120, 3), np.uint8), np.zeros((80, 120, 3), np.uint8)
    L = rtow.lib()
    L.rtow_render_rgb8.argtypes = [C.c_void_p, C.POINTER(rtow.Scene), C.POINTER(rtow.Config), C.c_void_p, C.POINTER(rtow.Stats)]
    # (the 8-bit entry point: several launches, so write_color runs as its own kernel instead of inside the reduce)
    rtow.check(L.rtow_render_rgb8(small._h, C.byref(scene.c), C.byref(fcfg), rgb8b.ctypes.data_as(C.c_void_p), None))
    rtow.check(L.rtow_render_rgb8(ctx._h, C.byref(scene.c), C.byref(fcfg), rgb8a.ctypes.data_as(C.c_void_p), None))
    small.close()
    assert np.array_equal(fwhole, fparts) and fp.samples == fw.samples == 120 * 80 * 120 and fp.segments == fw.segments
    assert np.array_equal(rgb8a, rgb8b)


def test_rtweekend_gpus_flag_partitions_like_one_device(ctx):
    """rtweekend --gpus N: one host thread and context per device, strips scattered into the host
    image.  With every rank on device 0 (test hook) the PPM must be the single-device PPM byte for
    byte, P3 and P6."""
    import subprocess

    from conftest import REPO

    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    base = [str(exe), "-w", "96", "-a", "1.5", "-s", "8", "-c", "20", "-t", "2", "--precision", "strict"]
    one = subprocess.run(base, capture_output=True, check=True)
    for n in ("2", "3"):
        many = subprocess.run(base + ["--gpus", n, "--gpus-same-device"], capture_output=True, check=True)
        assert many.stdout == one.stdout
    p6a = subprocess.run(base + ["--p6"], capture_output=True, check=True)
    p6b = subprocess.run(base + ["--p6", "--gpus", "4", "--gpus-same-device"], capture_output=True, check=True)
    assert p6a.stdout == p6b.stdout and p6a.stdout.startswith(b"P6")
    # a device index that does not exist is an error, not a silent fallback
    bad = subprocess.run(base + ["--gpus", "64"], capture_output=True)
    assert bad.returncode != 0


def _fast_default_image(scene, cfg, _cache={}):
    """Fast-build image of (scene, cfg) from a context created WITHOUT any knob in the environment."""
    import os
    import subprocess
    import sys
    import tempfile

    key = (scene.c.n_prims, cfg.image_width, cfg.image_height, cfg.samples_per_pixel)
    if key not in _cache:
        env = {k: v for k, v in os.environ.items() if not k.startswith("RTOW_")}
        with tempfile.TemporaryDirectory() as td:
            code = (
                "import sys, numpy as np; sys.path[:0] = %r\n"
                "import rtow\n"
                "scene = rtow.HostScene.cover(11, 1.5, %r) if %d < 900 else rtow.HostScene.obj(%r, 16 / 9)\n"
                "cfg = rtow.make_config(%d, %d, %d, 2, %d, seed=5, precision=rtow.F64_FAST)\n"
                "img, st = rtow.Context(0).render(scene, cfg)\n"
                "np.save(%r, img); open(%r, 'w').write(str(st.segments))\n"
            ) % ([p for p in sys.path if p], scene.c.n_moving > 0, scene.c.n_prims, str(GOLDEN / "suzanne.obj"), cfg.image_width, cfg.image_height,
                 cfg.samples_per_pixel, cfg.max_child_rays, td + "/i.npy", td + "/s.txt")
            subprocess.run([sys.executable, "-c", code], check=True, env=env, capture_output=True)
            _cache[key] = (np.load(td + "/i.npy"), int(open(td + "/s.txt").read()))
    return _cache[key]


@pytest.mark.parametrize("env", [
    {"RTOW_WALK_CAP": "off", "RTOW_LEAF_VOTES": "1", "RTOW_FETCH_VOTES": "1"},   # every scheduling measure off
    {"RTOW_WALK_CAP": "1,64", "RTOW_LEAF_VOTES": "64", "RTOW_FETCH_VOTES": "64"},  # ... at its extreme (the grid's cap is clamped to 3)
    {"RTOW_BVH4_SM": "1"},                                                           # state-machine BVH4 kernel
    {"RTOW_BVH4_SM": "1", "RTOW_SM4_VOTES": "64,64,64"},
    {"RTOW_BVH4_STACK_K": "2"},                                                      # nearly everything spills
    {"RTOW_NO_BVH4": "1"},                                                           # meshes on the binary walk
    {"RTOW_NO_SPEC": "1"},                                                           # the generic GRID kernel instead of the scene-class specialisation
], ids=["plain", "extreme", "sm4", "sm4-64", "spill", "bvh2", "generic-grid"])
def test_scheduling_knobs_do_not_change_the_image(monkeypatch, env):
    """Resumable walks, leaf / fetch quorums, the state-machine form of the BVH4 kernel, the size of the
    LDS stack and the scene-class specialisation of the GRID kernel (static cover: SPEC 1, moving cover: SPEC 2,
    against the generic instantiation) only decide WHEN a lane does its work or which dead code the kernel carries: with any setting the strict image is the oracle's, bit
    for bit, on the sphere scene (GRID) and on the mesh (BVH4), and the fast image is the fast image of the
    default setting.  (Knobs are read when a context is created.)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = rtow.Context(0)
    try:
        for scene, cfg in (
            (rtow.HostScene.cover(11, 1.5, True), rtow.make_config(240, 160, 8, 2, 50, seed=3, precision=rtow.F64_STRICT)),
            (rtow.HostScene.cover(11, 1.5, False), rtow.make_config(240, 160, 8, 2, 50, seed=6, precision=rtow.F64_STRICT)),
            (rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), rtow.make_config(320, 180, 8, 2, 20, seed=4, precision=rtow.F64_STRICT)),
        ):
            img, st = c.render(scene, cfg)
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=16, accel=True)
            assert st.segments == ost.segments and np.array_equal(img, ref), (env, int((img != ref).sum()))
            # the fast build (its own summation schedule) under the same setting: the image of the default setting
            fcfg = rtow.make_config(cfg.image_width, cfg.image_height, 24, 2, cfg.max_child_rays, seed=5, precision=rtow.F64_FAST)
            fimg, fst = c.render(scene, fcfg)
            want = _fast_default_image(scene, fcfg)
            assert np.array_equal(fimg, want[0]) and fst.segments == want[1], env
    finally:
        c.close()


def test_rtweekend_rccl_path_with_one_rank(ctx):
    """rtow_render_multi (the single-process multi-device form): with use_rccl the strips go through a
    real RCCL communicator (ncclCommInitAll, one ncclGather, one D2H).  A one-GPU box can only hold a
    one-rank communicator, so this checks that path end to end with N = 1 — library loaded on demand,
    communicator, collective, copy, reassembly — and the N = 3 partition without the collective; the
    frames equal the single-context frame bit for bit.  (N > 1 over RCCL needs N devices: unmeasured.)"""
    import subprocess

    from conftest import REPO

    scene = rtow.HostScene.cover(11, 1.5, True)
    cfg = rtow.make_config(150, 100, 12, 3, 30, seed=9, precision=rtow.F64_STRICT, tile_rows=8)
    whole, st = ctx.render(scene, cfg)
    one, s1 = rtow.render_multi([0], scene, cfg, use_rccl=True)
    assert np.array_equal(one, whole) and s1.segments == st.segments and s1.samples == st.samples
    three, s3 = rtow.render_multi([0, 0, 0], scene, cfg, use_rccl=False)
    assert np.array_equal(three, whole) and s3.segments == st.segments
    L = rtow.lib()
    ids = (C.c_int32 * 2)(0, 0)
    out = np.zeros((100, 150, 3))
    rc = L.rtow_render_multi(2, ids, C.byref(scene.c), C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_double)), None, 1)
    assert rc != 0  # RCCL refuses a communicator over the same device twice: an error, not a wrong image
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    base = [str(exe), "-w", "96", "-a", "1.5", "-s", "8", "-c", "20", "-t", "2", "--precision", "strict"]
    plain = subprocess.run(base, capture_output=True, check=True)
    rccl = subprocess.run(base + ["--rccl"], capture_output=True, check=True)
    assert rccl.stdout == plain.stdout
    p6 = subprocess.run(base + ["--p6"], capture_output=True, check=True)
    p6r = subprocess.run(base + ["--p6", "--rccl"], capture_output=True, check=True)
    assert p6r.stdout == p6.stdout


def test_fast_and_f32_builds_are_run_to_run_deterministic(ctx):
    """The image must not depend on which lane traced which sample: work is handed out dynamically and
    idle lanes take over samples at the end of a launch, so e.g. a multiply fused into the pixel
    accumulate on one path only would show up here as last-bit differences between two runs."""
    scene = rtow.HostScene.cover(11, 1.5, False)
    for prec in (rtow.F64_FAST, rtow.F32, rtow.F64_STRICT):
        for kernel in (rtow.KERNEL_GRID, rtow.KERNEL_BVH):
            cfg = rtow.make_config(480, 320, 6, 2, 50, seed=1, precision=prec, kernel=kernel)
            a, _ = ctx.render(scene, cfg)
            b, _ = ctx.render(scene, cfg)
            assert np.array_equal(a, b), (prec, kernel, int((a != b).sum()))


def test_build_defaults_that_the_measured_numbers_rest_on():
    """Guards for two build-time defaults whose effect is large and easy to lose: the mesh tree
    makes PAIRS of triangles per leaf, which also lets suzanne's whole 4-wide image into LDS with a few stack
    entries per lane (C4 +15 %), and the grid takes about one cell per small primitive, its cell size refitted over
    the axes that are not collapsed (C2 +3 %, moving cover +5 %).  The images themselves are covered by the parity
    tests; this pins their shape."""
    import struct

    c = rtow.Context(0)
    c.upload(rtow.HostScene.obj(GOLDEN / "suzanne.obj"))
    bi = c.build_info()
    assert 200 <= bi.bvh4_nodes <= 300, bi.bvh4_nodes              # 452 with single-triangle leaves
    assert bi.bvh4_image_bytes + 6 * 4096 <= 160 * 1024             # whole image + >= 6 stack entries per lane in LDS
    for moving in (False, True):
        c.upload(rtow.HostScene.cover(11, 1.5, moving))
        img = c.debug_image(1)
        n = struct.unpack_from("<3i", img, 36)
        assert n[1] == 1 and 20 <= n[0] <= 24 and 20 <= n[2] <= 24, n  # 35 x 1 x 35 before the refit
        assert struct.unpack_from("<I", img, 60)[0] == (80 if moving else 48)  # fat lists still fit
        assert len(img) <= 160 * 1024
    c.close()


def test_variant_and_world_scene_models_render_like_the_oracle(ctx):
    """scenes built on the reference's other two models (variant primitives,
    src/variant-primitives.h; the World of src/vmodel.h) through the product's host API render, strict
    build, to the oracle's image bit for bit — in the library and through `rtweekend --primitives`."""
    import subprocess

    from conftest import REPO

    scene_ref = orc.OrcScene.cover(5, 1.5, True)
    cfg = rtow.make_config(96, 64, 6, 3, 20, seed=31, precision=rtow.F64_STRICT)
    ref, ost = orc.render(scene_ref, cfg, orc.RNG_PHILOX, nthreads=4)
    for model in (rtow.MODEL_VARIANT, rtow.MODEL_WORLD):
        scene = rtow.HostScene.cover(5, 1.5, True, model=model)
        img, st = ctx.render(scene, cfg)
        assert np.array_equal(img, ref), model
        assert st.segments == ost.segments
    mesh_ref = orc.OrcScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    mcfg = rtow.make_config(64, 36, 4, 2, 20, seed=5, precision=rtow.F64_STRICT)
    mref, _ = orc.render(mesh_ref, mcfg, orc.RNG_PHILOX, nthreads=4)
    mimg, _ = ctx.render(rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9, model=rtow.MODEL_VARIANT), mcfg)
    assert np.array_equal(mimg, mref)
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    args = ["-w", "96", "-a", "1.5", "-s", "6", "-t", "3", "-c", "20", "-n", "5", "-m", "--seed", "31",
            "--precision", "strict"]
    want = orc.ppm_text(ref, 96, 64, 6)
    for flag in ("oo", "variant", "world"):
        r = subprocess.run([str(exe)] + args + ["--primitives", flag], capture_output=True, check=True)
        assert r.stdout == want, flag
    r = subprocess.run([str(exe), "-l", str(GOLDEN / "suzanne.obj"), "--primitives", "world", "-w", "32"],
                       capture_output=True)
    assert r.returncode == 1 and b"spheres only" in r.stderr


def test_fast_build_image_does_not_depend_on_nstreams(ctx):
    """Fast builds: the work items follow a schedule over the sample range, not Config::nthreads
    (include/rtow.h, rtow_debug_schedule): the same effective spp gives the same image bit for bit whatever
    nstreams is — a drop-in caller with the reference's default of 4 threads gets the bench's kernel — and it
    stays within the fast build's tolerance of the strict image (which does follow nstreams)."""
    import ctypes as C

    scene = rtow.HostScene.cover(11, 1.5, False)
    imgs = []
    for ns in (1, 4, 12):
        cfg = rtow.make_config(240, 160, 48, ns, 50, seed=5, precision=rtow.F64_FAST)
        imgs.append(ctx.render(scene, cfg)[0])
        pairs = (C.c_uint32 * 128)()
        n = rtow.lib().rtow_debug_schedule(ctx._h, C.byref(cfg), pairs, 64)
        sched = [(pairs[2 * i], pairs[2 * i + 1]) for i in range(n)]
        assert sum(c for _, c in sched) == 48 and sched[0][0] == 0
        assert all(a + c == b for (a, c), (b, _) in zip(sched, sched[1:]))  # contiguous sample ranges
        assert all(c == 12 for _, c in sched)  # the divisor of 48 nearest 10
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
    # a scene of triangles only aims at 16 samples per item (longer, resumable walks), again whatever nstreams is
    mesh = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    mimgs = []
    for ns in (1, 4):
        mcfg = rtow.make_config(96, 54, 64, ns, 20, seed=5, precision=rtow.F64_FAST)
        mimgs.append(ctx.render(mesh, mcfg)[0])
        pairs = (C.c_uint32 * 128)()
        n = rtow.lib().rtow_debug_schedule(ctx._h, C.byref(mcfg), pairs, 64)
        assert n == 4 and all(pairs[2 * i + 1] == 16 for i in range(n))
    assert np.array_equal(mimgs[0], mimgs[1])
    strict, _ = ctx.render(scene, rtow.make_config(240, 160, 48, 4, 50, seed=5, precision=rtow.F64_STRICT))
    assert np.abs(imgs[0] - strict).mean() / 48 <= 1e-4
    # a stream range is scheduled over its own samples: two accumulated halves stay within the same tolerance
    full = imgs[1]
    half = ctx.render(scene, rtow.make_config(240, 160, 48, 4, 50, seed=5, precision=rtow.F64_FAST, stream_first=0,
                                               stream_count=2))[0]
    assert np.abs(half - full).mean() / 48 > 1e-3  # (really half the samples)


def test_rtow_render_uploads_only_what_its_kernel_reads(ctx):
    """rtow_render / rtow_render_rgb8 know their config and build only the structures its kernel reads (the cover
    scene through the grid kernel: no BVH image, no binary32 images).  The image is the one a full upload gives;
    a later rtow_render_device that needs a structure the lean upload left out is refused (not a wrong image)
    until rtow_scene_upload — which builds everything — has run."""
    import torch

    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(120, 80, 8, 2, 50, seed=3, precision=rtow.F64_STRICT)
    c = rtow.Context(0)
    try:
        c.upload(scene)
        full_bi = c.build_info()
        assert full_bi.bvh_image_bytes > 0 and full_bi.grid_image_bytes > 0
        out = torch.zeros((80, 120, 3), dtype=torch.float64, device="cuda:0")
        c.render_device(cfg, out.data_ptr(), 0, True)
        torch.cuda.synchronize()
        want = out.cpu().numpy()
        img, st = c.render(scene, cfg)  # lean: the grid image only
        assert np.array_equal(img, want) and st.kernel_used == rtow.KERNEL_GRID
        lean_bi = c.build_info()
        assert lean_bi.bvh_image_bytes == 0 and lean_bi.grid_image_bytes == full_bi.grid_image_bytes
        bvh_cfg = rtow.make_config(120, 80, 8, 2, 50, seed=3, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
        with pytest.raises(rtow.RtowError, match="rtow_scene_upload"):
            c.render_device(bvh_cfg, out.data_ptr(), 0, True)
        f32_cfg = rtow.make_config(120, 80, 8, 2, 50, seed=3, precision=rtow.F32)
        with pytest.raises(rtow.RtowError):
            c.render_device(f32_cfg, out.data_ptr(), 0, True)
        via_bvh, _ = c.render(scene, bvh_cfg)  # rtow_render builds what THIS config needs
        assert np.array_equal(via_bvh, want)
        c.render(scene, f32_cfg)
        c.upload(scene)
        c.render_device(bvh_cfg, out.data_ptr(), 0, True)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        # the 8-bit entry point: write_color fused into the reduce kernel, bytes of the reference's PPM
        import ctypes as C2
        rgb8 = np.zeros((80, 120, 3), dtype=np.uint8)
        L = rtow.lib()
        L.rtow_render_rgb8.argtypes = [C2.c_void_p, C2.POINTER(rtow.Scene), C2.POINTER(rtow.Config), C2.c_void_p,
                                       C2.POINTER(rtow.Stats)]
        rtow.check(L.rtow_render_rgb8(c._h, C2.byref(scene.c), C2.byref(cfg), rgb8.ctypes.data_as(C2.c_void_p), None))
        text = rtow.ppm_text(want, 120, 80, 8).split(b"\n")[3:]
        ref8 = np.array([int(v) for line in text if line for v in line.split()], dtype=np.uint8).reshape(80, 120, 3)
        assert np.array_equal(rgb8, ref8)
    finally:
        c.close()


@pytest.mark.parametrize("ntri", [63, 64, 65, 96, 97, 968])
def test_stream_kernel_tiled_triangle_loop_equals_the_scalar_loop_and_the_oracle(ntri, tmp_path, monkeypatch):
    """The STREAM kernel streams a mesh of 64 triangles or more through LDS-staged tiles of 32 records (coalesced
    16-byte loads, double-buffered per wave, broadcast reads: csrc/rtow_trace_hit.h); below that, and with
    RTOW_STREAM_SCALAR set, through scalar loads.  Same test on the same operands in the same order: the two loops
    and the oracle agree bit for bit in the strict build — on tile counts with and without a ragged last tile (63: the
    scalar loop by size) — and the fast build's two loops agree bit for bit with each other."""
    lines = (GOLDEN / "suzanne.obj").read_text().splitlines()
    faces = [l for l in lines if l.startswith("f ")][:ntri]
    obj = tmp_path / f"part{ntri}.obj"
    obj.write_text("\n".join([l for l in lines if l.startswith("v ")] + faces) + "\n")
    scene = rtow.HostScene.obj(obj, 16 / 9)
    assert scene.c.n_triangles == ntri
    cfg = rtow.make_config(160, 90, 6, 2, 20, seed=17, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BRUTE)
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    out = {}
    for mode in ("tiled", "scalar"):
        if mode == "scalar":
            monkeypatch.setenv("RTOW_STREAM_SCALAR", "1")
        c = rtow.Context(0)  # (knobs are read at context creation)
        try:
            cfg.precision = rtow.F64_STRICT
            img, st = c.render(scene, cfg)
            assert st.kernel_used == rtow.KERNEL_BRUTE and st.segments == ost.segments
            assert np.array_equal(img, ref), (mode, int((img != ref).sum()))
            cfg.precision = rtow.F64_FAST
            out[mode], _ = c.render(scene, cfg)
        finally:
            c.close()
    assert np.array_equal(out["tiled"], out["scalar"])
