"""GPU: the device-side BVH build (csrc/rtow_build.hip).

The closest hit does not depend on the tree, so the bar is the same as for the host-built
tree: strict build bit-identical to the oracle (which walks the REFERENCE's median-split BVH,
src/render.cpp:73-110) and to the image the host SAH tree gives.  The emitted links are
validated on the device at upload (rtow_scene_upload fails otherwise).
"""
import ctypes as C
import subprocess
import sys

import numpy as np
import pytest

import orc
import rtow
from conftest import GOLDEN, REPO
from test_gpu_parity import CASES, _random_sphere_scene, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dctx():
    c = rtow.Context(0)
    c.set_builder(rtow.BUILDER_DEVICE_LBVH)
    yield c
    c.close()


@pytest.mark.parametrize("name", list(CASES))
def test_device_built_tree_strict_is_bit_identical_to_oracle(dctx, name):
    kind, args, w, aspect, spp, ns, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    cfg = rtow.make_config(w, rtow.image_height(w, aspect), spp, ns, depth, seed=seed,
                           precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
    img, st = dctx.render(scene, cfg)
    bi = dctx.build_info()
    assert bi.builder == rtow.BUILDER_DEVICE_LBVH and bi.bvh_nodes >= 1
    assert st.kernel_used == rtow.KERNEL_BVH
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    assert np.array_equal(img, ref), f"{int((img != ref).sum())} of {img.size} values differ"
    assert st.segments == ost.segments
    gold = np.load(GOLDEN / "oracle_philox.npz")
    assert np.array_equal(img, gold[name + "_img"])


@pytest.mark.parametrize("name,w,spp", [("cover_static", 480, 6), ("cover_moving", 480, 6), ("suzanne", 400, 4)])
def test_device_and_host_trees_give_the_same_image(ctx, dctx, name, w, spp):
    kind, args, _, aspect, _, _, depth, seed = CASES[name]
    scene = make_scene(kind, args)
    for precision in (rtow.F64_STRICT, rtow.F64_FAST):
        cfg = rtow.make_config(w, rtow.image_height(w, aspect), spp, 2, depth, seed=seed, precision=precision,
                               kernel=rtow.KERNEL_BVH)
        a, sa = ctx.render(scene, cfg)
        b, sb = dctx.render(scene, cfg)
        assert ctx.build_info().builder == rtow.BUILDER_HOST_SAH
        assert dctx.build_info().builder == rtow.BUILDER_DEVICE_LBVH
        assert sa.segments == sb.segments
        assert np.array_equal(a, b), f"{int((a != b).sum())} values differ"


def test_device_build_mesh100k_global_image_path(dctx, tmp_path):
    """96,800 triangles (BASELINE config C5's shape): image in global memory, tree from the GPU."""
    obj = tmp_path / "mesh.obj"
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(obj), "10"], check=True,
                   capture_output=True)
    scene = rtow.HostScene.obj(obj, 16 / 9)
    cfg = rtow.make_config(96, 54, 2, 1, 20, seed=13, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
    img, st = dctx.render(scene, cfg)
    bi = dctx.build_info()
    assert bi.bvh_nodes > 96800 // 2 and bi.bvh_image_bytes > 160 * 1024
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    assert st.segments == ost.segments
    assert np.array_equal(img, ref), f"{int((img != ref).sum())} values differ"


def _sphere_scene(geom, keep_alive):
    n = len(geom)
    base = rtow.HostScene.cover(0, 2.0, False)
    mats = (rtow.Material * 1)()
    mats[0].kind = rtow.MAT_LAMBERTIAN
    mats[0].albedo = (C.c_double * 3)(0.6, 0.5, 0.4)
    mats[0].ir = 1.5
    g = np.ascontiguousarray(np.asarray(geom, dtype=np.float64))
    mi = np.zeros(n, dtype=np.int32)
    idx = np.arange(n, dtype=np.int32)
    sc = rtow.Scene()
    sc.camera = base.c.camera
    sc.n_spheres = n
    sc.sphere_geom = g.ctypes.data_as(C.POINTER(C.c_double))
    sc.sphere_mat = mi.ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = 1
    sc.materials = mats
    sc.n_prims = n
    sc.prim_kind = mi.ctypes.data_as(C.POINTER(C.c_int32))
    sc.prim_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    keep_alive.extend([g, mi, idx, mats, base])
    return sc


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9, 33])
def test_device_build_tiny_scenes(dctx, n):
    """1 primitive (no inner node), <= leaf size (the root is one leaf record), just above."""
    keep = []
    rng = np.random.default_rng(n)
    geom = np.zeros((n, 4))
    geom[:, :3] = rng.uniform(-1.5, 1.5, size=(n, 3)) + [0, 1, 0]
    geom[:, 3] = rng.uniform(0.2, 0.6, size=n)
    sc = _sphere_scene(geom, keep)
    cfg = rtow.make_config(64, 32, 4, 2, 10, seed=n, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
    img, st = dctx.render(sc, cfg)
    ref, ost = orc.render(sc, cfg, orc.RNG_PHILOX, nthreads=2)
    assert st.segments == ost.segments and np.array_equal(img, ref)
    bi = dctx.build_info()
    assert bi.bvh_nodes == 1 if n <= 2 else bi.bvh_nodes >= 3


def test_device_build_equal_morton_keys(dctx):
    """Concentric and coincident spheres: every centroid (and Morton key) is the same, so the
    radix tree is decided by the index tie-break alone."""
    keep = []
    geom = [[0.0, 1.0, 0.0, 0.1 + 0.05 * k] for k in range(12)] + [[0.0, 1.0, 0.0, 0.3]] * 3
    sc = _sphere_scene(geom, keep)
    cfg = rtow.make_config(64, 32, 4, 2, 10, seed=2, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH)
    img, st = dctx.render(sc, cfg)
    brute, sb = dctx.render(sc, rtow.make_config(64, 32, 4, 2, 10, seed=2, precision=rtow.F64_STRICT,
                                                 kernel=rtow.KERNEL_BRUTE))
    assert st.segments == sb.segments and np.array_equal(img, brute)


def test_device_build_overlapping_spheres_and_hollow(dctx):
    for hollow in (False, True):
        sc, keep = _random_sphere_scene(hollow=hollow)
        imgs = []
        for kernel in (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH):
            cfg = rtow.make_config(80, 40, 6, 2, 30, seed=9, precision=rtow.F64_STRICT, kernel=kernel)
            imgs.append(dctx.render(sc, cfg)[0])
        assert np.array_equal(imgs[0], imgs[1])


def test_rtweekend_builder_flag(ctx):
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    args = [str(exe), "-w", "64", "-a", "1.7777777777777777", "-s", "4", "-c", "10", "-t", "2", "-n", "3",
            "--precision", "strict", "--kernel", "bvh"]
    a = subprocess.run(args + ["--builder", "host"], capture_output=True, check=True)
    b = subprocess.run(args + ["--builder", "device"], capture_output=True, check=True)
    assert a.stdout == b.stdout and a.stdout.startswith(b"P3\n")
    assert b"built on the device" in b.stderr and b"built on the host (SAH)" in a.stderr
    bad = subprocess.run(args + ["--builder", "nope"], capture_output=True)
    assert bad.returncode != 0


def test_set_builder_rejects_unknown_values(ctx):
    with pytest.raises(rtow.RtowError):
        ctx.set_builder(7)
    ctx.set_builder(rtow.BUILDER_AUTO)


# ---- the grid image, built on the device (csrc/rtow_build_grid.hip) -------------------------------
@pytest.mark.parametrize("moving", [False, True])
def test_device_built_grid_image_is_byte_identical(ctx, dctx, moving):
    """Same arithmetic on both sides (bounds, median split, grid_header(), cell ranges), lists in
    ascending id order: the device-built grid image equals the host-built one byte for byte —
    also the f32 build's image derived from it."""
    scene = rtow.HostScene.cover(11, 1.5, moving)
    ctx.set_builder(rtow.BUILDER_HOST_SAH)
    ctx.upload(scene)
    dctx.upload(scene)
    for which in (1, 3):
        a, b = ctx.debug_image(which), dctx.debug_image(which)
        assert len(a) == len(b) and len(a) > 1000
        assert a == b, f"image {which}: first difference at byte {next(i for i in range(len(a)) if a[i] != b[i])}"


def test_device_built_grid_other_scenes(ctx, dctx):
    import ctypes as C

    cases = [rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), rtow.HostScene.cover(3, 1.5, True)]
    sc, keep = _random_sphere_scene(hollow=True)
    cases.append(sc)
    for scene in cases:
        ctx.set_builder(rtow.BUILDER_HOST_SAH)
        ctx.upload(scene)
        dctx.upload(scene)
        assert ctx.debug_image(1) == dctx.debug_image(1)


def test_device_built_grid_strict_render_is_bit_identical_to_oracle(dctx):
    scene = rtow.HostScene.cover(11, 1.5, False)
    cfg = rtow.make_config(120, 80, 4, 2, 50, seed=1, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_GRID)
    img, st = dctx.render(scene, cfg)
    assert st.kernel_used == rtow.KERNEL_GRID
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8)
    assert st.segments == ost.segments and np.array_equal(img, ref)


def test_device_grid_build_falls_back_like_the_host(ctx, dctx):
    """A scene the grid does not suit (one primitive: no 'small' set after the median split is
    fine, but 70 equal spheres at one point exceed nothing; many large ones do): both builders
    must agree on whether there is a grid at all."""
    keep = []
    rng = np.random.default_rng(1)
    geom = np.zeros((80, 4))
    geom[:, :3] = rng.uniform(-3, 3, size=(80, 3))
    geom[:10, 3] = 0.01          # ten tiny spheres ...
    geom[10:, 3] = 1.5           # ... and seventy huge ones: more than 64 'large' primitives
    sc = _sphere_scene(geom, keep)
    ctx.set_builder(rtow.BUILDER_HOST_SAH)
    ctx.upload(sc)
    dctx.upload(sc)
    assert ctx.build_info().grid_image_bytes == dctx.build_info().grid_image_bytes


# ---- the 4-wide image of the BVH4 kernel, collapsed from the radix tree on the device (round 4) ---------------
def _write_obj(path, tris):
    with open(path, "w") as fh:
        for tri in tris:
            for p in tri:
                fh.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for i in range(len(tris)):
            fh.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")


@pytest.mark.parametrize("sm", [False, True])
def test_device_built_bvh4_image_strict_is_bit_identical_to_oracle(dctx, tmp_path, monkeypatch, sm):
    """`--builder device` no longer forces the binary walk on a triangle mesh: the 4-wide image (breadth-first
    nodes, triangle records in sorted = leaf order, pairs of triangles per leaf) is collapsed from the LBVH on the
    GPU.  Strict build against the oracle, bit for bit: suzanne (binary32 nodes, staged in LDS whole), suzanne
    subdivided 2x2 as it is and shrunk and moved off the origin (binary16 nodes in the mesh's own frame), the
    96,800-triangle mesh of BASELINE configs[4]; trip form and state-machine form of the kernel."""
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    if sm:
        monkeypatch.setenv("RTOW_BVH4_SM", "1")
    c = rtow.Context(0)
    c.set_builder(rtow.BUILDER_DEVICE_LBVH)
    try:
        v, f = make_mesh.load(GOLDEN / "suzanne.obj")
        cases = [("suzanne", GOLDEN / "suzanne.obj", 128, 968)]
        tris = make_mesh.subdivide(v, f, 2)
        _write_obj(tmp_path / "m2.obj", tris)
        cases.append(("2x2", tmp_path / "m2.obj", 64, 3872))
        _write_obj(tmp_path / "m2off.obj", tris * 0.55 + np.array((0.3125, -0.11, 0.27)))
        cases.append(("2x2 off-origin", tmp_path / "m2off.obj", 64, 3872))
        if not sm:
            subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(tmp_path / "m10.obj"), "10"],
                           check=True, capture_output=True)
            cases.append(("10x10", tmp_path / "m10.obj", 64, 96800))
        for k, (name, obj, node_bytes, nt) in enumerate(cases):
            scene = rtow.HostScene.obj(obj, 16 / 9)
            assert scene.c.n_triangles == nt
            w, h, spp = (96, 54, 4) if nt < 50000 else (96, 54, 2)
            cfg = rtow.make_config(w, h, spp, 2, 20, seed=5 + k, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_AUTO)
            img, st = c.render(scene, cfg)
            bi = c.build_info()
            assert bi.builder == rtow.BUILDER_DEVICE_LBVH and bi.bvh4_nodes > 0, name
            # the node format follows the image's size by the host builder's rule: binary32 planes when the whole image
            # fits LDS beside six stack entries per lane (round 5; suzanne's device-built tree has 276 nodes against the
            # host's 261 and is staged whole with seven)
            wide = bi.bvh4_image_bytes + (64 * bi.bvh4_nodes if bi.bvh4_node_bytes == 64 else 0)  # with 128-byte nodes
            assert (bi.bvh4_node_bytes == 128) == (wide + 6 * 4096 <= 160 * 1024), (name, bi.bvh4_image_bytes, bi.bvh4_nodes)
            if name == "suzanne":
                assert bi.bvh4_node_bytes == 128 and bi.bvh4_nodes <= 280, bi.bvh4_nodes
            if node_bytes == 64:
                assert bi.bvh4_node_bytes == 64, name
            assert st.kernel_used == rtow.KERNEL_BVH4, name  # AUTO takes the 4-wide walk with the device builder too
            ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=8, accel=nt > 5000)
            assert st.segments == ost.segments, name
            assert np.array_equal(img, ref), f"{name}: {int((img != ref).sum())} values differ"
            fast = rtow.make_config(w, h, spp, 2, 20, seed=5 + k, precision=rtow.F64_FAST)
            fimg, fst = c.render(scene, fast)
            assert fst.kernel_used == rtow.KERNEL_BVH4 and np.abs(fimg - ref).mean() / spp <= 2e-3, name
    finally:
        c.close()


def test_device_trees_radix_ploc_sah_give_the_same_image_and_the_sah_tree_is_the_best(tmp_path, monkeypatch):
    """Round 5: the device builder's binary tree comes from a binned SAH build (csrc/rtow_build.hip pass 3c, the
    default), from parallel locally-ordered clustering over the Morton order (pass 3b, RTOW_PLOC_RADIUS=8|16) or from
    Karras' radix tree (pass 3, RTOW_PLOC_RADIUS=0: rounds 1-4).  Same image bit for bit (the
    closest hit is tree-independent), fewer node tests per segment, and the device-side validation has counted every
    triangle record in exactly one leaf (rtow_scene_upload fails otherwise)."""
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    v, f = make_mesh.load(GOLDEN / "suzanne.obj")
    _write_obj(tmp_path / "m3.obj", make_mesh.subdivide(v, f, 3))
    for obj, nt in ((GOLDEN / "suzanne.obj", 968), (tmp_path / "m3.obj", 8712)):
        scene = rtow.HostScene.obj(obj, 16 / 9)
        out = {}
        for radius in ("0", "8", "16", "sah"):
            if radius == "sah":
                monkeypatch.delenv("RTOW_PLOC_RADIUS", raising=False)
                monkeypatch.setenv("RTOW_DEVICE_TREE", "sah")
            else:
                monkeypatch.setenv("RTOW_PLOC_RADIUS", radius)
            c = rtow.Context(0)  # (the knob is read at context creation)
            try:
                c.set_builder(rtow.BUILDER_DEVICE_LBVH)
                for kernel in (rtow.KERNEL_BVH4, rtow.KERNEL_BVH):
                    cfg = rtow.make_config(160, 90, 4, 2, 20, seed=21, precision=rtow.F64_STRICT, kernel=kernel)
                    img, st = c.render(scene, cfg)
                    assert st.kernel_used == kernel and c.build_info().builder == rtow.BUILDER_DEVICE_LBVH
                    out[(radius, kernel)] = (img, st.segments, st.node_tests / st.segments)
            finally:
                c.close()
        monkeypatch.delenv("RTOW_PLOC_RADIUS", raising=False)
        monkeypatch.delenv("RTOW_DEVICE_TREE", raising=False)
        ref = out[("0", rtow.KERNEL_BVH4)]
        for key, (img, seg, _) in out.items():
            assert seg == ref[1] and np.array_equal(img, ref[0]), (nt, key)
        # node tests per segment of the 4-wide walk: suzanne 11.2 (radix) -> 9.9 (PLOC, radius 16: the default up to 16,384
        # primitives; the host's SAH tree: 9.0); the 3x3 subdivision is indifferent at this image size (17.4 either way)
        radix = out[("0", rtow.KERNEL_BVH4)][2]
        if nt == 968:
            assert out[("16", rtow.KERNEL_BVH4)][2] < 0.95 * radix, (radix, out[("16", rtow.KERNEL_BVH4)][2])
        for radius in ("8", "16"):
            assert out[(radius, rtow.KERNEL_BVH4)][2] < 1.03 * radix, (nt, radius)
        # the binned SAH build on the device (the default): suzanne 9.0, the host tree's number
        assert out[("sah", rtow.KERNEL_BVH4)][2] < (0.85 if nt == 968 else 1.0) * radix, (nt, radix, out[("sah", rtow.KERNEL_BVH4)][2])


def test_auto_builder_takes_the_device_for_big_meshes(tmp_path):
    """RTOW_BUILDER_AUTO, the default of a new context (include/rtow.h): rtow_render* builds a mesh of 16,384 triangles
    or more on the device — the binned SAH build of csrc/rtow_build.hip makes the host builder's tree (same node and
    triangle tests per segment) in a third of its time — and everything else on the host (a small mesh: the device's
    two dozen launches cost more than the host's 0.4 ms; sphere scenes: the grid); rtow_scene_upload, which knows no
    config, takes the host builder.  Same image either way."""
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(tmp_path / "m5.obj"), "5"], check=True,
                   capture_output=True)
    mesh = rtow.HostScene.obj(tmp_path / "m5.obj", 16 / 9)  # 24,200 triangles
    small = rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9)
    cover = rtow.HostScene.cover(11, 1.5, False)
    c = rtow.Context(0)
    try:
        low = rtow.make_config(160, 90, 4, 2, 20, seed=3, precision=rtow.F64_STRICT)
        img, st = c.render(mesh, low)
        assert c.build_info().builder == rtow.BUILDER_DEVICE_LBVH and st.kernel_used == rtow.KERNEL_BVH4
        dev_tests = (st.node_tests / st.segments, st.prim_tests / st.segments)
        c.set_builder(rtow.BUILDER_HOST_SAH)
        himg, hst = c.render(mesh, low)
        assert c.build_info().builder == rtow.BUILDER_HOST_SAH and np.array_equal(img, himg)
        # the device's tree is as good as the host's: node and triangle tests per segment within 3 %
        host_tests = (hst.node_tests / hst.segments, hst.prim_tests / hst.segments)
        assert abs(dev_tests[0] / host_tests[0] - 1) < 0.03 and abs(dev_tests[1] / host_tests[1] - 1) < 0.03, (dev_tests, host_tests)
        c.set_builder(rtow.BUILDER_AUTO)
        c.render_rgb8(mesh, rtow.make_config(640, 360, 64, 4, 20, seed=3, precision=rtow.F64_FAST))  # a longer frame: still the device
        assert c.build_info().builder == rtow.BUILDER_DEVICE_LBVH
        c.render(small, low)
        assert c.build_info().builder == rtow.BUILDER_HOST_SAH  # 968 triangles
        c.render(cover, rtow.make_config(96, 64, 4, 2, 10, seed=3, precision=rtow.F64_FAST))
        assert c.build_info().builder == rtow.BUILDER_HOST_SAH  # spheres: the grid, host-built
        c.upload(mesh)
        assert c.build_info().builder == rtow.BUILDER_HOST_SAH  # no config: host
    finally:
        c.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 9, 40])
def test_device_built_bvh4_tiny_meshes(dctx, tmp_path, n):
    """One triangle (no inner node of the radix tree), a mesh that is ONE leaf (<= 2 triangles), just above, and
    small ones: the first n triangles of suzanne, strict build against the oracle."""
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    v, f = make_mesh.load(GOLDEN / "suzanne.obj")
    tris = make_mesh.subdivide(v, f, 1)[100:100 + n]
    _write_obj(tmp_path / "t.obj", tris)
    scene = rtow.HostScene.obj(tmp_path / "t.obj", 16 / 9)
    cfg = rtow.make_config(64, 36, 4, 2, 10, seed=n, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_BVH4)
    img, st = dctx.render(scene, cfg)
    assert st.kernel_used == rtow.KERNEL_BVH4 and dctx.build_info().bvh4_nodes >= 1
    ref, ost = orc.render(scene, cfg, orc.RNG_PHILOX, nthreads=2)
    assert st.segments == ost.segments and np.array_equal(img, ref)


def test_device_built_bvh4_coincident_triangles(dctx, tmp_path):
    """Forty copies of one triangle and forty of another: equal Morton keys, the radix tree is decided by the index
    tie-break alone and is a chain — the deepest 4-wide tree per triangle there is.  Against the binary walk of the
    same scene (exact-t ties between coincident triangles are tree-dependent, so not against the oracle)."""
    sys.path.insert(0, str(REPO / "scripts"))
    import make_mesh

    v, f = make_mesh.load(GOLDEN / "suzanne.obj")
    t = make_mesh.subdivide(v, f, 1)
    tris = np.concatenate([np.repeat(t[10:11], 40, axis=0), np.repeat(t[500:501], 40, axis=0), t[:60]])
    _write_obj(tmp_path / "c.obj", tris)
    scene = rtow.HostScene.obj(tmp_path / "c.obj", 16 / 9)
    cfg4 = rtow.make_config(64, 36, 4, 2, 10, seed=3, precision=rtow.F64_FAST, kernel=rtow.KERNEL_BVH4)
    a, sa = dctx.render(scene, cfg4)
    assert sa.kernel_used == rtow.KERNEL_BVH4 and np.isfinite(a).all()
    cfg2 = rtow.make_config(64, 36, 4, 2, 10, seed=3, precision=rtow.F64_FAST, kernel=rtow.KERNEL_BVH)
    b, sb = dctx.render(scene, cfg2)
    assert sb.kernel_used == rtow.KERNEL_BVH
    assert np.abs(a - b).mean() / 4 <= 2e-3


def test_device_builder_beyond_the_bvh4_limits_takes_the_binary_walk(dctx, tmp_path):
    subprocess.run([sys.executable, str(REPO / "scripts" / "make_mesh.py"), str(tmp_path / "m17.obj"), "17"], check=True,
                   capture_output=True)
    big = rtow.HostScene.obj(tmp_path / "m17.obj", 16 / 9)
    cfg = rtow.make_config(96, 54, 8, 2, 20, seed=13, precision=rtow.F64_FAST, kernel=rtow.KERNEL_BVH4)
    a, st = dctx.render(big, cfg)
    assert st.kernel_used == rtow.KERNEL_BVH and np.isfinite(a).all() and dctx.build_info().bvh4_nodes == 0
