"""GPU: random mixed scenes (spheres of very different sizes, moving spheres, triangles; all three
materials) through every kernel, both builders and the oracle.

The oracle walks the REFERENCE's tree and tests in insertion order; the device walks its own
structures.  The strict build must still return the same image bit for bit: the closest hit does
not depend on the acceleration structure.  (Insertion order = class-major order here, so prim_kind /
prim_index describe exactly the order the oracle's tree is built from.)
"""
import ctypes as C

import numpy as np
import pytest

import orc
import rtow

pytestmark = pytest.mark.gpu


def random_scene(seed, n_sph, n_mov, n_tri, keep):
    rng = np.random.default_rng(seed)
    base = rtow.HostScene.cover(0, 1.5, False)
    g = 0 if (n_sph == 0 and n_mov == 0) else 1  # triangle-only scenes stay triangle-only (the BVH4 kernel's case)
    sph = np.zeros((n_sph + g, 4))
    if g:
        sph[0] = [0.0, -200.5, 0.0, 200.0]  # a big ground sphere among small ones
    sph[g:, :3] = rng.uniform(-3, 3, size=(n_sph, 3)) * [1, 0.4, 1] + [0, 0.6, 0]
    sph[g:, 3] = rng.choice([0.05, 0.15, 0.4, 0.9], size=n_sph, p=[0.4, 0.4, 0.15, 0.05])
    mov = np.zeros((n_mov, 8))  # c0 xyz, c1 xyz, radius, pad
    mov[:, :3] = rng.uniform(-3, 3, size=(n_mov, 3)) * [1, 0.4, 1] + [0, 0.6, 0]
    mov[:, 3:6] = mov[:, :3] + rng.uniform(-0.4, 0.4, size=(n_mov, 3))
    mov[:, 6] = rng.uniform(0.05, 0.3, size=n_mov)
    tri = np.zeros((n_tri, 9))
    c = rng.uniform(-3, 3, size=(n_tri, 3)) * [1, 0.3, 1] + [0, 0.8, 0]
    for k in range(3):
        tri[:, 3 * k:3 * k + 3] = c + rng.uniform(-0.5, 0.5, size=(n_tri, 3))
    n_mat = 8
    mats = (rtow.Material * n_mat)()
    for i in range(n_mat):
        kind = [rtow.MAT_LAMBERTIAN, rtow.MAT_METAL, rtow.MAT_DIELECTRIC][i % 3]
        mats[i].kind = kind
        mats[i].albedo = (C.c_double * 3)(*rng.uniform(0.3, 0.95, 3))
        mats[i].fuzz = float(rng.uniform(0, 0.5)) if kind == rtow.MAT_METAL else 0.0
        mats[i].ir = 1.5
    ns, nm, nt = n_sph + g, n_mov, n_tri
    smat = rng.integers(0, n_mat, max(ns, 1)).astype(np.int32)
    smat[0] = 0  # Lambertian ground
    mmat = rng.integers(0, n_mat, max(nm, 1)).astype(np.int32)
    tmat = (rng.integers(0, n_mat // 3, max(nt, 1)) * 3).astype(np.int32)  # triangles: Lambertian only
    kinds = np.concatenate([np.zeros(ns), np.ones(nm), np.full(nt, 2)]).astype(np.int32)
    index = np.concatenate([np.arange(ns), np.arange(nm), np.arange(nt)]).astype(np.int32)
    sc = rtow.Scene()
    sc.camera = base.c.camera
    sc.n_spheres, sc.n_moving, sc.n_triangles = ns, nm, nt
    sph_c, mov_c, tri_c = np.ascontiguousarray(sph), np.ascontiguousarray(mov), np.ascontiguousarray(tri)
    sc.sphere_geom = sph_c.ctypes.data_as(C.POINTER(C.c_double))
    sc.sphere_mat = smat.ctypes.data_as(C.POINTER(C.c_int32))
    sc.moving_geom = mov_c.ctypes.data_as(C.POINTER(C.c_double))
    sc.moving_mat = mmat.ctypes.data_as(C.POINTER(C.c_int32))
    sc.triangle_geom = tri_c.ctypes.data_as(C.POINTER(C.c_double))
    sc.triangle_mat = tmat.ctypes.data_as(C.POINTER(C.c_int32))
    sc.n_materials = n_mat
    sc.materials = mats
    sc.n_prims = ns + nm + nt
    sc.prim_kind = kinds.ctypes.data_as(C.POINTER(C.c_int32))
    sc.prim_index = index.ctypes.data_as(C.POINTER(C.c_int32))
    keep.extend([base, sph_c, mov_c, tri_c, mats, smat, mmat, tmat, kinds, index])
    return sc


SHAPES = [(30, 0, 0), (60, 20, 0), (0, 0, 80), (40, 10, 60), (200, 0, 0), (5, 5, 5), (120, 40, 100), (0, 0, 1500)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"s{a}_m{b}_t{c}" for a, b, c in SHAPES])
def test_random_scene_all_kernels_and_builders_match_the_oracle(ctx, shape):
    keep = []
    scene = random_scene(sum(shape) + 17, *shape, keep)
    cfg0 = rtow.make_config(72, 48, 4, 2, 12, seed=shape[0] + 3, precision=rtow.F64_STRICT)
    ref, ost = orc.render(scene, cfg0, orc.RNG_PHILOX, nthreads=4)
    dctx = rtow.Context(0)
    dctx.set_builder(rtow.BUILDER_DEVICE_LBVH)
    try:
        for c in (ctx, dctx):
            for kernel in (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_GRID, rtow.KERNEL_BVH4):  # (BVH4: triangle-only scenes, host builder; else the binary walk)
                cfg = rtow.make_config(72, 48, 4, 2, 12, seed=shape[0] + 3, precision=rtow.F64_STRICT, kernel=kernel)
                img, st = c.render(scene, cfg)
                if kernel == rtow.KERNEL_BVH4 and c is ctx and shape[0] == 0 and shape[1] == 0:
                    assert st.kernel_used == rtow.KERNEL_BVH4
                assert st.segments == ost.segments, (kernel, st.kernel_used)
                assert np.array_equal(img, ref), (kernel, st.kernel_used, int((img != ref).sum()))
            # the opt-in kernel that walks the reference's own tree: the oracle's walk, test for test
            rimg, rst = c.render(scene, rtow.make_config(72, 48, 4, 2, 12, seed=shape[0] + 3, precision=rtow.F64_STRICT,
                                                         kernel=rtow.KERNEL_REFTREE))
            assert rst.kernel_used == rtow.KERNEL_REFTREE and np.array_equal(rimg, ref) and rst.segments == ost.segments
            assert rst.node_tests == ost.node_tests and rst.prim_tests == ost.prim_tests
            # the fast and f32 builds: finite, close, and the same from both builders
            fa, _ = c.render(scene, rtow.make_config(72, 48, 16, 4, 12, seed=5, precision=rtow.F64_FAST))
            f3, _ = c.render(scene, rtow.make_config(72, 48, 16, 4, 12, seed=5, precision=rtow.F32))
            assert np.isfinite(fa).all() and np.isfinite(f3).all()
            assert np.abs(np.sqrt(np.clip(fa / 16, 0, 1)) - np.sqrt(np.clip(f3 / 16, 0, 1))).mean() < 2.0 / 255.0
    finally:
        dctx.close()
