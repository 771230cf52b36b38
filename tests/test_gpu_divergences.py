"""The documented divergences from the reference's TREE, each constructed on purpose.

The device finds closest hits through its own acceleration structures, so wherever the reference's
result depends on its BVH (src/render.cpp:52-110) rather than on its primitive tests, the device does
what the primitive tests alone would do.  Every case states what the reference does (through the
oracle, which restates its tree, boxes and slab test), what the device does, and checks that all
three device kernels agree with each other (STREAM tests every primitive with no boxes at all).
None of these occurs in the reference's own scenes."""
import ctypes as C
import math

import numpy as np
import pytest

import orc
import rtow

pytestmark = pytest.mark.gpu


def camera(lookfrom, lookat, vup, vfov, aspect, aperture=0.0, focus=None, t0=0.0, t1=1.0):
    """Camera::Camera (src/common-model.cpp:136-154) for the test scenes (both sides get the same struct)."""
    lf, la, up = (np.array(x, dtype=np.float64) for x in (lookfrom, lookat, vup))
    norm = lambda v: v * (1.0 / math.sqrt(float(v @ v)))
    w = norm(lf - la)
    u = norm(np.cross(up, w))
    v = norm(np.cross(w, u))
    vh = 2.0 * math.tan(vfov * math.pi / 180 / 2)
    vw = aspect * vh
    fd = focus if focus is not None else math.sqrt(float((lf - la) @ (lf - la)))
    hor, ver = fd * vw * u, fd * vh * v
    llc = lf - hor / 2.0 - ver / 2.0 - fd * w
    cam = rtow.Camera()
    for name, val in (("origin", lf), ("u", u), ("v", v), ("w", w), ("horizontal", hor), ("vertical", ver),
                      ("lower_left_corner", llc)):
        setattr(cam, name, (C.c_double * 3)(*val))
    cam.lens_radius, cam.t0, cam.t1 = aperture / 2, t0, t1
    return cam


def scene_of(cam, spheres=(), triangles=(), materials=()):
    """spheres: (cx, cy, cz, r, mat); triangles: (9 coords..., mat); materials: (kind, albedo, fuzz, ir).
    Insertion order: spheres first, then triangles."""
    keep = {}
    sc = rtow.Scene()
    sc.camera = cam
    mats = (rtow.Material * len(materials))()
    for i, (kind, alb, fuzz, ir) in enumerate(materials):
        mats[i].kind, mats[i].fuzz, mats[i].ir = kind, fuzz, ir
        mats[i].albedo = (C.c_double * 3)(*alb)
    sg = np.ascontiguousarray([s[:4] for s in spheres], dtype=np.float64).reshape(-1, 4)
    sm = np.ascontiguousarray([s[4] for s in spheres], dtype=np.int32)
    tg = np.ascontiguousarray([t[:9] for t in triangles], dtype=np.float64).reshape(-1, 9)
    tm = np.ascontiguousarray([t[9] for t in triangles], dtype=np.int32)
    n = len(spheres) + len(triangles)
    kinds = np.array([rtow.PRIM_SPHERE] * len(spheres) + [rtow.PRIM_TRIANGLE] * len(triangles), dtype=np.int32)
    index = np.array(list(range(len(spheres))) + list(range(len(triangles))), dtype=np.int32)
    keep.update(mats=mats, sg=sg, sm=sm, tg=tg, tm=tm, kinds=kinds, index=index)
    pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    sc.n_spheres, sc.sphere_geom, sc.sphere_mat = len(spheres), sg.ctypes.data_as(pd), sm.ctypes.data_as(pi)
    sc.n_triangles, sc.triangle_geom, sc.triangle_mat = len(triangles), tg.ctypes.data_as(pd), tm.ctypes.data_as(pi)
    sc.n_materials, sc.materials = len(materials), mats
    sc.n_prims, sc.prim_kind, sc.prim_index = n, kinds.ctypes.data_as(pi), index.ctypes.data_as(pi)
    return sc, keep


LAMB = lambda rgb: (rtow.MAT_LAMBERTIAN, rgb, 0.0, 0.0)
KERNELS = (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_GRID)
TRI_KERNELS = (rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_GRID, rtow.KERNEL_BVH4)  # triangle-only scenes


def device_images(ctx, sc, cfgargs, kernels=KERNELS):
    out = []
    for k in kernels:
        cfg = rtow.make_config(*cfgargs, precision=rtow.F64_STRICT, kernel=k)
        img, st = ctx.render(sc, cfg)
        out.append((img, st))
    for img, st in out[1:]:
        assert np.array_equal(img, out[0][0]) and st.segments == out[0][1].segments
    return out[0]


def reftree_image(ctx, sc, cfgargs):
    """The opt-in kernel that walks the reference's own tree (RTOW_KERNEL_REFTREE, strict build)."""
    return ctx.render(sc, rtow.make_config(*cfgargs, precision=rtow.F64_STRICT, kernel=rtow.KERNEL_REFTREE))


def test_zero_thickness_leaf_box_the_reference_never_hits_a_flat_leaf(ctx):
    """Aabb::hit rejects with `t_max <= t_min` (src/common-model.h:80), and a leaf's box is
    the union of its primitives' boxes with the default box at the origin (src/render.cpp:76-78).  A
    leaf whose triangles all lie in a coordinate plane THROUGH THE ORIGIN therefore has a box of zero
    thickness, which no ray ever enters: the reference renders such a scene as empty sky.
    Device: hits the triangle (its boxes are padded; STREAM uses no boxes) — what Triangle::hit alone
    says.  Moved off the plane by 0.25 the box has thickness and both sides agree bit for bit."""
    cam = camera((0.3, 0.2, 3.0), (0, 0, 0), (0, 1, 0), 40.0, 1.5)
    args = (96, 64, 8, 2, 10, 3)
    flat, k1 = scene_of(cam, triangles=[(-1, -1, 0, 1, -1, 0, 0, 1, 0, 0)], materials=[LAMB((0.8, 0.3, 0.3))])
    ref, ost = orc.render(flat, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
    assert ost.segments == ost.samples  # the reference: every ray misses, pure sky
    img, st = device_images(ctx, flat, args, kernels=TRI_KERNELS)
    assert st.segments > st.samples and not np.array_equal(img, ref)  # the device: the triangle is there
    rimg, rst = reftree_image(ctx, flat, args)  # ... unless it is told to walk the reference's tree
    assert np.array_equal(rimg, ref) and rst.segments == ost.segments
    # the device image is what the reference's own hit test gives once its box has thickness:
    # the same triangle and camera translated by +0.25 in z (exactly representable)
    cam2 = camera((0.3, 0.2, 3.25), (0, 0, 0.25), (0, 1, 0), 40.0, 1.5)
    lifted, k2 = scene_of(cam2, triangles=[(-1, -1, 0.25, 1, -1, 0.25, 0, 1, 0.25, 0)], materials=[LAMB((0.8, 0.3, 0.3))])
    ref2, ost2 = orc.render(lifted, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
    img2, st2 = device_images(ctx, lifted, args, kernels=TRI_KERNELS)
    assert np.array_equal(img2, ref2) and st2.segments == ost2.segments
    assert st2.segments > st2.samples


def test_exact_ties_in_t_resolve_like_the_reference_leaf_scan(ctx):
    """Two coincident primitives with different materials: every hit is an exact tie in t.  The reference's
    leaf scan accepts `t <= tmax` (src/common-model.cpp:76,115 via the shrinking tmax of src/render.cpp:57-65),
    so the LATER primitive in ITS traversal order wins; with <= 6 primitives the whole scene is one leaf in
    insertion order, in larger scenes the order is whatever its std::sort left.
    Device: the same `<=` rule in each kernel's own test order.  STREAM (ascending primitive id) and GRID
    (large list, then cells in ascending id) reproduce the reference's one-leaf order exactly.  The BVH kernel
    tests leaves in its tree's fixed depth-first order, so ONE of the two coincident primitives wins for every
    ray — the image is the reference image of one of the two insertion orders (checked: which one depends on
    the builder, as it does on std::sort in the reference; neither is specified behaviour)."""
    cam = camera((0, 0.5, 4.0), (0, 0, 0), (0, 1, 0), 35.0, 1.5)
    args = (90, 60, 6, 2, 8, 11)
    red, blue = LAMB((0.9, 0.1, 0.1)), LAMB((0.1, 0.1, 0.9))
    tri = (-1.5, -1, 0.5, 1.5, -1, 0.5, 0, 1.2, 0.5)

    def case(make, exact_kernels):
        refs, bvh = [], []
        for order in ((0, 1), (1, 0)):
            sc, keep = make(order)
            ref, ost = orc.render(sc, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
            img, st = device_images(ctx, sc, args, kernels=exact_kernels)
            assert np.array_equal(img, ref) and st.segments == ost.segments
            for k in (rtow.KERNEL_BVH, rtow.KERNEL_BVH4):  # (BVH4 runs the binary walk when the scene has spheres)
                b, bst = ctx.render(sc, rtow.make_config(*args, precision=rtow.F64_STRICT, kernel=k))
                assert bst.segments == ost.segments
                bvh.append(b)
            rimg, rst = reftree_image(ctx, sc, args)  # the reference's own leaf order: its winner, in every case
            assert np.array_equal(rimg, ref) and rst.segments == ost.segments
            refs.append(ref)
        assert not np.array_equal(refs[0], refs[1])  # the winner really is the later one
        for b in bvh:
            assert np.array_equal(b, refs[0]) or np.array_equal(b, refs[1])

    case(lambda o: scene_of(cam, triangles=[tri + (o[0],), tri + (o[1],)], materials=[red, blue]),
         (rtow.KERNEL_BRUTE,))
    # coincident spheres over a ground sphere (the grid lists both in every cell they touch)
    case(lambda o: scene_of(cam, spheres=[(0, 0, 0, 0.8, o[0]), (0, 0, 0, 0.8, o[1]), (0, -100.8, 0, 100, 0)],
                            materials=[red, blue]),
         (rtow.KERNEL_BRUTE, rtow.KERNEL_GRID))


def test_float_rounded_triangle_box_drops_a_stripe_in_the_reference(ctx):
    """Triangle::bounding_box goes through glm::vec3 = float (src/common-model.cpp:127-134), so
    a box corner is the vertex ROUNDED TO NEAREST float and can lie inside the triangle.  At coordinates
    near 2^24 the float grid is 1.0 wide: a vertex at x = 2^24 + 1 rounds to 2^24 and the reference's box
    ends one unit short of the triangle: a ray that starts beyond the box (x > 2^24) and runs away from it
    never enters the box, although it hits the triangle's last unit-wide stripe.
    Device: exact f64 bounds (padded) — the whole triangle, identical on all kernels.  Shifted by -1 in x
    (every vertex representable in float) the box is exact and both sides agree bit for bit."""
    X = float(2**24)
    z0 = -20.0
    args = (120, 80, 4, 1, 5, 2)

    def build(shift):
        # the camera sits at x = 2^24 + 0.5: rays into the right half of the image have d.x > 0
        cam = camera((X + 0.5 + shift, 0, 0), (X + 0.5 + shift, 0, z0), (0, 1, 0), 40.0, 1.5)
        tri = (X - 7 + shift, -4, z0, X + 1 + shift, -4, z0, X + 1 + shift, 4, z0, 0)
        return scene_of(cam, triangles=[tri], materials=[LAMB((0.2, 0.8, 0.2))])

    sc, keep = build(0.0)
    assert float(np.float32(X + 1)) == X  # the premise: 2^24 + 1 is not a float
    ref, ost = orc.render(sc, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
    img, st = device_images(ctx, sc, args, kernels=(rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_BVH4))
    assert st.segments > ost.segments  # the device also hits the stripe the reference's box cuts off
    # the difference is confined to the columns whose rays reach the triangle's plane at x > 2^24 (the
    # half-unit stripe right of the camera axis is ~3 pixels wide each way: columns 56..62 of 120);
    # everywhere else the rays enter the reference's box and both sides agree bit for bit
    diff = np.any(img != ref, axis=(0, 2))
    assert diff[56:63].any() and not diff[:55].any() and not diff[64:].any()
    rimg, rst = reftree_image(ctx, sc, args)  # the reference's float-rounded box: its stripe missing too
    assert np.array_equal(rimg, ref) and rst.segments == ost.segments
    sc1, keep1 = build(-1.0)
    ref1, ost1 = orc.render(sc1, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
    img1, st1 = device_images(ctx, sc1, args, kernels=(rtow.KERNEL_BRUTE, rtow.KERNEL_BVH, rtow.KERNEL_BVH4))
    assert np.array_equal(img1, ref1) and st1.segments == ost1.segments


def test_reference_tree_kernel_equals_the_oracles_reference_tree_everywhere(ctx):
    """RTOW_KERNEL_REFTREE against the oracle's restatement of the reference tree (accel = 0: its median-split
    build, boxes and Aabb::hit), bit for bit, on the reference's own scene classes (sort ties and all), on a
    negative-radius sphere (signed-radius box, src/common-model.cpp:168-171: the reference's own tree loses the
    hollow sphere's far side where the default kernels keep it) and through the `rtweekend` CLI; the tree's
    "stupid volume" diagnostic equals the survey's numbers for the reference."""
    import subprocess

    from conftest import GOLDEN, REPO

    cases = [(rtow.HostScene.cover(11, 1.5, False), orc.OrcScene.cover(11, 1.5, False), (120, 80, 4, 2, 50, 7), 2150.93),
             (rtow.HostScene.cover(11, 1.5, True), orc.OrcScene.cover(11, 1.5, True), (120, 80, 4, 2, 50, 8), None),
             (rtow.HostScene.obj(GOLDEN / "suzanne.obj", 16 / 9), orc.OrcScene.obj(GOLDEN / "suzanne.obj", 16 / 9),
              (96, 54, 4, 2, 20, 9), 34.6011)]
    for scene, oscene, args, volume in cases:
        ref, ost = orc.render(oscene, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
        img, st = reftree_image(ctx, scene, args)
        assert np.array_equal(img, ref) and st.segments == ost.segments
        assert st.kernel_used == rtow.KERNEL_REFTREE
        # the reference's walk visits what the oracle's walk visits
        assert st.node_tests == ost.node_tests and st.prim_tests == ost.prim_tests
        if volume is not None:
            assert abs(ctx.build_info().ref_tree_stupid_volume - volume) < 5e-3 * max(1.0, volume / 1000)
    # negative radius: glass shell (outer r = 0.5, inner r = -0.45) over a ground sphere
    cam = camera((0, 0.4, 3.0), (0, 0, 0), (0, 1, 0), 35.0, 1.5)
    glass = (rtow.MAT_DIELECTRIC, (1, 1, 1), 0.0, 1.5)
    sc, keep = scene_of(cam, spheres=[(0, -100.5, 0, 100, 0), (0, 0, 0, 0.5, 1), (0, 0, 0, -0.45, 1)],
                        materials=[LAMB((0.8, 0.8, 0.0)), glass])
    args = (90, 60, 8, 2, 20, 4)
    ref, ost = orc.render(sc, rtow.make_config(*args), orc.RNG_PHILOX, nthreads=4)
    img, st = reftree_image(ctx, sc, args)
    assert np.array_equal(img, ref) and st.segments == ost.segments
    # fast build: refused, not silently replaced
    with pytest.raises(rtow.RtowError):
        ctx.render(sc, rtow.make_config(*args, precision=rtow.F64_FAST, kernel=rtow.KERNEL_REFTREE))
    exe = REPO / "raytracing-one-weekend_amd" / "rtweekend"
    r = subprocess.run([str(exe), "-w", "90", "-a", "1.5", "-s", "6", "-t", "3", "-c", "12", "-n", "4", "-m", "--seed", "77",
                        "--precision", "strict", "--kernel", "reftree"], capture_output=True, check=True)
    cfg = rtow.make_config(90, 60, 6, 3, 12, seed=77)
    ref, _ = orc.render(orc.OrcScene.cover(4, 1.5, True), cfg, orc.RNG_PHILOX, nthreads=4)
    assert r.stdout == orc.ppm_text(ref, 90, 60, 6)
    assert b"Total BVH stupid volume" in r.stderr
