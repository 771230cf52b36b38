"""ctypes binding of the CPU oracle (oracle/liboracle.so) — test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "raytracing-one-weekend_amd"))
import rtow  # noqa: E402  (struct mirrors of include/rtow.h only)

ORC_DIR = REPO / "oracle"
ORC_LIB = ORC_DIR / "liboracle.so"
RNG_MT19937, RNG_PHILOX = 0, 1

_pd = C.POINTER(C.c_double)


class OrcStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("segments", C.c_uint64), ("prim_tests", C.c_uint64),
        ("node_tests", C.c_uint64), ("rng_doubles", C.c_uint64),
        ("bvh_stupid_volume", C.c_double), ("bvh_nodes", C.c_int32), ("bvh_leaves", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src_newer = (not ORC_LIB.exists()) or any(
        (ORC_DIR / f).stat().st_mtime > ORC_LIB.stat().st_mtime
        for f in ("rtow_oracle.cpp", "rtow_oracle.h"))
    if src_newer:
        subprocess.run(["make", "-C", str(ORC_DIR)], check=True, capture_output=True)
    L = C.CDLL(str(ORC_LIB))
    L.orc_mt_reset.restype = None
    L.orc_mt_burn.argtypes = [C.c_uint64]
    L.orc_mt_burn.restype = None
    L.orc_scene_cover.argtypes = [C.c_int, C.c_double, C.c_int, C.POINTER(C.POINTER(rtow.Scene))]
    L.orc_scene_obj.argtypes = [C.c_char_p, C.c_double, C.POINTER(C.POINTER(rtow.Scene))]
    L.orc_scene_free.argtypes = [C.POINTER(rtow.Scene)]
    L.orc_scene_free.restype = None
    L.orc_render.argtypes = [C.POINTER(rtow.Scene), C.POINTER(rtow.Config), C.c_int, C.c_int, _pd,
                             C.POINTER(OrcStats)]
    L.orc_render_ex.argtypes = [C.POINTER(rtow.Scene), C.POINTER(rtow.Config), C.c_int, C.c_int, C.c_int, _pd,
                                C.POINTER(OrcStats)]
    L.orc_ppm.argtypes = [_pd, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p),
                          C.POINTER(C.c_uint64)]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_free.restype = None
    L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
    L.orc_philox4x32_10.restype = None
    L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), C.c_int]
    L.orc_philox4x32.restype = None
    L.orc_philox_request.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                     C.c_int]
    L.orc_philox_request.restype = C.c_double
    L.orc_mt_random_double.argtypes = [C.c_double, C.c_double]
    L.orc_mt_random_double.restype = C.c_double
    d3 = _pd
    L.orc_sphere_hit.argtypes = [d3, C.c_double, d3, d3, C.c_double, C.c_double, _pd, _pd, _pd,
                                 C.POINTER(C.c_int)]
    L.orc_triangle_hit.argtypes = [d3, d3, d3, d3, d3, C.c_double, C.c_double, _pd, _pd, _pd]
    L.orc_aabb_hit.argtypes = [d3, d3, d3, d3, C.c_double, C.c_double]
    _lib = L
    return L


class OrcScene:
    def __init__(self, ptr):
        self.ptr = ptr

    @property
    def c(self):
        return self.ptr.contents

    @classmethod
    def cover(cls, nsqrt=11, aspect=1.5, moving=False, reset_rng=True, burn=0):
        L = lib()
        if reset_rng:
            L.orc_mt_reset()
        if burn:
            L.orc_mt_burn(burn)
        out = C.POINTER(rtow.Scene)()
        rc = L.orc_scene_cover(nsqrt, aspect, int(moving), C.byref(out))
        assert rc == 0, rc
        return cls(out)

    @classmethod
    def obj(cls, path, aspect=16.0 / 9.0, reset_rng=True):
        L = lib()
        if reset_rng:
            L.orc_mt_reset()
        out = C.POINTER(rtow.Scene)()
        rc = L.orc_scene_obj(str(path).encode(), aspect, C.byref(out))
        assert rc == 0, rc
        return cls(out)

    def close(self):
        if self.ptr:
            lib().orc_scene_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(scene, cfg: rtow.Config, rng_mode=RNG_PHILOX, nthreads=1, into=None, accel=False):
    """Returns (sums [rows, W, 3] float64, OrcStats).  `into`: sums to accumulate onto.
    `accel`: closest hit through the checker's own tree instead of the reference's (same image,
    see oracle/rtow_oracle.h orc_render_ex; for renders at BASELINE sizes)."""
    s = scene.c if hasattr(scene, "c") else scene
    tile = max(cfg.tile_rows, 1)
    nr = max(cfg.nranks, 1)
    rows = sum(1 for i in range(cfg.image_height) if (i // tile) % nr == cfg.rank)
    out = np.zeros((rows, cfg.image_width, 3), dtype=np.float64) if into is None else into
    st = OrcStats()
    rc = lib().orc_render_ex(C.byref(s), C.byref(cfg), rng_mode, nthreads, int(bool(accel)),
                             out.ctypes.data_as(_pd), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return out, st


def ppm_text(rgb_sums, width, height, spp_eff) -> bytes:
    a = np.ascontiguousarray(rgb_sums, dtype=np.float64)
    txt = C.c_char_p()
    n = C.c_uint64()
    L = lib()
    rc = L.orc_ppm(a.ctypes.data_as(_pd), width, height, spp_eff, C.byref(txt), C.byref(n))
    assert rc == 0
    try:
        return C.string_at(txt, n.value)
    finally:
        L.orc_free(txt)


def scene_arrays(s: rtow.Scene) -> dict:
    """Copy a flattened scene into numpy arrays (for comparisons)."""
    def arr(p, n, dt):
        if n == 0:
            return np.zeros((0,), dtype=dt)
        return np.ctypeslib.as_array(p, shape=(n,)).astype(dt).copy()

    mats = np.zeros((s.n_materials, 6), dtype=np.float64)
    for i in range(s.n_materials):
        m = s.materials[i]
        mats[i] = [m.albedo[0], m.albedo[1], m.albedo[2], m.fuzz, m.ir, m.kind]
    cam = s.camera
    camv = np.array(list(cam.origin) + list(cam.u) + list(cam.v) + list(cam.w) +
                    list(cam.horizontal) + list(cam.vertical) + list(cam.lower_left_corner) +
                    [cam.lens_radius, cam.t0, cam.t1])
    return dict(
        camera=camv,
        sphere_geom=arr(s.sphere_geom, s.n_spheres * 4, np.float64),
        sphere_mat=arr(s.sphere_mat, s.n_spheres, np.int32),
        moving_geom=arr(s.moving_geom, s.n_moving * 8, np.float64),
        moving_mat=arr(s.moving_mat, s.n_moving, np.int32),
        triangle_geom=arr(s.triangle_geom, s.n_triangles * 9, np.float64),
        triangle_mat=arr(s.triangle_mat, s.n_triangles, np.int32),
        materials=mats,
        prim_kind=arr(s.prim_kind, s.n_prims, np.int32),
        prim_index=arr(s.prim_index, s.n_prims, np.int32),
    )
