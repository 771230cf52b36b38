"""ctypes view of the C-ABI in include/rtow.h (librtow.so).

Plumbing only: it mirrors the structs, loads the in-tree shared library and adds
thin helpers for device buffers (torch is used for device memory, streams and
torch.distributed — not for any of the rendering arithmetic).

There is no CPU fallback here by design: if librtow.so is missing or no HIP
device is usable the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
LIB_PATH = Path(os.environ.get("RTOW_LIB", PKG_DIR / "librtow.so"))  # RTOW_LIB: A/B against another build

RTOW_ABI_VERSION = 7
RTOW_OK, RTOW_EINVAL, RTOW_ENODEV, RTOW_EHIP, RTOW_ENOSCENE, RTOW_EEMPTY, RTOW_ENOMEM = 0, -1, -2, -3, -4, -5, -6
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC = 0, 1, 2
PRIM_SPHERE, PRIM_MOVING_SPHERE, PRIM_TRIANGLE = 0, 1, 2
F64_STRICT, F64_FAST, F32 = 0, 1, 2
KERNEL_AUTO, KERNEL_BRUTE, KERNEL_BVH, KERNEL_GRID, KERNEL_BVH4, KERNEL_REFTREE = 0, 1, 2, 3, 4, 5
MODEL_OO, MODEL_VARIANT, MODEL_WORLD = 0, 1, 2  # scene models of the host scene scripts (include/rtow.h)
BUILDER_HOST_SAH, BUILDER_DEVICE_LBVH, BUILDER_AUTO = 0, 1, 2

d3 = C.c_double * 3
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int32)


class Camera(C.Structure):
    _fields_ = [
        ("origin", d3), ("u", d3), ("v", d3), ("w", d3),
        ("horizontal", d3), ("vertical", d3), ("lower_left_corner", d3),
        ("lens_radius", C.c_double), ("t0", C.c_double), ("t1", C.c_double),
    ]


class Material(C.Structure):
    _fields_ = [
        ("albedo", d3), ("fuzz", C.c_double), ("ir", C.c_double),
        ("kind", C.c_int32), ("pad_", C.c_int32),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("camera", Camera),
        ("n_spheres", C.c_int32), ("sphere_geom", _pd), ("sphere_mat", _pi),
        ("n_moving", C.c_int32), ("moving_geom", _pd), ("moving_mat", _pi),
        ("n_triangles", C.c_int32), ("triangle_geom", _pd), ("triangle_mat", _pi),
        ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
        ("n_prims", C.c_int32), ("prim_kind", _pi), ("prim_index", _pi),
    ]


class Config(C.Structure):
    _fields_ = [
        ("image_width", C.c_int32), ("image_height", C.c_int32),
        ("samples_per_pixel", C.c_int32), ("nstreams", C.c_int32),
        ("max_child_rays", C.c_int32), ("precision", C.c_int32), ("kernel", C.c_int32),
        ("rank", C.c_int32), ("nranks", C.c_int32), ("tile_rows", C.c_int32),
        ("seed", C.c_uint64),
        ("stream_first", C.c_int32), ("stream_count", C.c_int32), ("accumulate", C.c_int32),
        ("pad_", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64), ("segments", C.c_uint64),
        ("prim_tests", C.c_uint64), ("node_tests", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
        ("local_rows", C.c_int32), ("kernel_used", C.c_int32),
    ]


class BuildInfo(C.Structure):
    _fields_ = [
        ("builder", C.c_int32), ("bvh_nodes", C.c_int32),
        ("bvh_image_bytes", C.c_int32), ("grid_image_bytes", C.c_int32),
        ("bvh_build_ms", C.c_double), ("grid_build_ms", C.c_double), ("upload_ms", C.c_double),
        ("bvh4_nodes", C.c_int32), ("bvh4_image_bytes", C.c_int32),
        ("ref_tree_nodes", C.c_int32), ("bvh4_node_bytes", C.c_int32),
        ("ref_tree_stupid_volume", C.c_double), ("ref_tree_build_ms", C.c_double),
    ]


class HostConfig(C.Structure):
    _fields_ = [
        ("number_of_balls_sqrt", C.c_int32), ("aspect_ratio", C.c_double),
        ("moving_spheres", C.c_int32),
    ]


# every symbol include/rtow.h declares
EXPORTS = [
    "rtow_abi_version", "rtow_last_error", "rtow_ctx_create", "rtow_ctx_destroy",
    "rtow_scene_upload", "rtow_local_rows", "rtow_local_row_list", "rtow_render_device",
    "rtow_render", "rtow_host_scene_cover", "rtow_host_scene_obj", "rtow_host_scene_free",
    "rtow_host_rng_reset", "rtow_host_ppm", "rtow_host_free", "rtow_tonemap_device",
    "rtow_profile_collect", "rtow_debug_counters", "rtow_render_rgb8",
    "rtow_ctx_set_builder", "rtow_build_info", "rtow_debug_image", "rtow_render_multi",
    "rtow_debug_schedule", "rtow_host_scene_cover_model", "rtow_host_scene_obj_model",
    "rtow_multi_create", "rtow_multi_set_builder", "rtow_multi_upload", "rtow_multi_build_info",
    "rtow_multi_render", "rtow_multi_destroy", "rtow_host_reftree_info",
    "rtow_render_device_rgb8", "rtow_multi_render_rgb8", "rtow_multi_frame_breakdown",
]
MULTI_BREAKDOWN = ("total", "handoff_enqueue", "place_enqueue", "wait_and_copy", "wait_only", "dev_trace", "dev_gather",
                   "dev_place_copy")  # RTOW_MB_* of include/rtow.h, milliseconds


class RtowError(RuntimeError):
    pass


_lib = None


def lib():
    """Load librtow.so (built in-tree by __graft_entry__.build / make)."""
    global _lib
    if _lib is not None:
        return _lib
    # Load-order guard: torch bundles its own copy of the HIP runtime, librtow.so links /opt/rocm's.  Both
    # live in one process whenever torch supplies device memory and streams (bench.py, some tests); that is
    # reliable when torch's libraries are mapped FIRST (bench.py's order) and was not the other way round
    # ("No HIP GPUs are available" / "no ROCm-capable device" from whichever runtime came second).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise RtowError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                        " (there is no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    L.rtow_abi_version.restype = C.c_int
    L.rtow_last_error.restype = C.c_char_p
    L.rtow_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.rtow_ctx_destroy.argtypes = [C.c_void_p]
    L.rtow_ctx_destroy.restype = None
    L.rtow_scene_upload.argtypes = [C.c_void_p, C.POINTER(Scene)]
    L.rtow_local_rows.argtypes = [C.POINTER(Config)]
    L.rtow_local_row_list.argtypes = [C.POINTER(Config), _pi, C.c_int32]
    L.rtow_render_device.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.c_void_p,
                                     C.POINTER(Stats)]
    L.rtow_render.argtypes = [C.c_void_p, C.POINTER(Scene), C.POINTER(Config), _pd,
                              C.POINTER(Stats)]
    L.rtow_host_scene_cover.argtypes = [C.POINTER(HostConfig), C.POINTER(C.POINTER(Scene))]
    L.rtow_host_scene_obj.argtypes = [C.POINTER(HostConfig), C.c_char_p,
                                      C.POINTER(C.POINTER(Scene))]
    if hasattr(L, "rtow_host_scene_cover_model"):
        L.rtow_host_scene_cover_model.argtypes = [C.POINTER(HostConfig), C.c_int32, C.POINTER(C.POINTER(Scene))]
        L.rtow_host_scene_obj_model.argtypes = [C.POINTER(HostConfig), C.c_char_p, C.c_int32,
                                                C.POINTER(C.POINTER(Scene))]
    L.rtow_host_scene_free.argtypes = [C.POINTER(Scene)]
    L.rtow_host_scene_free.restype = None
    L.rtow_host_rng_reset.restype = None
    L.rtow_host_ppm.argtypes = [_pd, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_char_p),
                                C.POINTER(C.c_uint64)]
    L.rtow_host_free.argtypes = [C.c_void_p]
    L.rtow_host_free.restype = None
    L.rtow_tonemap_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    L.rtow_ctx_set_builder.argtypes = [C.c_void_p, C.c_int32]
    L.rtow_build_info.argtypes = [C.c_void_p, C.POINTER(BuildInfo)]
    L.rtow_render_multi.argtypes = [C.c_int32, _pi, C.POINTER(Scene), C.POINTER(Config), _pd, C.POINTER(Stats),
                                    C.c_int32]
    L.rtow_debug_image.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    if hasattr(L, "rtow_host_reftree_info"):
        L.rtow_host_reftree_info.argtypes = [C.POINTER(Scene), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_double)]
    if hasattr(L, "rtow_multi_create"):
        L.rtow_multi_create.argtypes = [C.c_int32, _pi, C.c_int32, C.POINTER(C.c_void_p)]
        L.rtow_multi_set_builder.argtypes = [C.c_void_p, C.c_int32]
        L.rtow_multi_upload.argtypes = [C.c_void_p, C.POINTER(Scene)]
        L.rtow_multi_build_info.argtypes = [C.c_void_p, C.POINTER(BuildInfo)]
        L.rtow_multi_render.argtypes = [C.c_void_p, C.POINTER(Config), _pd, C.POINTER(Stats)]
        L.rtow_multi_destroy.argtypes = [C.c_void_p]
        L.rtow_multi_destroy.restype = None
    if hasattr(L, "rtow_multi_render_rgb8"):
        L.rtow_multi_render_rgb8.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.POINTER(Stats)]
        L.rtow_render_device_rgb8.argtypes = [C.c_void_p, C.POINTER(Config), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    L.rtow_render_rgb8.argtypes = [C.c_void_p, C.POINTER(Scene), C.POINTER(Config), C.c_void_p, C.POINTER(Stats)]
    L.rtow_profile_collect.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    if hasattr(L, "rtow_debug_schedule"):  # (absent from older builds loaded through RTOW_LIB for A/B runs)
        L.rtow_debug_schedule.argtypes = [C.c_void_p, C.POINTER(Config), C.POINTER(C.c_uint32), C.c_int32]
    ver = L.rtow_abi_version()
    if ver != RTOW_ABI_VERSION and not ("RTOW_LIB" in os.environ and ver >= 4):  # (older builds: A/B runs only)
        raise RtowError("librtow.so ABI version mismatch")
    _lib = L
    return L


def check(rc: int, what: str = "rtow"):
    if rc != 0:
        msg = lib().rtow_last_error()
        raise RtowError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def image_height(width: int, aspect_ratio: float) -> int:
    """int(image_width / aspect_ratio) — reference src/render.cpp:137."""
    return int(width / aspect_ratio)


def make_config(width, height, spp, nstreams=1, max_child_rays=50, seed=1, precision=F64_FAST,
                kernel=KERNEL_AUTO, rank=0, nranks=1, tile_rows=8, stream_first=0, stream_count=0,
                accumulate=0) -> Config:
    return Config(width, height, spp, nstreams, max_child_rays, precision, kernel, rank, nranks,
                  tile_rows, seed, stream_first, stream_count, accumulate, 0)


def spp_effective(cfg: Config) -> int:
    """samples_per_pixel / nthreads * nthreads — reference src/render.cpp:185."""
    return cfg.samples_per_pixel // cfg.nstreams * cfg.nstreams


def local_rows(cfg: Config):
    L = lib()
    n = L.rtow_local_rows(C.byref(cfg))
    if n < 0:
        check(n, "rtow_local_rows")
    buf = (C.c_int32 * max(n, 1))()
    got = L.rtow_local_row_list(C.byref(cfg), buf, n)
    if got < 0:
        check(got, "rtow_local_row_list")
    return list(buf[:n])


class HostScene:
    """A flattened scene built by the product's host-side scene scripts."""

    def __init__(self, ptr):
        self.ptr = ptr

    @property
    def c(self) -> Scene:
        return self.ptr.contents

    @classmethod
    def cover(cls, nsqrt=11, aspect=1.5, moving=False, reset_rng=True, model=MODEL_OO):
        """`model`: which of the reference's scene models the script builds (MODEL_OO / _VARIANT / _WORLD)."""
        L = lib()
        if reset_rng:
            L.rtow_host_rng_reset()
        hc = HostConfig(nsqrt, aspect, int(moving))
        out = C.POINTER(Scene)()
        if model == MODEL_OO:
            check(L.rtow_host_scene_cover(C.byref(hc), C.byref(out)), "rtow_host_scene_cover")
        else:
            check(L.rtow_host_scene_cover_model(C.byref(hc), model, C.byref(out)), "rtow_host_scene_cover_model")
        return cls(out)

    @classmethod
    def obj(cls, path, aspect=16.0 / 9.0, reset_rng=True, model=MODEL_OO):
        L = lib()
        if reset_rng:
            L.rtow_host_rng_reset()
        hc = HostConfig(0, aspect, 0)
        out = C.POINTER(Scene)()
        if model == MODEL_OO:
            check(L.rtow_host_scene_obj(C.byref(hc), str(path).encode(), C.byref(out)), "rtow_host_scene_obj")
        else:
            check(L.rtow_host_scene_obj_model(C.byref(hc), str(path).encode(), model, C.byref(out)),
                  "rtow_host_scene_obj_model")
        return cls(out)

    def close(self):
        if self.ptr:
            lib().rtow_host_scene_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ppm_text(rgb_sums, width, height, spp_eff) -> bytes:
    """write_color + P3 framing (reference src/render.cpp:11-20,182-186)."""
    import numpy as np

    a = np.ascontiguousarray(rgb_sums, dtype=np.float64)
    assert a.size == width * height * 3
    txt = C.c_char_p()
    n = C.c_uint64()
    L = lib()
    check(L.rtow_host_ppm(a.ctypes.data_as(_pd), width, height, spp_eff, C.byref(txt), C.byref(n)),
          "rtow_host_ppm")
    try:
        return C.string_at(txt, n.value)
    finally:
        L.rtow_host_free(txt)


class Context:
    """One HIP device (one process per GPU)."""

    def __init__(self, device_id: int = 0):
        self._h = C.c_void_p()
        check(lib().rtow_ctx_create(device_id, C.byref(self._h)), "rtow_ctx_create")
        self.device_id = device_id

    def upload(self, scene):
        s = scene.c if isinstance(scene, HostScene) else scene
        check(lib().rtow_scene_upload(self._h, C.byref(s)), "rtow_scene_upload")

    def set_builder(self, builder: int):
        """BUILDER_AUTO (default of a new context), BUILDER_HOST_SAH or BUILDER_DEVICE_LBVH; applies from the next upload."""
        check(lib().rtow_ctx_set_builder(self._h, builder), "rtow_ctx_set_builder")

    def debug_image(self, which: int) -> bytes:
        """A resident scene image (0 BVH, 1 grid, 2/3 the f32 build's), for tests."""
        n = C.c_int64()
        check(lib().rtow_debug_image(self._h, which, None, 0, C.byref(n)), "rtow_debug_image")
        buf = (C.c_ubyte * max(n.value, 1))()
        check(lib().rtow_debug_image(self._h, which, buf, n.value, C.byref(n)), "rtow_debug_image")
        return bytes(buf[:n.value])

    def build_info(self) -> BuildInfo:
        bi = BuildInfo()
        check(lib().rtow_build_info(self._h, C.byref(bi)), "rtow_build_info")
        return bi

    def render_device(self, cfg: Config, d_ptr: int, stream: int = 0, want_stats=False):
        st = Stats() if want_stats else None
        check(lib().rtow_render_device(self._h, C.byref(cfg), C.c_void_p(d_ptr),
                                       C.c_void_p(stream), C.byref(st) if st is not None else None),
              "rtow_render_device")
        return st

    def tonemap_device(self, d_sums: int, n_values: int, spp_eff: int, d_rgb8: int, stream: int = 0):
        """write_color on the device: 8-bit RGB from radiance sums (both device pointers)."""
        check(lib().rtow_tonemap_device(self._h, C.c_void_p(d_sums), n_values, spp_eff,
                                        C.c_void_p(d_rgb8), C.c_void_p(stream)), "rtow_tonemap_device")

    def render(self, scene, cfg: Config, into=None):
        """Upload + render + D2H: returns (numpy [rows, W, 3] float64 sums, Stats).  `into`: an
        existing sums array to accumulate onto (cfg.accumulate)."""
        import numpy as np

        s = scene.c if isinstance(scene, HostScene) else scene
        rows = lib().rtow_local_rows(C.byref(cfg))
        if rows < 0:
            check(rows, "rtow_local_rows")
        out = np.zeros((rows, cfg.image_width, 3), dtype=np.float64) if into is None else into
        st = Stats()
        check(lib().rtow_render(self._h, C.byref(s), C.byref(cfg), out.ctypes.data_as(_pd),
                                C.byref(st)), "rtow_render")
        return out, st

    def render_rgb8(self, scene, cfg: Config, want_stats=True):
        """Upload + render + write_color on the device + D2H of the bytes: (numpy [rows, W, 3] uint8, Stats | None)."""
        import numpy as np

        s = scene.c if isinstance(scene, HostScene) else scene
        rows = lib().rtow_local_rows(C.byref(cfg))
        if rows < 0:
            check(rows, "rtow_local_rows")
        out = np.zeros((rows, cfg.image_width, 3), dtype=np.uint8)
        st = Stats() if want_stats else None
        check(lib().rtow_render_rgb8(self._h, C.byref(s), C.byref(cfg), out.ctypes.data_as(C.c_void_p),
                                     C.byref(st) if st is not None else None), "rtow_render_rgb8")
        return out, st

    def render_device_rgb8(self, cfg: Config, d_ptr: int, stream: int = 0, want_stats=False):
        """This rank's rows as bytes on the device (write_color fused into the reduce kernel)."""
        st = Stats() if want_stats else None
        check(lib().rtow_render_device_rgb8(self._h, C.byref(cfg), C.c_void_p(d_ptr), C.c_void_p(stream),
                                            C.byref(st) if st is not None else None), "rtow_render_device_rgb8")
        return st

    def profile_collect(self):
        """(trace-kernel ms since the last collect, launches); raises if a launch dropped samples."""
        ms, n = C.c_double(), C.c_int32()
        check(lib().rtow_profile_collect(self._h, C.byref(ms), C.byref(n)), "rtow_profile_collect")
        return ms.value, n.value

    def close(self):
        if self._h:
            lib().rtow_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_multi(device_ids, scene, cfg: Config, use_rccl=True):
    """One frame over several devices from this process (rtow_render_multi): returns ([H, W, 3] sums, Stats)."""
    import numpy as np

    s = scene.c if isinstance(scene, HostScene) else scene
    ids = (C.c_int32 * len(device_ids))(*device_ids)
    out = np.zeros((cfg.image_height, cfg.image_width, 3), dtype=np.float64)
    st = Stats()
    check(lib().rtow_render_multi(len(device_ids), ids, C.byref(s), C.byref(cfg), out.ctypes.data_as(_pd), C.byref(st),
                                  int(bool(use_rccl))), "rtow_render_multi")
    return out, st


class MultiContext:
    """Persistent multi-device handle (rtow_multi_*): contexts, streams, buffers, worker threads and the RCCL
    communicator live as long as the object; upload once, render many frames."""

    def __init__(self, device_ids, use_rccl=True):
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        check(lib().rtow_multi_create(len(device_ids), ids, int(bool(use_rccl)), C.byref(self._h)), "rtow_multi_create")
        self.n = len(device_ids)

    def upload(self, scene):
        s = scene.c if isinstance(scene, HostScene) else scene
        check(lib().rtow_multi_upload(self._h, C.byref(s)), "rtow_multi_upload")

    def build_info(self) -> BuildInfo:
        bi = BuildInfo()
        check(lib().rtow_multi_build_info(self._h, C.byref(bi)), "rtow_multi_build_info")
        return bi

    def render(self, cfg: Config, want_stats=True):
        import numpy as np

        out = np.zeros((cfg.image_height, cfg.image_width, 3), dtype=np.float64)
        st = Stats() if want_stats else None
        check(lib().rtow_multi_render(self._h, C.byref(cfg), out.ctypes.data_as(_pd),
                                      C.byref(st) if st is not None else None), "rtow_multi_render")
        return out, st

    def render_rgb8(self, cfg: Config, want_stats=True):
        """The frame as the PPM's bytes: write_color by the owning rank, bytes gathered, rows placed on the first
        device, one copy.  ([H, W, 3] uint8, Stats | None)"""
        import numpy as np

        out = np.zeros((cfg.image_height, cfg.image_width, 3), dtype=np.uint8)
        st = Stats() if want_stats else None
        check(lib().rtow_multi_render_rgb8(self._h, C.byref(cfg), out.ctypes.data_as(C.c_void_p),
                                           C.byref(st) if st is not None else None), "rtow_multi_render_rgb8")
        return out, st

    def frame_breakdown(self) -> dict:
        """Where the last frame's time went (rtow_multi_frame_breakdown), milliseconds by name."""
        v = (C.c_double * len(MULTI_BREAKDOWN))()
        L = lib()
        L.rtow_multi_frame_breakdown.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32]
        n = L.rtow_multi_frame_breakdown(self._h, v, len(MULTI_BREAKDOWN))
        if n < 0:
            check(n, "rtow_multi_frame_breakdown")
        return {k: float(v[i]) for i, k in enumerate(MULTI_BREAKDOWN[:n])}

    def close(self):
        if self._h:
            lib().rtow_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def have_gpu() -> bool:
    """True when a HIP device can be bound (used by tests to choose markers only)."""
    try:
        c = Context(0)
        c.close()
        return True
    except Exception:
        return False
