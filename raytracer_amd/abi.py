"""ctypes binding of the C ABI (include/rt_abi.h) in raytracer_amd/librt_amd.so.

This is plumbing only: every computation happens in the HIP library.  There is no
CPU fallback -- if the library is missing, or no GPU is present when a context is
created, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .scene import CameraArgs, CameraDesc, SceneData, SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: a diagnostic build of the same library (tools/build_stamps.sh), never a fallback
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_STATE, RT_ERR_IO = range(6)
INF = np.float32(1000000.0)
SCENE_HIT_NOALLOC_MAX_RAYS = 1 << 20     # RT_SCENE_HIT_NOALLOC_MAX_RAYS

# every symbol include/rt_abi.h declares
ABI_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_upload_scene", "rt_set_camera", "rt_set_partition", "rt_partition_pixels", "rt_reset",
    "rt_render", "rt_read_framebuffer", "rt_local_pixel_count", "rt_local_pixel_indices",
    "rt_export_local_rgba", "rt_scene_hit", "rt_scene_hit_noalloc", "rt_postprocess", "rt_get_stats", "rt_reset_stats",
    "rt_set_timing", "rt_set_culling", "rt_host_bounds", "rt_last_error", "rt_obj_load", "rt_obj_free", "rt_write_ppm", "rt_write_pfm",
]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("bounce_rays", C.c_uint64), ("path_launches", C.c_uint64),
                ("path_ms", C.c_double), ("render_ms", C.c_double), ("n_prims", C.c_uint64), ("streams", C.c_uint64), ("coop_chunks", C.c_uint64),
                ("batch_launches", C.c_uint64), ("batch_listed", C.c_uint64), ("batch_early", C.c_uint64)]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        FP, IP, VP = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
        L.rt_create.argtypes = [C.POINTER(VP), C.c_int, C.c_int, C.c_int]
        L.rt_destroy.argtypes = [VP]
        L.rt_upload_scene.argtypes = [VP, C.POINTER(SceneDesc)]
        L.rt_set_camera.argtypes = [VP, C.POINTER(CameraDesc)]
        L.rt_set_partition.argtypes = [VP, C.c_int, C.c_int, C.c_int]
        L.rt_partition_pixels.argtypes = [C.c_int] * 5 + [IP, C.POINTER(C.c_int64)]
        L.rt_reset.argtypes = [VP]
        L.rt_render.argtypes = [VP, C.c_int, C.c_int]
        L.rt_read_framebuffer.argtypes = [VP, FP, IP, FP]
        L.rt_local_pixel_count.argtypes = [VP, C.POINTER(C.c_int64)]
        L.rt_local_pixel_indices.argtypes = [VP, IP]
        L.rt_export_local_rgba.argtypes = [VP, VP]
        L.rt_scene_hit.argtypes = [VP, C.c_int, FP, FP, FP, IP, IP]
        L.rt_scene_hit_noalloc.argtypes = [VP, C.c_int, VP, VP, VP, VP, VP]
        L.rt_postprocess.argtypes = [VP, FP, FP]
        L.rt_get_stats.argtypes = [VP, C.POINTER(Stats)]
        L.rt_reset_stats.argtypes = [VP]
        L.rt_set_timing.argtypes = [VP, C.c_int]
        L.rt_set_culling.argtypes = [VP, C.c_int]
        L.rt_host_bounds.argtypes = [C.POINTER(SceneDesc), FP, IP, FP]
        L.rt_last_error.argtypes = [VP]
        L.rt_last_error.restype = C.c_char_p
        L.rt_obj_load.argtypes = [C.c_char_p, C.POINTER(FP), C.POINTER(FP), C.POINTER(FP),
                                  C.POINTER(C.c_int32)]
        L.rt_obj_free.argtypes = [FP]
        L.rt_obj_free.restype = None
        L.rt_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, FP]
        L.rt_write_pfm.argtypes = [C.c_char_p, C.c_int, C.c_int, FP]
        _lib = L
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


class Context:
    """One rendering context on one GPU (mirrors the state initCUDA builds, kernel.cu:193-256)."""

    def __init__(self, width, height, device=0):
        self._h = C.c_void_p()
        self.width, self.height = int(width), int(height)
        rc = lib().rt_create(C.byref(self._h), device, self.width, self.height)
        if rc != RT_OK:
            msg = lib().rt_last_error(None).decode()
            self._h = C.c_void_p()
            raise RtError(rc, msg)

    def _chk(self, rc):
        if rc != RT_OK:
            raise RtError(rc, lib().rt_last_error(self._h).decode())

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # Scene::initializeScene (Scene/Scene.cu:126-202)
    def upload_scene(self, scene: SceneData):
        d = scene.desc()
        self._chk(lib().rt_upload_scene(self._h, C.byref(d)))

    # Camera ctor + cudaMemcpyToSymbol (kernel.cu:228-229, 239)
    def set_camera(self, cam: CameraArgs):
        d = CameraDesc.from_args(cam)
        self._chk(lib().rt_set_camera(self._h, C.byref(d)))

    def set_partition(self, rank, world, tile=64):
        self._chk(lib().rt_set_partition(self._h, rank, world, tile))

    def reset(self):
        self._chk(lib().rt_reset(self._h))

    # n graph replays (kernel.cu:122-123)
    def render(self, n_samples=1, max_depth=0):
        self._chk(lib().rt_render(self._h, n_samples, max_depth))

    def read_framebuffer(self, rgba=True, prim_id=True, t=True):
        h, w = self.height, self.width
        a = np.zeros((h, w, 4), np.float32) if rgba else None
        b = np.zeros((h, w), np.int32) if prim_id else None
        c = np.zeros((h, w), np.float32) if t else None
        self._chk(lib().rt_read_framebuffer(self._h, _fp(a), _ip(b), _fp(c)))
        return dict(rgba=a, prim_id=b, t=c)

    def local_pixel_count(self):
        n = C.c_int64()
        self._chk(lib().rt_local_pixel_count(self._h, C.byref(n)))
        return n.value

    def local_pixel_indices(self):
        idx = np.zeros(self.local_pixel_count(), np.int32)
        self._chk(lib().rt_local_pixel_indices(self._h, _ip(idx)))
        return idx

    def export_local_rgba(self, device_ptr):
        self._chk(lib().rt_export_local_rgba(self._h, C.c_void_p(device_ptr)))

    # Scene::hit (Scene/Scene.cu:8-67)
    def scene_hit(self, org, dir):
        org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        n = org.shape[0]
        t = np.zeros(n, np.float32)
        pid = np.zeros(n, np.int32)
        hit = np.zeros(n, np.int32)
        self._chk(lib().rt_scene_hit(self._h, n, _fp(org), _fp(dir), _fp(t), _ip(pid), _ip(hit)))
        return t, pid, hit

    def scene_hit_noalloc(self, n, d_org, d_dir, d_t, d_prim_id, d_ishit):
        """rt_scene_hit_noalloc: all five arguments are DEVICE addresses (ints) of caller-owned buffers."""
        self._chk(lib().rt_scene_hit_noalloc(self._h, int(n), C.c_void_p(d_org), C.c_void_p(d_dir),
                                             C.c_void_p(d_t), C.c_void_p(d_prim_id), C.c_void_p(d_ishit)))

    def postprocess(self, hdr=True, filtered=True):
        h, w = self.height, self.width
        a = np.zeros((h, w, 4), np.float32) if hdr else None
        b = np.zeros((h, w, 4), np.float32) if filtered else None
        self._chk(lib().rt_postprocess(self._h, _fp(a), _fp(b)))
        return a, b

    def stats(self) -> Stats:
        s = Stats()
        self._chk(lib().rt_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        self._chk(lib().rt_reset_stats(self._h))

    def set_culling(self, enabled=True):
        self._chk(lib().rt_set_culling(self._h, 1 if enabled else 0))

    def set_timing(self, enabled=True):
        self._chk(lib().rt_set_timing(self._h, 1 if enabled else 0))


def partition_pixels(width, height, rank, world, tile=64):
    """Global pixel indices owned by `rank` of `world` (host-only)."""
    n = C.c_int64()
    rc = lib().rt_partition_pixels(width, height, rank, world, tile, None, C.byref(n))
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error(None).decode())
    idx = np.zeros(n.value, np.int32)
    rc = lib().rt_partition_pixels(width, height, rank, world, tile, _ip(idx), C.byref(n))
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error(None).decode())
    return idx


def host_bounds(scene: SceneData):
    """(bounds[n,6] = lo.xyz hi.xyz, position[n], geom_extent) of rt_host_bounds: host-only."""
    d = scene.desc()
    b = np.zeros((scene.n_tris, 6), np.float32)
    pos = np.zeros(scene.n_tris, np.int32)
    ext = C.c_float()
    rc = lib().rt_host_bounds(C.byref(d), _fp(b), _ip(pos), C.byref(ext))
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error(None).decode())
    return b, pos, float(ext.value)


def obj_load(path):
    """Mesh::readFile (Model/Model.cu:139-284) -> (pos[n,3,3], nrm[n,3,3], uv[n,3,2])."""
    FP = C.POINTER(C.c_float)
    p, q, u = FP(), FP(), FP()
    n = C.c_int32()
    rc = lib().rt_obj_load(os.fsencode(path), C.byref(p), C.byref(q), C.byref(u), C.byref(n))
    if rc != RT_OK:
        raise RtError(rc, f"rt_obj_load({path}) failed")
    try:
        k = n.value
        pos = np.ctypeslib.as_array(p, shape=(k * 9,)).reshape(k, 3, 3).copy()
        nrm = np.ctypeslib.as_array(q, shape=(k * 9,)).reshape(k, 3, 3).copy()
        uv = np.ctypeslib.as_array(u, shape=(k * 6,)).reshape(k, 3, 2).copy()
    finally:
        lib().rt_obj_free(p)
        lib().rt_obj_free(q)
        lib().rt_obj_free(u)
    return pos, nrm, uv


def write_ppm(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    rc = lib().rt_write_ppm(os.fsencode(path), w, h, _fp(rgba))
    if rc != RT_OK:
        raise RtError(rc, f"rt_write_ppm({path}) failed")


def write_pfm(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    rc = lib().rt_write_pfm(os.fsencode(path), w, h, _fp(rgba))
    if rc != RT_OK:
        raise RtError(rc, f"rt_write_pfm({path}) failed")
