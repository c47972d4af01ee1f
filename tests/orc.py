"""ctypes wrapper around oracle/librt_oracle.so -- the CHECKER used by tests,
smoke() and bench.py's cpu_baseline leg.  Never imported by raytracer_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd.scene import CameraArgs, CameraDesc, SceneData, SceneDesc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "librt_oracle.so")
INF = np.float32(1000000.0)


class Counts(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("shadow", C.c_uint64), ("bounce", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_render.argtypes = [C.POINTER(SceneDesc), C.POINTER(CameraDesc)] + [C.c_int] * 9 + \
            [FP, IP, FP, C.POINTER(Counts)]
        L.orc_render.restype = C.c_int
        L.orc_render_pixels.argtypes = [C.POINTER(SceneDesc), C.POINTER(CameraDesc), C.c_int, C.c_int,
                                        C.c_int, IP, C.c_int, C.c_int, C.c_int, FP, IP, FP,
                                        C.POINTER(Counts)]
        L.orc_render_pixels.restype = C.c_int
        L.orc_scene_hit.argtypes = [C.POINTER(SceneDesc), C.c_int, FP, FP, FP, IP, IP]
        L.orc_camera_ray.argtypes = [C.POINTER(CameraDesc), C.c_int, FP, FP, FP]
        L.orc_trace_ray.argtypes = [C.POINTER(SceneDesc), FP, FP, C.c_int, C.c_uint32, C.c_uint32,
                                    FP, FP, IP]
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_pixel_uniform4.argtypes = [C.c_uint32, C.c_uint32, FP]
        L.orc_sinf.argtypes = [C.c_float]
        L.orc_sinf.restype = C.c_float
        L.orc_cosf.argtypes = [C.c_float]
        L.orc_cosf.restype = C.c_float
        L.orc_tonemap_aces.argtypes = [C.c_int, FP, FP]
        L.orc_median3x3.argtypes = [C.c_int, C.c_int, FP, FP]
        L.orc_num_threads.restype = C.c_int
        L.orc_texture_levels.argtypes = [C.c_int, C.c_int]
        L.orc_texture_mip.argtypes = [FP, C.c_int, C.c_int, C.c_int, FP, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_tex2d_grad.argtypes = [FP, C.c_int, C.c_int, C.c_float, C.c_float, FP, FP, FP]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def render(scene: SceneData, cam: CameraArgs, width, height, spp=1, max_depth=0, crop=None,
           threads=0):
    """-> dict(rgba[h,w,4], prim_id[h,w], t[h,w], primary, shadow, bounce)."""
    x0, y0, x1, y1 = crop if crop else (0, 0, width, height)
    h, w = y1 - y0, x1 - x0
    rgba = np.zeros((h, w, 4), np.float32)
    pid = np.zeros((h, w), np.int32)
    t = np.zeros((h, w), np.float32)
    cnt = Counts()
    sd, cd = scene.desc(), CameraDesc.from_args(cam)
    rc = lib().orc_render(C.byref(sd), C.byref(cd), width, height, x0, y0, x1, y1, spp,
                          max_depth, threads, _fp(rgba), _ip(pid), _fp(t), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return dict(rgba=rgba, prim_id=pid, t=t, primary=cnt.primary, shadow=cnt.shadow,
                bounce=cnt.bounce)


def render_pixels(scene: SceneData, cam: CameraArgs, width, height, xy, spp=1, max_depth=0, threads=0):
    """The pixels xy[n, 2] = (x, y) of a width x height frame -> dict(rgba[n,4], prim_id[n], t[n], ...)."""
    xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
    n = xy.shape[0]
    rgba = np.zeros((n, 4), np.float32)
    pid = np.zeros(n, np.int32)
    t = np.zeros(n, np.float32)
    cnt = Counts()
    sd, cd = scene.desc(), CameraDesc.from_args(cam)
    rc = lib().orc_render_pixels(C.byref(sd), C.byref(cd), width, height, n, _ip(xy), spp, max_depth,
                                 threads, _fp(rgba), _ip(pid), _fp(t), C.byref(cnt))
    if rc != 0:
        raise RuntimeError(f"orc_render_pixels failed: {rc}")
    return dict(rgba=rgba, prim_id=pid, t=t, primary=cnt.primary, shadow=cnt.shadow, bounce=cnt.bounce)


def scene_hit(scene: SceneData, org, dir):
    org = np.ascontiguousarray(org, np.float32).reshape(-1, 3)
    dir = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
    n = org.shape[0]
    t = np.zeros(n, np.float32)
    pid = np.zeros(n, np.int32)
    hit = np.zeros(n, np.int32)
    sd = scene.desc()
    rc = lib().orc_scene_hit(C.byref(sd), n, _fp(org), _fp(dir), _fp(t), _ip(pid), _ip(hit))
    assert rc == 0
    return t, pid, hit


def camera_ray(cam: CameraArgs, xy):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    n = xy.shape[0]
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    cd = CameraDesc.from_args(cam)
    assert lib().orc_camera_ray(C.byref(cd), n, _fp(xy), _fp(o), _fp(d)) == 0
    return o, d


def trace_ray(scene: SceneData, org, dir, max_depth=0, pixel_index=0, draw0=0):
    org = np.ascontiguousarray(org, np.float32)
    dir = np.ascontiguousarray(dir, np.float32)
    rgb = np.zeros(3, np.float32)
    t = C.c_float()
    prim = C.c_int32()
    sd = scene.desc()
    assert lib().orc_trace_ray(C.byref(sd), _fp(org), _fp(dir), max_depth, pixel_index, draw0,
                               _fp(rgb), C.byref(t), C.byref(prim)) == 0
    return rgb, np.float32(t.value), prim.value


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def pixel_uniform4(pixel, draw):
    o = np.zeros(4, np.float32)
    lib().orc_pixel_uniform4(pixel, draw, _fp(o))
    return o


def tonemap(rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros_like(rgba)
    lib().orc_tonemap_aces(rgba.size // 4, _fp(rgba), _fp(out))
    return out


def median3x3(rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[:2]
    out = np.zeros_like(rgba)
    lib().orc_median3x3(w, h, _fp(rgba), _fp(out))
    return out


def texture_levels(w, h):
    return lib().orc_texture_levels(w, h)


def texture_mip(img, level):
    """Level `level` of the mip chain generateMipMaps builds over img[h, w, 4] (Texture.cu:48-133)."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    lw, lh = C.c_int(), C.c_int()
    assert lib().orc_texture_mip(_fp(img), w, h, level, None, C.byref(lw), C.byref(lh)) == 0
    out = np.zeros((lh.value, lw.value, 4), np.float32)
    assert lib().orc_texture_mip(_fp(img), w, h, level, _fp(out), None, None) == 0
    return out


def tex2d_grad(img, u, v, dpdx=(0.0, 0.0), dpdy=(0.0, 0.0)):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    dx, dy, rgb = np.float32(dpdx), np.float32(dpdy), np.zeros(3, np.float32)
    assert lib().orc_tex2d_grad(_fp(img), w, h, u, v, _fp(dx), _fp(dy), _fp(rgb)) == 0
    return rgb
