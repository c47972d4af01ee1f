"""Flat scene / camera containers shared by the C-ABI binding, tests and bench.

``SceneData`` is the 11-argument call of the reference's
``Scene::initializeScene`` (Scene/Scene.cuh:16-17) flattened to numpy arrays, in
the reference's traversal order (Scene/Scene.cu:13-41): standalone ``Triangle``
models first, then each ``Mesh``'s triangles, then ``Quadratic`` models.
``CameraArgs`` is the 7-argument ``Camera`` constructor (Camera/Camera.cuh:11).

``SceneDesc``/``CameraDesc`` are the ctypes mirrors of ``rt_scene_desc`` /
``rt_camera_desc`` in include/rt_abi.h.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

MAT_LAMBERTIAN = 1  # material::LAMBERTIAN, Material/Object.cuh:15
MAT_FRESNEL, MAT_COOK, MAT_GGX, MAT_OREN_NAYAR = 7, 9, 17, 25   # Object.cuh:16-19
QUAD_SPHERE = 0     # quadraticType::Sphere, Model/Model.cuh:100-103


def _f32(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a.reshape(shape)


def _i32(a, shape):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    return a.reshape(shape)


def identity():
    return np.eye(4, dtype=np.float32)


def translation(x, y, z):
    """Matrix/Matrix.cu:143-151."""
    m = np.eye(4, dtype=np.float32)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return m


def scale(x, y, z):
    """Matrix/Matrix.cu:170-179."""
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = x, y, z
    return m


@dataclass
class CameraArgs:
    pos: tuple = (0.0, 0.0, 0.0)
    lookat: tuple = (0.0, 0.0, -1.0)
    fov: float = 2.0
    near: float = 0.1
    far: float = 1000.0
    res: tuple = (0, 0)          # the reference passes (W/2, H/2), kernel.cu:228-229
    up: tuple = (0.0, 1.0, 0.0)

    @staticmethod
    def for_image(width, height, **kw):
        return CameraArgs(res=(width // 2, height // 2), **kw)


class CameraDesc(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("lookat", C.c_float * 3), ("fov", C.c_float),
                ("near_", C.c_float), ("far_", C.c_float), ("res_x", C.c_int32),
                ("res_y", C.c_int32), ("up", C.c_float * 3)]

    @staticmethod
    def from_args(a: CameraArgs) -> "CameraDesc":
        d = CameraDesc()
        d.pos[:] = [float(v) for v in a.pos]
        d.lookat[:] = [float(v) for v in a.lookat]
        d.fov, d.near_, d.far_ = float(a.fov), float(a.near), float(a.far)
        d.res_x, d.res_y = int(a.res[0]), int(a.res[1])
        d.up[:] = [float(v) for v in a.up]
        return d


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)


class SceneDesc(C.Structure):
    _fields_ = [("n_tri_models", C.c_int32), ("n_meshes", C.c_int32), ("n_quads", C.c_int32),
                ("n_tris", C.c_int32), ("tri_pos", _FP), ("tri_nrm", _FP), ("tri_model", _IP),
                ("mesh_first", _IP), ("mesh_count", _IP), ("model_xf", _FP),
                ("model_mat_type", _IP), ("model_color", _FP), ("quad_coeff", _FP),
                ("quad_type", _IP), ("n_point_lights", C.c_int32), ("pl_pos", _FP),
                ("pl_illum", _FP), ("n_dir_lights", C.c_int32), ("dl_dir", _FP), ("dl_illum", _FP),
                ("n_tri_lights", C.c_int32), ("tl_pos", _FP), ("tl_illum", _FP), ("tl_two_side", _IP),
                ("tl_mat_type", _IP), ("tl_color", _FP), ("model_param", _FP),
                ("n_env_lights", C.c_int32), ("env_width", C.c_int32), ("env_height", C.c_int32),
                ("env_image", _FP), ("tri_uv", _FP), ("n_textures", C.c_int32), ("mesh_texture", _IP),
                ("tex_size", _IP), ("tex_data", _FP)]


@dataclass
class SceneData:
    """Host-side scene, caller-owned; uploads copy it (as the reference does)."""
    n_tri_models: int = 0
    tri_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3), np.float32))
    tri_nrm: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3), np.float32))
    tri_model: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    mesh_first: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    mesh_count: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    model_xf: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 4), np.float32))
    model_mat_type: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    model_color: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    # (roughness, eta): second constructor argument of GGX / Cook_Torrance and of Fresnel (material.cuh:48,88,112)
    model_param: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    quad_coeff: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    quad_type: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    pl_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    pl_illum: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    dl_dir: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    dl_illum: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    # TriangleLight(posa, posb, posc, light_color, two) + its own Material (Light.cuh:71, Scene.cu:151-153)
    tl_pos: np.ndarray = field(default_factory=lambda: np.zeros((0, 3, 3), np.float32))
    tl_illum: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    tl_two_side: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    tl_mat_type: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    tl_color: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    # EnvironmentLight(texture, w, h) (Light.cuh:108): [h, w, 4] floats as stbi_loadf(.., 4) returns them; None = no such light
    env_image: np.ndarray = None
    # Triangle::vText (Model.cuh:75) per triangle; None = zeros
    tri_uv: np.ndarray = None
    # Mesh::initTexture(data, w, h) (Model.cu:315-319): the images, [h, w, 4] floats each, and per mesh the index
    # of the one it bound (-1: none, H5)
    textures: list = field(default_factory=list)
    mesh_texture: np.ndarray = None

    @property
    def n_tris(self):
        return int(self.tri_pos.shape[0])

    @property
    def n_meshes(self):
        return int(self.mesh_first.shape[0])

    @property
    def n_quads(self):
        return int(self.quad_coeff.shape[0])

    @property
    def n_models(self):
        return self.n_tri_models + self.n_meshes + self.n_quads

    @property
    def n_prims(self):
        return self.n_tris + self.n_quads

    @property
    def n_lights(self):
        """Lights in Scene::sampleAllLight order (Scene.cu:83-114): point, directional, triangle."""
        return int(self.pl_pos.shape[0]) + int(np.asarray(self.dl_dir).reshape(-1, 3).shape[0]) + \
            self.n_tri_lights + (0 if self.env_image is None else 1)

    @property
    def n_tri_lights(self):
        return int(np.asarray(self.tl_pos).reshape(-1, 3, 3).shape[0])

    def normalise(self):
        n = self.tri_pos.shape[0] if self.tri_pos.ndim == 3 else np.asarray(self.tri_pos).size // 9
        self.tri_pos = _f32(self.tri_pos, (n, 3, 3))
        self.tri_nrm = _f32(self.tri_nrm, (n, 3, 3))
        self.tri_model = _i32(self.tri_model, (n,))
        self.mesh_first = _i32(self.mesh_first, (-1,))
        self.mesh_count = _i32(self.mesh_count, (-1,))
        self.quad_coeff = _f32(self.quad_coeff, (-1, 3))
        self.quad_type = _i32(self.quad_type, (-1,))
        nm = self.n_tri_models + self.mesh_first.shape[0] + self.quad_coeff.shape[0]
        self.model_xf = _f32(self.model_xf, (nm, 4, 4))
        self.model_mat_type = _i32(self.model_mat_type, (nm,))
        self.model_color = _f32(self.model_color, (nm, 3))
        if np.asarray(self.model_param).size != 2 * nm:
            self.model_param = np.zeros((nm, 2), np.float32)
        self.model_param = _f32(self.model_param, (nm, 2))
        self.pl_pos = _f32(self.pl_pos, (-1, 3))
        self.pl_illum = _f32(self.pl_illum, (-1, 3))
        self.dl_dir = _f32(self.dl_dir, (-1, 3))
        self.dl_illum = _f32(self.dl_illum, (-1, 3))
        self.tl_pos = _f32(self.tl_pos, (-1, 3, 3))
        nl = self.tl_pos.shape[0]
        self.tl_illum = _f32(self.tl_illum, (nl, 3))
        self.tl_two_side = _i32(self.tl_two_side, (nl,))
        self.tl_mat_type = _i32(self.tl_mat_type, (nl,))
        self.tl_color = _f32(self.tl_color, (nl, 3))
        if self.env_image is not None:
            self.env_image = np.ascontiguousarray(self.env_image, np.float32)
            assert self.env_image.ndim == 3 and self.env_image.shape[2] == 4
        if self.tri_uv is not None:
            self.tri_uv = _f32(self.tri_uv, (n, 3, 2))
        if self.textures:
            self.textures = [np.ascontiguousarray(t, np.float32) for t in self.textures]
            assert all(t.ndim == 3 and t.shape[2] == 4 for t in self.textures)
            if self.mesh_texture is None:
                self.mesh_texture = np.full(self.mesh_first.shape[0], -1, np.int32)
            self.mesh_texture = _i32(self.mesh_texture, (self.mesh_first.shape[0],))
            self._tex_size = np.array([[t.shape[1], t.shape[0]] for t in self.textures], np.int32)
            self._tex_data = np.concatenate([t.reshape(-1) for t in self.textures]).astype(np.float32)
        assert self.pl_pos.shape == self.pl_illum.shape
        assert self.dl_dir.shape == self.dl_illum.shape
        assert self.quad_type.shape[0] == self.quad_coeff.shape[0]
        assert int(self.mesh_count.sum()) + self.n_tri_models == n
        return self

    def desc(self) -> SceneDesc:
        """ctypes view; keeps no copies, so `self` must outlive the call."""
        self.normalise()
        d = SceneDesc()
        d.n_tri_models, d.n_meshes, d.n_quads = self.n_tri_models, self.n_meshes, self.n_quads
        d.n_tris = self.n_tris
        fp = lambda a: a.ctypes.data_as(_FP)
        ip = lambda a: a.ctypes.data_as(_IP)
        d.tri_pos, d.tri_nrm, d.tri_model = fp(self.tri_pos), fp(self.tri_nrm), ip(self.tri_model)
        d.mesh_first, d.mesh_count = ip(self.mesh_first), ip(self.mesh_count)
        d.model_xf, d.model_mat_type = fp(self.model_xf), ip(self.model_mat_type)
        d.model_color = fp(self.model_color)
        d.model_param = fp(self.model_param)
        d.quad_coeff, d.quad_type = fp(self.quad_coeff), ip(self.quad_type)
        d.n_point_lights = int(self.pl_pos.shape[0])
        d.pl_pos, d.pl_illum = fp(self.pl_pos), fp(self.pl_illum)
        d.n_dir_lights = int(self.dl_dir.shape[0])
        d.dl_dir, d.dl_illum = fp(self.dl_dir), fp(self.dl_illum)
        d.n_tri_lights = self.n_tri_lights
        d.tl_pos, d.tl_illum, d.tl_color = fp(self.tl_pos), fp(self.tl_illum), fp(self.tl_color)
        d.tl_two_side, d.tl_mat_type = ip(self.tl_two_side), ip(self.tl_mat_type)
        if self.env_image is not None:
            d.n_env_lights = 1
            d.env_height, d.env_width = int(self.env_image.shape[0]), int(self.env_image.shape[1])
            d.env_image = fp(self.env_image)
        if self.tri_uv is not None:
            d.tri_uv = fp(self.tri_uv)
        if self.textures:
            d.n_textures = len(self.textures)
            d.mesh_texture, d.tex_size, d.tex_data = ip(self.mesh_texture), ip(self._tex_size), fp(self._tex_data)
        return d


class SceneBuilder:
    """Mirrors how the reference's host code assembles a scene before
    ``initializeScene`` (kernel.cu:268-313): models by type, one material each."""

    def __init__(self):
        self._tris = []      # (pos[3,3], nrm[3,3], xf, color)
        self._meshes = []    # (pos[n,3,3], nrm[n,3,3], xf, color, uv[n,3,2] or None, texture index or -1)
        self._textures = []
        self._quads = []     # (coeff, type, xf, color)
        self._lights = []
        self._dir_lights = []
        self._tri_lights = []

    def add_triangle(self, pos, nrm, xf=None, color=(0.7, 0.7, 0.7)):
        self._tris.append((_f32(pos, (3, 3)), _f32(nrm, (3, 3)),
                           identity() if xf is None else _f32(xf, (4, 4)), color))
        return self

    def add_mesh(self, pos, nrm, xf=None, color=(0.7, 0.7, 0.7), uv=None, texture=None):
        """A Mesh as readFile leaves it (Model.cu:139-284); ``uv`` = its triangles' vText, ``texture`` = the
        [h, w, 4] float image Mesh::initTexture binds (Model.cu:315-319)."""
        pos = _f32(pos, (-1, 3, 3))
        tex = -1
        if texture is not None:
            self._textures.append(np.ascontiguousarray(texture, np.float32))
            tex = len(self._textures) - 1
        self._meshes.append((pos, _f32(nrm, pos.shape),
                             identity() if xf is None else _f32(xf, (4, 4)), color,
                             None if uv is None else _f32(uv, (pos.shape[0], 3, 2)), tex))
        return self

    def add_sphere(self, coeff_x, xf=None, color=(0.7, 0.7, 0.7)):
        """Quadratic(make_float3(c, 0, 0), Sphere): radius = 1/c (Model.cu:341-346)."""
        self._quads.append(((coeff_x, 0.0, 0.0), QUAD_SPHERE,
                            identity() if xf is None else _f32(xf, (4, 4)), color))
        return self

    def add_point_light(self, pos, illum):
        self._lights.append((pos, illum))
        return self

    def add_directional_light(self, direction, illum):
        """DirectionalLight(direction, color) (Light/Light.cuh:57)."""
        self._dir_lights.append((direction, illum))
        return self

    def add_triangle_light(self, pos, illum, two_side=False, color=(1.0, 1.0, 1.0)):
        """TriangleLight(posa, posb, posc, light_color, two) with a Lambertian material of its
        own (Light/Light.cuh:71, Light.cu:112-135; kernel.cu:281-283)."""
        self._tri_lights.append((_f32(pos, (3, 3)), illum, bool(two_side), color))
        return self

    def build(self) -> SceneData:
        s = SceneData()
        s.n_tri_models = len(self._tris)
        pos, nrm, model, xf, col, uvs = [], [], [], [], [], []
        for i, (p, n, m, c) in enumerate(self._tris):
            pos.append(p[None]); nrm.append(n[None]); model.append(np.full(1, i, np.int32))
            xf.append(m); col.append(c); uvs.append(np.zeros((1, 3, 2), np.float32))
        first, count = [], []
        cursor = len(self._tris)
        for j, (p, n, m, c, uv, _) in enumerate(self._meshes):
            pos.append(p); nrm.append(n)
            uvs.append(np.zeros((p.shape[0], 3, 2), np.float32) if uv is None else uv)
            model.append(np.full(p.shape[0], s.n_tri_models + j, np.int32))
            first.append(cursor); count.append(p.shape[0]); cursor += p.shape[0]
            xf.append(m); col.append(c)
        for (_, _, m, c) in self._quads:
            xf.append(m); col.append(c)
        s.tri_pos = np.concatenate(pos) if pos else np.zeros((0, 3, 3), np.float32)
        s.tri_nrm = np.concatenate(nrm) if nrm else np.zeros((0, 3, 3), np.float32)
        s.tri_model = np.concatenate(model) if model else np.zeros((0,), np.int32)
        s.mesh_first, s.mesh_count = np.array(first, np.int32), np.array(count, np.int32)
        s.model_xf = np.array(xf, np.float32).reshape(-1, 4, 4)
        s.model_mat_type = np.full(len(xf), MAT_LAMBERTIAN, np.int32)
        s.model_color = np.array(col, np.float32).reshape(-1, 3)
        s.quad_coeff = np.array([q[0] for q in self._quads], np.float32).reshape(-1, 3)
        s.quad_type = np.array([q[1] for q in self._quads], np.int32)
        s.pl_pos = np.array([l[0] for l in self._lights], np.float32).reshape(-1, 3)
        s.pl_illum = np.array([l[1] for l in self._lights], np.float32).reshape(-1, 3)
        s.dl_dir = np.array([l[0] for l in self._dir_lights], np.float32).reshape(-1, 3)
        s.dl_illum = np.array([l[1] for l in self._dir_lights], np.float32).reshape(-1, 3)
        s.tl_pos = np.array([l[0] for l in self._tri_lights], np.float32).reshape(-1, 3, 3)
        s.tl_illum = np.array([l[1] for l in self._tri_lights], np.float32).reshape(-1, 3)
        s.tl_two_side = np.array([l[2] for l in self._tri_lights], np.int32)
        s.tl_mat_type = np.full(len(self._tri_lights), MAT_LAMBERTIAN, np.int32)
        s.tl_color = np.array([l[3] for l in self._tri_lights], np.float32).reshape(-1, 3)
        if self._textures:
            s.tri_uv = np.concatenate(uvs)
            s.textures = list(self._textures)
            s.mesh_texture = np.array([m[5] for m in self._meshes], np.int32)
        return s.normalise()
