"""The reference-side bridge of INTEGRATION.md section 1, compiled with g++ against
include/rt_abi.h and linked with librt_amd.so (tests/host/rt_bridge_test.cpp, built by
__graft_entry__.build()), run as a subprocess: OBJ file -> rt_obj_load -> rt_upload_scene ->
rt_set_camera -> rt_render (one call per sample, like the reference's frame loop) ->
rt_read_framebuffer -> rt_write_pfm.  Its image must equal the committed fixture c2.npz."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

from .golden.make_golden import load_case
from .test_parity_gpu import assert_radiance

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRIDGE = os.path.join(ROOT, "tests", "host", "rt_bridge_test")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(v) for v in f.readline().split())
        assert float(f.readline()) < 0                     # little-endian
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)


@pytest.mark.parametrize("depth", [0, 12])
def test_cpp_bridge_renders_the_cornell_fixture(tmp_path, depth):
    assert os.path.exists(BRIDGE), "tests/host/rt_bridge_test is not built (run __graft_entry__.build())"
    scene, cam, w, h, crop, z, meta = load_case("c2")
    assert scene.n_meshes == 1 and scene.n_tri_models == 0 and scene.n_quads == 0
    obj = str(tmp_path / "cornell.obj")
    scenes.write_obj(obj, scene.tri_pos, scene.tri_nrm)
    out = str(tmp_path / "frame.pfm")
    spp = 2
    r = subprocess.run([BRIDGE, obj, out, str(w), str(h), str(spp), str(depth)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "rt_bridge_test ok: 36 triangles" in r.stdout
    img = read_pfm(out)                                    # rows in framebuffer order (row 0 first)
    x0, y0, x1, y1 = crop
    assert_radiance(img[y0:y1, x0:x1], z[f"s{spp}_d{depth}_rgba"][..., :3])
    rays = z[f"s{spp}_d{depth}_rays"]
    assert "%d primary" % (w * h * spp) in r.stdout and rays[0] > 0


def test_cpp_bridge_renders_a_textured_mesh(tmp_path):
    """The same bridge with an OBJ that carries vt lines and a texture bound to the mesh (Mesh::readFile's
    vText, Mesh::initTexture): rt_obj_load's uv -> tri_uv, the image -> tex_data; the frame must equal the
    oracle's render of the same scene assembled in Python."""
    from tests import orc
    scene, cam, w, h, crop, z, meta = load_case("c2")
    p = scene.tri_pos
    uv = np.stack([(p[..., 0] + np.float32(11.0)) / np.float32(22.0), (p[..., 1] + np.float32(11.0)) / np.float32(22.0)],
                  axis=2).astype(np.float32)
    obj = str(tmp_path / "cornell_uv.obj")
    scenes.write_obj(obj, scene.tri_pos, scene.tri_nrm, uv)
    tw, th = 16, 8
    yy, xx = np.mgrid[0:th, 0:tw]
    tex = np.zeros((th, tw, 4), np.float32)
    tex[..., 0] = np.where((xx ^ yy) & 1, np.float32(0.9), np.float32(0.2))
    tex[..., 1] = xx.astype(np.float32) / np.float32(tw)
    tex[..., 2] = yy.astype(np.float32) / np.float32(th)
    tex[..., 3] = 1.0
    scene.tri_uv, scene.textures, scene.mesh_texture = uv, [tex], np.int32([0])
    scene.normalise()
    out = str(tmp_path / "frame_uv.pfm")
    spp, depth = 2, 3
    r = subprocess.run([BRIDGE, obj, out, str(w), str(h), str(spp), str(depth), str(tw), str(th)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = read_pfm(out)
    ref = orc.render(scene, cam, w, h, spp=spp, max_depth=depth)
    assert_radiance(img, ref["rgba"][..., :3])
    plain = z["s2_d0_rgba"]                                  # and the texture did change the picture
    x0, y0, x1, y1 = crop
    assert np.abs(img[y0:y1, x0:x1] - plain[..., :3]).max() > 0.05
