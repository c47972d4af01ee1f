"""CPU tests of the C-ABI library (no compute calls): it loads, exports every symbol
include/rt_abi.h declares, and its host-side surface (OBJ loader, image writers, partition
map) behaves like the reference's."""
import os
import re

import numpy as np
import pytest

from raytracer_amd import abi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(abi.ABI_SYMBOLS)
    L = abi.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_no_torch_types_and_no_oracle_in_product():
    """The product must not route through the oracle or any CPU fallback."""
    for dp, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "librt_oracle" not in txt and "rt_oracle.h" not in txt, f
                assert "import orc" not in txt and "tests.orc" not in txt, f
    hdr = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    assert "torch" not in hdr and "at::" not in hdr


@pytest.mark.skipif(_has_gpu(), reason="checks the loud failure on a box without a GPU")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(abi.RtError) as e:
        abi.Context(8, 8)
    assert e.value.code == abi.RT_ERR_DEVICE


def test_obj_loader_roundtrip(tmp_path):
    """Mesh::readFile semantics (Model.cu:139-284): v / vn / f a//c, 1-based."""
    pos, nrm = scenes.blob_mesh(7, 5, 3.0, seed=3)
    path = tmp_path / "blob.obj"
    scenes.write_obj(str(path), pos, nrm)
    p, n, uv = abi.obj_load(str(path))
    assert np.array_equal(p.view(np.uint32), pos.view(np.uint32))
    assert np.array_equal(n.view(np.uint32), nrm.view(np.uint32))
    assert (uv == 0).all()


def test_obj_loader_with_texcoords_and_errors(tmp_path):
    txt = ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0.25 0.5\nvt 1 0\nvt 0 1\nvn 0 0 1\n"
           "# comment\nf 1/1/1 2/2/1 3/3/1\nf 1//1 3//1 2//1\n")
    path = tmp_path / "t.obj"
    path.write_text(txt)
    p, n, uv = abi.obj_load(str(path))
    assert p.shape == (2, 3, 3)
    assert np.array_equal(p[0], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    assert np.array_equal(p[1], [[0, 0, 0], [0, 1, 0], [1, 0, 0]])
    assert (n == [0, 0, 1]).all()
    assert np.array_equal(uv[0], [[0.25, 0.5], [1, 0], [0, 1]])
    assert (uv[1] == 0).all()                     # "a//c" faces get (0, 0) (Model.cu:232)
    with pytest.raises(abi.RtError) as e:         # missing file: readFile returns false (:146)
        abi.obj_load(str(tmp_path / "nope.obj"))
    assert e.value.code == abi.RT_ERR_IO
    quad = tmp_path / "q.obj"                     # the reference indexes V[3] by arity (:219-221)
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n")
    with pytest.raises(abi.RtError) as e:
        abi.obj_load(str(quad))
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    nonrm = tmp_path / "n.obj"                    # H6: faces must carry normals
    nonrm.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(abi.RtError) as e:
        abi.obj_load(str(nonrm))
    assert e.value.code == abi.RT_ERR_INVALID
    empty = tmp_path / "e.obj"                    # no faces: readFile returns false (:207-208)
    empty.write_text("v 0 0 0\n")
    with pytest.raises(abi.RtError):
        abi.obj_load(str(empty))


def test_image_writers(tmp_path):
    img = np.zeros((2, 3, 4), np.float32)
    img[0, 0, :3] = [1.0, 0.5, 0.0]
    img[1, 2, :3] = [2.0, -1.0, 0.25]
    abi.write_ppm(str(tmp_path / "a.ppm"), img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n3 2\n255\n")
    px = np.frombuffer(raw[len(b"P6\n3 2\n255\n"):], np.uint8).reshape(2, 3, 3)
    assert list(px[1, 0]) == [255, 128, 0]        # row 0 is written last (bottom-up image)
    assert list(px[0, 2]) == [255, 0, 64]
    abi.write_pfm(str(tmp_path / "a.pfm"), img)
    raw = (tmp_path / "a.pfm").read_bytes()
    hdr = b"PF\n3 2\n-1.0\n"
    assert raw.startswith(hdr)
    f = np.frombuffer(raw[len(hdr):], np.float32).reshape(2, 3, 3)
    assert np.array_equal(f, img[..., :3])


@pytest.mark.parametrize("w,h,world,tile", [(200, 136, 3, 64), (64, 64, 1, 64), (37, 23, 2, 8),
                                            (1920, 1080, 8, 64)])
def test_partition_is_a_disjoint_cover(w, h, world, tile):
    seen = np.zeros(w * h, np.int32)
    sizes = []
    for r in range(world):
        idx = abi.partition_pixels(w, h, r, world, tile)
        assert idx.min(initial=0) >= 0 and idx.max(initial=0) < w * h
        seen[idx] += 1
        sizes.append(len(idx))
        # 64 consecutive local pixels = one wave = an 8x8 block (interior tiles)
        if len(idx) >= 64 and w % 8 == 0 and h % 8 == 0:
            blk = idx[:64]
            xs, ys = blk % w, blk // w
            assert xs.max() - xs.min() == 7 and ys.max() - ys.min() == 7
    assert (seen == 1).all()
    if world == 8:
        assert max(sizes) - min(sizes) <= 64 * 64
