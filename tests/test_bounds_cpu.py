"""Host-side culling bounds (rt_host_bounds, no GPU): whenever the oracle's Triangle::hit
accepts a (ray, triangle) pair, the conservative box test the closest-hit kernel runs
(rt_device.h node_test, restated here in numpy float32) must let that pair through."""
import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.scene import SceneBuilder, identity

from . import orc


def node_pass(box, org, dirn, extent_geom, best=np.float32(1e6)):
    """node_test of rt_device.h for n rays against one box, float32 throughout."""
    f = np.float32
    extent = f(16.0) * max(f(extent_geom), f(1.0))
    lo, hi = box[:3].astype(f), box[3:].astype(f)
    d = dirn.copy()
    tiny = np.abs(d) < f(1e-30)
    d[tiny] = np.copysign(f(1e-30), d[tiny])
    inv = f(1.0) / d
    oi = (org * inv).astype(f)                               # RayPair::px/py/pz
    # one fma per plane: fl(lo * inv - oi); the float64 product of two float32 is exact
    a = (lo[None, :].astype(np.float64) * inv.astype(np.float64) - oi.astype(np.float64)).astype(f)
    b = (hi[None, :].astype(np.float64) * inv.astype(np.float64) - oi.astype(np.float64)).astype(f)
    tn = np.maximum(np.minimum(a, b).max(1), f(0.0))
    tf = np.minimum(np.maximum(a, b).min(1) * f(1.000004), best * f(1.000004))
    rej = tn > tf
    wild = ~(np.abs(org) <= extent).all(1)                   # zero inverse in the kernel: passes
    return ~rej | wild


def normalize32(d):
    """Ray::Ray: dir * (1 / sqrt(dot)) in float32 (Ray.cu:2, helper_math.h:62-65,1309)."""
    f = np.float32
    dd = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    inv = f(1.0) / np.sqrt(dd, dtype=f)
    return (d * inv[:, None]).astype(f)


def random_triangle(rng, kind):
    c = rng.uniform(-300, 300, 3)
    if kind == "tiny":
        s = rng.uniform(0.05, 0.3)
    elif kind == "large":
        s = rng.uniform(30, 200)
    else:
        s = rng.uniform(0.5, 8)
    a = c + rng.normal(size=3) * s
    b = c + rng.normal(size=3) * s
    d = c + rng.normal(size=3) * s
    if kind == "sliver":
        d = a + (b - a) * rng.uniform(0.2, 0.8) + rng.normal(size=3) * s * 0.01
    return np.array([a, b, d], np.float32)


@pytest.mark.parametrize("kind", ["normal", "tiny", "large", "sliver"])
def test_accepted_pairs_pass_the_box_test(kind):
    rng = np.random.default_rng({"normal": 1, "tiny": 2, "large": 3, "sliver": 4}[kind])
    n_accept = 0
    for _ in range(60):
        tri = random_triangle(rng, kind)
        b = SceneBuilder()
        b.add_triangle(tri, [(0, 0, 1)] * 3, identity(), (0.7, 0.7, 0.7))
        scene = b.build()
        bounds, pos, ext = abi.host_bounds(scene)
        assert pos[0] == 0
        # targets: inside, on the edges, and in a band just outside them (the "fat" rim)
        n = 3000
        w = rng.dirichlet([1, 1, 1], n)
        w[n // 3:] = rng.dirichlet([1, 1, 0.02], n - n // 3)           # hug an edge
        w = w[np.arange(n)[:, None], rng.permuted(np.tile(np.arange(3), (n, 1)), axis=1)]
        tgt = (w[:, :, None] * tri[None].astype(np.float64)).sum(1)
        edge = np.linalg.norm(tri[1] - tri[0]) + 1e-9
        tgt += rng.normal(size=(n, 3)) * rng.choice([0.0, 1e-4, 1e-3, 1e-2, 0.1], (n, 1)) / max(edge, 0.05)
        org = (tgt + rng.normal(size=(n, 3)) * rng.choice([2.0, 40.0, 600.0], (n, 1))).astype(np.float32)
        d = (tgt - org.astype(np.float64)).astype(np.float32)
        t, pid, hit = orc.scene_hit(scene, org, d)
        acc = pid == 0
        if not acc.any():
            continue
        n_accept += int(acc.sum())
        ok = node_pass(bounds[0], org[acc], normalize32(d[acc]), ext)
        assert ok.all(), f"{(~ok).sum()} accepted pairs would be culled (box={bounds[0]})"
        # the depth test at the moment of acceptance: best just above the accepted t
        ok2 = node_pass(bounds[0], org[acc], normalize32(d[acc]), ext, best=np.nextafter(t[acc], np.float32(2e6)))
        assert ok2.all()
    assert n_accept > (200 if kind != "tiny" else 0)


def test_box_test_has_teeth():
    """The same check fails once the boxes are shrunk by 3 %: the bound is not vacuous."""
    rng = np.random.default_rng(9)
    fails = 0
    for _ in range(20):
        tri = random_triangle(rng, "normal")
        b = SceneBuilder()
        b.add_triangle(tri, [(0, 0, 1)] * 3, identity(), (0.7, 0.7, 0.7))
        scene = b.build()
        bounds, pos, ext = abi.host_bounds(scene)
        n = 3000
        w = rng.dirichlet([1, 1, 0.02], n)
        tgt = (w[:, :, None] * tri[None].astype(np.float64)).sum(1)
        org = (tgt + rng.normal(size=(n, 3)) * 40).astype(np.float32)
        d = (tgt - org.astype(np.float64)).astype(np.float32)
        t, pid, hit = orc.scene_hit(scene, org, d)
        acc = pid == 0
        box = bounds[0].copy()
        c, h = (box[:3] + box[3:]) / 2, (box[3:] - box[:3]) / 2 * 0.97
        box[:3], box[3:] = c - h, c + h
        fails += int((~node_pass(box, org[acc], normalize32(d[acc]), ext)).sum())
    assert fails > 100


def test_invisible_triangles_are_marked():
    """H1: 2*area below 0.01 can never pass Model.cu:99 -> dead box (lo > hi)."""
    b = SceneBuilder()
    b.add_triangle([(0, 0, -5), (0.05, 0, -5), (0, 0.05, -5)], [(0, 0, 1)] * 3, identity(), (0.7,) * 3)
    b.add_triangle([(0, 0, -5), (2, 0, -5), (0, 2, -5)], [(0, 0, 1)] * 3, identity(), (0.7,) * 3)
    bounds, pos, ext = abi.host_bounds(b.build())
    assert (bounds[0][:3] > bounds[0][3:]).all()
    assert np.isfinite(bounds[1]).all() and (bounds[1][3:5] - bounds[1][:2] > 2.0).all()
    assert sorted(pos.tolist()) == [0, 1]


def test_position_is_a_permutation():
    from raytracer_amd import scenes
    cfg = scenes.c3(64, 36, lon=20, lat=20, ground=(6, 3))
    bounds, pos, ext = abi.host_bounds(cfg.scene)
    assert sorted(pos.tolist()) == list(range(cfg.scene.n_tris))
    assert np.isfinite(bounds).all() and (bounds[:, 3:] >= bounds[:, :3]).all() and ext >= 125.0
