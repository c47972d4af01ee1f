"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bars (BASELINE.json north_star): primitive ids and t bit-exact;
radiance within 1e-4 relative per channel (tolerance written in `assert_radiance`)."""
import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.scene import CameraArgs, SceneBuilder, identity, translation

from . import orc
from .golden.make_golden import CASES, load_case

pytestmark = pytest.mark.gpu

RADIANCE_RTOL = 1e-4     # north_star: "radiance must be within 1e-4 relative per channel"
RADIANCE_ATOL = 1e-7     # floor for channels that are (near) zero


def assert_radiance(got, ref):
    err = np.abs(got - ref)
    tol = RADIANCE_RTOL * np.abs(ref) + RADIANCE_ATOL
    bad = err > tol
    assert not bad.any(), (f"{int(bad.sum())} channel values beyond 1e-4 relative; "
                           f"worst rel err {float((err / np.maximum(np.abs(ref), 1e-12)).max()):.3g}")


def render_both(ctx_factory, scene, cam, w, h, spp, depth):
    ctx = ctx_factory(w, h)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.reset()
    ctx.reset_stats()
    ctx.render(spp, depth)
    got = ctx.read_framebuffer()
    st = ctx.stats()
    ref = orc.render(scene, cam, w, h, spp=spp, max_depth=depth)
    return got, ref, st


def check_images(got, ref, st=None):
    assert np.array_equal(got["prim_id"], ref["prim_id"])
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
    assert_radiance(got["rgba"], ref["rgba"])
    if st is not None:
        assert st.primary_rays == ref["primary"]
        assert st.shadow_rays == ref["shadow"]
        assert st.bounce_rays == ref["bounce"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_fixtures(gpu_ctx_factory, name):
    """HIP path against the committed fixtures (tests/golden, scene arrays stored inside)."""
    scene, cam, w, h, crop, z, meta = load_case(name)
    x0, y0, x1, y1 = crop
    ctx = gpu_ctx_factory(w, h)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    for spp, depth in meta["runs"]:
        ctx.reset()
        ctx.render(spp, depth)
        got = ctx.read_framebuffer()
        k = f"s{spp}_d{depth}"
        assert np.array_equal(got["prim_id"][y0:y1, x0:x1], z[k + "_prim"])
        assert np.array_equal(got["t"][y0:y1, x0:x1].view(np.uint32), z[k + "_t"].view(np.uint32))
        assert_radiance(got["rgba"][y0:y1, x0:x1], z[k + "_rgba"])


def test_scene_hit_mixed(gpu_ctx_factory):
    """Scene::hit on random rays: every primitive type, transforms, misses."""
    scene, rng = scenes.mixed_test_scene()
    ctx = gpu_ctx_factory(16, 16)
    ctx.upload_scene(scene)
    n = 20000
    org = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    org[:, 2] += 40.0
    tgt = rng.uniform([-350, -150, -900], [350, 300, -150], (n, 3)).astype(np.float32)
    d = tgt - org
    t, pid, hit = ctx.scene_hit(org, d)
    rt, rpid, rhit = orc.scene_hit(scene, org, d)
    assert np.array_equal(pid, rpid)
    assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    assert np.array_equal(hit, rhit)
    assert 0.2 < hit.mean() < 1.0 and len(np.unique(pid)) > 50


def soup_scene(rng, n_tris):
    """Unordered triangle soup: sizes from invisible (H1) to huge, slivers, duplicates (exact
    ties in t), coplanar pairs; plus two spheres.  Index order is spatially random, so the
    device order (the tree's rank partition) differs from the reference order everywhere."""
    b = SceneBuilder()
    P = []
    for i in range(n_tris):
        c = rng.uniform([-250, -150, -700], [250, 200, -200])
        s = rng.choice([0.03, 1.0, 6.0, 40.0, 160.0], p=[0.05, 0.3, 0.4, 0.2, 0.05])
        a, bb, d = (c + rng.normal(size=3) * s for _ in range(3))
        if i % 17 == 0:
            d = a + (bb - a) * 0.5 + rng.normal(size=3) * s * 0.02          # sliver
        P.append(np.array([a, bb, d], np.float32))
    for i in range(0, n_tris, 11):                                           # exact duplicates
        P[(i + 5) % n_tris] = P[i].copy()
    P = np.stack(P)
    N = np.broadcast_to(np.float32([0, 0, 1]), P.shape).copy()
    b.add_triangle(P[0], N[0], identity(), (0.7, 0.6, 0.5))
    b.add_triangle(P[1], N[1], translation(3, -2, 5), (0.5, 0.6, 0.7))
    h = (n_tris - 2) // 2
    b.add_mesh(P[2:2 + h], N[2:2 + h], identity(), (0.7, 0.7, 0.7))
    b.add_mesh(P[2 + h:], N[2 + h:], identity(), (0.6, 0.7, 0.6))
    b.add_sphere(1.0 / 35.0, translation(-60.0, 10.0, -420.0), (0.7, 0.2, 0.1))
    b.add_sphere(1.0 / 20.0, translation(90.0, -40.0, -300.0), (0.3, 0.3, 0.8))
    b.add_point_light((-300.0, 300.0, -100.0), (3.0e5, 2.5e5, 2.0e5))
    b.add_point_light((250.0, 150.0, -200.0), (1.0e5, 1.5e5, 2.0e5))
    return b.build()


@pytest.mark.parametrize("n_tris,seed", [(5, 11), (300, 12), (4000, 13)])
def test_culling_is_results_identical(gpu_ctx_factory, n_tris, seed):
    """Box-tree traversal == scan of every triangle == oracle, bit for bit, on a soup
    with ties, slivers and invisible triangles; rays from inside and far outside the scene
    (the far ones exceed the culling extent and take the unculled path)."""
    rng = np.random.default_rng(seed)
    scene = soup_scene(rng, n_tris)
    ctx = gpu_ctx_factory(16, 16)
    ctx.upload_scene(scene)
    n = 30000
    org = rng.uniform([-300, -200, -750], [300, 250, 100], (n, 3)).astype(np.float32)
    org[:200] *= 50.0                                   # far outside: "wild" rays
    tgt = rng.uniform([-250, -150, -700], [250, 200, -200], (n, 3)).astype(np.float32)
    d = tgt - org
    rt, rpid, rhit = orc.scene_hit(scene, org, d)
    for cull in (1, 0):
        ctx.set_culling(cull)
        t, pid, hit = ctx.scene_hit(org, d)
        assert np.array_equal(pid, rpid), f"cull={cull}: {(pid != rpid).sum()} ids differ"
        assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
        assert np.array_equal(hit, rhit)
    ctx.set_culling(1)
    assert hit.mean() > (0.3 if n_tris >= 300 else 0.01)


def test_culling_render_equals_scan(gpu_ctx_factory):
    """Whole render (primary + shadow + bounces) with and without culling: identical bits."""
    scene = soup_scene(np.random.default_rng(21), 1500)
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    out = []
    for cull in (1, 0):
        ctx.set_culling(cull)
        ctx.reset()
        ctx.render(2, 3)
        out.append(ctx.read_framebuffer())
    ctx.set_culling(1)
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    ref = orc.render(scene, cam, 96, 64, spp=2, max_depth=3)
    check_images(out[0], ref)


def test_heavy_chunk_feedback_and_cooperative_form(gpu_ctx_factory):
    """The path kernel over a deep tree with bounces (max_depth > 0: the depth-0 pipeline does not
    apply).  From the second progressive sample on, chunks the previous sample found heavy are
    handed out first, one per BLOCK, and shared by its four waves (subtrees by hash, hits merged
    by a 64-bit ds_min in LDS, barriers around every traversal).  rt_stats.coop_chunks proves
    that form ran.  Six one-sample calls at depth 12 must match the oracle bit for bit in id / t
    (radiance 1e-4, ray counts equal) and the plain scan in every bit of the image."""
    cfg = scenes.c3(256, 144, f=12, ground=(10, 6))
    ctx = gpu_ctx_factory(256, 144)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    out, stats = [], []
    for cull in (1, 0):
        ctx.set_culling(cull)
        ctx.reset()
        ctx.reset_stats()
        for _ in range(6):
            ctx.render(1, 12)
        out.append(ctx.read_framebuffer())
        stats.append(ctx.stats())
    ctx.set_culling(1)
    assert stats[0].coop_chunks > 0, "no chunk ran in the cooperative form"
    assert stats[1].coop_chunks == 0                        # the scan has no feedback, hence no heavy chunks
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    assert stats[0].shadow_rays == stats[1].shadow_rays and stats[0].bounce_rays == stats[1].bounce_rays
    ref = orc.render(cfg.scene, cfg.camera, 256, 144, spp=6, max_depth=12)
    check_images(out[0], ref, stats[0])


def test_depth0_pipeline_on_small_deep_tree(gpu_ctx_factory):
    """render(2, 0) three times over the same deep tree: the batch kernel (k_batch: phases P,
    H, S of a unit) in batches of two, against the oracle and the plain scan."""
    cfg = scenes.c3(256, 144, f=12, ground=(10, 6))
    ctx = gpu_ctx_factory(256, 144)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    out = []
    for cull in (1, 0):
        ctx.set_culling(cull)
        ctx.reset()
        ctx.reset_stats()
        for _ in range(3):
            ctx.render(2, 0)
        out.append(ctx.read_framebuffer())
        st = ctx.stats()
    ctx.set_culling(1)
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    ref = orc.render(cfg.scene, cfg.camera, 256, 144, spp=6, max_depth=0)
    check_images(out[0], ref, st)


def flicker_scene(rng, n_tris):
    """Small triangles scattered through the view with RANDOM vertex normals: most pixels flip
    between hit and miss from sample to sample, and many hits have wo.n < 0 (an unusable
    BRDF sample, PathTracer.cu:77) -- every draw count a sample can have, in every order."""
    b = SceneBuilder()
    c = rng.uniform([-220, -130, -520], [220, 170, -380], (n_tris, 1, 3))
    P = (c + rng.normal(size=(n_tris, 3, 3)) * rng.choice([3.0, 8.0, 16.0], (n_tris, 1, 1))).astype(np.float32)
    N = rng.normal(size=(n_tris, 3, 3)).astype(np.float32)
    N /= np.linalg.norm(N, axis=2, keepdims=True)
    h = n_tris // 2
    b.add_mesh(P[:h], N[:h], identity(), (0.7, 0.7, 0.7))
    b.add_mesh(P[h:], N[h:], identity(), (0.6, 0.7, 0.6))
    b.add_point_light((-300.0, 300.0, -100.0), (3.0e5, 2.5e5, 2.0e5))
    b.add_point_light((250.0, 150.0, -200.0), (1.0e5, 1.5e5, 2.0e5))
    return b.build()


@pytest.mark.parametrize("spp", [16, 21, 70, 140])
def test_sample_batches_on_flickering_pixels(gpu_ctx_factory, spp):
    """rt_render(n, 0) over a deep tree traces all samples of a batch at once from PREDICTED
    RNG counters and settles mispredicted pixels by enumeration (phases P and H of k_batch).
    Batches are of 16 samples, of 32 from 64 samples on (70 = 32 + 32 + 6), of 64 from 128 on
    (140 = 64 + 64 + 12).
    On a scene built to defeat the prediction it must equal (a) the oracle, (b) the same
    samples rendered one call at a time or in uneven calls, (c) the unculled per-sample path
    kernel -- in every
    bit of image, id and t buffers, and in the ray counts."""
    scene = flicker_scene(np.random.default_rng(31), 3000)
    cam = CameraArgs.for_image(80, 48, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    ctx = gpu_ctx_factory(80, 48)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    out, stats = [], []
    for mode in ("batched", "one by one", "scan", "uneven calls"):
        ctx.set_culling(0 if mode == "scan" else 1)
        ctx.reset()
        ctx.reset_stats()
        if mode == "one by one":
            for _ in range(spp):
                ctx.render(1, 0)
        elif mode == "uneven calls":                    # batches of 5, 3 and the rest (16s / 32s + remainder)
            ctx.render(5, 0)
            ctx.render(3, 0)
            ctx.render(spp - 8, 0)
        else:
            ctx.render(spp, 0)
        out.append(ctx.read_framebuffer())
        stats.append(ctx.stats())
    ctx.set_culling(1)
    for other in (1, 2, 3):
        for k in ("rgba", "t", "prim_id"):
            assert np.array_equal(out[0][k].view(np.uint32), out[other][k].view(np.uint32)), (other, k)
        assert stats[0].shadow_rays == stats[other].shadow_rays
    hit = out[0]["prim_id"] >= 0
    assert 0.15 < hit.mean() < 0.85                     # the last sample's hits: a coin flip per pixel
    # the oracle at every batch shape: 16 (8 px x 16), 21 = 16 + 5, 70 = 32 + 32 + 6 (4 px x 32,
    # the enumeration's four rounds), 140 = 64 + 64 + 12 (2 px x 64, eight rounds)
    ref = orc.render(scene, cam, 80, 48, spp=spp, max_depth=0)
    check_images(out[0], ref, stats[0])


@pytest.mark.parametrize("batch", ["1", "5", "64"])
def test_batch_size_does_not_change_the_image(gpu_ctx_factory, monkeypatch, batch):
    """RT_BATCH (read at rt_create) caps the samples per batch; any cap gives the same bits."""
    scene = flicker_scene(np.random.default_rng(37), 2000)
    cam = CameraArgs.for_image(64, 40, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    out = []
    for env in (None, batch):
        if env is None:
            monkeypatch.delenv("RT_BATCH", raising=False)
        else:
            monkeypatch.setenv("RT_BATCH", env)
        ctx = gpu_ctx_factory(64, 40)
        ctx.upload_scene(scene)
        ctx.set_camera(cam)
        ctx.render(130, 0)
        out.append(ctx.read_framebuffer())
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k


def test_sample_batches_with_spheres_and_ties(gpu_ctx_factory):
    """The batch kernels over the triangle soup (exact duplicates, slivers, invisible triangles)
    plus two quadric spheres: 18 samples at depth 0 against the oracle and against the unculled
    path kernel."""
    scene = soup_scene(np.random.default_rng(23), 1500)
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.set_timing(True)                                  # launch counts come with the timing events
    out, stats = [], []
    for cull in (1, 0):
        ctx.set_culling(cull)
        ctx.reset()
        ctx.reset_stats()
        ctx.render(18, 0)
        out.append(ctx.read_framebuffer())
        stats.append(ctx.stats())
    ctx.set_culling(1)
    ctx.set_timing(False)
    assert stats[0].batch_launches > 0 and stats[1].batch_launches == 0       # batch kernel vs path kernel
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    ref = orc.render(scene, cam, 96, 64, spp=18, max_depth=0)
    check_images(out[0], ref, stats[0])
    assert (out[0]["prim_id"] >= scene.n_tris).any()                          # a sphere is visible


@pytest.mark.parametrize("depth,spp", [(0, 1), (0, 3), (12, 2)])
def test_c1_sphere(gpu_ctx_factory, depth, spp):
    cfg = scenes.c1(160, 120)
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 160, 120, spp, depth)
    check_images(got, ref, st)
    assert (got["prim_id"] >= 0).sum() > 100


@pytest.mark.parametrize("depth", [0, 12])
def test_c1_full_size(gpu_ctx_factory, depth):
    """C1 as BASELINE.json states it: 640x480, 1 spp, the whole frame against the oracle."""
    cfg = scenes.c1()
    assert (cfg.width, cfg.height, cfg.spp) == (640, 480, 1)
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 640, 480, cfg.spp, depth)
    check_images(got, ref, st)
    assert (got["prim_id"] >= 0).sum() > 1000


@pytest.mark.parametrize("depth,spp", [(0, 1), (0, 4), (12, 2)])
def test_c2_cornell(gpu_ctx_factory, depth, spp):
    cfg = scenes.c2(192, 108)
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 192, 108, spp, depth)
    check_images(got, ref, st)
    assert (got["prim_id"] >= 0).mean() > 0.95


@pytest.mark.parametrize("depth,spp", [(0, 2), (0, 16), (12, 2)])
def test_c3_decimated(gpu_ctx_factory, depth, spp):
    cfg = scenes.c3(128, 72, lon=24, lat=24, ground=(8, 4))
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 128, 72, spp, depth)
    check_images(got, ref, st)


@pytest.mark.parametrize("depth,spp", [(0, 2), (5, 2)])
def test_mixed_scene(gpu_ctx_factory, depth, spp):
    scene, _ = scenes.mixed_test_scene()
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    got, ref, st = render_both(gpu_ctx_factory, scene, cam, 96, 64, spp, depth)
    check_images(got, ref, st)


@pytest.mark.parametrize("depth,spp", [(0, 3), (6, 2)])
def test_directional_lights(gpu_ctx_factory, depth, spp):
    """Point and directional lights together (the delta lights of Scene.cu:93-102), in
    sampleAllLight's order; the directional ones cast shadow rays 10 000 units long."""
    scene, _ = scenes.sun_test_scene()
    assert scene.n_lights == 4 and scene.dl_dir.shape[0] == 2
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    got, ref, st = render_both(gpu_ctx_factory, scene, cam, 96, 64, spp, depth)
    check_images(got, ref, st)
    assert st.shadow_rays > 4000


def test_directional_light_deep_tree(gpu_ctx_factory):
    """The same through the depth-0 pipeline (deep tree) with only a directional light."""
    cfg = scenes.c3(128, 72, lon=24, lat=24, ground=(8, 4))
    cfg.scene.pl_pos = cfg.scene.pl_pos[:1].copy()
    cfg.scene.pl_illum = cfg.scene.pl_illum[:1].copy()
    cfg.scene.dl_dir = np.float32([[0.3, -1.0, -0.2]])
    cfg.scene.dl_illum = np.float32([[1.5, 1.4, 1.2]])
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 128, 72, 3, 0)
    check_images(got, ref, st)


def test_many_lights(gpu_ctx_factory):
    """More lights than one 64-bit shadow mask holds: exercises the multi-pass light loop."""
    cfg = scenes.c5(64, 36, nx=12, nz=10, n_light_side=9)       # 81 lights
    assert cfg.scene.n_lights == 81
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 64, 36, 2, 0)
    check_images(got, ref, st)


def test_ragged_and_empty(gpu_ctx_factory):
    """Image sizes that are not multiples of the 8x8 wave block / 64-pixel tile; empty scene."""
    cfg = scenes.c2(37, 23)
    got, ref, st = render_both(gpu_ctx_factory, cfg.scene, cfg.camera, 37, 23, 2, 0)
    check_images(got, ref, st)
    empty = SceneBuilder().build()
    got, ref, st = render_both(gpu_ctx_factory, empty, cfg.camera, 37, 23, 1, 0)
    check_images(got, ref, st)
    assert (got["prim_id"] == -1).all() and (got["rgba"][..., :3] == 0).all()


def test_partition_union_equals_whole(gpu_ctx_factory):
    """Tile partition (multi-GPU sharding): rank images are disjoint and their union is the
    single-context image, bit for bit (RNG keyed by the global pixel index)."""
    cfg = scenes.c2(200, 136)
    whole = gpu_ctx_factory(200, 136)
    whole.upload_scene(cfg.scene)
    whole.set_camera(cfg.camera)
    whole.render(2, 0)
    ref = whole.read_framebuffer()
    acc = np.zeros_like(ref["rgba"])
    seen = np.zeros((136, 200), np.int32)
    part = gpu_ctx_factory(200, 136)
    part.upload_scene(cfg.scene)
    part.set_camera(cfg.camera)
    for rank in range(3):
        part.set_partition(rank, 3, 64)
        part.render(2, 0)
        fb = part.read_framebuffer()
        own = fb["prim_id"] != -2
        assert part.local_pixel_count() == own.sum()
        seen += own
        acc += fb["rgba"]
    assert (seen == 1).all()
    assert np.array_equal(acc.view(np.uint32), ref["rgba"].view(np.uint32))


def test_progressive_accumulation(gpu_ctx_factory):
    """render(1) three times == render(3): the frame counter and RNG streams persist."""
    cfg = scenes.c2(64, 40)
    a = gpu_ctx_factory(64, 40)
    a.upload_scene(cfg.scene)
    a.set_camera(cfg.camera)
    a.reset()
    for _ in range(3):
        a.render(1, 0)
    x = a.read_framebuffer()
    a.reset()
    a.render(3, 0)
    y = a.read_framebuffer()
    assert np.array_equal(x["rgba"].view(np.uint32), y["rgba"].view(np.uint32))


def test_postprocess(gpu_ctx_factory):
    cfg = scenes.c2(96, 54)
    ctx = gpu_ctx_factory(96, 54)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.reset()
    ctx.render(2, 0)
    raw = ctx.read_framebuffer()["rgba"]
    hdr, flt = ctx.postprocess()
    rh = orc.tonemap(raw)
    assert np.array_equal(hdr.view(np.uint32), rh.view(np.uint32))
    assert np.array_equal(flt.view(np.uint32), orc.median3x3(rh).view(np.uint32))


def test_errors(gpu_ctx_factory):
    ctx = abi.Context(8, 8)
    with pytest.raises(abi.RtError) as e:
        ctx.render(1, 0)
    assert e.value.code == abi.RT_ERR_STATE
    cfg = scenes.c1(8, 8)
    cfg.scene.model_mat_type[:] = 5           # not a material::MATERIAL_TYPE (Object.cuh:13-21)
    with pytest.raises(abi.RtError) as e:
        ctx.upload_scene(cfg.scene)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    ctx.close()


# ---------------------------------------------------------------- BASELINE.json's full sizes
def _oracle_crops(cfg, crops, spp, depth=0):
    out = []
    for (x0, y0, cw, ch) in crops:
        out.append(((x0, y0, cw, ch),
                    orc.render(cfg.scene, cfg.camera, cfg.width, cfg.height, spp=spp,
                               max_depth=depth, crop=(x0, y0, x0 + cw, y0 + ch))))
    return out


def _check_crops(got, refs):
    for (x0, y0, cw, ch), ref in refs:
        sl = (slice(y0, y0 + ch), slice(x0, x0 + cw))
        assert np.array_equal(got["prim_id"][sl], ref["prim_id"]), (x0, y0)
        assert np.array_equal(got["t"][sl].view(np.uint32), ref["t"].view(np.uint32)), (x0, y0)
        assert_radiance(got["rgba"][sl], ref["rgba"])


def test_full_size_c2(gpu_ctx_factory):
    """C2 as BASELINE.json states it (1920x1080, 16 spp): oracle on windows of the frame."""
    cfg = scenes.c2()
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.render(cfg.spp, 0)
    got = ctx.read_framebuffer()
    crops = [(0, 0, 40, 24), (940, 520, 48, 32), (1872, 1050, 48, 30), (300, 900, 32, 32)]
    _check_crops(got, _oracle_crops(cfg, crops, cfg.spp))
    assert (got["prim_id"] >= 0).mean() > 0.9


def test_full_size_c3(gpu_ctx_factory):
    """C3 at its full size (69 451 triangles, 3 lights, 1920x1080): the depth-0 pipeline over
    several batches against the oracle on windows at the silhouette, the ground and the sky;
    and rendering in two calls equals rendering in one (counters and streams persist)."""
    cfg = scenes.c3()
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    spp = 3
    ctx.render(spp, 0)
    got = ctx.read_framebuffer()
    hit = got["prim_id"] >= 0
    ys, xs = np.nonzero(hit[:, :-1] != hit[:, 1:])          # silhouette / horizon columns
    k = len(xs) // 2
    crops = [(int(min(max(xs[k] - 8, 0), cfg.width - 16)), int(min(max(ys[k] - 4, 0), cfg.height - 8)), 16, 8),
             (952, 536, 16, 8), (40, 1040, 16, 8), (900, 4, 16, 4)]
    _check_crops(got, _oracle_crops(cfg, crops, spp))
    ctx.reset()
    ctx.render(2, 0)
    ctx.render(1, 0)
    again = ctx.read_framebuffer()
    for key in ("rgba", "t", "prim_id"):
        assert np.array_equal(got[key].view(np.uint32), again[key].view(np.uint32)), key


def test_full_size_c3_with_bounces(gpu_ctx_factory):
    """C3 at its full size at the reference's default MAX_DEPTH = 12 (PathTracer.cu:3-5): the bounce kernel
    with chunk feedback, cooperative chunks and the per-lane traversal of scattered rays over the 69 451-
    triangle tree, three one-sample launches, against the oracle on windows at the silhouette, on the mesh,
    on the ground and in the sky."""
    cfg = scenes.c3()
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    spp, depth = 3, 12
    ctx.render(spp, depth)
    got = ctx.read_framebuffer()
    st = ctx.stats()
    assert st.bounce_rays > 0 and st.batch_launches == 0
    hit = got["prim_id"] >= 0
    ys, xs = np.nonzero(hit[:, :-1] != hit[:, 1:])
    k = len(xs) // 3
    crops = [(int(min(max(xs[k] - 6, 0), cfg.width - 12)), int(min(max(ys[k] - 3, 0), cfg.height - 6)), 12, 6),
             (954, 538, 12, 6), (40, 1040, 12, 6), (900, 4, 12, 3)]
    _check_crops(got, _oracle_crops(cfg, crops, spp, depth))


def test_full_size_c4_rays(gpu_ctx_factory, c4_full):
    """C4's full mesh (871 414 triangles): Scene::hit with the tree equals the scan of every
    triangle, bit for bit, on camera-like and random rays; a handful of them against the oracle."""
    cfg = c4_full
    ctx = gpu_ctx_factory(64, 64)
    ctx.upload_scene(cfg.scene)
    rng = np.random.default_rng(5)
    n = 4096
    pos = np.asarray(cfg.camera.pos, np.float32)
    look = np.asarray(cfg.camera.lookat, np.float32)
    org = np.tile(pos, (n, 1)).astype(np.float32)
    span = np.abs(cfg.scene.tri_pos).max()
    tgt = look + rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32) * np.float32(span)
    org[n // 2:] = rng.uniform(-1.0, 1.0, (n - n // 2, 3)).astype(np.float32) * np.float32(span)
    d = (tgt - org).astype(np.float32)
    ctx.set_culling(1)
    t1, id1, h1 = ctx.scene_hit(org, d)
    ctx.set_culling(0)
    t0, id0, h0 = ctx.scene_hit(org, d)
    ctx.set_culling(1)
    assert np.array_equal(id1, id0) and np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
    assert np.array_equal(h1, h0) and 0.05 < h1.mean() < 1.0
    pick = np.concatenate([np.nonzero(h1)[0][:12], np.nonzero(h1 == 0)[0][:4]])
    rt, rid, rh = orc.scene_hit(cfg.scene, org[pick], d[pick])
    assert np.array_equal(id1[pick], rid) and np.array_equal(t1[pick].view(np.uint32), rt.view(np.uint32))


def test_full_size_c5_rays(gpu_ctx_factory, c5_full):
    """C5's 10 M-triangle mesh: tree == scan on rays from the camera and towards the lights."""
    cfg = c5_full
    assert cfg.scene.n_tris == 10_000_000 and cfg.scene.n_lights == 64
    ctx = gpu_ctx_factory(64, 64)
    ctx.upload_scene(cfg.scene)
    rng = np.random.default_rng(9)
    n = 1024
    pos = np.asarray(cfg.camera.pos, np.float32)
    look = np.asarray(cfg.camera.lookat, np.float32)
    span = np.float32(np.abs(cfg.scene.tri_pos[::997]).max())
    org = np.tile(pos, (n, 1)).astype(np.float32)
    tgt = look + rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32) * span
    # second half: from points just above the surface towards the lights (shadow-ray like)
    v = cfg.scene.tri_pos[rng.integers(0, cfg.scene.n_tris, n // 2), 0, :]
    org[n // 2:] = v + np.float32([0.0, 0.5, 0.0])
    tgt[n // 2:] = cfg.scene.pl_pos[rng.integers(0, 64, n // 2)]
    d = (tgt - org).astype(np.float32)
    ctx.set_culling(1)
    t1, id1, h1 = ctx.scene_hit(org, d)
    ctx.set_culling(0)
    t0, id0, h0 = ctx.scene_hit(org, d)
    ctx.set_culling(1)
    assert np.array_equal(id1, id0) and np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
    assert np.array_equal(h1, h0) and h1.mean() > 0.1


def _check_windows(got, cfg, crops, spp, depth=0):
    """All pixels of the windows (x0, y0, w, h) in ONE oracle call, one task per pixel."""
    xy = np.array([(x0 + i, y0 + j) for (x0, y0, cw, ch) in crops for j in range(ch) for i in range(cw)],
                  np.int32)
    ref = orc.render_pixels(cfg.scene, cfg.camera, cfg.width, cfg.height, xy, spp=spp, max_depth=depth)
    ys, xs = xy[:, 1], xy[:, 0]
    assert np.array_equal(got["prim_id"][ys, xs], ref["prim_id"])
    assert np.array_equal(got["t"][ys, xs].view(np.uint32), ref["t"].view(np.uint32))
    assert_radiance(got["rgba"][ys, xs], ref["rgba"])
    return ref


def _silhouette_window(got, cfg, cw, ch):
    """A cw x ch window straddling the hit / miss boundary nearest the middle of the frame."""
    hit = got["prim_id"] >= 0
    ys, xs = np.nonzero(hit[:, :-1] != hit[:, 1:])
    assert len(xs) > 0
    k = len(xs) // 2
    return (int(min(max(xs[k] - cw // 2, 0), cfg.width - cw)),
            int(min(max(ys[k] - ch // 2, 0), cfg.height - ch)), cw, ch)


def test_full_size_c3_bench_shape(gpu_ctx_factory, monkeypatch):
    """C3 exactly as the bench renders it -- 1920x1080, 64 spp in one call = one batch of 64
    samples (2 pixels x 64 samples per wave; open pixels take the enumeration's eight rounds) --
    against the oracle on windows at the silhouette, on the mesh, the ground and the sky."""
    cfg = scenes.c3()
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.reset_stats()
    ctx.set_timing(True)
    ctx.render(cfg.spp, 0)
    st = ctx.stats()
    ctx.set_timing(False)
    assert cfg.spp == 64 and st.batch_launches == st.streams   # one 64-sample batch per stream through k_batch
    got = ctx.read_framebuffer()
    crops = [_silhouette_window(got, cfg, 8, 4), (956, 538, 8, 4), (44, 1042, 8, 4), (904, 4, 8, 2)]
    ref = _check_windows(got, cfg, crops, cfg.spp)
    assert 0 < (ref["prim_id"] >= 0).sum() < len(ref["prim_id"])        # hits and misses both
    # The first launch measures the mean unit cost, the second lists the units far above it (the
    # silhouette's), the third hands those out first, one pixel per wave, and skips them in the
    # bands -- every pixel still gets each sample once.
    assert st.batch_listed == 0
    more = (cfg.spp, cfg.spp, 32, 32, 32)                   # (another shape: the 64-sample list is not for it)
    for i, spp in enumerate(more):
        ctx.render(spp, 0)
        assert (ctx.stats().batch_listed > 0) == (i not in (2,))
    got = ctx.read_framebuffer()
    _check_windows(got, cfg, crops, cfg.spp + sum(more))
    # ... and the whole frame, bit for bit, against the same renders with the ordering off
    monkeypatch.setenv("RT_FEEDBACK", "0")
    plain = gpu_ctx_factory(cfg.width, cfg.height)
    monkeypatch.delenv("RT_FEEDBACK")
    plain.upload_scene(cfg.scene)
    plain.set_camera(cfg.camera)
    for spp in (cfg.spp,) + more:
        plain.render(spp, 0)
    assert plain.stats().batch_listed == 0
    ref = plain.read_framebuffer()
    assert np.array_equal(got["rgba"].view(np.uint32), ref["rgba"].view(np.uint32))
    assert np.array_equal(got["prim_id"], ref["prim_id"])
    assert np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))


@pytest.fixture(scope="module")
def c4_full():
    return scenes.c4()


@pytest.fixture(scope="module")
def c5_full():
    return scenes.c5()


def test_full_size_c4_render(gpu_ctx_factory, c4_full):
    """C4 at its full size (871 414 triangles, 3840x2160) through rt_render: k_batch
    over the real tree.  (a) 2 spp against the oracle on windows at
    the silhouette, on the mesh, the ground and the sky; (b) 64-sample batches (2 pixels x 64
    samples per wave: the shape the bench runs) equal the same samples rendered one call at a time,
    in every bit of the image, id and t buffers and in the shadow-ray count -- three batches in a
    row, so the third is handed the second's lists (costly units first, by pixel, then the early
    ones); (c) the first 64-sample batch against the oracle on smaller windows of the same places
    (22 pixels x 64 samples x up to four scans of 871 k triangles: what the oracle does in half a
    minute on 16 threads)."""
    cfg = c4_full
    assert cfg.scene.n_tris == 871_414 and (cfg.width, cfg.height) == (3840, 2160)
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.render(2, 0)
    got = ctx.read_framebuffer()
    sil = _silhouette_window(got, cfg, 8, 4)
    crops = [sil, (1916, 1078, 8, 4), (90, 2084, 8, 4), (1800, 8, 8, 2)]
    ref = _check_windows(got, cfg, crops, 2)
    assert 0 < (ref["prim_id"] >= 0).sum() < len(ref["prim_id"])
    assert 0.2 < (got["prim_id"] >= 0).mean() < 1.0
    small = [(sil[0] + 2, sil[1] + 1, 4, 2), (1918, 1079, 4, 2), (92, 2085, 2, 2), (1803, 8, 2, 1)]
    out, shadow = [], []
    for mode in ("batch", "one by one"):
        ctx.reset()
        ctx.reset_stats()
        if mode == "batch":
            ctx.render(64, 0)
            first = ctx.read_framebuffer()
            ctx.render(64, 0)
            ctx.render(64, 0)
            st = ctx.stats()
            assert st.batch_launches == 3 and st.batch_listed + st.batch_early > 0
        else:
            for _ in range(192):
                ctx.render(1, 0)
        out.append(ctx.read_framebuffer())
        shadow.append(ctx.stats().shadow_rays)
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), k
    assert shadow[0] == shadow[1] and shadow[0] > 0
    ref = _check_windows(first, cfg, small, 64)
    assert 0 < (ref["prim_id"] >= 0).sum() < len(ref["prim_id"])


def test_full_size_c5_render(gpu_ctx_factory, c5_full):
    """C5 at its full size (10 M triangles, 64 lights, 3840x2160) through rt_render: 1 spp
    against the oracle on a handful of pixels (a hit costs the oracle 65 scans of 10 M
    triangles); a 16-sample batch against one call per sample; and 64-sample batches (the
    configuration's own sample count: 2 pixels x 64 samples per wave through the five-wave build of
    k_batch) against one call per sample -- three batches in a row, so the third takes the lists
    the second wrote."""
    cfg = c5_full
    assert cfg.scene.n_tris == 10_000_000 and cfg.scene.n_lights == 64
    ctx = gpu_ctx_factory(cfg.width, cfg.height)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.render(1, 0)
    got = ctx.read_framebuffer()
    hit = got["prim_id"] >= 0
    ys, xs = np.nonzero(hit[:, :-4])                      # pixels on the height field, near and far
    near, far = len(xs) // 8, (7 * len(xs)) // 8
    miss_y, miss_x = np.nonzero(~hit[:, :-2])
    crops = [_silhouette_window(got, cfg, 4, 1), (int(xs[near]), int(ys[near]), 4, 1),
             (int(xs[far]), int(ys[far]), 2, 1), (int(miss_x[len(miss_x) // 2]), int(miss_y[len(miss_y) // 2]), 2, 1)]
    ref = _check_windows(got, cfg, crops, 1)
    assert (ref["prim_id"] >= 0).sum() >= 6 and ref["shadow"] > 64
    for batch, n_batches in ((16, 1), (64, 3)):
        out, shadow = [], []
        for mode in ("batch", "one by one"):
            ctx.reset()
            ctx.reset_stats()
            if mode == "batch":
                for _ in range(n_batches):
                    ctx.render(batch, 0)
                st = ctx.stats()
                assert st.batch_launches == n_batches
                assert n_batches < 3 or st.batch_listed + st.batch_early > 0
            else:
                for _ in range(batch * n_batches):
                    ctx.render(1, 0)
            out.append(ctx.read_framebuffer())
            shadow.append(ctx.stats().shadow_rays)
        for k in ("rgba", "t", "prim_id"):
            assert np.array_equal(out[0][k].view(np.uint32), out[1][k].view(np.uint32)), (batch, k)
        assert shadow[0] == shadow[1] and shadow[0] > 0


def test_five_wave_batch_kernel_against_oracle(gpu_ctx_factory, monkeypatch):
    """The five-wave build of k_batch (what scenes beyond the last-level cache take: C5) on a small deep
    scene, forced with RT_BATCH_WAVES=5: whole frames against the oracle at 64 samples per launch, three
    launches in a row -- the third is handed lists (costly units first, by pixel; early units whole) --
    and bit-equal to the four-wave build's frames."""
    cfg = scenes.c3(64, 48, lon=48, lat=40, ground=(10, 6))
    frames = {}
    for waves in ("5", "4"):
        monkeypatch.setenv("RT_BATCH_WAVES", waves)
        ctx = gpu_ctx_factory(cfg.width, cfg.height)
        monkeypatch.delenv("RT_BATCH_WAVES")
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.reset_stats()
        for _ in range(3):
            ctx.render(64, 0)
        st = ctx.stats()
        assert st.batch_launches == 3
        frames[waves] = (ctx.read_framebuffer(), st)
    got, st = frames["5"]
    assert st.batch_listed + st.batch_early > 0
    ref = orc.render(cfg.scene, cfg.camera, cfg.width, cfg.height, spp=192, max_depth=0)
    check_images(got, ref, st)
    for k in ("rgba", "t", "prim_id"):
        assert np.array_equal(got[k].view(np.uint32), frames["4"][0][k].view(np.uint32)), k


# ----------------------------------------------------------------------------- triangle lights (N3)
def test_triangle_lights_full_frames(gpu_ctx_factory):
    """Scenes with TriangleLights (hit-able, emissive, sampled with MIS: Light.cu:35-105,
    PathTracer.cu:33-52, Scene.cu:43-64,242-284) through k_path_general: whole frames against the
    oracle with ray counts, at depth 0 and with bounces off the lights' own surfaces; the tree
    and the scan agree bit for bit; rendering in two calls equals one."""
    scene, _ = scenes.area_test_scene()
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    for spp, depth in [(1, 0), (3, 4)]:
        got, ref, st = render_both(gpu_ctx_factory, scene, cam, 96, 64, spp, depth)
        check_images(got, ref, st)
        assert st.batch_launches == 0
        n_prims = scene.n_tris + scene.n_quads
        assert (got["prim_id"] >= n_prims).sum() > 50         # the camera sees a light
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.set_culling(0)
    ctx.render(2, 4)
    ctx.render(1, 4)
    scan = ctx.read_framebuffer()
    assert np.array_equal(scan["rgba"].view(np.uint32), got["rgba"].view(np.uint32))
    assert np.array_equal(scan["prim_id"], got["prim_id"])


def test_reference_demo_scene(gpu_ctx_factory):
    """The reference's own demo scene (kernel.cu:268-313: two spheres under a two-sided triangle
    light, camera of kernel.cu:228-229) without its environment light, 320x240, 4 spp, depth 12."""
    scene = scenes.demo_scene()
    cam = CameraArgs.for_image(320, 240)
    got, ref, st = render_both(gpu_ctx_factory, scene, cam, 320, 240, 4, 12)
    check_images(got, ref, st)
    assert (got["prim_id"] == 2).sum() > 1000 and st.bounce_rays > 0


def test_scene_hit_with_triangle_lights(gpu_ctx_factory):
    """Scene::hit tests the triangle lights after every model (Scene.cu:43-58)."""
    scene, rng = scenes.area_test_scene()
    ctx = gpu_ctx_factory(16, 16)
    ctx.upload_scene(scene)
    n = 20000
    org = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    org[:, 2] += 40.0
    tgt = rng.uniform([-350, -150, -900], [350, 300, -150], (n, 3)).astype(np.float32)
    tgt[: n // 4] = rng.uniform([100, -70, -360], [200, 100, -290], (n // 4, 3))     # at the standing light
    d = tgt - org
    rt, rpid, rhit = orc.scene_hit(scene, org, d)
    for cull in (1, 0):
        ctx.set_culling(cull)
        t, pid, hit = ctx.scene_hit(org, d)
        assert np.array_equal(pid, rpid)
        assert np.array_equal(t.view(np.uint32), rt.view(np.uint32))
        assert np.array_equal(hit, rhit)
    assert (pid >= scene.n_tris + scene.n_quads).sum() > 200


def test_triangle_light_material_must_be_lambertian(gpu_ctx_factory):
    scene = scenes.demo_scene()
    scene.tl_mat_type = np.int32([3])
    ctx = gpu_ctx_factory(16, 16)
    with pytest.raises(abi.RtError) as e:
        ctx.upload_scene(scene)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED


# ----------------------------------------------------------------------------- every BRDF (N4)
def test_every_brdf_full_frames(gpu_ctx_factory):
    """GGX, Cook-Torrance, Oren-Nayar and Fresnel (material.cu:84-254, material.cuh:108-164) next to
    Lambertian, under point, directional and triangle lights: whole frames against the oracle with
    ray counts, at depth 0 (the Fresnel sphere's last ray is traced for the light it may land on)
    and with bounces; tree == scan bit for bit; two calls == one."""
    scene, _ = scenes.material_test_scene()
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    for spp, depth in [(2, 0), (3, 6)]:
        got, ref, st = render_both(gpu_ctx_factory, scene, cam, 96, 64, spp, depth)
        check_images(got, ref, st)
        assert st.batch_launches == 0
    for model_prims in (range(0, 1), range(1, 2), range(2, 282), range(282, 408), range(408, 409), range(409, 410)):
        seen = np.isin(got["prim_id"], list(model_prims))
        assert seen.sum() > 20 and got["rgba"][..., :3][seen].max() > 0        # every material is in view and lit
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.set_culling(0)
    ctx.render(1, 6)
    ctx.render(2, 6)
    scan = ctx.read_framebuffer()
    assert np.array_equal(scan["rgba"].view(np.uint32), got["rgba"].view(np.uint32))


def test_material_parameters_are_required(gpu_ctx_factory):
    from raytracer_amd.scene import SceneDesc
    import ctypes as C
    cfg = scenes.c1(8, 8)
    cfg.scene.model_mat_type[:] = 17
    d = cfg.scene.desc()
    d.model_param = C.POINTER(C.c_float)()                   # GGX without its roughness
    ctx = gpu_ctx_factory(8, 8)
    rc = abi.lib().rt_upload_scene(ctx._h, C.byref(d))
    assert rc == abi.RT_ERR_INVALID


def test_sample_runs_equal_single_samples(gpu_ctx_factory):
    """Over a shallow tree (C1, C2) and in scan mode one k_path launch renders a run of samples per chunk
    (path_run): bit-identical to one launch per sample, at depth 0 and with bounces, and the launch count
    says which happened."""
    cfg = scenes.c2(160, 90)
    for depth, cull in [(0, 1), (5, 1), (3, 0)]:
        a = gpu_ctx_factory(cfg.width, cfg.height)
        a.upload_scene(cfg.scene)
        a.set_camera(cfg.camera)
        a.set_culling(cull)
        a.reset_stats()
        a.render(7, depth)                                   # runs of 4 and 3
        one = a.read_framebuffer()
        assert a.stats().path_launches == 2 * a.stats().streams
        a.reset()
        a.reset_stats()
        for n in (1, 1, 2, 3):                               # runs of 1, 1, 2, 3
            a.render(n, depth)
        assert a.stats().path_launches == 4 * a.stats().streams
        split = a.read_framebuffer()
        assert np.array_equal(one["rgba"].view(np.uint32), split["rgba"].view(np.uint32))
        assert np.array_equal(one["prim_id"], split["prim_id"])
        ref = orc.render(cfg.scene, cfg.camera, cfg.width, cfg.height, spp=7, max_depth=depth)
        check_images(one, ref)
    # a deep tree with bounces keeps one launch per sample (feedback, cooperative chunks)
    deep = scenes.c3(96, 54, lon=24, lat=24, ground=(8, 4))
    b = gpu_ctx_factory(deep.width, deep.height)
    b.upload_scene(deep.scene)
    b.set_camera(deep.camera)
    b.reset_stats()
    b.render(3, 4)
    assert b.stats().path_launches == 3 * b.stats().streams


def test_lane_traversal_equals_wave_traversal(gpu_ctx_factory, monkeypatch):
    """Scattered rays walk the tree lane by lane (traverse_lane) or with the whole wave (traverse): the same
    node test, triangle test and tie rule either way, so every bit of the frame is the same -- over a soup
    with exact ties and spheres, through k_path (both builds of the bounce kernel) and through the general
    kernel (triangle lights, every BRDF)."""
    soup = soup_scene(np.random.default_rng(33), 1200)
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    for scene in (soup, scenes.material_test_scene()[0]):
        frames = []
        for knobs in ({"RT_LANE_TRAVERSAL": "0"}, {}, {"RT_BOUNCE_WAVES": "2"}, {"RT_BOUNCE_QUEUE": "1"}, {"RT_BOUNCE_QUEUE": "2"}):
            for k, val in knobs.items():
                monkeypatch.setenv(k, val)
            ctx = gpu_ctx_factory(96, 64)
            ctx.upload_scene(scene)
            ctx.set_camera(cam)
            ctx.render(3, 7)
            frames.append(ctx.read_framebuffer())
            for k in knobs:
                monkeypatch.delenv(k)
        for other in frames[1:]:
            assert np.array_equal(frames[0]["rgba"].view(np.uint32), other["rgba"].view(np.uint32))
            assert np.array_equal(frames[0]["prim_id"], other["prim_id"])
    ref = orc.render(soup, cam, 96, 64, spp=3, max_depth=7)
    # (frames: of the materials scene; the soup against the oracle once more)
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(soup)
    ctx.set_camera(cam)
    ctx.render(3, 7)
    check_images(ctx.read_framebuffer(), ref)


def test_bounce_queue_equals_one_kernel_form(gpu_ctx_factory, monkeypatch):
    """Bounces over a deep tree as two launches per sample (RT_BOUNCE_QUEUE=1: k_path's EMIT build up to the BRDF
    sample of the first hit, then k_bounce's persistent waves, which take the remaining paths from a queue lane by
    lane, one or two paths per lane) or as one (the default: k_path carries every path to its end).  Same arithmetic per path, so the same bits, the
    same ray counts and the same Philox counters (a second render call continues from them): a decimated C3
    whole and as the union of three partitions, with few persistent blocks too, and against the oracle."""
    cfg = scenes.c3(160, 96, lon=40, lat=30, ground=(8, 5))
    spp, depth = 3, 12
    frames, counts = [], []
    for knobs in ({"RT_BOUNCE_QUEUE": "0"}, {"RT_BOUNCE_QUEUE": "1"}, {"RT_BOUNCE_QUEUE": "1", "RT_MAX_BLOCKS": "2"},
                  {"RT_BOUNCE_QUEUE": "1", "RT_BOUNCE_PAIRS": "1"}, {"RT_BOUNCE_QUEUE": "2"},
                  {"RT_BOUNCE_QUEUE": "2", "RT_MAX_BLOCKS": "1"}):
        for k, val in knobs.items():
            monkeypatch.setenv(k, val)
        ctx = gpu_ctx_factory(cfg.width, cfg.height)
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.render(2, depth)
        ctx.render(spp - 2, depth)                      # continues from the counters the first call left
        frames.append(ctx.read_framebuffer())
        st = ctx.stats()
        counts.append((st.primary_rays, st.shadow_rays, st.bounce_rays))
        for k in knobs:
            monkeypatch.delenv(k)
    assert counts[0][2] > 0
    for other, cnt in zip(frames[1:], counts[1:]):
        assert np.array_equal(frames[0]["rgba"].view(np.uint32), other["rgba"].view(np.uint32))
        assert np.array_equal(frames[0]["prim_id"], other["prim_id"])
        assert cnt == counts[0]
    check_images(frames[1], orc.render(cfg.scene, cfg.camera, cfg.width, cfg.height, spp=spp, max_depth=depth))
    # three ranks' partitions, gathered: the whole frame
    union = np.zeros_like(frames[1]["rgba"])
    monkeypatch.setenv("RT_BOUNCE_QUEUE", "2")
    part = gpu_ctx_factory(cfg.width, cfg.height)
    part.upload_scene(cfg.scene)
    part.set_camera(cfg.camera)
    for rank in range(3):
        part.set_partition(rank, 3, 16)
        part.render(spp, depth)
        fb = part.read_framebuffer()
        union += np.where((fb["prim_id"] != -2)[..., None], fb["rgba"], 0.0).astype(np.float32)
    assert np.array_equal(union.view(np.uint32), frames[1]["rgba"].view(np.uint32))


def test_general_kernel_with_few_persistent_waves(gpu_ctx_factory, monkeypatch):
    """The general kernel's waves are persistent: with the grid cut to three blocks every wave works through
    a dozen chunks -- regenerating paths slot by slot over the shallow demo scene (slots straddle chunks,
    bounce numbers differ within a wave), taking whole chunks over the deeper materials scene -- and the
    frames still equal the oracle's, ray counts included."""
    monkeypatch.setenv("RT_MAX_BLOCKS", "3")
    cases = [(scenes.demo_scene_with_environment(), CameraArgs.for_image(160, 120), 160, 120, 3, 5),
             (scenes.demo_scene(), CameraArgs.for_image(161, 97), 161, 97, 2, 0),
             (scenes.material_test_scene()[0],
              CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0)), 96, 64, 2, 6)]
    for scene, cam, w, h, spp, depth in cases:
        got, ref, st = render_both(gpu_ctx_factory, scene, cam, w, h, spp, depth)
        check_images(got, ref, st)
    monkeypatch.setenv("RT_GENERAL_WAVES", "1")              # the one-wave build over a tree, too
    scene, cam, w, h, spp, depth = cases[2]
    got, ref, st = render_both(gpu_ctx_factory, scene, cam, w, h, spp, depth)
    check_images(got, ref, st)


# ----------------------------------------------------------------------------- mesh textures (N4)
def test_textured_meshes_full_frames(gpu_ctx_factory):
    """Mesh textures with mip LOD (Model.cu:60-88,304-309, Texture/Texture.cu): the mip chain built on the
    device (k_mip_level = d_mipmap) and the software tex2DGrad (DESIGN H13) against the oracle's, whole
    frames with ray counts at depth 0 and with bounces, every BRDF taking the texel as its colour; spheres
    in front of a textured mesh inherit its colour; tree == scan bit for bit; a partition's pixels equal
    the full frame's."""
    scene, _ = scenes.textured_test_scene()
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    for spp, depth in [(2, 0), (3, 6)]:
        got, ref, st = render_both(gpu_ctx_factory, scene, cam, 96, 64, spp, depth)
        check_images(got, ref, st)
        assert st.batch_launches == 0
    import copy
    plain = copy.deepcopy(scene)
    plain.textures, plain.mesh_texture = [], None
    base = orc.render(plain, cam, 96, 64, spp=3, max_depth=6)
    changed = np.abs(base["rgba"] - ref["rgba"])[..., :3].max(axis=2) > 1e-6
    for lo, hi in [(2, 282), (282, 408), (scene.n_tris, scene.n_tris + 1)]:      # blob, ground, a sphere in front
        seen = (got["prim_id"] >= lo) & (got["prim_id"] < hi)
        assert (seen & changed).sum() > 20
    ctx = gpu_ctx_factory(96, 64)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.set_culling(0)
    ctx.render(1, 6)
    ctx.render(2, 6)
    scan = ctx.read_framebuffer()
    assert np.array_equal(scan["rgba"].view(np.uint32), got["rgba"].view(np.uint32))
    part = gpu_ctx_factory(96, 64)
    part.upload_scene(scene)
    part.set_camera(cam)
    part.set_partition(1, 3, 16)
    part.render(3, 6)
    sub = part.read_framebuffer()
    mine = sub["prim_id"] != -2
    assert mine.sum() > 1000
    assert np.array_equal(sub["rgba"][mine].view(np.uint32), got["rgba"][mine].view(np.uint32))


def test_texture_upload_limits(gpu_ctx_factory):
    import ctypes as C
    scene, _ = scenes.textured_test_scene()
    ctx = gpu_ctx_factory(8, 8)
    d = scene.desc()
    d.tex_data = C.POINTER(C.c_float)()
    assert abi.lib().rt_upload_scene(ctx._h, C.byref(d)) == abi.RT_ERR_INVALID
    scene.mesh_texture = np.int32([0, 2, -1])                   # only two images
    with pytest.raises(abi.RtError) as e:
        ctx.upload_scene(scene)
    assert e.value.code == abi.RT_ERR_INVALID
    scene.mesh_texture = np.int32([1, -1, 0])                   # any mesh may bind any image; NULL uv = zeros
    scene.tri_uv = None
    ctx.upload_scene(scene)


# ----------------------------------------------------------------------------- environment light (N3)
def test_reference_demo_scene_with_environment(gpu_ctx_factory):
    """The reference's demo scene whole (kernel.cu:268-313): two spheres, a two-sided triangle light
    and an EnvironmentLight (Light.cu:137-234, Distribution2D of PiecewiseConstant.cu; a synthetic
    image stands in for environment.png): rays that miss pick up getLe, every hit samples the
    environment through the Distribution2D and evaluates the MIS branch.  Whole frame against the
    oracle with ray counts; the scan agrees bit for bit."""
    scene = scenes.demo_scene_with_environment()
    cam = CameraArgs.for_image(160, 120)
    for spp, depth in [(2, 0), (3, 5)]:
        got, ref, st = render_both(gpu_ctx_factory, scene, cam, 160, 120, spp, depth)
        check_images(got, ref, st)
    sky = got["prim_id"] < 0
    assert sky.sum() > 1000 and (got["rgba"][..., :3][sky] > 0).all()      # the environment is visible
    ctx = gpu_ctx_factory(160, 120)
    ctx.upload_scene(scene)
    ctx.set_camera(cam)
    ctx.set_culling(0)
    ctx.render(3, 5)
    scan = ctx.read_framebuffer()
    assert np.array_equal(scan["rgba"].view(np.uint32), got["rgba"].view(np.uint32))


def test_environment_light_limits(gpu_ctx_factory):
    import ctypes as C
    scene = scenes.demo_scene_with_environment()
    d = scene.desc()
    d.n_env_lights = 2                                       # Scene::getEnvironmentLight reads one
    ctx = gpu_ctx_factory(8, 8)
    assert abi.lib().rt_upload_scene(ctx._h, C.byref(d)) == abi.RT_ERR_UNSUPPORTED
    d.n_env_lights = 1
    d.env_image = C.POINTER(C.c_float)()
    assert abi.lib().rt_upload_scene(ctx._h, C.byref(d)) == abi.RT_ERR_INVALID


@pytest.mark.parametrize("size", [(8, 8), (40, 24), (130, 70), (131, 69, 5)])
def test_batch_kernel_equals_path_kernel_on_odd_shapes(gpu_ctx_factory, monkeypatch, size):
    """k_batch against the per-sample path kernel (RT_BATCH_MIN_TOP keeps a context off k_batch) on
    images and partitions that leave units ragged or lanes without a pixel: tiny frames, 1 / 3 / 8
    ranks, sample counts that end a batch early -- every bit of the three buffers, and the RNG state
    (a second call continues identically)."""
    w, h = size[:2]
    if len(size) > 2:                                        # the five-wave build (large launches' choice) on a small one
        monkeypatch.setenv("RT_BATCH_WAVES", str(size[2]))
    cfg = scenes.c3(w, h, f=10, ground=(8, 5))
    monkeypatch.setenv("RT_BATCH_MIN_TOP", "99")
    plain = gpu_ctx_factory(w, h)
    monkeypatch.delenv("RT_BATCH_MIN_TOP")
    fast = gpu_ctx_factory(w, h)
    for ctx in (plain, fast):
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
    for world in (1, 3, 8):
        for rank in sorted({0, world - 1}):
            for ctx in (plain, fast):
                ctx.set_partition(rank, world, 8)
                ctx.reset_stats()
            for spp in (1, 5, 70):
                out = []
                for ctx in (plain, fast):
                    ctx.render(spp, 0)
                    out.append(ctx.read_framebuffer())
                assert np.array_equal(out[0]["rgba"].view(np.uint32), out[1]["rgba"].view(np.uint32)), (world, rank, spp)
                assert np.array_equal(out[0]["prim_id"], out[1]["prim_id"])
                assert np.array_equal(out[0]["t"].view(np.uint32), out[1]["t"].view(np.uint32))
            sp, sf = plain.stats(), fast.stats()
            assert sp.batch_launches == 0 and (sf.batch_launches > 0 or fast.local_pixel_count() == 0)
            assert (sp.primary_rays, sp.shadow_rays) == (sf.primary_rays, sf.shadow_rays)


def test_general_kernel_partition_union(gpu_ctx_factory):
    """The general kernel under the tile partition: three ranks' images of the every-BRDF scene under
    an environment light add up to the single-context image bit for bit (RNG streams are keyed by the
    global pixel index), and scene_hit still answers on a partitioned context."""
    scene, _ = scenes.material_test_scene()
    scene.env_image = scenes.synthetic_environment(32, 16, seed=5)
    cam = CameraArgs.for_image(96, 64, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    whole = gpu_ctx_factory(96, 64)
    whole.upload_scene(scene)
    whole.set_camera(cam)
    whole.render(3, 4)
    ref = whole.read_framebuffer()
    acc = np.zeros_like(ref["rgba"])
    seen = np.zeros(ref["prim_id"].shape, bool)
    part = gpu_ctx_factory(96, 64)
    part.upload_scene(scene)
    part.set_camera(cam)
    for rank in range(3):
        part.set_partition(rank, 3, 8)
        part.render(2, 4)
        part.render(1, 4)
        got = part.read_framebuffer()
        mine = got["prim_id"] != -2
        assert not (seen & mine).any()
        seen |= mine
        acc[mine] = got["rgba"][mine]
        assert np.array_equal(got["prim_id"][mine], ref["prim_id"][mine])
    assert seen.all()
    assert np.array_equal(acc.view(np.uint32), ref["rgba"].view(np.uint32))
