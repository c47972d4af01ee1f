"""rt_scene_hit_noalloc: Scene::hit for callers that cannot allocate device memory (its scratch
is a static device array; include/rt_abi.h).  Through the C ABI, against rt_scene_hit on the
same rays, bit for bit, on two calls in a row -- the second with a smaller input, so whatever
the first left in the shared scratch array must not matter -- and over a shallow and a deep
tree; sizes beyond the maximum are refused."""
import numpy as np
import pytest
import torch

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu


def _rays(rng, n, scene):
    org = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    org[:, 2] += 40.0
    tgt = rng.uniform([-350, -150, -900], [350, 300, -150], (n, 3)).astype(np.float32)
    return org, (tgt - org).astype(np.float32)


def _call(ctx, org, d):
    n = org.shape[0]
    dev = torch.device("cuda", 0)
    d_org = torch.from_numpy(org).to(dev).contiguous()
    d_dir = torch.from_numpy(d).to(dev).contiguous()
    d_t = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    d_id = torch.full((n,), -7, dtype=torch.int32, device=dev)
    d_hit = torch.full((n,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.scene_hit_noalloc(n, d_org.data_ptr(), d_dir.data_ptr(), d_t.data_ptr(), d_id.data_ptr(),
                          d_hit.data_ptr())
    return d_t.cpu().numpy(), d_id.cpu().numpy(), d_hit.cpu().numpy()


@pytest.mark.parametrize("kind", ["mixed", "deep"])
def test_scene_hit_noalloc_equals_scene_hit(gpu_ctx_factory, kind):
    if kind == "mixed":
        scene, rng = scenes.mixed_test_scene()          # triangles, meshes, spheres: shallow tree
    else:
        scene = scenes.c3(64, 64, f=12, ground=(10, 6)).scene   # 3 001 triangles: deep tree, dispensers in use
        rng = np.random.default_rng(3)
    ctx = gpu_ctx_factory(16, 16)
    ctx.upload_scene(scene)
    for n in (50_000, 1_234):                           # the second, smaller call reuses the same scratch
        org, d = _rays(rng, n, scene)
        if kind == "deep":
            org[:] = (0.0, 0.0, 0.0)
            d = (rng.uniform([-300, -200, -900], [300, 200, -300], (n, 3))).astype(np.float32)
        t0, id0, hit0 = ctx.scene_hit(org, d)
        t1, id1, hit1 = _call(ctx, org, d)
        assert np.array_equal(id1, id0)
        assert np.array_equal(t1.view(np.uint32), t0.view(np.uint32))
        assert np.array_equal(hit1, hit0)
        assert 0.05 < hit0.mean() < 1.0


def test_scene_hit_noalloc_limits(gpu_ctx_factory):
    scene, _ = scenes.mixed_test_scene()
    ctx = gpu_ctx_factory(16, 16)
    ctx.upload_scene(scene)
    buf = torch.zeros(16, dtype=torch.float32, device="cuda")
    with pytest.raises(abi.RtError) as e:
        ctx.scene_hit_noalloc(abi.SCENE_HIT_NOALLOC_MAX_RAYS + 1, buf.data_ptr(), buf.data_ptr(),
                              buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
    assert e.value.code == abi.RT_ERR_INVALID and "RT_SCENE_HIT_NOALLOC_MAX_RAYS" in str(e.value)
    ctx.scene_hit_noalloc(0, 0, 0, 0, 0, 0)             # nothing to do: fine
    with pytest.raises(abi.RtError):
        ctx.scene_hit_noalloc(4, 0, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
