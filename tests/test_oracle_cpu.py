"""CPU tests of the oracle (no GPU): Philox known answers, the one known answer that comes
from the reference's own code, the fixed sinf/cosf, and the committed
golden fixtures."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.scene import CameraArgs, SceneBuilder, translation

from . import orc
from .golden.make_golden import CASES, load_case


def test_philox_random123_known_answers():
    """Random123 kat_vectors for philox4x32-10 (the generator behind curandStatePhilox4_32_10_t)."""
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                      [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_mapping_and_stream_layout():
    """curand_uniform4: x * 2^-32 + 2^-33 in (0, 1]; key = idx + 332, subsequence = idx."""
    z = np.load(__file__.replace("test_oracle_cpu.py", "golden/philox_uniform4.npz"))
    for pi, p in enumerate(z["pixels"]):
        for d in range(16):
            u = orc.pixel_uniform4(int(p), d)
            assert np.array_equal(u, z["u"][pi, d])
            raw = orc.philox([d, 0, int(p), 0], [int(p) + 332, 0])
            exp = (np.array(raw, np.uint32).astype(np.float32) * np.float32(2.3283064e-10)
                   + np.float32(2.3283064e-10) / np.float32(2.0))
            assert np.array_equal(u, exp.astype(np.float32))
            assert (u > 0).all() and (u <= 1).all()


def test_survey_known_answer():
    """The only output of the REFERENCE ITSELF available: SURVEY.md section 8(c) recorded, from
    the reference's per-pixel code built as host C++, sphere + point light at 64x48, centre
    pixel: t = 5.005611, radiance (0.244406, 0.010100, 0.011608).  The sphere/light/camera are
    kernel.cu:285-291,301,228-229; the ray is the pixel centre (offset 0.5, 0.5); the
    Lambertian colour that reproduces the recorded channel ratios is (0.7, 0.2, 0.1)."""
    b = SceneBuilder()
    b.add_sphere(0.33, translation(0, 0, -8), color=(0.7, 0.2, 0.1))
    b.add_point_light((-8, 0, 0), (233.7, 33.8, 77.7))
    s = b.build()
    cam = CameraArgs.for_image(64, 48)
    o, d = orc.camera_ray(cam, [[0.5, 0.5]])
    for depth in (0, 12):
        rgb, t, prim = orc.trace_ray(s, o[0], d[0], depth, pixel_index=24 * 64 + 32)
        assert prim == 0
        assert abs(float(t) - 5.005611) < 5e-7
        assert np.allclose(rgb, [0.244406, 0.010100, 0.011608], rtol=0, atol=5e-7)


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7fffffff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fffffff), bi)
    return np.abs(ai - bi)


def test_fixed_sincos_within_1ulp_of_libm():
    """material.cu:46 calls the platform's sinf/cosf; the build fixes them (double Taylor,
    rounded).  They must stay within 1 ulp of a correctly rounded result on sampleHemi's range."""
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-np.pi / 4, 3 * np.pi / 4, 20000),
                        [-np.pi / 4, 0.0, np.pi / 4, np.pi / 2, 3 * np.pi / 4, 1e-8, -1e-8]])
    x = x.astype(np.float32)
    s = np.array([orc.lib().orc_sinf(float(v)) for v in x], np.float32)
    c = np.array([orc.lib().orc_cosf(float(v)) for v in x], np.float32)
    rs = np.sin(x.astype(np.float64)).astype(np.float32)
    rc = np.cos(x.astype(np.float64)).astype(np.float32)
    big = np.abs(rc) > 1e-6          # relative ulp is meaningless next to cos's zero crossing
    assert _ulp_diff(s, rs).max() <= 1
    assert _ulp_diff(c[big], rc[big]).max() <= 1
    assert np.abs(c[~big] - rc[~big]).max() < 1e-12 if (~big).any() else True


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    """The oracle, rebuilt on this machine, reproduces the committed fixtures bit for bit."""
    scene, cam, w, h, crop, z, meta = load_case(name)
    for spp, depth in meta["runs"]:
        r = orc.render(scene, cam, w, h, spp=spp, max_depth=depth, crop=crop)
        k = f"s{spp}_d{depth}"
        assert np.array_equal(r["prim_id"], z[k + "_prim"])
        assert np.array_equal(r["t"].view(np.uint32), z[k + "_t"].view(np.uint32))
        assert np.array_equal(r["rgba"].view(np.uint32), z[k + "_rgba"].view(np.uint32))
        assert [r["primary"], r["shadow"], r["bounce"]] == list(z[k + "_rays"])


def test_oracle_crop_and_threads_invariant():
    """A crop equals the same window of the full render; thread count does not matter."""
    cfg = scenes.c2(64, 36)
    full = orc.render(cfg.scene, cfg.camera, 64, 36, spp=2, max_depth=3, threads=1)
    crop = orc.render(cfg.scene, cfg.camera, 64, 36, spp=2, max_depth=3, crop=(8, 4, 40, 30))
    assert np.array_equal(full["rgba"][4:30, 8:40], crop["rgba"])
    assert np.array_equal(full["prim_id"][4:30, 8:40], crop["prim_id"])


def test_oracle_exact_tie_keeps_first():
    """H8: two coincident triangles -- strict `t < rec.t` keeps the first in traversal order."""
    tri = [(-5, -5, -20), (5, -5, -20), (0, 6, -20)]
    b = SceneBuilder()
    b.add_triangle(tri, [(0, 0, 1)] * 3)
    b.add_triangle(tri, [(0, 0, 1)] * 3)
    s = b.build()
    t, pid, hit = orc.scene_hit(s, [[0, 0, 0]], [[0, 0, -1]])
    assert hit[0] == 1 and pid[0] == 0 and t[0] == np.float32(20.0)


def test_oracle_postprocess_reference_values():
    """ACES curve (PostProcess.cuh:29-37) against a float64 evaluation; median of a 3x3 ramp."""
    x = np.linspace(0, 4, 64, dtype=np.float32)
    img = np.stack([x, x * 0.5, x * 0.25, np.ones_like(x)], -1).reshape(8, 8, 4)
    out = orc.tonemap(img)
    xd = img[..., :3].astype(np.float64)
    ref = np.clip((xd * (2.51 * xd + 0.03)) / (xd * (2.43 * xd + 0.59) + 0.14), 0, 1)
    assert np.allclose(out[..., :3], ref, atol=2e-7)
    assert (out[..., 3] == 1).all()
    med = orc.median3x3(img)
    # interior pixel: 9 samples sorted by luma descending, element 4 = the middle one
    lum = (0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2])
    w = sorted(lum[2:5, 3:6].reshape(-1), reverse=True)[4]
    assert np.isclose(0.299 * med[3, 4, 0] + 0.587 * med[3, 4, 1] + 0.114 * med[3, 4, 2], w)


def test_directional_light_closed_form():
    """DirectionalLight::lightIllumi (Light/Light.cu:21-33) through the delta-light branch of
    evaluateDirectLight: an unoccluded Lambertian plate facing the light receives
    illum * |cos| * albedo / pi (pdf_light = 1, no distance falloff); an occluder between plate
    and light blocks it (Scene.cu:227); a light from behind the plate gives black
    (Lambertian::f needs wi.y > 0)."""
    def plate_scene(direction, blocker=False):
        b = SceneBuilder()
        b.add_triangle([(-500, -500, -100), (500, -500, -100), (0, 600, -100)], [(0, 0, 1)] * 3,
                       None, (0.7, 0.5, 0.3))
        if blocker:
            b.add_triangle([(-50, -50, -50), (50, -50, -50), (0, 60, -50)], [(0, 0, 1)] * 3,
                           None, (0.7, 0.7, 0.7))
        b.add_directional_light(direction, (2.0, 1.0, 0.5))
        return b.build()
    org, d = np.float32([3.0, 2.0, 0.0]), np.float32([0.0, 0.0, -1.0])
    rgb, t, prim = orc.trace_ray(plate_scene((0.0, 0.0, -3.0)), org, d, 0)
    assert prim == 0 and abs(t - 100.0) < 1e-3
    want = np.float32([2.0, 1.0, 0.5]) * np.float32([0.7, 0.5, 0.3]) * np.float32(0.318309892)
    assert np.allclose(rgb, want, rtol=2e-6)
    # oblique: cos = 0.8
    rgb, _, _ = orc.trace_ray(plate_scene((0.6, 0.0, -0.8)), org, d, 0)
    assert np.allclose(rgb, want * np.float32(0.8), rtol=1e-5)
    # from behind the plate: Lambertian f = 0
    rgb, _, _ = orc.trace_ray(plate_scene((0.0, 0.0, 1.0)), org, d, 0)
    assert (rgb == 0).all()
    # occluded (Scene.cu:227): a ray that passes beside the blocker and lands below its interior
    rgb2, _, prim2 = orc.trace_ray(plate_scene((0.0, 0.0, -1.0), blocker=True),
                                   np.float32([0.0, 200.0, 0.0]), np.float32([0.0, -1.98, -1.0]), 0)
    # that ray lands at (0, 2, -100), straight below the blocker's interior: blocked
    assert prim2 == 0 and (rgb2 == 0).all()


def test_squared_length_thresholds():
    """The HIP path decides `length(x) < 0.001f` (Scene.cu:3-5) and `length(f) < 0.0001f`
    (PathTracer.cu:77) on the squared length (rt_device.h kLenSqBelow1e3 / 1e4): with a correctly
    rounded square root, sqrtf(y) < c exactly for the floats y below the thresholds."""
    for c, thr in ((np.float32(0.001), float.fromhex("0x1.0c6f7ap-20")), (np.float32(0.0001), float.fromhex("0x1.5798ecp-27"))):
        t = np.float32(thr)
        assert float(t) == thr                                        # representable
        below = np.nextafter(t, np.float32(0.0))
        assert np.sqrt(below, dtype=np.float32) < c and not (np.sqrt(t, dtype=np.float32) < c)
        y = np.linspace(0.0, 4.0 * thr, 200001).astype(np.float32)
        assert np.array_equal(np.sqrt(y, dtype=np.float32) < c, y < t)


def test_triangle_light_emission_and_sidedness():
    """TriangleLight (Light/Light.cu:35-105, PathTracer.cu:33-52): a camera ray that lands on the
    light picks up L(-dir) -- illum from either side of a two-sided light, from the side its
    normal = normalize(cross(p1 - p0, p2 - p0)) points to of a one-sided one -- plus whatever
    sampleAllLight finds at that surface point; the hit reports id n_tris + n_quads + k."""
    from raytracer_amd.scene import SceneBuilder
    cam = CameraArgs.for_image(32, 24)
    tri = [(-30.0, -30.0, -8.0), (30.0, -30.0, -8.0), (0.0, 40.0, -8.0)]   # normal +z: faces the camera
    for two_side, verts, lit in [(False, tri, True), (False, tri[::-1], False), (True, tri[::-1], True)]:
        b = SceneBuilder()
        b.add_triangle_light(verts, (5.0, 6.0, 7.0), two_side=two_side, color=(0.5, 0.5, 0.5))
        r = orc.render(b.build(), cam, 32, 24, spp=1, max_depth=0, crop=(14, 10, 18, 14))
        assert np.all(r["prim_id"] == 0)                      # no models: the light is id 0
        want = (5.0, 6.0, 7.0) if lit else (0.0, 0.0, 0.0)
        # a lone flat light does not light itself: the sampled point lies in the surface's own plane
        np.testing.assert_allclose(r["rgba"][..., :3], np.broadcast_to(np.float32(want), r["rgba"][..., :3].shape),
                                   rtol=1e-6)


def test_triangle_light_blocks_and_is_sampled():
    """A floor under a one-sided light facing down is lit through the MIS branch (Scene.cu:242-284);
    turned upwards the same light leaves it black, and a second light's triangle in between casts a
    shadow like any model (Scene.cu:227: lightidx != idx)."""
    from raytracer_amd.scene import SceneBuilder, identity
    cam = CameraArgs.for_image(32, 24, pos=(0.0, 2.0, 0.0), lookat=(0.0, -1.0, -6.0))
    floor = ([(-12.0, -3.0, 2.0), (12.0, -3.0, 2.0), (0.0, -3.0, -30.0)], [(0.0, 1.0, 0.0)] * 3)
    down = [(-2.0, 6.0, -8.0), (0.0, 6.0, -11.0), (2.0, 6.0, -8.0)]         # cross(p1-p0, p2-p0) = -y
    def mean_floor(lights):
        b = SceneBuilder()
        b.add_triangle(floor[0], floor[1], identity(), (0.7, 0.7, 0.7))
        for verts, two in lights:
            b.add_triangle_light(verts, (40.0, 40.0, 40.0), two_side=two, color=(0.5, 0.5, 0.5))
        r = orc.render(b.build(), cam, 32, 24, spp=8, max_depth=0, crop=(8, 0, 24, 8))                 # (row 0 is the bottom of the image)
        floor_px = r["prim_id"] == 0
        assert floor_px.sum() > 100
        return float(r["rgba"][..., 0][floor_px].mean())
    lit = mean_floor([(down, False)])
    assert lit > 0.05
    assert mean_floor([(down[::-1], False)]) == 0.0                          # faces up: L = BLACK
    blocker = [(-12.0, 4.0, 2.0), (12.0, 4.0, 2.0), (0.0, 4.0, -30.0)]       # one-sided, faces up, below the light
    assert mean_floor([(down, False), (blocker, False)]) < 0.02 * lit


def test_rough2alpha_and_specular_emission():
    """rough2alpha (material.cuh:7-13): log(1) = 0 leaves the constant term, and the clamp at 1e-3.
    A Fresnel (specular) pane in front of a two-sided triangle light: its surface gathers no
    direct light (PathTracer.cu:67-74), but the ray that leaves it picks up the light's emission
    (PathTracer.cu:33-52) even at max_depth 0, where a Lambertian pane's last ray reads nothing."""
    import ctypes as C
    from raytracer_amd.scene import MAT_FRESNEL, SceneBuilder
    f = orc.lib().orc_rough2alpha
    f.argtypes, f.restype = [C.c_float], C.c_float
    assert f(1.0) == np.float32(1.62142)
    assert f(1e-9) == f(1e-3)
    cam = CameraArgs.for_image(32, 24)
    def centre(mat):
        b = SceneBuilder()
        b.add_triangle([(-30.0, -30.0, -8.0), (30.0, -30.0, -8.0), (0.0, 40.0, -8.0)], [(0.0, 0.0, 1.0)] * 3,
                       color=(1.0, 1.0, 1.0))
        b.add_triangle_light([(-60.0, -60.0, -20.0), (0.0, 80.0, -20.0), (60.0, -60.0, -20.0)], (3.0, 3.0, 3.0),
                             two_side=True, color=(0.0, 0.0, 0.0))
        s = b.build()
        s.model_mat_type[:] = mat
        s.model_param = np.float32([[0.0, 1.5]])
        r = orc.render(s, cam, 32, 24, spp=16, max_depth=0, crop=(15, 11, 17, 13))
        assert np.all(r["prim_id"] == 0)
        return r["rgba"][..., :3].mean()
    glass, matte = centre(MAT_FRESNEL), centre(1)
    # (Fresnel::sample_f feeds |wo.y|, not 1 - |wo.y|, into Schlick's polynomial -- material.cuh:127-128 --
    # so head-on rays almost always reflect; the few that refract carry the light's emission)
    assert glass > 0.05
    assert matte == 0.0           # the Lambertian front face cannot see the light behind it


def test_environment_light_emission_and_sampling():
    """EnvironmentLight (Light.cu:137-234): with nothing to hit every camera ray returns getLe -- a
    constant image comes back as that constant whatever the direction (the filter's weights sum to
    one), a two-band image puts each band where L's (phi, theta) -> (u, v) mapping says: v = theta /
    pi, theta measured from +y, so the top rows of the image are straight up.  A Lambertian floor
    under a bright top half is lit through the Distribution2D + MIS branch (Scene.cu:242-284)."""
    from raytracer_amd.scene import SceneBuilder, identity
    cam = CameraArgs.for_image(32, 24)
    img = np.zeros((8, 16, 4), np.float32)
    img[..., :3] = (0.25, 0.5, 0.75)
    img[..., 3] = 1.0
    s = SceneBuilder().build()
    s.env_image = img
    r = orc.render(s, cam, 32, 24, spp=2, max_depth=0)
    assert np.all(r["prim_id"] == -1)
    np.testing.assert_allclose(r["rgba"][..., :3], np.broadcast_to(np.float32((0.25, 0.5, 0.75)), (24, 32, 3)), rtol=1e-6)
    img2 = img.copy()
    img2[:4, :, :3] = (9.0, 9.0, 9.0)                        # theta < pi/2: the upper hemisphere
    img2[4:, :, :3] = (0.0, 0.0, 0.0)
    s.env_image = img2
    r = orc.render(s, cam, 32, 24, spp=1, max_depth=0)
    up, down = r["rgba"][20:, :, 0].mean(), r["rgba"][:4, :, 0].mean()      # row 0 is the bottom of the image
    assert up > 8.0 and down < 1.0
    b = SceneBuilder()
    b.add_triangle([(-12.0, -3.0, 2.0), (12.0, -3.0, 2.0), (0.0, -3.0, -30.0)], [(0.0, 1.0, 0.0)] * 3, identity(),
                   (0.7, 0.7, 0.7))
    s = b.build()
    cam2 = CameraArgs.for_image(32, 24, pos=(0.0, 2.0, 0.0), lookat=(0.0, -1.0, -6.0))
    def floor_mean(image):
        s.env_image = image
        rr = orc.render(s, cam2, 32, 24, spp=16, max_depth=0, crop=(8, 0, 24, 8))
        assert np.all(rr["prim_id"] == 0)
        return float(rr["rgba"][..., 0].mean())
    lit, unlit = floor_mean(img2), floor_mean(np.ascontiguousarray(img2[::-1]))      # bright half above / below the floor
    # (not black below: Distribution::sample returns offset + fraction, PiecewiseConstant.cu:39, so the sampled
    # theta = sample.y * pi wraps around the sphere many times, Light.cu:140)
    assert lit > 0.5 and lit > 3.0 * unlit


# ----------------------------------------------------------------------------- mesh textures (N4)
def test_texture_mip_chain_and_filter():
    """initImages / generateMipMaps (Texture.cu:2-13,48-186) and the software tex2DGrad (DESIGN H13):
    level count and sizes, a constant image stays constant down the chain, averaged texels above 1 are
    clamped (Texture.cu:37), a 2x2 box average where the taps land on texel centres' midpoints, edge
    clamping, and the level the gradients select."""
    assert [orc.texture_levels(w, h) for w, h in [(1, 1), (2, 2), (5, 7), (64, 32), (1024, 3)]] == [1, 2, 3, 7, 11]
    img = np.full((7, 5, 4), 0.25, np.float32)
    for lvl, (lh, lw) in enumerate([(7, 5), (3, 2), (1, 1)]):
        m = orc.texture_mip(img, lvl)
        assert m.shape == (lh, lw, 4) and np.all(m == np.float32(0.25))
    hot = np.full((4, 4, 4), 3.0, np.float32)
    assert np.all(orc.texture_mip(hot, 0) == 3.0) and np.all(orc.texture_mip(hot, 1) == 1.0)
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 0.9, (8, 8, 4)).astype(np.float32)
    m1 = orc.texture_mip(img, 1)
    # tap x of the 4x4 level reads the 8x8 one at texel coordinate 2x - 0.5: the mean of texels 2x-1 and 2x
    # (clamped at the edge); taps x and x + 1 together cover texels 2x-1 .. 2x+2 with equal weights
    pad = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="edge")
    want = np.zeros_like(m1)
    for y in range(4):
        for x in range(4):
            want[y, x] = pad[2 * y:2 * y + 4, 2 * x:2 * x + 4].astype(np.float64).mean(axis=(0, 1))
    assert np.allclose(m1, want, rtol=0, atol=2e-7)
    # texel centres return the texel; coordinates outside [0, 1] clamp to the edge texel
    assert np.array_equal(orc.tex2d_grad(img, (3 + 0.5) / 8, (5 + 0.5) / 8), img[5, 3, :3])
    assert np.array_equal(orc.tex2d_grad(img, -0.7, 1.9), img[7, 0, :3])
    assert np.array_equal(orc.tex2d_grad(img, float("nan"), 0.5 / 8), img[0, 0, :3])
    # gradients of one texel per pixel or less: level 0; two texels: level 1; huge or infinite: the 1x1 level
    c = (0.4, 0.6)
    lvl = [orc.tex2d_grad(orc.texture_mip(img, l), *c) for l in range(4)]
    assert np.array_equal(orc.tex2d_grad(img, *c, (1 / 8, 0), (0, 1 / 8)), lvl[0])
    assert np.array_equal(orc.tex2d_grad(img, *c, (2 / 8, 0), (0, 0.1 / 8)), lvl[1])
    assert np.array_equal(orc.tex2d_grad(img, *c, (0, 0), (float("inf"), 0)), lvl[3])
    assert np.array_equal(orc.tex2d_grad(img, *c, (float("nan"), 0), (0, 0)), lvl[0])
    mid = orc.tex2d_grad(img, *c, (2 ** 0.5 / 8, 0), (0, 0))                  # lambda = 0.5
    assert np.allclose(mid, 0.5 * lvl[0] + 0.5 * lvl[1], atol=1e-7)


def test_mesh_texture_colours_the_record():
    """Mesh::hit's lookup (Model.cu:301-310): a textured mesh's hit is shaded with the texel, not the BRDF's
    colour; a sphere in front of it inherits that colour (nothing resets rec.color); an untextured mesh in
    front resets it (H5)."""
    from raytracer_amd.scene import SceneBuilder
    tex = np.zeros((4, 4, 4), np.float32)
    tex[..., 0], tex[..., 3] = 1.0, 1.0                                       # pure red everywhere
    quad_p = np.float32([[(-50, -50, -100), (50, -50, -100), (50, 50, -100)], [(-50, -50, -100), (50, 50, -100), (-50, 50, -100)]])
    quad_n = np.float32([[(0, 0, 1)] * 3] * 2)
    quad_uv = np.float32([[(0, 0), (1, 0), (1, 1)], [(0, 0), (1, 1), (0, 1)]])

    def scene(sphere=False, cover=False, textured=True):
        b = SceneBuilder()
        b.add_mesh(quad_p, quad_n, color=(0.0, 1.0, 0.0), uv=quad_uv, texture=tex if textured else None)
        if cover:
            b.add_mesh(quad_p * np.float32([0.2, 0.2, 0.5]), quad_n, color=(0.0, 0.0, 1.0))
        if sphere:
            b.add_sphere(1.0 / 5.0, scenes_translation(0.0, 0.0, -40.0), color=(0.0, 0.0, 1.0))
        b.add_point_light((0.0, 0.0, 0.0), (1.0e4, 1.0e4, 1.0e4))
        return b.build()

    from raytracer_amd.scene import translation as scenes_translation
    cam = CameraArgs.for_image(128, 128)
    centre = lambda s: orc.render(s, cam, 128, 128, spp=1, max_depth=0, crop=(64, 64, 65, 65))["rgba"][0, 0, :3]
    plain, red = centre(scene(textured=False)), centre(scene())
    assert plain[1] > 0 and plain[0] == 0 and plain[2] == 0                   # the BRDF's green
    assert red[0] > 0 and red[1] == 0 and red[2] == 0                         # the texture's red
    ball = centre(scene(sphere=True))
    assert ball[0] > 0 and ball[1] == 0 and ball[2] == 0                      # blue sphere shaded red
    assert centre(scene(sphere=True, textured=False))[2] > 0
    cover = centre(scene(cover=True))
    assert cover[2] > 0 and cover[0] == 0                                     # untextured mesh in front: its own blue
