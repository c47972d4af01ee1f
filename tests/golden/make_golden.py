"""Generates tests/golden/*.npz with the CPU oracle (oracle/librt_oracle.so).

The reference holds no golden vectors, test data or images, and its
sources cannot be built in this image ("Oracle"), so these fixtures are outputs of
the ORACLE -- they pin the oracle and the HIP path against regressions and against compiler /
machine differences, not against the reference.  The one value that does come from the
reference's own code is the known answer recorded in SURVEY.md section 8(c); it is checked in
tests/test_oracle_cpu.py::test_survey_known_answer.

Every scene comes from raytracer_amd/scenes.py (seeded); because numpy's float64 sin/cos may
differ in the last bit between CPUs, the exact scene arrays and camera arguments are stored IN
the fixture (`scene_*`, `camera`) and the tests load them from there.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.scene import CameraArgs  # noqa: E402
from tests import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (builder kwargs, width, height, crop, [(spp, depth), ...])
CASES = {
    "c1": (("c1", {}), 160, 120, (56, 36, 104, 84), [(1, 0), (2, 12)]),
    "c2": (("c2", {}), 192, 108, (64, 40, 128, 88), [(2, 0), (2, 12)]),
    "c3": (("c3", dict(lon=24, lat=24, ground=(8, 4))), 128, 72, (40, 16, 88, 64), [(2, 0), (2, 12)]),
    "c4": (("c4", dict(lon=20, lat=16, ground=(6, 1))), 128, 72, (40, 16, 88, 64), [(1, 0)]),
    "c5": (("c5", dict(nx=12, nz=10)), 64, 36, (16, 8, 48, 28), [(1, 0)]),
    "mixed": (("mixed", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (2, 5)]),
    "sun": (("sun", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (2, 5)]),
    # triangle lights: the reference's demo scene without its environment light
    # (kernel.cu:268-313, camera kernel.cu:228-229) and the mixed scene under every light type
    "demo": (("demo", {}), 160, 120, (40, 20, 120, 100), [(2, 0), (4, 6)]),
    "area": (("area", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (2, 5)]),
    # every BRDF of material.cu, one model each
    "materials": (("materials", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (3, 6)]),
    # the reference's demo scene with its environment light (the reference's environment.png) and the materials
    # scene under one: EnvironmentLight + Distribution2D
    "demo_env": (("demo_env", {}), 160, 120, (40, 20, 120, 100), [(2, 0), (3, 6)]),
    "materials_env": (("materials_env", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (2, 6)]),
    # mesh textures with mip LOD: the materials scene with two textured meshes
    "textured": (("textured", {}), 96, 64, (16, 8, 80, 56), [(2, 0), (3, 6)]),
}


SCENE_FIELDS = ["tri_pos", "tri_nrm", "tri_model", "mesh_first", "mesh_count", "model_xf",
                "model_mat_type", "model_color", "quad_coeff", "quad_type", "pl_pos", "pl_illum"]
REFERENCE_ROOT = os.environ.get("RT_REFERENCE_ROOT", "/root/reference")   # read by THIS script only (fixture generation)
OPTIONAL_SCENE_FIELDS = ["dl_dir", "dl_illum",        # absent from fixtures written before them
                         "tl_pos", "tl_illum", "tl_two_side", "tl_mat_type", "tl_color",
                         "model_param", "env_image", "tri_uv", "mesh_texture"]


def load_case(name):
    """-> (SceneData, CameraArgs, width, height, crop, {key: array}) from the fixture."""
    from raytracer_amd.scene import SceneData
    z = np.load(os.path.join(HERE, f"{name}.npz"))
    scene = SceneData()
    scene.n_tri_models = int(z["scene_n_tri_models"])
    for f in SCENE_FIELDS:
        setattr(scene, f, z["scene_" + f])
    for f in OPTIONAL_SCENE_FIELDS:
        if "scene_" + f in z:
            setattr(scene, f, z["scene_" + f])
    k = 0
    while f"scene_texture_{k}" in z:                       # Mesh::initTexture images
        scene.textures.append(z[f"scene_texture_{k}"])
        k += 1
    scene.normalise()
    c = json.loads(str(z["camera"]))
    cam = CameraArgs(pos=tuple(c["pos"]), lookat=tuple(c["lookat"]), fov=c["fov"], near=c["near"],
                     far=c["far"], res=tuple(c["res"]), up=tuple(c["up"]))
    meta = json.loads(str(z["meta"]))
    return scene, cam, meta["width"], meta["height"], tuple(meta["crop"]), z, meta


def build_case(name):
    (kind, kw), w, h, crop, runs = CASES[name]
    if kind in ("mixed", "sun", "area", "materials", "textured"):
        scene, _ = {"mixed": scenes.mixed_test_scene, "sun": scenes.sun_test_scene,
                    "area": scenes.area_test_scene, "materials": scenes.material_test_scene,
                    "textured": scenes.textured_test_scene}[kind]()
        cam = CameraArgs.for_image(w, h, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    elif kind == "materials_env":
        scene, _ = scenes.material_test_scene()
        scene.env_image = scenes.synthetic_environment(32, 16, seed=5)
        cam = CameraArgs.for_image(w, h, near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    elif kind in ("demo", "demo_env"):
        if kind == "demo":
            scene = scenes.demo_scene()
        else:
            # the reference's own environment map, a data file beside its driver (kernel.cu:271-272 loads it with
            # stbi_loadf(.., 4)): decoded here with PIL and converted the way stb does; the fixture carries the
            # float image, so nothing reads the reference tree when the tests run
            from PIL import Image
            png = os.path.join(REFERENCE_ROOT, "src", "RayTracer", "environment.png")
            scene = scenes.demo_scene_with_environment(
                scenes.stb_ldr_to_hdr(np.asarray(Image.open(png).convert("RGBA"))))
        cam = CameraArgs.for_image(w, h)
    else:
        cfg = scenes.CONFIGS[kind](w, h, **kw)
        scene, cam = cfg.scene, cfg.camera
    return scene, cam, w, h, crop, runs


def main():
    import sys
    only = sys.argv[1:]                   # e.g. `python -m tests.golden.make_golden sun`
    for name in CASES:
        if only and name not in only:
            continue
        scene, cam, w, h, crop, runs = build_case(name)
        out = {}
        for spp, depth in runs:
            r = orc.render(scene, cam, w, h, spp=spp, max_depth=depth, crop=crop)
            k = f"s{spp}_d{depth}"
            out[k + "_rgba"] = r["rgba"]
            out[k + "_prim"] = r["prim_id"]
            out[k + "_t"] = r["t"]
            out[k + "_rays"] = np.array([r["primary"], r["shadow"], r["bounce"]], np.int64)
        for f in SCENE_FIELDS + OPTIONAL_SCENE_FIELDS:
            if getattr(scene, f) is not None:
                out["scene_" + f] = getattr(scene, f)
        for ti, img in enumerate(scene.textures):
            out[f"scene_texture_{ti}"] = img
        out["scene_n_tri_models"] = np.array(scene.n_tri_models)
        out["camera"] = np.array(json.dumps(dict(pos=cam.pos, lookat=cam.lookat, fov=cam.fov,
                                                 near=cam.near, far=cam.far, res=cam.res,
                                                 up=cam.up)))
        meta = dict(case=name, builder=CASES[name][0][0], kwargs=CASES[name][0][1], width=w,
                    height=h, crop=crop, runs=runs, n_tris=scene.n_tris, n_quads=scene.n_quads,
                    n_lights=scene.n_lights)
        print(name, "primary/shadow/bounce rays of last run:", out[k + "_rays"])
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, {k: v.shape for k, v in out.items() if k != "meta"})
    if only:
        return
    # RNG fixture: first 16 uniform4 draws of 4 pixel streams (H7 convention)
    pix = [0, 1, 1919, 2073599]
    u = np.array([[orc.pixel_uniform4(p, d) for d in range(16)] for p in pix], np.float32)
    np.savez_compressed(os.path.join(HERE, "philox_uniform4.npz"), pixels=np.array(pix), u=u)


if __name__ == "__main__":
    main()
