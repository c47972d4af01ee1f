// rt_bridge_test.cpp -- the reference-side bridge of INTEGRATION.md section 1 as a program that
// compiles with a plain C++ compiler against include/rt_abi.h and links librt_amd.so: what a
// maintainer of UI233/RayTracer adds in place of initCUDA / renderScene / the upload step of
// test_for_initialize_scene (kernel.cu:115-123, 193-256, 268-313).  The host objects below
// carry exactly the fields the reference's Triangle / Mesh / Quadratic / PointLight keep on
// the host before they are cudaMemcpy'd.
//
//   rt_bridge_test <mesh.obj> <out.pfm> <width> <height> <spp> <max_depth>
//
// loads the OBJ with the library's Mesh::readFile counterpart, uploads it as ONE mesh model with
// the Cornell configuration's two point lights and camera (raytracer_amd/scenes.py c2), renders
// and writes the frame as PFM.  tests/test_bridge_gpu.py runs it and compares the image with
// tests/golden/c2.npz.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rt_abi.h"

struct HostModel {            // Model: transformation + material (Model.cuh:42-60)
    float transformation[16]; // row-major mat4 (Model.cuh:57)
    int material_type;        // material::MATERIAL_TYPE (Object.cuh:4-22)
    float lambert_color[3];   // Lambertian::color (material.cuh:39)
};
struct HostTriangle {         // Triangle (Model.cuh:62-78)
    float pos[9], normal[9];
    int model;
};
struct HostQuadric { float coefficient[3]; int type; };       // Quadratic (Model.cuh:100-121)
struct HostPointLight { float pos[3], illum[3]; };            // PointLight (Light.cuh:45-50)

struct RtBridge {
    rt_context *ctx = nullptr;

    // replaces initCUDA()'s surface / RNG / counter set-up           (kernel.cu:193-256)
    bool init(int device, int width, int height) { return rt_create(&ctx, device, width, height) == RT_OK; }

    // replaces Scene::initializeScene(light_size, model_size, pointl, dirl, tril, envl,
    //                                 tri, mesh, quad, mat_type, mat)   (Scene/Scene.cu:126-202)
    // Host arrays are copied; the caller keeps ownership, as with the reference.
    bool upload(int n_standalone_triangles,
                const std::vector<HostTriangle> &tris,   // standalone Triangle models, then every
                                                         // Mesh's meshTable in order (Scene.cu:13-30)
                const std::vector<int> &mesh_first, const std::vector<int> &mesh_count,
                const std::vector<HostModel> &models,    // triangles, meshes, quadrics: the order
                                                         // initializeScene walks mat_type[] (Scene.cu:143-149)
                const std::vector<HostQuadric> &quads, const std::vector<HostPointLight> &lights,
                // Mesh::initTexture(data, w, h) of the first mesh (Model.cu:315-319) and its triangles' vText
                const float *tri_uv = nullptr, const float *tex_data = nullptr, int tex_w = 0, int tex_h = 0)
    {
        std::vector<float> pos, nrm, xf, col, qc, lp, li;
        std::vector<int32_t> tmodel, mtype, qtype, mfirst(mesh_first.begin(), mesh_first.end()),
            mcount(mesh_count.begin(), mesh_count.end());
        for (auto &t : tris) { pos.insert(pos.end(), t.pos, t.pos + 9);
                               nrm.insert(nrm.end(), t.normal, t.normal + 9);
                               tmodel.push_back(t.model); }
        for (auto &m : models) { xf.insert(xf.end(), m.transformation, m.transformation + 16);
                                 mtype.push_back(m.material_type);
                                 col.insert(col.end(), m.lambert_color, m.lambert_color + 3); }
        for (auto &q : quads) { qc.insert(qc.end(), q.coefficient, q.coefficient + 3); qtype.push_back(q.type); }
        for (auto &l : lights) { lp.insert(lp.end(), l.pos, l.pos + 3); li.insert(li.end(), l.illum, l.illum + 3); }
        rt_scene_desc d{};
        d.n_tri_models = n_standalone_triangles;  d.n_meshes = (int)mfirst.size();
        d.n_quads = (int)quads.size();            d.n_tris = (int)tris.size();
        d.tri_pos = pos.data(); d.tri_nrm = nrm.data(); d.tri_model = tmodel.data();
        d.mesh_first = mfirst.data(); d.mesh_count = mcount.data();
        d.model_xf = xf.data(); d.model_mat_type = mtype.data(); d.model_color = col.data();
        d.quad_coeff = qc.data(); d.quad_type = qtype.data();
        d.n_point_lights = (int)lights.size(); d.pl_pos = lp.data(); d.pl_illum = li.data();
        std::vector<int32_t> mesh_tex(mfirst.size(), -1);
        const int32_t tex_size[2] = {tex_w, tex_h};
        if (tex_data && !mesh_tex.empty()) {
            mesh_tex[0] = 0;
            d.tri_uv = tri_uv; d.n_textures = 1; d.mesh_texture = mesh_tex.data();
            d.tex_size = tex_size; d.tex_data = tex_data;
        }
        // DirectionalLight objects (Light.cuh:57) go the same way: d.n_dir_lights, d.dl_dir (the
        // constructor's `direction` argument; the library normalises it as Light.cu:21 does), d.dl_illum
        return rt_upload_scene(ctx, &d) == RT_OK;
    }

    // replaces Camera cam(...); cudaMemcpyToSymbol(globalCam, &cam, ...)   (kernel.cu:228-229,239)
    bool camera(const float pos[3], const float lookat[3], float fov, float near_, float far_,
                int res_x, int res_y, const float up[3])
    {
        rt_camera_desc c{{pos[0], pos[1], pos[2]}, {lookat[0], lookat[1], lookat[2]}, fov, near_, far_,
                         res_x, res_y, {up[0], up[1], up[2]}};
        return rt_set_camera(ctx, &c) == RT_OK;
    }

    // replaces cudaGraphLaunch + cudaStreamSynchronize in renderScene()     (kernel.cu:122-123)
    bool sample(int max_depth = 12) { return rt_render(ctx, 1, max_depth) == RT_OK; }   // MAX_DEPTH (PathTracer.cu:3-5)

    // the reference shows the image through GL interop (kernel.cu:203-213); with this library the
    // frame comes back to the host and can be uploaded with glTexSubImage2D or written out
    bool frame(float *rgba) { return rt_read_framebuffer(ctx, rgba, nullptr, nullptr) == RT_OK; }
    // render -> HDR -> filter of the reference's graph                        (kernel.cu:249-251)
    bool display_frame(float *rgba) { return rt_postprocess(ctx, nullptr, rgba) == RT_OK; }

    ~RtBridge() { rt_destroy(ctx); }
};

static int die(const RtBridge &b, const char *what)
{
    std::fprintf(stderr, "rt_bridge_test: %s failed: %s\n", what, rt_last_error(b.ctx));
    return 1;
}

int main(int argc, char **argv)
{
    if (argc != 7 && argc != 9) {
        std::fprintf(stderr, "usage: %s mesh.obj out.pfm width height spp max_depth [tex_w tex_h]\n", argv[0]);
        return 2;
    }
    const int tex_w = argc == 9 ? std::atoi(argv[7]) : 0, tex_h = argc == 9 ? std::atoi(argv[8]) : 0;
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), spp = std::atoi(argv[5]),
              depth = std::atoi(argv[6]);

    // Mesh::readFile (Model.cu:139-284): the OBJ stays a host-side load
    float *pos = nullptr, *nrm = nullptr, *uv = nullptr;
    int32_t n = 0;
    if (rt_obj_load(argv[1], &pos, &nrm, &uv, &n) != RT_OK) {
        std::fprintf(stderr, "rt_bridge_test: cannot load %s\n", argv[1]);
        return 1;
    }
    std::vector<HostTriangle> tris((size_t)n);
    for (int i = 0; i < n; i++) {
        std::memcpy(tris[i].pos, pos + 9 * (size_t)i, sizeof(tris[i].pos));
        std::memcpy(tris[i].normal, nrm + 9 * (size_t)i, sizeof(tris[i].normal));
        tris[i].model = 0;                                   // the one Mesh model
    }
    // with [tex_w tex_h]: the mesh binds a procedural image (what stbi_loadf would hand to initTexture)
    // and its triangles carry the OBJ's vt coordinates
    std::vector<float> tri_uv(uv, uv + 6 * (size_t)n), tex;
    for (int y = 0; y < tex_h; y++)
        for (int x = 0; x < tex_w; x++) {
            tex.push_back(((x ^ y) & 1) ? 0.9f : 0.2f);
            tex.push_back((float)x / (float)tex_w);
            tex.push_back((float)y / (float)tex_h);
            tex.push_back(1.0f);
        }
    rt_obj_free(pos); rt_obj_free(nrm); rt_obj_free(uv);

    HostModel mesh{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, 1 /* LAMBERTIAN */, {0.7f, 0.7f, 0.7f}};
    std::vector<HostPointLight> lights = {{{-7.5f, 10.0f, -50.0f}, {550.0f, 500.0f, 400.0f}},
                                          {{8.5f, 9.0f, -47.5f}, {300.0f, 350.0f, 450.0f}}};
    RtBridge b;
    if (!b.init(0, W, H)) { std::fprintf(stderr, "rt_bridge_test: rt_create failed: %s\n", rt_last_error(nullptr)); return 1; }
    if (!b.upload(0, tris, {0}, {n}, {mesh}, {}, lights, tri_uv.data(), tex.empty() ? nullptr : tex.data(), tex_w, tex_h))
        return die(b, "upload");
    const float cpos[3] = {0, 0, 0}, look[3] = {0, 0, -1}, up[3] = {0, 1, 0};
    if (!b.camera(cpos, look, 2.0f, 1.0f, 1000.0f, W / 2, H / 2, up)) return die(b, "camera");   // kernel.cu:228-229
    for (int s = 0; s < spp; s++)                            // the frame loop: one more sample per iteration
        if (!b.sample(depth)) return die(b, "render");
    std::vector<float> rgba((size_t)W * H * 4);
    if (!b.frame(rgba.data())) return die(b, "read_framebuffer");
    if (rt_write_pfm(argv[2], W, H, rgba.data()) != RT_OK) return die(b, "write_pfm");
    rt_stats st;
    if (rt_get_stats(b.ctx, &st) != RT_OK) return die(b, "stats");
    std::printf("rt_bridge_test ok: %d triangles, %llu primary + %llu shadow + %llu bounce rays\n", n,
                (unsigned long long)st.primary_rays, (unsigned long long)st.shadow_rays,
                (unsigned long long)st.bounce_rays);
    return 0;
}
