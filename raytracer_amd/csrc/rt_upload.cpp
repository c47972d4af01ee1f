// rt_upload.cpp -- host-side scene upload: AoS reference objects -> SoA device records.
// Pure host C++ (compiled -ffp-contract=off): every value placed in a device record
// is a ray-independent sub-expression of the reference's per-ray code, evaluated
// here once with the reference's own operation order (rt_math.h), so the GPU
// kernels consume bit-identical operands without redoing the work per ray.
#include "rt_upload.h"

#include <algorithm>
#include <cstring>
#include <thread>

namespace rt {

static m4 load_m4(const float *p)
{
    m4 m;
    std::memcpy(m.v, p, sizeof(m.v));
    return m;
}

static void build_tri(const rt_scene_desc *d, int g, TriRec &T, TriCold &C)
{
    const float *pp = d->tri_pos + 9 * (size_t)g;
    const float *nn = d->tri_nrm + 9 * (size_t)g;
    int model = d->tri_model[g];
    m4 M = load_m4(d->model_xf + 16 * (size_t)model);
    v3 pos[3] = {mk3(pp[0], pp[1], pp[2]), mk3(pp[3], pp[4], pp[5]), mk3(pp[6], pp[7], pp[8])};
    // Model.cu:94  ta, tb, tc = transformation(pos[i])
    v3 ta = apply3(M, pos[0]), tb = apply3(M, pos[1]), tc = apply3(M, pos[2]);
    // Model.cu:97  norma = cross(ta - tc, tb - tc)
    v3 n = cross(sub(ta, tc), sub(tb, tc));
    // Model.cu:413-420 area(): rpos[i] = transformation(pos[i]) (same values as ta,tb,tc)
    float area = 0.5f * length(cross(sub(tc, ta), sub(tb, ta)));
    T.ax = ta.x; T.ay = ta.y; T.az = ta.z;
    T.bx = tb.x; T.by = tb.y; T.bz = tb.z;
    T.cx = tc.x; T.cy = tc.y; T.cz = tc.z;
    T.nx = n.x; T.ny = n.y; T.nz = n.z;
    T.dn = dot(ta, n);          // Model.cu:102 second term
    T.S = 2.0f * area;          // Model.cu:106
    T.pad0 = 0.0f;
    T.pad1 = 0.0f;
    std::memcpy(C.n0, nn, 3 * sizeof(float));
    std::memcpy(C.n1, nn + 3, 3 * sizeof(float));
    std::memcpy(C.n2, nn + 6, 3 * sizeof(float));
    C.model = model;
    C.pad[0] = C.pad[1] = 0;
}

int build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err)
{
    if (!d) { err = "scene is NULL"; return RT_ERR_INVALID; }
    if (d->n_tri_models < 0 || d->n_meshes < 0 || d->n_quads < 0 || d->n_tris < 0 ||
        d->n_point_lights < 0) {
        err = "negative count in scene description";
        return RT_ERR_INVALID;
    }
    const int n_models = d->n_tri_models + d->n_meshes + d->n_quads;
    if (d->n_tris > 0 && (!d->tri_pos || !d->tri_nrm || !d->tri_model)) {
        err = "triangle arrays are NULL"; return RT_ERR_INVALID;
    }
    if (d->n_meshes > 0 && (!d->mesh_first || !d->mesh_count)) {
        err = "mesh_first/mesh_count are NULL"; return RT_ERR_INVALID;
    }
    if (n_models > 0 && (!d->model_xf || !d->model_mat_type || !d->model_color)) {
        err = "model arrays are NULL"; return RT_ERR_INVALID;
    }
    if (d->n_quads > 0 && (!d->quad_coeff || !d->quad_type)) {
        err = "quadric arrays are NULL"; return RT_ERR_INVALID;
    }
    if (d->n_point_lights > 0 && (!d->pl_pos || !d->pl_illum)) {
        err = "point-light arrays are NULL"; return RT_ERR_INVALID;
    }
    // traversal order: standalone triangles, then mesh tables back to back
    long long cursor = d->n_tri_models;
    for (int j = 0; j < d->n_meshes; j++) {
        if (d->mesh_count[j] < 0 || d->mesh_first[j] != cursor) {
            err = "mesh tables must follow the standalone triangles contiguously, in order";
            return RT_ERR_INVALID;
        }
        cursor += d->mesh_count[j];
    }
    if (cursor != d->n_tris) {
        err = "n_tris does not equal n_tri_models + sum(mesh_count)";
        return RT_ERR_INVALID;
    }
    for (int g = 0; g < d->n_tris; g++) {
        int m = d->tri_model[g];
        if (m < 0 || m >= d->n_tri_models + d->n_meshes) {
            err = "tri_model out of range"; return RT_ERR_INVALID;
        }
    }
    for (int m = 0; m < n_models; m++)
        if (d->model_mat_type[m] != kMatLambertian) {
            err = "only LAMBERTIAN materials are on the hot path (GGX/Cook/Oren-Nayar/Fresnel: "
                  "SURVEY section 8(f) N4)";
            return RT_ERR_UNSUPPORTED;
        }

    out.tris.resize((size_t)d->n_tris);
    out.cold.resize((size_t)d->n_tris);
    {
        unsigned hw = std::thread::hardware_concurrency();
        int nthreads = (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
        if (d->n_tris < 65536) nthreads = 1;
        std::vector<std::thread> pool;
        for (int w = 0; w < nthreads; w++)
            pool.emplace_back([&, w]() {
                size_t lo = (size_t)d->n_tris * w / nthreads, hi = (size_t)d->n_tris * (w + 1) / nthreads;
                for (size_t g = lo; g < hi; g++)
                    build_tri(d, (int)g, out.tris[g], out.cold[g]);
            });
        for (auto &t : pool) t.join();
    }

    out.models.resize((size_t)n_models);
    for (int m = 0; m < n_models; m++) {
        m4 M = load_m4(d->model_xf + 16 * (size_t)m);
        out.models[m].cas = transpose(inverse(M));        // Model.cu:126
        std::memcpy(out.models[m].color, d->model_color + 3 * (size_t)m, 3 * sizeof(float));
        out.models[m].mat_type = d->model_mat_type[m];
    }

    out.spheres.resize((size_t)d->n_quads);
    for (int q = 0; q < d->n_quads; q++) {
        SphereRec &S = out.spheres[q];
        int model = d->n_tri_models + d->n_meshes + q;
        S.M = load_m4(d->model_xf + 16 * (size_t)model);
        S.inv = inverse(S.M);                             // Model.cu:352
        S.type = d->quad_type[q];
        S.rad = (S.type == 0) ? (1 / d->quad_coeff[3 * q + 0]) : 0.0f;   // Model.cu:341-346
        S.rad2 = S.rad * S.rad;                           // Model.cu:364
        S.model = model;
    }

    out.lights.resize((size_t)d->n_point_lights);
    for (int j = 0; j < d->n_point_lights; j++) {
        std::memcpy(out.lights[j].pos, d->pl_pos + 3 * (size_t)j, 3 * sizeof(float));
        std::memcpy(out.lights[j].illum, d->pl_illum + 3 * (size_t)j, 3 * sizeof(float));
    }
    return RT_OK;
}

CameraRec make_camera(const rt_camera_desc *a)
{
    CameraRec c;
    v3 pos = mk3(a->pos[0], a->pos[1], a->pos[2]);
    v3 lookat = mk3(a->lookat[0], a->lookat[1], a->lookat[2]);
    c.pos = pos;
    c.up = normalize(mk3(a->up[0], a->up[1], a->up[2]));       // Camera.cu:6
    c.front = normalize(sub(lookat, pos));                     // Camera.cu:7
    c.res_x = a->res_x;                                        // Camera.cu:9
    c.res_y = a->res_y;
    c.n = a->near_;                                            // Camera.cu:11
    c.right = normalize(cross(c.front, c.up));                 // Camera.cu:12
    c.up = normalize(cross(c.right, c.front));                 // Camera.cu:13
    return c;
}

} // namespace rt
