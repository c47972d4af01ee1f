// rt_upload.cpp -- host-side scene upload: AoS reference objects -> SoA device records.
// Pure host C++ (compiled -ffp-contract=off): every value placed in a device record
// is a ray-independent sub-expression of the reference's per-ray code, evaluated
// here once with the reference's own operation order (rt_math.h), so the GPU
// kernels consume bit-identical operands without redoing the work per ray.
#include "rt_upload.h"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

namespace rt {

static m4 load_m4(const float *p)
{
    m4 m;
    std::memcpy(m.v, p, sizeof(m.v));
    return m;
}

static void hit_operands(v3 ta, v3 tb, v3 tc, int orig, TriRec &T);

static void build_tri(const rt_scene_desc *d, int g, TriRec &T, TriCold &C)
{
    const float *pp = d->tri_pos + 9 * (size_t)g;
    const float *nn = d->tri_nrm + 9 * (size_t)g;
    int model = d->tri_model[g];
    m4 M = load_m4(d->model_xf + 16 * (size_t)model);
    v3 pos[3] = {mk3(pp[0], pp[1], pp[2]), mk3(pp[3], pp[4], pp[5]), mk3(pp[6], pp[7], pp[8])};
    // Model.cu:94  ta, tb, tc = transformation(pos[i])
    v3 ta = apply3(M, pos[0]), tb = apply3(M, pos[1]), tc = apply3(M, pos[2]);
    hit_operands(ta, tb, tc, g, T);
    std::memcpy(C.n0, nn, 3 * sizeof(float));
    std::memcpy(C.n1, nn + 3, 3 * sizeof(float));
    std::memcpy(C.n2, nn + 6, 3 * sizeof(float));
    C.model = model;
    C.pad[0] = C.pad[1] = 0;
}

// The ray-independent operands of Triangle::hit (Model.cu:97-113) -- and of TriangleLight::hit
// (Light.cu:72-87), which is the same code on the light's vertices -- for world-space vertices.
static void hit_operands(v3 ta, v3 tb, v3 tc, int orig, TriRec &T)
{
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
    T.orig = orig;
    // Largest |cross|^2 (Model.cu:107-109) with which the area test (:113) can pass:
    // one s_i alone above (S + 0.001)(1 + 1e-5) already puts s1+s2+s3-S beyond 0.001.
    {
        double q = ((double)T.S + 0.001) * (1.0 + 1e-5);
        q *= q;
        float qf = (float)q;
        if (!(qf >= q)) qf = std::nextafter(qf, std::numeric_limits<float>::infinity());
        T.qmax = std::isfinite(q) ? qf : std::numeric_limits<float>::infinity();
    }
}

// ------------------------------------------------------------------ culling bounds
// Everything below runs in double on the float values the kernels will see; results
// are rounded OUTWARD to float.  DESIGN.md ("Culling is results-identical") has the
// argument; the short form:
//   Triangle::hit accepts a point p only if fl(s1+s2+s3) - S <= 0.001 (Model.cu:113).
//   In exact arithmetic s1+s2+s3 - S = 2 * sum over edges p lies outside of
//   (edge length * distance to the edge line) for p in the plane, and grows like
//   sqrt(S^2 + P^2 h^2) - S with the height h above it.  With x = 0.001 + the fp32
//   evaluation error, p must therefore lie within
//       delta = x / k + sqrt(2 S x + x^2) / P,   k = 2 e_min sin(theta_min / 2)
//   of the triangle, hence inside the triangle's bounding box grown by delta.
namespace {

struct Box { double lo[3], hi[3]; };      // lo > hi: dead; (-inf, +inf): never culled
const double kInfD = std::numeric_limits<double>::infinity();
const Box kDeadBox = {{kInfD, kInfD, kInfD}, {-kInfD, -kInfD, -kInfD}};
const Box kOpenBox = {{-kInfD, -kInfD, -kInfD}, {kInfD, kInfD, kInfD}};

inline double len3(const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
inline void cross3d(const double *a, const double *b, double *r)
{
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
inline float round_dn(double v)
{
    float f = (float)v;
    if ((double)f > v) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double v)
{
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

// Box of one triangle's acceptance region, WITHOUT the coordinate-magnitude slack
// (added by build_tree once the extent of the whole scene is known).
Box leaf_box(const TriRec &T)
{
    const double a[3] = {T.ax, T.ay, T.az}, b[3] = {T.bx, T.by, T.bz}, c[3] = {T.cx, T.cy, T.cz};
    const double n1 = std::fabs((double)T.nx) + std::fabs((double)T.ny) + std::fabs((double)T.nz);
    for (int i = 0; i < 3; i++)
        if (!std::isfinite(a[i]) || !std::isfinite(b[i]) || !std::isfinite(c[i])) return kOpenBox;
    if (!std::isfinite(n1) || !std::isfinite((double)T.S)) return kOpenBox;
    // |dot(norma, dir)| <= |norma|_1 (1 + ulps) < 0.01: Model.cu:99 rejects every ray (H1)
    if (n1 * 1.00001 < 0.01) return kDeadBox;
    double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    double v[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    double w[3] = {c[0] - b[0], c[1] - b[1], c[2] - b[2]};
    double eab = len3(u), eca = len3(v), ebc = len3(w);
    double emin = std::min(eab, std::min(eca, ebc)), emax = std::max(eab, std::max(eca, ebc));
    double P = eab + eca + ebc;
    double cr[3];
    cross3d(u, v, cr);
    double Sd = len3(cr);
    if (!(emin > 0.0) || !(Sd > 0.0)) return kOpenBox;
    // smallest vertex angle, from the law of cosines
    auto ang = [](double opp, double s1, double s2) {
        double cs = (s1 * s1 + s2 * s2 - opp * opp) / (2.0 * s1 * s2);
        return std::acos(std::max(-1.0, std::min(1.0, cs)));
    };
    double th = std::min(ang(ebc, eab, eca), std::min(ang(eca, eab, ebc), ang(eab, eca, ebc)));
    double k = 2.0 * emin * std::sin(0.5 * th);
    if (!(k > 0.0)) return kOpenBox;
    // fixed point of delta = f(x(delta)); gamma bounds the fp32 error of one |cross|,
    // D the distance from an accepted p to any vertex
    const double gamma = 1.0 / 1048576.0;
    auto f = [&](double delta) {
        double D = emax + delta;
        double x = 0.001 * (1.0 + 1e-6) + std::fabs((double)T.S - Sd) + 6.0 * gamma * D * D +
                   gamma * (Sd + 0.001);
        return x / k + std::sqrt(2.0 * Sd * x + x * x) / P;
    };
    double delta = 0.0;
    for (int it = 0; it < 16; it++) delta = f(delta);
    delta = 1.5 * delta + 1e-6 * emax;
    if (!std::isfinite(delta) || f(delta) > delta) return kOpenBox;   // no consistent bound
    Box B;
    for (int i = 0; i < 3; i++) {
        B.lo[i] = std::min(a[i], std::min(b[i], c[i])) - delta;
        B.hi[i] = std::max(a[i], std::max(b[i], c[i])) + delta;
    }
    return B;
}

inline bool box_dead(const Box &b) { return b.lo[0] > b.hi[0]; }

template <class F>
void parallel_for(size_t n, F fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
    if (n < 65536) nthreads = 1;
    std::vector<std::thread> pool;
    for (int w = 0; w < nthreads; w++)
        pool.emplace_back([&, w]() {
            size_t lo = n * w / nthreads, hi = n * (w + 1) / nthreads;
            for (size_t g = lo; g < hi; g++) fn(g);
        });
    for (auto &t : pool) t.join();
}

// Order the triangles by recursive rank partition of their box centres and stack the
// implicit 4-ary box tree on top.  `tris` arrives in reference traversal order.
void build_tree(HostScene &hs)
{
    const size_t n = hs.tris.size();
    std::vector<Box> lb(n);
    parallel_for(n, [&](size_t g) { lb[g] = leaf_box(hs.tris[g]); });

    double lo[3] = {kInfD, kInfD, kInfD}, hi[3] = {-kInfD, -kInfD, -kInfD}, ext = 0.0;
    for (size_t g = 0; g < n; g++) {
        const TriRec &T = hs.tris[g];
        const float vv[9] = {T.ax, T.ay, T.az, T.bx, T.by, T.bz, T.cx, T.cy, T.cz};
        for (float x : vv)
            if (std::isfinite(x)) ext = std::max(ext, (double)std::fabs(x));
        if (!box_dead(lb[g]) && std::isfinite(lb[g].lo[0]))
            for (int i = 0; i < 3; i++) {
                lo[i] = std::min(lo[i], 0.5 * (lb[g].lo[i] + lb[g].hi[i]));
                hi[i] = std::max(hi[i], 0.5 * (lb[g].lo[i] + lb[g].hi[i]));
            }
    }
    for (const SphereRec &S : hs.spheres) {
        double r = std::fabs((double)S.rad);
        for (int i = 0; i < 3; i++) {
            double row = std::fabs((double)S.M.v[i][0]) + std::fabs((double)S.M.v[i][1]) +
                         std::fabs((double)S.M.v[i][2]);
            double e = std::fabs((double)S.M.v[i][3]) + r * row;
            if (std::isfinite(e)) ext = std::max(ext, e);
        }
    }
    hs.geom_extent = (float)std::min(std::max(ext, 1.0), 1e30);
    // fp32 slack of the ray side: p = fl(o + t d) lies within 2^-23 (|o| + |p|) of the
    // exact line, and ray origins are culled only inside [-E, E]^3, E = kCullExtent * extent
    const double slack = (double)hs.geom_extent * kCullExtent * (1.0 / 524288.0);

    // Device order = the leaf order of the implicit tree.  A group of level l has four
    // children of 4^l consecutive positions each, so the order is fixed by a recursive
    // partition by RANK: the triangles of a node are cut into its children's position
    // ranges with nth_element on their centres along one axis (first in two at the middle
    // child boundary, then each half again along an axis of its own); the axis is the one
    // of three whose halves have the smallest count-weighted box area (C3 -7 %, C4 -10 %
    // render time against "longest axis").  Compared with a space-filling-curve order this
    // keeps sibling boxes disjoint in the cut axis at every level, which is what prunes
    // the traversal.
    std::vector<float> ctr(3 * n);
    parallel_for(n, [&](size_t g) {
        const bool ok = !box_dead(lb[g]) && std::isfinite(lb[g].lo[0]) && std::isfinite(lb[g].hi[0]);
        for (int i = 0; i < 3; i++) {
            double c = ok ? 0.5 * (lb[g].lo[i] + lb[g].hi[i]) : 0.0;
            ctr[3 * g + i] = std::isfinite(c) ? (float)c : 0.0f;
        }
    });
    std::vector<uint32_t> order(n);
    for (size_t g = 0; g < n; g++) order[g] = (uint32_t)g;
    int levels = 0;                                  // children of the top group cover 4^levels positions
    for (size_t cap = kTreeArity; cap < n; cap *= kTreeArity) levels++;
    auto half_area = [&](const uint32_t *first, const uint32_t *last) {
        double mn[3] = {kInfD, kInfD, kInfD}, mx[3] = {-kInfD, -kInfD, -kInfD};
        for (const uint32_t *k = first; k != last; ++k) {
            const Box &b = lb[*k];
            if (box_dead(b)) continue;
            for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], b.lo[i]); mx[i] = std::max(mx[i], b.hi[i]); }
        }
        if (mx[0] < mn[0]) return 0.0;
        const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto cut = [&](size_t lo, size_t mid, size_t hi) {
        // order[lo, hi) -> the `mid - lo` smallest centres along one axis first: the axis whose two
        // halves have the smallest count-weighted box area (the surface-area heuristic at the
        // one split position an implicit tree allows); the longest axis for tiny ranges
        if (mid <= lo || mid >= hi) return;
        float mn[3] = {3e38f, 3e38f, 3e38f}, mx[3] = {-3e38f, -3e38f, -3e38f};
        for (size_t k = lo; k < hi; k++)
            for (int i = 0; i < 3; i++) {
                float c = ctr[3 * (size_t)order[k] + i];
                mn[i] = std::min(mn[i], c);
                mx[i] = std::max(mx[i], c);
            }
        int ax = 0;
        if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
        if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
        auto split = [&](int axis) {
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                             [&](uint32_t x, uint32_t y) {
                                 float cx = ctr[3 * (size_t)x + axis], cy = ctr[3 * (size_t)y + axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        };
        if (hi - lo > 8) {
            double best = kInfD;
            for (int axis = 0; axis < 3; axis++) {
                split(axis);
                const double a = half_area(order.data() + lo, order.data() + mid) * (double)(mid - lo) +
                                 half_area(order.data() + mid, order.data() + hi) * (double)(hi - mid);
                if (a < best) { best = a; ax = axis; }
            }
        }
        split(ax);
    };
    std::function<void(size_t, size_t, int, int)> part = [&](size_t lo, size_t hi, int l, int depth) {
        // node whose four children cover 4^l positions each, holding order[lo, hi)
        if (l <= 0 || hi - lo <= 1) return;
        size_t c = 1;
        for (int k = 0; k < l; k++) c *= kTreeArity;
        const size_t m2 = std::min(hi, lo + 2 * c);
        cut(lo, m2, hi);
        cut(lo, std::min(m2, lo + c), m2);
        cut(m2, std::min(hi, lo + 3 * c), hi);
        std::vector<std::thread> pool;
        for (int k = 0; k < kTreeArity; k++) {
            const size_t a = std::min(hi, lo + (size_t)k * c), b = std::min(hi, a + c);
            if (b <= a) continue;
            if (depth < 2 && b - a > 65536)
                pool.emplace_back([&part, a, b, l, depth]() { part(a, b, l - 1, depth + 1); });
            else
                part(a, b, l - 1, depth + 1);
        }
        for (auto &t : pool) t.join();
    };
    part(0, n, levels, 0);

    const size_t n_pad = std::max<size_t>(kTreeArity, (n + kTreeArity - 1) / kTreeArity * kTreeArity);
    std::vector<TriRec> sorted(n_pad);
    std::memset(sorted.data(), 0, n_pad * sizeof(TriRec));     // padding: norma = 0, never hit
    hs.pos_of.assign(n, 0u);
    std::vector<Box> level(n_pad, kDeadBox);
    parallel_for(n, [&](size_t pos) {
        uint32_t g = order[pos];
        sorted[pos] = hs.tris[g];
        hs.pos_of[g] = (uint32_t)pos;
        Box b = lb[g];
        if (!box_dead(b))
            for (int i = 0; i < 3; i++) {
                b.lo[i] = (double)round_dn(b.lo[i] - slack);
                b.hi[i] = (double)round_up(b.hi[i] + slack);
            }
        level[pos] = b;
    });
    for (size_t pos = n; pos < n_pad; pos++) sorted[pos].orig = -1;
    hs.tris.swap(sorted);

    hs.groups.clear();
    hs.level_off.clear();
    hs.level_cnt.clear();
    for (;;) {
        const size_t ng = level.size() / kTreeArity;
        hs.level_off.push_back(hs.groups.size());
        hs.level_cnt.push_back((uint32_t)ng);
        hs.groups.resize(hs.groups.size() + ng);
        NodeGroup *out = hs.groups.data() + hs.level_off.back();
        const size_t np_pad = ng <= 1 ? 0 : (ng + kTreeArity - 1) / kTreeArity * kTreeArity;
        std::vector<Box> up(np_pad, kDeadBox);
        parallel_for(ng, [&](size_t i) {
            Box u = kDeadBox;
            for (int j = 0; j < kTreeArity; j++) {
                const Box &b = level[i * kTreeArity + j];
                for (int a = 0; a < 3; a++) {
                    out[i].lo[a][j] = (float)b.lo[a];        // already float-representable
                    out[i].hi[a][j] = (float)b.hi[a];
                }
                if (box_dead(b)) continue;
                for (int a = 0; a < 3; a++) {
                    u.lo[a] = std::min(u.lo[a], b.lo[a]);
                    u.hi[a] = std::max(u.hi[a], b.hi[a]);
                }
            }
            if (np_pad) up[i] = u;
        });
        if (ng <= 1) break;
        level.swap(up);
    }
    hs.top = (int)hs.level_cnt.size() - 1;
}

} // namespace

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
    if (d->n_dir_lights < 0 || (d->n_dir_lights > 0 && (!d->dl_dir || !d->dl_illum))) {
        err = "directional-light arrays are NULL or their count negative"; return RT_ERR_INVALID;
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
    if (d->n_tri_lights < 0 || (d->n_tri_lights > 0 && (!d->tl_pos || !d->tl_illum || !d->tl_two_side ||
                                                        !d->tl_mat_type || !d->tl_color))) {
        err = "triangle-light arrays are NULL or their count negative"; return RT_ERR_INVALID;
    }
    for (int k = 0; k < d->n_tri_lights; k++)
        if (d->tl_mat_type[k] != kMatLambertian) {
            err = "a triangle light's own material must be LAMBERTIAN (Light.cu:115-125)";
            return RT_ERR_UNSUPPORTED;
        }
    if (d->n_env_lights < 0 || d->n_env_lights > 1) {
        err = "at most one environment light (Scene::getEnvironmentLight reads el[0], Scene.cuh:28-31)";
        return RT_ERR_UNSUPPORTED;
    }
    if (d->n_env_lights == 1 && (!d->env_image || d->env_width < 1 || d->env_height < 1)) {
        err = "environment light without an image"; return RT_ERR_INVALID;
    }
    out.general = d->n_tri_lights > 0 || d->n_env_lights > 0;
    // Mesh::initTexture (Model.cu:315-319) -> initImages (Texture.cu:135-186)
    if (d->n_textures < 0 || (d->n_textures > 0 && (!d->mesh_texture || !d->tex_size || !d->tex_data))) {
        err = "texture arrays are NULL or their count negative"; return RT_ERR_INVALID;
    }
    out.n_tri_models = d->n_tri_models;
    out.textures.clear(); out.tex_level0.clear(); out.tex_level0_at.clear(); out.mesh_tex.clear(); out.uv.clear();
    out.tex_texels = 0;
    if (d->n_textures > 0) {
        const float *img = d->tex_data;
        size_t at0 = 0;
        for (int k = 0; k < d->n_textures; k++) {
            const int w = d->tex_size[2 * k], h = d->tex_size[2 * k + 1];
            if (w < 1 || h < 1 || w > 32768 || h > 32768) {
                err = "texture size must be 1..32768 texels a side"; return RT_ERR_INVALID;
            }
            TexRec T{};
            T.w = w; T.h = h;
            size_t sz = (size_t)std::max(w, h);                        // getMipMapLevels, Texture.cu:2-13
            while (sz) { sz /= 2; T.levels++; }
            int lw = w, lh = h;
            for (int l = 0; l < T.levels; l++) {
                if (out.tex_texels > 0xffffffffull - (size_t)lw * lh) { err = "textures too large"; return RT_ERR_UNSUPPORTED; }
                T.off[l] = (uint32_t)out.tex_texels;
                out.tex_texels += (size_t)lw * lh;
                lw = std::max(1, lw / 2); lh = std::max(1, lh / 2);    // Texture.cu:55-58
            }
            out.textures.push_back(T);
            out.tex_level0_at.push_back((uint32_t)at0);
            out.tex_level0.insert(out.tex_level0.end(), img, img + 4 * (size_t)w * h);
            at0 += (size_t)w * h;
            img += 4 * (size_t)w * h;
        }
        for (int j = 0; j < d->n_meshes; j++) {
            if (d->mesh_texture[j] < -1 || d->mesh_texture[j] >= d->n_textures) {
                err = "mesh_texture out of range"; return RT_ERR_INVALID;
            }
            out.mesh_tex.push_back(d->mesh_texture[j]);
        }
        out.uv.assign((size_t)d->n_tris, TriUv{});
        if (d->tri_uv)
            std::memcpy(out.uv.data(), d->tri_uv, sizeof(TriUv) * (size_t)d->n_tris);
        out.general = true;
    }
    for (int m = 0; m < n_models; m++) {
        const int t = d->model_mat_type[m];
        if (t != kMatLambertian && t != kMatFresnel && t != kMatCook && t != kMatGGX && t != kMatOrenNayar) {
            err = "unknown material::MATERIAL_TYPE (Object.cuh:13-21: 1, 7, 9, 17, 25)";
            return RT_ERR_UNSUPPORTED;
        }
        if ((t == kMatFresnel || t == kMatCook || t == kMatGGX) && !d->model_param) {
            err = "model_param is NULL but a GGX / Cook_Torrance / Fresnel material needs its roughness / eta";
            return RT_ERR_INVALID;
        }
        if (t != kMatLambertian) out.general = true;
    }

    out.tris.resize((size_t)d->n_tris);
    out.cold.resize((size_t)d->n_tris);
    parallel_for((size_t)d->n_tris,
                 [&](size_t g) { build_tri(d, (int)g, out.tris[g], out.cold[g]); });

    out.models.resize((size_t)n_models);
    for (int m = 0; m < n_models; m++) {
        m4 M = load_m4(d->model_xf + 16 * (size_t)m);
        out.models[m].cas = transpose(inverse(M));        // Model.cu:126
        std::memcpy(out.models[m].color, d->model_color + 3 * (size_t)m, 3 * sizeof(float));
        out.models[m].mat_type = d->model_mat_type[m];
    }

    // the BRDF objects as their constructors leave them (material.cuh:7-13,48,88,112); logf as
    // the oracle fixes it: the float rounding of the double-precision log
    out.mats.assign((size_t)n_models + (size_t)d->n_tri_lights, MatRec{});
    for (int m = 0; m < n_models; m++) {
        float rough = d->model_param ? d->model_param[2 * (size_t)m] : 0.0f;
        rough = std::fmax(rough, 1e-3f);
        const float x = (float)std::log((double)rough);
        const float alpha = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x +
                            0.000640711f * x * x * x * x;
        out.mats[m].m = alpha;
        out.mats[m].a2 = alpha * alpha;
        out.mats[m].eta = d->model_param ? d->model_param[2 * (size_t)m + 1] : 0.0f;
    }
    // triangle lights and their Materials (ModelRec n_models + k; cas is not read for them)
    out.tlights.assign((size_t)d->n_tri_lights, TriLightRec{});
    out.models.resize((size_t)n_models + (size_t)d->n_tri_lights);
    for (int k = 0; k < d->n_tri_lights; k++) {
        const float *pp = d->tl_pos + 9 * (size_t)k;
        const v3 p0 = mk3(pp[0], pp[1], pp[2]), p1 = mk3(pp[3], pp[4], pp[5]), p2 = mk3(pp[6], pp[7], pp[8]);
        TriLightRec &L = out.tlights[k];
        hit_operands(p0, p1, p2, d->n_tris + d->n_quads + k, L.T);     // Light.cu:72-81
        const v3 n = normalize(cross(sub(p1, p0), sub(p2, p0)));       // Light.cu:39
        L.normal[0] = n.x; L.normal[1] = n.y; L.normal[2] = n.z;
        L.area = 0.5f * length(cross(sub(p2, p0), sub(p1, p0)));       // Light.cuh:76-79
        std::memcpy(L.illum, d->tl_illum + 3 * (size_t)k, 3 * sizeof(float));
        L.two_side = d->tl_two_side[k] != 0;
        const v3 e1 = sub(p1, p0), e2 = sub(p2, p0);
        L.e1[0] = e1.x; L.e1[1] = e1.y; L.e1[2] = e1.z;
        L.e2[0] = e2.x; L.e2[1] = e2.y; L.e2[2] = e2.z;
        L.model = n_models + k;
        ModelRec &M = out.models[(size_t)n_models + k];
        M = ModelRec{};
        std::memcpy(M.color, d->tl_color + 3 * (size_t)k, 3 * sizeof(float));
        M.mat_type = d->tl_mat_type[k];
    }

    // EnvironmentLight::setUp (Light.cu:201-234): Distribution2D over the image's luminance
    // (PiecewiseConstant.cu:65-96), each row and the row integrals a Distribution (:8-22)
    out.env_n = d->n_env_lights;
    if (d->n_env_lights == 1) {
        const int w = d->env_width, h = d->env_height;
        const float *img = d->env_image;
        out.env_w = w; out.env_h = h;
        out.env_img.assign(img, img + 4 * (size_t)w * h);
        auto make = [](int n, const float *v, float *value, float *cdf) -> float {
            std::memcpy(value, v, sizeof(float) * (size_t)n);
            cdf[0] = 0.0f;
            for (int i = 1; i <= n; i++) cdf[i] = cdf[i - 1] + value[i - 1] / n;
            const float funInt = cdf[n];
            for (int i = 1; i <= n; i++) cdf[i] = funInt == 0 ? (float)i / n : cdf[i] / funInt;
            cdf[n] = 1.0f;
            return funInt;
        };
        out.env_cond_value.assign((size_t)w * h, 0.0f);
        out.env_cond_cdf.assign((size_t)(w + 1) * h, 0.0f);
        out.env_cond_funint.assign((size_t)h, 0.0f);
        out.env_marg_value.assign((size_t)h, 0.0f);
        out.env_marg_cdf.assign((size_t)h + 1, 0.0f);
        std::vector<float> ctmp((size_t)w), vtmp((size_t)h);
        for (int i = 0; i < h; i++) {
            for (int j = 0; j < w; j++) {
                const size_t idx = 4 * ((size_t)i * w + j);
                // rgb2y_xyz (PiecewiseConstant.cu:3-6): the 0.06007 is a double
                ctmp[j] = (float)((double)(img[idx] + 4.5906f * img[idx + 1]) + 0.06007 * (double)img[idx + 2]);
            }
            vtmp[i] = make(w, ctmp.data(), &out.env_cond_value[(size_t)i * w], &out.env_cond_cdf[(size_t)i * (w + 1)]);
            out.env_cond_funint[i] = vtmp[i];
        }
        out.env_marg_funint = make(h, vtmp.data(), out.env_marg_value.data(), out.env_marg_cdf.data());
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

    build_tree(out);
    if (out.top >= kMaxLevels) { err = "too many triangles for the traversal tree"; return RT_ERR_UNSUPPORTED; }

    out.lights.assign((size_t)d->n_point_lights + (size_t)d->n_dir_lights, LightRec{});
    for (int j = 0; j < d->n_point_lights; j++) {
        std::memcpy(out.lights[j].pos, d->pl_pos + 3 * (size_t)j, 3 * sizeof(float));
        std::memcpy(out.lights[j].illum, d->pl_illum + 3 * (size_t)j, 3 * sizeof(float));
        out.lights[j].type = kLightPoint;
    }
    for (int j = 0; j < d->n_dir_lights; j++) {
        LightRec &L = out.lights[(size_t)d->n_point_lights + j];
        const float *a = d->dl_dir + 3 * (size_t)j;
        v3 dir = normalize(mk3(a[0], a[1], a[2]));            // Light.cu:21: dir(normalize(direction))
        L.pos[0] = dir.x; L.pos[1] = dir.y; L.pos[2] = dir.z;
        std::memcpy(L.illum, d->dl_illum + 3 * (size_t)j, 3 * sizeof(float));
        L.type = kLightDirectional;
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
    c.aspect = (float)c.res_x / c.res_y;                       // Camera.cu:19, hoisted
    c.right = normalize(cross(c.front, c.up));                 // Camera.cu:12
    c.up = normalize(cross(c.right, c.front));                 // Camera.cu:13
    return c;
}

} // namespace rt
