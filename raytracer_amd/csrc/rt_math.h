// rt_math.h -- fp32 vector / mat4 arithmetic shared by the host-side upload code and
// the gfx950 kernels.  Every function restates the operation ORDER of the reference
// (helper_math.h, Matrix/Matrix.cu) so that values computed at upload time on the
// host and values computed per ray on the GPU carry the same bits the reference's
// per-ray code would produce.  Compile with -ffp-contract=off: no FMA contraction.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct m4 { float v[4][4]; };

RT_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD v4 mk4(float x, float y, float z, float w) { v4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// helper_math.h:349, 579, 804, 814, 997 -- componentwise
RT_HD v3 add(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD v3 sub(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD v3 mul(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD v3 scale(v3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
RT_HD v3 divs(v3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
RT_HD v3 neg(v3 a) { return mk3(-a.x, -a.y, -a.z); }
// helper_math.h:1248: (ax*bx + ay*by) + az*bz
RT_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float dot(v4 a, v4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
// helper_math.h:1420
RT_HD v3 cross(v3 a, v3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// helper_math.h:62-65: the oracle is the reference built as HOST code, where
// rsqrtf(x) is 1.0f / sqrtf(x) (two correctly rounded operations).
RT_HD float rsqrt_ref(float x) { return 1.0f / sqrtf(x); }
RT_HD float length(v3 v) { return sqrtf(dot(v, v)); }                     // :1291
RT_HD v3 normalize(v3 v) { float inv = rsqrt_ref(dot(v, v)); return scale(v, inv); }  // :1309
RT_HD v4 normalize(v4 v)                                                  // :1314
{
    float inv = rsqrt_ref(dot(v, v));
    return mk4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

// Matrix/Matrix.cu:73-83
RT_HD v4 apply4(const m4 &m, v4 a)
{
    v4 r;
    r.x = m.v[0][0] * a.x + m.v[0][1] * a.y + m.v[0][2] * a.z + m.v[0][3] * a.w;
    r.y = m.v[1][0] * a.x + m.v[1][1] * a.y + m.v[1][2] * a.z + m.v[1][3] * a.w;
    r.z = m.v[2][0] * a.x + m.v[2][1] * a.y + m.v[2][2] * a.z + m.v[2][3] * a.w;
    r.w = m.v[3][0] * a.x + m.v[3][1] * a.y + m.v[3][2] * a.z + m.v[3][3] * a.w;
    return r;
}
// Matrix/Matrix.cu:85-91: homogeneous point, divided by w
RT_HD v3 apply3(const m4 &m, v3 a)
{
    v4 r = apply4(m, mk4(a.x, a.y, a.z, 1.0f));
    return mk3(r.x / r.w, r.y / r.w, r.z / r.w);
}
// direction: make_float3(M(make_float4(v, 0)))
RT_HD v3 apply_dir(const m4 &m, v3 a)
{
    v4 r = apply4(m, mk4(a.x, a.y, a.z, 0.0f));
    return mk3(r.x, r.y, r.z);
}
// Matrix/Matrix.cu:235-243
RT_HD m4 transpose(const m4 &m)
{
    m4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.v[i][j] = m.v[j][i];
    return r;
}
// Matrix/Matrix.cu:181-233 -- cofactors, scale by 1/det, transpose.  The
// reference's factor11 repeats factor07's expression; kept.
RT_HD m4 inverse(const m4 &mm)
{
    const float(*m)[4] = mm.v;
    float f00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float f01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float f02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float f03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float f04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float f05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float f06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float f07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float f08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float f09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float f10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float f11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float f12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float f13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float f14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float f15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float f16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float f17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float f18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    m4 I;
    I.v[0][0] = +(m[1][1] * f00 - m[1][2] * f01 + m[1][3] * f02);
    I.v[0][1] = -(m[1][0] * f00 - m[1][2] * f03 + m[1][3] * f04);
    I.v[0][2] = +(m[1][0] * f01 - m[1][1] * f03 + m[1][3] * f05);
    I.v[0][3] = -(m[1][0] * f02 - m[1][1] * f04 + m[1][2] * f05);

    I.v[1][0] = -(m[0][1] * f00 - m[0][2] * f01 + m[0][3] * f02);
    I.v[1][1] = +(m[0][0] * f00 - m[0][2] * f03 + m[0][3] * f04);
    I.v[1][2] = -(m[0][0] * f01 - m[0][1] * f03 + m[0][3] * f05);
    I.v[1][3] = +(m[0][0] * f02 - m[0][1] * f04 + m[0][2] * f05);

    I.v[2][0] = +(m[0][1] * f06 - m[0][2] * f07 + m[0][3] * f08);
    I.v[2][1] = -(m[0][0] * f06 - m[0][2] * f09 + m[0][3] * f10);
    I.v[2][2] = +(m[0][0] * f11 - m[0][1] * f09 + m[0][3] * f12);
    I.v[2][3] = -(m[0][0] * f08 - m[0][1] * f10 + m[0][2] * f12);

    I.v[3][0] = -(m[0][1] * f13 - m[0][2] * f14 + m[0][3] * f15);
    I.v[3][1] = +(m[0][0] * f13 - m[0][2] * f16 + m[0][3] * f17);
    I.v[3][2] = -(m[0][0] * f14 - m[0][1] * f16 + m[0][3] * f18);
    I.v[3][3] = +(m[0][0] * f15 - m[0][1] * f17 + m[0][2] * f18);

    float det = +m[0][0] * I.v[0][0] + m[0][1] * I.v[0][1] + m[0][2] * I.v[0][2] +
                m[0][3] * I.v[0][3];
    float b = 1.0f / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            I.v[i][j] *= b;
    return transpose(I);
}

// ---- constants of the path (reference file:line) ----
constexpr float kInf = 1000000.0f;          // Ray/IntersectionRecord.cuh:7-9  INF
constexpr float kEps = 1e-2f;               // Model/Model.cuh:24-26           FLOAT_EPISLON
constexpr float kInvPi = 0.318309892f;      // Material/material.cu:5
constexpr float kPiOver4 = 0.78539816339744830961f;  // material.cu:7
constexpr float kPiOver2 = 1.57079632679489661923f;  // material.cu:8
constexpr int kMatLambertian = 1;           // Material/Object.cuh:15
constexpr int kMatSpecularBit = 4;          // Material/Object.cuh:10
constexpr int kMatFresnel = 7, kMatCook = 9, kMatGGX = 17, kMatOrenNayar = 25;   // Object.cuh:16-19

// ---- device-resident scene records (built on the host at upload) ----
// One 64-byte record per triangle.  Every arithmetic field is a ray-independent
// sub-expression of Triangle::hit (Model/Model.cu:91-137) evaluated with the
// reference's own operation order:
//   a,b,c = transformation(pos[i])             (:94, Matrix.cu:85-91)
//   n     = cross(a - c, b - c)                (:97)
//   dn    = dot(a, n)                          (:102, second term)
//   S     = 2.0f * area()                      (:106, :413-420)
// The records are stored in TRAVERSAL-TREE order (rt_upload.cpp build_tree: recursive rank
// partition of the leaf boxes' centres); `orig` is the triangle's index in the reference's own
// traversal order (Scene.cu:13-41), which is what the id buffer reports and what
// breaks exact ties in t (hazard H8: the reference keeps the first one).
//   qmax  = upper bound on any |cross|^2 of Model.cu:107-109 for which the area test
//           :113 can still pass: ((S + 0.001) * (1 + 1e-5))^2 rounded up.
struct TriRec {
    float ax, ay, az, dn;
    float bx, by, bz, S;
    float cx, cy, cz;
    int32_t orig;
    float nx, ny, nz, qmax;
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

// Conservative axis-aligned bounds of everything Triangle::hit can accept, four nodes
// per record in struct-of-arrays form (96 bytes = one s_load_dwordx16 + one
// s_load_dwordx8): lo[axis][child], hi[axis][child].  Leaves (level 0) bound one
// triangle's acceptance region each -- the triangle's box grown by the distance
// `delta` within which the area test of Model.cu:113 can still pass, plus the fp32
// slack of p = o + t d; inner nodes are exact unions.  A ray whose parameter interval
// inside the box misses (0, best) cannot be accepted by any triangle below the node
// ("Culling is results-identical").  lo > hi: dead (padding, or a triangle
// Model.cu:99 rejects for every ray); lo = -inf, hi = +inf: never culled.
struct NodeGroup { float lo[3][4]; float hi[3][4]; };
static_assert(sizeof(NodeGroup) == 96, "NodeGroup must be 96 bytes");

constexpr float kCullExtent = 16.0f; // rays are culled while their origin stays inside
                                     // [-E, E]^3, E = kCullExtent * max |scene coordinate|
constexpr int kTreeArity = 4;        // children per node = one 64-byte scalar load
constexpr int kMaxLevels = 16;       // 4^16 leaves

// Read only for the winning triangle (Model.cu:122-131).
struct TriCold {
    float n0[3], n1[3], n2[3];   // Triangle::normal[3]
    int32_t model;
    int32_t pad[2];
};
static_assert(sizeof(TriCold) == 48, "TriCold must be 48 bytes");

struct ModelRec {
    m4 cas;             // transpose(inverse(transformation)), Model.cu:126
    float color[3];     // Lambertian::color
    int32_t mat_type;   // material::MATERIAL_TYPE
};

// The BRDF object behind a model's Material as its constructor leaves it (material.cuh:48,88,112);
// read only by the general kernel (non-Lambertian materials, triangle lights).
struct MatRec {
    float a2;           // GGX: rough2alpha(roughness)^2; Cook_Torrance: m * m
    float m;            // Cook_Torrance: rough2alpha(rough)
    float eta;          // Fresnel
    float pad;
};

struct SphereRec {
    m4 M;               // transformation
    m4 inv;             // inverse(transformation), Model.cu:352
    float rad;          // getRadius() = 1 / coefficient.x, Model.cu:341-346
    float rad2;         // getRadius() * getRadius(), Model.cu:364
    int32_t model;
    int32_t type;       // quadraticType; only Sphere (0) intersects
};

// Delta lights in Scene::sampleAllLight order (Scene.cu:83-102): point lights, then
// directional ones.  type 0: pos = PointLight::pos; type 1: pos = DirectionalLight::dir,
// normalised at upload as its constructor does (Light.cu:21).
struct LightRec { float pos[3]; float illum[3]; int32_t type; int32_t pad; };
constexpr int kLightPoint = 0, kLightDirectional = 1;

// One TriangleLight (Light/Light.cuh:66-103).  Its hit test (Light.cu:68-92) is Triangle::hit's
// (Model.cu:97-120) on the light's own vertices, operand for operand, so T holds the same
// ray-independent sub-expressions: n = cross(p0 - p2, p1 - p2) -- the UNNORMALISED local of
// Light.cu:72 that also becomes the record's normal (:97) --, dn = dot(p0, n), S = 2 area().
// T.orig = n_tris + n_quads + k: what the id buffer reports for a hit on light k.
struct TriLightRec {
    TriRec T;
    float normal[3];    // the member: normalize(cross(p1 - p0, p2 - p0)) (Light.cu:39)
    float area;         // area() (Light.cuh:76-79)
    float illum[3];
    int32_t two_side;
    float e1[3];        // p1 - p0, p2 - p0: the tangent's operands (Light.cu:100)
    int32_t model;      // ModelRec of the light's own Material: n_models + k (Scene.cu:151-153)
    float e2[3];
    int32_t pad;
};
static_assert(sizeof(TriLightRec) == 128, "TriLightRec must be 128 bytes");

// EnvironmentLight (Light/Light.cuh:105-129) as setUp leaves it (Light.cu:201-234): the image (the
// texture object's texels) and the Distribution2D tables (PiecewiseConstant.cu:8-22, 65-96), rows
// back to back.  n = 0: no such light.
struct EnvView {
    const float *img;           // [h][w][4]
    const float *cond_value;    // [h][w]      condition_w[i].value
    const float *cond_cdf;      // [h][w + 1]  condition_w[i].cdf
    const float *cond_funint;   // [h]         condition_w[i].funInt
    const float *marg_value;    // [h]         hmargin->value
    const float *marg_cdf;      // [h + 1]
    float marg_funint;
    int w, h, n;
};

// Mesh textures (Model.cu:304-319, Texture/Texture.cu:135-186).  One TexRec per image bound with
// Mesh::initTexture: its mip chain (levels = getMipMapLevels, Texture.cu:2-13; level l + 1 from level l
// by d_mipmap, :14-45) lies in TexView::data, level l at float4 offset off[l].  uv = Triangle::vText in
// the reference's traversal order; mesh_tex[j] = the image Mesh j bound, -1 = none (H5).  n = 0: no
// textures in the scene.
constexpr int kMaxTexLevels = 16;    // images up to 32768 texels a side
struct TexRec { uint32_t off[kMaxTexLevels]; int32_t w, h, levels, pad; };
struct TriUv { float u0, v0, u1, v1, u2, v2; };
struct TexView {
    const TexRec *recs;
    const float *data;
    const int32_t *mesh_tex;
    const TriUv *uv;
    int n, n_tri_models;
};

// Camera after its constructor (Camera/Camera.cu:4-14); pers/raster2world are dead.
struct CameraRec {
    v3 pos, up, front, right;
    int32_t res_x, res_y;
    float n;
    float aspect;      // (float)res_x / res_y as Camera::generateRay computes it per ray (Camera.cu:19): a ray-independent
                       // IEEE division, done once at rt_set_camera -- the same bits
};

} // namespace rt
