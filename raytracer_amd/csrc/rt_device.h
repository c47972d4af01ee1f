// rt_device.h -- per-ray device functions of the render path (gfx950).
// Each function names the reference code it restates (paths relative to
// src/RayTracer/).  Ray-independent sub-expressions are NOT recomputed here:
// they arrive precomputed in TriRec / ModelRec / SphereRec (rt_math.h) with the
// reference's own operation order, so the bits are the same.
#pragma once

#include "rt_math.h"

namespace rt {

struct Ray { v3 o, d; };

// Ray::Ray(o, r): direction = normalize(r)   (Ray/Ray.cu:2)
__device__ __forceinline__ Ray make_ray(v3 o, v3 r) { Ray q; q.o = o; q.d = normalize(r); return q; }
__device__ __forceinline__ v3 ray_pos(const Ray &r, float t) { return add(r.o, scale(r.d, t)); }  // Ray.cu:4-7

// ------------------------------------------------------------------ RNG (H7)
// Philox4x32-10 as cuRAND's curandStatePhilox4_32_10_t generates it
// (core/PathTracer.cuh:4; call sites PathTracer.cu:27, Scene.cu:87).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int round = 0; round < 10; round++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// curand_init(seed = idx + 332, subsequence = idx, offset = 0) (kernel.cu:46-47) then
// the `draw`-th curand_uniform4: x * 2^-32 + 2^-33.
__device__ __forceinline__ void pixel_uniform4(uint32_t pixel, uint32_t draw, float u[4])
{
    uint64_t seed = (uint64_t)pixel + 332u;
    uint32_t r[4];
    philox4x32_10(draw, 0u, pixel, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int i = 0; i < 4; i++)
        u[i] = (float)r[i] * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// ------------------------------------------------------------------ sinf / cosf
// material.cu:46 calls the platform libm; the build fixes both as the float
// rounding of a double Taylor evaluation (IEEE mul/add only), identical on the
// CPU oracle and here.  Argument range in sampleHemi: (-pi/4, 3pi/4].
__device__ __forceinline__ double sin_poly(double x)
{
    double x2 = x * x;
    double p = -7.6471637318198164759e-13;
    p = p * x2 + 1.6059043836821614599e-10;
    p = p * x2 + -2.5052108385441718775e-08;
    p = p * x2 + 2.7557319223985890653e-06;
    p = p * x2 + -1.9841269841269841270e-04;
    p = p * x2 + 8.3333333333333333333e-03;
    p = p * x2 + -1.6666666666666666667e-01;
    return x + (x * x2) * p;
}
__device__ __forceinline__ double cos_poly(double x)
{
    double x2 = x * x;
    double p = 4.7794773323873852974e-14;
    p = p * x2 + -1.1470745597729724714e-11;
    p = p * x2 + 2.0876756987868098979e-09;
    p = p * x2 + -2.7557319223985890653e-07;
    p = p * x2 + 2.4801587301587301587e-05;
    p = p * x2 + -1.3888888888888888889e-03;
    p = p * x2 + 4.1666666666666666667e-02;
    p = p * x2 + -0.5;
    return 1.0 + x2 * p;
}
__device__ __forceinline__ float det_sinf(float x)
{
    double xd = (double)x;
    if (xd > 0.78539816339744830962)
        return (float)cos_poly(xd - 1.5707963267948966192);
    return (float)sin_poly(xd);
}
__device__ __forceinline__ float det_cosf(float x)
{
    double xd = (double)x;
    if (xd > 0.78539816339744830962)
        return (float)(-sin_poly(xd - 1.5707963267948966192));
    return (float)cos_poly(xd);
}

// ------------------------------------------------------------------ camera
// Camera::generateRay (Camera/Camera.cu:16-24)
__device__ __forceinline__ Ray camera_ray(const CameraRec &c, float x, float y)
{
    const float step = 0.25f;
    float aspect = c.aspect;                                           // (float)res_x / res_y (Camera.cu:19), rt_upload.cpp
    v3 dir = add(add(scale(c.front, c.n),
                     scale(scale(scale(c.right, (float)x / c.res_x), step), aspect)),
                 scale(scale(c.up, (float)y / c.res_y), step));
    return make_ray(c.pos, normalize(dir));
}

// ------------------------------------------------------------------ triangle
// Triangle::hit (Model/Model.cu:91-137) for one (ray, triangle) pair, against the
// running closest t.  The acceptance conditions are a pure conjunction, so the
// cheap window test of :120 is evaluated before the area test of :106-113.
// Returns true and the new t when the reference would update colideRec.
__device__ __forceinline__ bool tri_hit(const TriRec &T, const v3 &o, const v3 &d, float best,
                                        float &t_out)
{
    float dnd = T.nx * d.x + T.ny * d.y + T.nz * d.z;                 // :98
    if (fabsf(dnd) < kEps)                                             // :99
        return false;
    float t = (-(o.x * T.nx + o.y * T.ny + o.z * T.nz) + T.dn) / dnd;  // :102
    if (!(t > kEps && t < best))                                       // :120
        return false;
    float px = o.x + t * d.x, py = o.y + t * d.y, pz = o.z + t * d.z;  // :104
    float ax = px - T.ax, ay = py - T.ay, az = pz - T.az;
    float bx = px - T.bx, by = py - T.by, bz = pz - T.bz;
    float cx = px - T.cx, cy = py - T.cy, cz = pz - T.cz;
    v3 c1 = cross(mk3(ax, ay, az), mk3(bx, by, bz));                   // :107
    v3 c2 = cross(mk3(cx, cy, cz), mk3(ax, ay, az));                   // :108
    v3 c3 = cross(mk3(cx, cy, cz), mk3(bx, by, bz));                   // :109
    float s1 = length(c1), s2 = length(c2), s3 = length(c3);
    if (fabsf(s1 + s2 + s3 - T.S) > 0.001f)                            // :113
        return false;
    t_out = t;
    return true;
}

// ------------------------------------------------------------------ ray pairs
// Two rays per lane share every wave-uniform operand.  Their arithmetic is written as plain fp32 instructions, one
// per ray: a v_pk_*_f32 occupies the SIMD for four cycles like its two halves (tools/valu_microbench.hip), and with
// packed ops in the stream plain ops no longer issue beside the half-rate ones (v_max / v_min / v_cmp /
// v_cndmask); without them they do.  Measured at the round's end: C3 -2.2 %, C4 -1.5 %, C2 -1.1 %, C5 -0.3 %
// of the step against the packed form (profiles/r03_ab_unpacked.txt), same bits.  The packed form -- ext_vector
// float2 arithmetic, v_pk_{mul,add,fma}_f32 -- stays as the A/B build: tools/build_variant.sh packed -DRT_PACKED
// -fslp-vectorize (the product is compiled with -fno-slp-vectorize so that the compiler does not pair them up again).
#ifdef RT_PACKED
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2 { float x, y; };
__device__ __forceinline__ f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 operator+(f2 a, float b) { return f2{a.x + b, a.y + b}; }
__device__ __forceinline__ f2 operator-(f2 a, float b) { return f2{a.x - b, a.y - b}; }
__device__ __forceinline__ f2 operator*(f2 a, float b) { return f2{a.x * b, a.y * b}; }
__device__ __forceinline__ f2 operator*(float a, f2 b) { return f2{a * b.x, a * b.y}; }
__device__ __forceinline__ f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#endif

struct RayPair {
    f2 ox, oy, oz, dx, dy, dz;
    f2 ix, iy, iz;    // 1 / d per axis (|d| clamped away from 0) for the box tests only;
                      // 0 for a ray that is never culled ("wild": origin outside the culling
                      // extent or not finite): every plane parameter is then 0 and every box passes
    f2 px, py, pz;    // o * (1 / d), so that a plane parameter is one fma: lo * (1/d) - o * (1/d)
    f2 best;          // running closest t per ray (IntersectRecord::t)
    f2 bestp;         // best * kBestInfl: conservative side of every pre-test
    f2 stop;          // shadow rays: a hit with t below this ends the ray (-1: never)
    int idx, idy;     // reference traversal index of the closest primitive, -1 = none
    int tidx, tidy;   // traverse<.., LIGHTS> only: the closest TRIANGLE before quadrics and lights were
                      // tested -- the mesh whose texture colour the record carries (Model.cu:301-310)
    uint32_t oct;     // wave-uniform.  bit 3: every culled ray of the wave runs the same way along
                      // every axis (they share an octant); bits 0-2: towards -x, -y, -z
};

constexpr float kBestInfl = 1.000004f;
constexpr float kEpsDefl = kEps * 0.999999f;

// 1 / d for the slab test.  A component below 1e-30 in magnitude is replaced by +-1e-30:
// over the t range of the path (t < 1e6) that moves the ray by less than 1e-24, far
// inside the slack the boxes carry.
// The hardware reciprocal (1 ulp) is enough: both planes of an axis and the ray's origin are scaled by
// the same value, so a box's interval only stretches by 1 +- 2^-23 as a whole, and its comparison with
// the closest hit so far carries the factor kBestInfl = 1 + 4e-6.
__device__ __forceinline__ float slab_inv(float d)
{
    float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}

// Conservative slab test of one box against both rays: true = some triangle below the
// node may still be accepted.  [tn, tf] is the parameter interval of o + t d inside
// the box; the node is skipped when it is empty, ends behind the origin, or starts
// beyond the closest hit so far: tn > min(tf, best) * (1 + 4e-6).  The relative factor
// covers the rounding of the comparison's two sides; the ABSOLUTE error of a plane
// parameter computed as fma(lo, 1/d, -(o * 1/d)) is at most 2^-23 |o| |1/d|, i.e. the
// plane is misplaced by at most 2^-23 |o| <= 2^-23 E in space, which the boxes' slack of
// 2^-19 E (rt_upload.cpp) covers sixteen times over.  A NaN (0 * inf, inf - inf) drops
// out of v_min/v_max, which only widens the interval.  `ex`, `ey` return the entry
// parameter, which orders the traversal.
__device__ __forceinline__ void node_test(const RayPair &r, float lx, float ly, float lz, float hx,
                                          float hy, float hz, bool &px, bool &py, float &ex,
                                          float &ey)
{
    f2 ax = f2_fma(f2{lx, lx}, r.ix, -r.px), bx = f2_fma(f2{hx, hx}, r.ix, -r.px);
    f2 ay = f2_fma(f2{ly, ly}, r.iy, -r.py), by = f2_fma(f2{hy, hy}, r.iy, -r.py);
    f2 az = f2_fma(f2{lz, lz}, r.iz, -r.pz), bz = f2_fma(f2{hz, hz}, r.iz, -r.pz);
    float tnx = fmaxf(fmaxf(fminf(ax.x, bx.x), fminf(ay.x, by.x)), fmaxf(fminf(az.x, bz.x), 0.0f));
    float tny = fmaxf(fmaxf(fminf(ax.y, bx.y), fminf(ay.y, by.y)), fmaxf(fminf(az.y, bz.y), 0.0f));
    float tfx = fminf(fminf(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x)), fmaxf(az.x, bz.x));
    float tfy = fminf(fminf(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y)), fmaxf(az.y, bz.y));
    tfx = fminf(tfx * kBestInfl, r.bestp.x);
    tfy = fminf(tfy * kBestInfl, r.bestp.y);
    px = !(tnx > tfx);
    py = !(tny > tfy);
    ex = tnx;
    ey = tny;
}

// The same test when every culled ray of the wave shares an octant (RayPair::oct): which of a
// box's two planes per axis a ray enters through is then known for the whole wave, so the planes
// arrive already ordered -- n* the entry side, f* the exit side -- and the six min / max per ray
// that sort them go away.  With 1/d > 0, fma(lo, 1/d, -o/d) <= fma(hi, 1/d, -o/d) (rounding is
// monotone), so the near / far values are the ones node_test's min / max pick, bit for bit; rays
// that are never culled (1/d = 0) and parked rays get the same value from both planes of an axis.
__device__ __forceinline__ void node_test_sorted(const RayPair &r, float nx, float ny, float nz,
                                                 float fx, float fy, float fz, bool &px, bool &py,
                                                 float &ex, float &ey)
{
    f2 ax = f2_fma(f2{nx, nx}, r.ix, -r.px), bx = f2_fma(f2{fx, fx}, r.ix, -r.px);
    f2 ay = f2_fma(f2{ny, ny}, r.iy, -r.py), by = f2_fma(f2{fy, fy}, r.iy, -r.py);
    f2 az = f2_fma(f2{nz, nz}, r.iz, -r.pz), bz = f2_fma(f2{fz, fz}, r.iz, -r.pz);
    float tnx = fmaxf(fmaxf(ax.x, ay.x), fmaxf(az.x, 0.0f));
    float tny = fmaxf(fmaxf(ax.y, ay.y), fmaxf(az.y, 0.0f));
    float tfx = fminf(fminf(bx.x, by.x), bz.x);
    float tfy = fminf(fminf(bx.y, by.y), bz.y);
    tfx = fminf(tfx * kBestInfl, r.bestp.x);
    tfy = fminf(tfy * kBestInfl, r.bestp.y);
    px = !(tnx > tfx);
    py = !(tny > tfy);
    ex = tnx;
    ey = tny;
}

// Triangle::hit (Model/Model.cu:91-137) for both rays of a pair against one
// (wave-uniform) triangle.  The acceptance conditions are a pure conjunction; two of
// them get a cheap conservative pre-test so that the IEEE divide and the three IEEE
// square roots run only for rays that can still pass:
//   window  t in (eps, best): tested as (num - eps' dnd)(best' dnd - num) >= 0 first
//   area    s1 + s2 + s3 - S <= 0.001: impossible once one |cross|^2 exceeds T.qmax
// Exact ties in t go to the lower reference index (H8: the reference, scanning in
// index order with strict <, keeps the first).
__device__ __forceinline__ void tri_test_pair(const TriRec &T, RayPair &r, bool cx, bool cy)
{
    f2 dnd = T.nx * r.dx + T.ny * r.dy + T.nz * r.dz;                         // :98
    cx = cx & !(fabsf(dnd.x) < kEps);                                          // :99
    cy = cy & !(fabsf(dnd.y) < kEps);
    f2 num = -(r.ox * T.nx + r.oy * T.ny + r.oz * T.nz) + T.dn;               // :102 numerator
    f2 u = num - kEpsDefl * dnd, w = r.bestp * dnd - num;
    f2 uw = u * w;
    cx = cx & !(uw.x < 0.0f);
    cy = cy & !(uw.y < 0.0f);
    if (!(cx | cy))
        return;
    f2 t;
    t.x = num.x / dnd.x;                                                       // :102
    t.y = num.y / dnd.y;
    cx = cx & (t.x > kEps) & ((t.x < r.best.x) | ((t.x == r.best.x) & (T.orig < r.idx)));   // :120
    cy = cy & (t.y > kEps) & ((t.y < r.best.y) | ((t.y == r.best.y) & (T.orig < r.idy)));
    if (!(cx | cy))
        return;
    f2 px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;       // :104
    f2 ax = px - T.ax, ay = py - T.ay, az = pz - T.az;
    f2 bx = px - T.bx, by = py - T.by, bz = pz - T.bz;
    f2 ex = px - T.cx, ey = py - T.cy, ez = pz - T.cz;
    // cross(p - a, p - b), cross(p - c, p - a), cross(p - c, p - b)             :107-109
    f2 c1x = ay * bz - az * by, c1y = az * bx - ax * bz, c1z = ax * by - ay * bx;
    f2 c2x = ey * az - ez * ay, c2y = ez * ax - ex * az, c2z = ex * ay - ey * ax;
    f2 c3x = ey * bz - ez * by, c3y = ez * bx - ex * bz, c3z = ex * by - ey * bx;
    f2 q1 = c1x * c1x + c1y * c1y + c1z * c1z;
    f2 q2 = c2x * c2x + c2y * c2y + c2z * c2z;
    f2 q3 = c3x * c3x + c3y * c3y + c3z * c3z;
    cx = cx & !((q1.x > T.qmax) | (q2.x > T.qmax) | (q3.x > T.qmax));
    cy = cy & !((q1.y > T.qmax) | (q2.y > T.qmax) | (q3.y > T.qmax));
    if (!(cx | cy))
        return;
    // The area test of :113 compares a sum of three correctly rounded square roots with 0.001.  Its verdict is
    // almost never close: inside the triangle the sum equals S up to rounding, outside it exceeds S by twice the
    // area the point adds.  So the hardware square root (v_sqrt_f32, 1 ulp, a sixth of the correctly rounded
    // expansion's issue time) decides whenever its answer is further from 0.001 than every way the two computations
    // can differ -- 1.5 ulp per root and one rounding per addition, bounded by 16 * 2^-23 (s1 + s2 + s3 + S), plus a
    // floor for roots of denormal squares the instruction may flush -- and only the remaining pairs take the
    // reference's arithmetic.  NaN and infinity compare false on both sure sides and take it too.  Same verdicts.
    bool ax_ = false, ay_ = false, ux = false, uy = false;                        // accepted / undecided, per ray
    if (cx) {
        const float s1 = __builtin_amdgcn_sqrtf(q1.x), s2 = __builtin_amdgcn_sqrtf(q2.x), s3 = __builtin_amdgcn_sqrtf(q3.x);
        const float sum = s1 + s2 + s3, df = fabsf(sum - T.S), m = (sum + T.S) * 1.9073486e-6f + 1e-12f;
        ax_ = df < 0.001f - m;
        ux = !ax_ & !(df > 0.001f + m);
    }
    if (cy) {
        const float s1 = __builtin_amdgcn_sqrtf(q1.y), s2 = __builtin_amdgcn_sqrtf(q2.y), s3 = __builtin_amdgcn_sqrtf(q3.y);
        const float sum = s1 + s2 + s3, df = fabsf(sum - T.S), m = (sum + T.S) * 1.9073486e-6f + 1e-12f;
        ay_ = df < 0.001f - m;
        uy = !ay_ & !(df > 0.001f + m);
    }
    if (ux | uy) {
        if (ux) {
            float s1 = sqrtf(q1.x), s2 = sqrtf(q2.x), s3 = sqrtf(q3.x);
            ax_ = !(fabsf(s1 + s2 + s3 - T.S) > 0.001f);                          // :113
        }
        if (uy) {
            float s1 = sqrtf(q1.y), s2 = sqrtf(q2.y), s3 = sqrtf(q3.y);
            ay_ = !(fabsf(s1 + s2 + s3 - T.S) > 0.001f);
        }
    }
    if (ax_) {
        r.best.x = t.x;
        r.bestp.x = t.x * kBestInfl;
        r.idx = T.orig;
    }
    if (ay_) {
        r.best.y = t.y;
        r.bestp.y = t.y * kBestInfl;
        r.idy = T.orig;
    }
}

// ------------------------------------------------------------------ sphere
// Quadratic::hit (Model/Model.cu:349-410) up to the acceptance test :396.
__device__ __forceinline__ bool sphere_hit(const SphereRec &S, const v3 &o, const v3 &d, float best,
                                           float &t_out)
{
    if (S.type != 0)
        return false;
    v3 Lo = apply3(S.inv, o);                                          // :353
    v3 Ld = normalize(apply_dir(S.inv, d));                            // :355-356
    v3 oc = sub(Lo, mk3(0.0f, 0.0f, 0.0f));                            // :357
    float dotOCD = dot(Ld, oc);                                        // :358
    if (dotOCD > 0)                                                    // :360
        return false;
    float dotOC = dot(oc, oc);
    float disc = dotOCD * dotOCD - dotOC + S.rad2;                     // :364
    float t0, t1;
    if (disc < 0)
        return false;
    else if (disc < kEps)
        t0 = t1 = -dotOCD;
    else {
        disc = sqrtf(disc);
        t0 = -dotOCD - disc;
        t1 = -dotOCD + disc;
        if (t0 < kEps)
            t0 = t1;
    }
    if (t0 > kEps && t0 < best) {                                      // :396
        t_out = t0;
        return true;
    }
    return false;
}

// ------------------------------------------------------------------ hit record
// The fields of IntersectRecord (Ray/IntersectionRecord.cuh:18-36) the path reads.
struct Hit {
    v3 pos, normal, tangent;
    v3 color;      // rec.color; (-1,-1,-1) = "use the BRDF's colour" (material.cu:54-55, H5)
    int model;
};

// Winner-only part of Triangle::hit (Model.cu:116-132): pure functions of
// (triangle, ray, t).
__device__ __forceinline__ void fill_tri_hit(const TriRec &T, const TriCold &C, const ModelRec &M,
                                             const Ray &r, Hit &h)
{
    const v3 &o = r.o, &d = r.d;
    float dnd = T.nx * d.x + T.ny * d.y + T.nz * d.z;
    float t = (-(o.x * T.nx + o.y * T.ny + o.z * T.nz) + T.dn) / dnd;
    v3 p = ray_pos(r, t);
    v3 ta = mk3(T.ax, T.ay, T.az), tb = mk3(T.bx, T.by, T.bz), tc = mk3(T.cx, T.cy, T.cz);
    float s2 = length(cross(sub(p, tc), sub(p, ta)));
    float s3 = length(cross(sub(p, tc), sub(p, tb)));
    float m1 = s3 / T.S, m2 = s2 / T.S, m3 = 1.0f - m1 - m2;           // :116
    v3 n0 = mk3(C.n0[0], C.n0[1], C.n0[2]), n1 = mk3(C.n1[0], C.n1[1], C.n1[2]),
       n2 = mk3(C.n2[0], C.n2[1], C.n2[2]);
    v3 n = normalize(add(add(scale(n0, m1), scale(n1, m2)), scale(n2, m3)));     // :125
    v4 tmp = normalize(apply4(M.cas, mk4(n.x, n.y, n.z, 0.0f)));       // :126-127 (float4 normalize)
    h.normal = mk3(tmp.x, tmp.y, tmp.z);                               // :128
    h.pos = ray_pos(r, t);                                             // :129
    h.tangent = normalize(cross(h.normal, mk3((float)0.3, (float)0.4, (float)-0.5)));  // :131
    h.color = mk3(-1.0f, -1.0f, -1.0f);                                // H5: unbound mesh texture
    h.model = C.model;
}

// Winner-only part of Quadratic::hit (Model.cu:382-404).
__device__ __forceinline__ void fill_sphere_hit(const SphereRec &S, const Ray &r, Hit &h)
{
    v3 Lo = apply3(S.inv, r.o);
    v3 Ld = normalize(apply_dir(S.inv, r.d));
    v3 oc = sub(Lo, mk3(0.0f, 0.0f, 0.0f));
    float dotOCD = dot(Ld, oc);
    float dotOC = dot(oc, oc);
    float disc = dotOCD * dotOCD - dotOC + S.rad2;
    float t0, t1;
    if (disc < kEps)
        t0 = t1 = -dotOCD;
    else {
        disc = sqrtf(disc);
        t0 = -dotOCD - disc;
        t1 = -dotOCD + disc;
        if (t0 < kEps)
            t0 = t1;
    }
    v3 pos = add(scale(Ld, t0), Lo);                                   // :382
    v3 normal = normalize(apply_dir(S.M, pos));                        // :383
    v3 tangent;
    if (pos.x == 0.0f && pos.y == 0.0f)                                // :385-386
        tangent = apply_dir(S.M, mk3(0.0f, 0.0f, -1.0f));
    else                                                               // :387-391, UP_VEC :2
        tangent = cross(normal, cross(apply_dir(S.M, sub(mk3(0.0f, S.rad, 0.0f), pos)), normal));
    tangent = normalize(tangent);                                      // :393
    h.pos = apply3(S.M, pos);                                          // :394
    h.normal = normal;
    h.tangent = normalize(tangent);                                    // :404
    h.color = mk3(-1.0f, -1.0f, -1.0f);
    h.model = S.model;
}

// Winner-only part of TriangleLight::hit (Light/Light.cu:90-100) for the accepted t.  The
// record's normal is the unnormalised cross product of Light.cu:72 (it shadows the member),
// flipped towards the ray for two-sided lights (:97).
__device__ __forceinline__ void fill_light_hit(const TriLightRec &L, const Ray &r, float t, Hit &h)
{
    const TriRec &T = L.T;
    v3 p = ray_pos(r, t);                                              // :79
    v3 ta = mk3(T.ax, T.ay, T.az), tb = mk3(T.bx, T.by, T.bz), tc = mk3(T.cx, T.cy, T.cz);
    float s2 = length(cross(sub(p, tc), sub(p, ta)));                  // :83
    float s3 = length(cross(sub(p, tc), sub(p, tb)));                  // :84
    float m1 = s3 / T.S, m2 = s2 / T.S, m3 = 1.0f - m1 - m2;           // :90
    v3 n = mk3(T.nx, T.ny, T.nz);
    h.normal = L.two_side ? (dot(n, r.d) > 0 ? neg(n) : n) : n;        // :97
    h.pos = ray_pos(r, t);                                             // :98
    h.tangent = normalize(add(scale(mk3(L.e1[0], L.e1[1], L.e1[2]), m2),
                              scale(mk3(L.e2[0], L.e2[1], L.e2[2]), m3)));   // :100
    h.color = mk3(-1.0f, -1.0f, -1.0f);
    h.model = L.model;
}
// TriangleLight::PDF (Light/Light.cuh:81-85) and ::L (:86-91)
__device__ __forceinline__ float light_pdf(const TriLightRec &L, float rec_t, v3 wi)
{
    const float c = fabsf(dot(mk3(L.normal[0], L.normal[1], L.normal[2]), wi));
    if (c < 0.001f)
        return 0.0f;
    return rec_t * rec_t / (L.area * c);
}
__device__ __forceinline__ v3 light_L(const TriLightRec &L, v3 r)
{
    const v3 illum = mk3(L.illum[0], L.illum[1], L.illum[2]);
    if (L.two_side)
        return illum;
    return dot(r, mk3(L.normal[0], L.normal[1], L.normal[2])) > 0 ? illum : mk3(0.0f, 0.0f, 0.0f);
}

// World-space position the reference stores in rec.pos for a hit (triangle:
// Model.cu:129; sphere: Model.cu:382,394) -- what the shadow test reads (Scene.cu:227).
__device__ __forceinline__ v3 hit_pos_sphere(const SphereRec &S, const Ray &r, float t0)
{
    v3 Lo = apply3(S.inv, r.o);
    v3 Ld = normalize(apply_dir(S.inv, r.d));
    return apply3(S.M, add(scale(Ld, t0), Lo));
}

// ------------------------------------------------------------------ spawnRay
// offsetFromPoint (Ray/IntersectionRecord.cu:34-49).  findFloatUp/Down
// (:2-32) leave the value unchanged (`*t++` moves the pointer) except that a
// zero comes back as -0.0f through findFloatDown (:22-23).
__device__ __forceinline__ float nudge(float v)
{
    if (v > 0.0f)
        return v;                       // findFloatUp: v > 0 is never -0.0f
    if (v == 0.0f)
        return -0.0f;                   // findFloatDown(+-0) -> -0.0f
    return v;
}
__device__ __forceinline__ v3 offset_from_point(v3 origin, v3 normal, v3 eb, v3 d)
{
    v3 n = mk3(fabsf(normal.x), fabsf(normal.y), fabsf(normal.z));
    v3 offset = scale(normal, dot(n, eb));
    if (dot(d, normal) < 0)
        offset = neg(offset);
    v3 po = add(origin, offset);
    return mk3(nudge(po.x), nudge(po.y), nudge(po.z));
}
// IntersectRecord::spawnRay (Ray/IntersectionRecord.cuh:38-42)
__device__ __forceinline__ Ray spawn_ray(const Hit &h, v3 w)
{
    return make_ray(offset_from_point(h.pos, h.normal, mk3(0.01f, 0.01f, 0.01f), w), w);
}

// ------------------------------------------------------------------ material
// Material::world2Local / local2World (Material/Object.cu:107-115)
__device__ __forceinline__ float local_y(const Hit &h, v3 w) { return dot(w, h.normal); }
__device__ __forceinline__ v3 local2world(const Hit &h, v3 l)
{
    return add(add(scale(h.tangent, l.x), scale(h.normal, l.y)),
               scale(normalize(cross(h.tangent, h.normal)), l.z));
}
// Lambertian::f (Material/material.cu:52-57) on the local y components.
__device__ __forceinline__ v3 lambert_f(float wo_y, float wi_y, v3 c, const ModelRec &M)
{
    if (c.x < 0.0f)
        c = mk3(M.color[0], M.color[1], M.color[2]);
    return (wo_y > 0.0f && wi_y > 0.0f) ? scale(c, kInvPi) : mk3(0.0f, 0.0f, 0.0f);
}
// Material::f (Material/Object.cu:3-32); only LAMBERTIAN is in scope, every other
// type evaluates to BLACK here (rt_upload_scene rejects them up front).
__device__ __forceinline__ v3 rec_f(const Hit &h, const ModelRec &M, v3 wo, v3 wi)
{
    if (M.mat_type != kMatLambertian)
        return mk3(0.0f, 0.0f, 0.0f);
    return lambert_f(local_y(h, wo), local_y(h, wi), h.color, M);
}
// The same with the outgoing direction already reduced to its local y component
// (wo_y = dot(wo, normal)), which is all Lambertian::f reads of it.
__device__ __forceinline__ v3 rec_f_y(const Hit &h, const ModelRec &M, float wo_y, v3 wi)
{
    if (M.mat_type != kMatLambertian)
        return mk3(0.0f, 0.0f, 0.0f);
    return lambert_f(wo_y, local_y(h, wi), h.color, M);
}
// sampleHemi (Material/material.cu:28-50)
__device__ __forceinline__ v3 sample_hemi(float sx, float sy)
{
    float rx = 2.0f * sx - 1.0f, ry = 2.0f * sy - 1.0f;
    if (rx == 0.0f && ry == 0.0f)
        return mk3(0.0f, 0.0f, 0.0f);
    float r, theta;
    if (fabsf(rx) > fabsf(ry)) {
        r = rx;
        theta = kPiOver4 * (ry / rx);
    } else {
        r = ry;
        theta = kPiOver2 - kPiOver4 * (rx / ry);
    }
    float dx = r * det_sinf(theta), dy = r * det_cosf(theta);
    float z = sqrtf(fmaxf(0.0f, 1 - dx * dx - dy * dy));
    return mk3(dx, z, dy);
}
// Material::sample_f -> Lambertian::sample_f (Object.cu:34-72, material.cu:59-82)
__device__ __forceinline__ v3 rec_sample_f(const Hit &h, const ModelRec &M, v3 wo, v3 &wi,
                                           float &pdf, float sx, float sy)
{
    v3 res = mk3(0.0f, 0.0f, 0.0f);
    v3 wl = mk3(0.0f, 0.0f, 0.0f);
    pdf = 0.0f;
    if (M.mat_type == kMatLambertian) {
        float wo_y = local_y(h, wo);
        wl = sample_hemi(sx, sy);
        pdf = wl.y > +0.0f ? wl.y * kInvPi : 0.0f;
        res = lambert_f(wo_y, wl.y, h.color, M);
    }
    wi = local2world(h, wl);
    return res;
}

__device__ __forceinline__ v3 rec_sample_f_y(const Hit &h, const ModelRec &M, float wo_y, v3 &wi,
                                             float &pdf, float sx, float sy)
{
    v3 res = mk3(0.0f, 0.0f, 0.0f);
    v3 wl = mk3(0.0f, 0.0f, 0.0f);
    pdf = 0.0f;
    if (M.mat_type == kMatLambertian) {
        wl = sample_hemi(sx, sy);
        pdf = wl.y > +0.0f ? wl.y * kInvPi : 0.0f;
        res = lambert_f(wo_y, wl.y, h.color, M);
    }
    wi = local2world(h, wl);
    return res;
}

// ------------------------------------------------------------------ every BRDF (general kernel)
// acosf / sinf / cosf / logf / powf of material.cu:84-254 are not bit-specified; the build fixes
// each as the float rounding of the double-precision function, here and in the oracle alike.
__device__ __forceinline__ float m_acosf(float x) { return (float)acos((double)x); }
__device__ __forceinline__ float m_sinf(float x) { return (float)sin((double)x); }
__device__ __forceinline__ float m_cosf(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float m_powf(float x, float y) { return (float)pow((double)x, (double)y); }

__device__ __forceinline__ v3 world2local(const Hit &h, v3 w)                     // Object.cu:107-110
{
    return mk3(dot(w, h.tangent), dot(w, h.normal), dot(w, normalize(cross(h.tangent, h.normal))));
}
// GGX::SchlickFresnel (material.cu:85-90) = Cook_Torrance::F (material.cuh:96-101)
__device__ __forceinline__ v3 schlick(v3 r0, float r)
{
    float t = (1.0f - r);
    v3 a = mk3(1.0f - r0.x, 1.0f - r0.y, 1.0f - r0.z);
    a = scale(scale(scale(a, t * t), t * t), t);
    return add(r0, a);
}
// GGX::G (material.cu:92-98) = Cook_Torrance::G (:192-198)
__device__ __forceinline__ float smith_g(float a2, v3 wi, v3 wo)
{
    float A = wo.y * sqrtf(a2 + (1.0f - a2) * wi.y * wi.y);
    float B = wi.y * sqrtf(a2 + (1.0f - a2) * wo.y * wo.y);
    return 2.0f * wo.y * wi.y / (A + B);
}
// GGX::D (material.cu:100-103; M_PI makes the expression double)
__device__ __forceinline__ float ggx_d(float a2, v3 wh)
{
    float k = wh.y * wh.y * (a2 - 1) + 1;
    return (float)((double)a2 / (3.14159265358979323846 * (double)k * (double)k));
}
__device__ __forceinline__ v3 ggx_f(float a2, v3 wo, v3 wi, v3 c)                // material.cu:105-114
{
    float dotwin_won = fmaxf(0.0001f, fabsf(4.0f * wi.y * wo.y));
    v3 wh = normalize(add(wo, wi));
    return divs(scale(scale(schlick(c, dot(wi, wh)), ggx_d(a2, wh)), smith_g(a2, wi, wo)), dotwin_won);
}
__device__ __forceinline__ float ggx_pdf(float a2, v3 wo, v3 wi)                 // material.cu:132-137
{
    v3 wh = normalize(add(wo, wi));
    return wi.y > 0.0f && dot(wi, wh) > 0.0f ? ggx_d(a2, wh) * wh.y * 0.25f / wo.y : 0.0f;
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : hi < v ? hi : v; }
// Oren_Nayar::f (material.cu:140-170), sigma = 0.5; A, B, TanBeta are double expressions
// narrowed to float; SinPhiO divides by SinThetaI as written (:154)
__device__ __forceinline__ v3 oren_f(v3 wo, v3 wi, v3 c)
{
    float sigma = 0.5;
    float A = (float)(1 - 0.5 * sigma * sigma / (sigma * sigma + 0.33));
    float B = (float)(0.45 * sigma * sigma / (sigma * sigma + 0.09));
    float CosThetaI = wi.y, CosThetaO = wo.y;
    float SinThetaI = sqrtf(1 - CosThetaI * CosThetaI);
    float SinThetaO = sqrtf(1 - CosThetaO * CosThetaO);
    float CosPhiI = (SinThetaI == 0) ? 1 : clampf(wi.x / SinThetaI, -1, 1);
    float CosPhiO = (SinThetaO == 0) ? 0 : clampf(wo.x / SinThetaO, -1, 1);
    float SinPhiI = (SinThetaI == 0) ? 1 : clampf(wi.z / SinThetaI, -1, 1);
    float SinPhiO = (SinThetaO == 0) ? 0 : clampf(wo.z / SinThetaI, -1, 1);
    float dCos = CosPhiI * CosPhiO + SinPhiI * SinPhiO;
    float maxCos = (dCos > 0) ? dCos : 0;
    float SinAlpha, TanBeta;
    if (CosThetaI * CosThetaI > CosThetaO * CosThetaO) {
        SinAlpha = SinThetaO;
        TanBeta = (float)(SinThetaI * 1.0 / ((CosThetaI > 0) ? CosThetaI : (-1 * CosThetaI)));
    } else {
        SinAlpha = SinThetaI;
        TanBeta = (float)(SinThetaI * 1.0 / ((CosThetaO > 0) ? CosThetaO : (-1 * CosThetaO)));
    }
    return (wo.y > 0.0f && wi.y > 0.0f) ? scale(scale(c, kInvPi), A + B * maxCos * SinAlpha * TanBeta)
                                        : mk3(0.0f, 0.0f, 0.0f);
}
// Cook_Torrance::D (material.cu:241-245): pow(float, int) is double, pow(float, float) powf
__device__ __forceinline__ float cook_d(float m, v3 wo)
{
    float c = (float)(1.0 / ((double)(m * m) * pow((double)wo.y, 4.0)));
    return c * m_powf(2.718281828f, -(m_acosf(wo.y) * m_acosf(wo.y)) / (m * m));
}
__device__ __forceinline__ v3 cook_f(const MatRec &P, v3 wo, v3 wi, v3 c)        // material.cu:200-208
{
    float dotwin_won = fmaxf(0.0001f, fabsf(4.0f * wi.y * wo.y));
    v3 wh = normalize(add(wo, wi));
    return divs(scale(scale(schlick(c, dot(wi, wh)), cook_d(P.m, wh)), smith_g(P.a2, wi, wo)), dotwin_won);
}
__device__ __forceinline__ float cook_pdf(const MatRec &P, v3 wo, v3 wi)         // material.cu:246-254
{
    if (!(wo.y * wi.y > 0))
        return 0;
    v3 wh = normalize(add(wo, wi));
    if (dot(wo, wh) == 0.0f)
        return 0.0f;
    return cook_d(P.m, wh) / (4 * dot(wo, wh));
}
// Fresnel::sample_f (material.cuh:118-152)
__device__ __forceinline__ v3 fresnel_sample_f(float eta, v3 wo, v3 &wi, float &pdf, float sx, v3 c)
{
    bool entering = wo.y > 0.0f;
    float n1 = 1.0f, n2 = eta;
    float R = (n1 - n2) / (n1 + n2);
    R *= R;
    float t = fabsf(wo.y);
    float F = R + (t * t) * (t * t) * t * (1.0f - R);
    if (sx < F) {
        pdf = F;
        wi = mk3(-wo.x, wo.y, -wo.z);
        return divs(scale(c, F), fmaxf(fabsf(wi.y), 0.0000001f));
    }
    pdf = 1 - F;
    float etas;
    float sin2theta = fmaxf(0.001f, 1.0f - wo.y * wo.y);
    if (!entering)
        etas = 1.0f / eta;
    else
        etas = eta;
    if (sin2theta * etas * etas >= 1.0f)
        return mk3(0.0f, 0.0f, 0.0f);
    float costhetaT = sqrtf(1.0f - etas * etas * sin2theta);
    wi = add(scale(wo, -etas), scale(mk3(0.0f, 1.0f, 0.0f), etas * wo.y - costhetaT));
    wi = normalize(wi);
    return divs(scale(c, 1.0f - F), fmaxf(fabsf(wi.y), 0.0000001f));
}

__device__ __forceinline__ v3 brdf_color(const Hit &h, const ModelRec &M)
{
    return h.color.x < 0.0f ? mk3(M.color[0], M.color[1], M.color[2]) : h.color;   // `if (c.x < 0.0f) c = color`
}
// Material::f (Object.cu:3-32), world-space directions
__device__ __forceinline__ v3 brdf_f(const Hit &h, const ModelRec &M, const MatRec &P, v3 wo, v3 wi)
{
    const v3 lo = world2local(h, wo), li = world2local(h, wi), c = brdf_color(h, M);
    switch (M.mat_type) {
    case kMatLambertian: return (lo.y > 0.0f && li.y > 0.0f) ? scale(c, kInvPi) : mk3(0.0f, 0.0f, 0.0f);
    case kMatGGX: return ggx_f(P.a2, lo, li, c);
    case kMatCook: return cook_f(P, lo, li, c);
    case kMatOrenNayar: return oren_f(lo, li, c);
    default: return mk3(0.0f, 0.0f, 0.0f);                                       // FRESNEL: BLACK
    }
}
// Material::PDF (Object.cu:74-104)
__device__ __forceinline__ float brdf_pdf(const Hit &h, const ModelRec &M, const MatRec &P, v3 wo, v3 wi)
{
    const v3 lo = world2local(h, wo), li = world2local(h, wi);
    switch (M.mat_type) {
    case kMatLambertian:
    case kMatOrenNayar: return li.y > +0.0f ? li.y * kInvPi : 0.0f;
    case kMatGGX: return ggx_pdf(P.a2, lo, li);
    case kMatCook: return cook_pdf(P, lo, li);
    default: return 0.0f;
    }
}
// Material::sample_f (Object.cu:34-72).  wi is in/out: Cook_Torrance::sample_f overwrites its
// sampled half vector with normalize(wo + *wi), *wi being the caller's variable as it comes in
// (material.cu:229, hazard H9).
__device__ __forceinline__ v3 brdf_sample_f(const Hit &h, const ModelRec &M, const MatRec &P, v3 wo, v3 &wi,
                                            float &pdf, float sx, float sy)
{
    const v3 lo = world2local(h, wo), c = brdf_color(h, M);
    v3 res = mk3(0.0f, 0.0f, 0.0f);
    switch (M.mat_type) {
    case kMatLambertian:
        wi = sample_hemi(sx, sy);
        pdf = wi.y > +0.0f ? wi.y * kInvPi : 0.0f;
        res = (lo.y > 0.0f && wi.y > 0.0f) ? scale(c, kInvPi) : mk3(0.0f, 0.0f, 0.0f);
        break;
    case kMatOrenNayar:                                                          // material.cu:171-188
        wi = sample_hemi(sx, sy);
        pdf = wi.y > +0.0f ? wi.y * kInvPi : 0.0f;
        res = oren_f(lo, wi, c);
        break;
    case kMatFresnel:
        res = fresnel_sample_f(P.eta, lo, wi, pdf, sx, c);
        break;
    case kMatGGX: {                                                              // material.cu:116-130
        float theta = m_acosf(sqrtf((1.0f - sx) / ((P.a2 - 1.0f) * sx + 1.0f)));
        float phi = (float)((double)sy * 3.14159265358979323846 * (double)2.0f);
        v3 wh = mk3(m_sinf(theta) * m_sinf(phi), m_cosf(theta), m_cosf(phi) * m_sinf(theta));
        wi = sub(scale(wh, 2.0f * dot(lo, wh)), lo);
        pdf = ggx_pdf(P.a2, lo, wi);
        res = ggx_f(P.a2, lo, wi, c);
        break;
    }
    case kMatCook: {                                                             // material.cu:209-234
        v3 wh = normalize(add(lo, wi));                                          // :229
        wi = sub(scale(wh, 2.0f * dot(lo, wh)), lo);                             // :230
        pdf = cook_pdf(P, lo, wi);
        res = cook_f(P, lo, wi, c);
        break;
    }
    default: break;
    }
    wi = local2world(h, wi);
    return res;
}

// ------------------------------------------------------------------ EnvironmentLight (general kernel)
__device__ __forceinline__ float m_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
constexpr double kPiD = 3.14159265358979323846;
constexpr float kEnvInvPi = 0.3183098861837907f;                                  // Light/Light.cu:2-4
// Distribution::sample (PiecewiseConstant.cu:24-40): the search as written, and the return value is
// offset + fraction, not divided by n
__device__ __forceinline__ float dist1_sample(const float *value, const float *cdf, float funint, int n,
                                              float u, float &pdf, int &offset)
{
    int l = 0, r = n;
    while (r > l) {
        const int mid = l + (r - l) / 2;
        if (cdf[mid] > u)
            r = mid - 1;
        else
            l = mid + 1;
    }
    if (r == n)
        --r;
    if (r < 0)
        r = 0;
    offset = r;
    pdf = value[r] / funint;
    return cdf[r + 1] == cdf[r] ? (float)r : (float)r + (u - cdf[r]) / (cdf[r + 1] - cdf[r]);
}
// Distribution2D::sample (PiecewiseConstant.cu:108-118)
__device__ __forceinline__ void env_dist_sample(const EnvView &e, float ux, float uy, float &d0, float &d1, float &pdf)
{
    float p0, p1;
    int v;
    d1 = dist1_sample(e.marg_value, e.marg_cdf, e.marg_funint, e.h, uy, p1, v);
    d0 = dist1_sample(e.cond_value + (size_t)v * e.w, e.cond_cdf + (size_t)v * (e.w + 1), e.cond_funint[v], e.w, ux, p0, v);
    pdf = p0 * p1;
}
// Distribution2D::PDF (PiecewiseConstant.cu:98-106); H12: indices clamped to the arrays
__device__ __forceinline__ float env_dist_pdf(const EnvView &e, float sx, float sy)
{
    int i1 = (int)(e.w * sx), i2 = (int)(e.h * sy);
    i1 = min(max(i1, 0), e.w - 1);
    i2 = min(max(i2, 0), e.h - 1);
    return e.cond_value[(size_t)i2 * e.w + i1] / e.marg_funint;
}
// tex2D<float4>(img, x, y) under the texture object of Light.cu:224-232 (normalised coordinates,
// wrap, linear): hazard H11, the filter of the CUDA programming guide restated in software
__device__ __forceinline__ v3 env_tex2d(const EnvView &e, float x, float y)
{
    float xb = (x - floorf(x)) * (float)e.w - 0.5f, yb = (y - floorf(y)) * (float)e.h - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f, b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    int i0 = (int)fi, j0 = (int)fj;
    int i[2] = {((i0 % e.w) + e.w) % e.w, (((i0 + 1) % e.w) + e.w) % e.w};
    int j[2] = {((j0 % e.h) + e.h) % e.h, (((j0 + 1) % e.h) + e.h) % e.h};
    const float4 *img = (const float4 *)e.img;
    const float4 t00 = img[(size_t)j[0] * e.w + i[0]], t10 = img[(size_t)j[0] * e.w + i[1]];
    const float4 t01 = img[(size_t)j[1] * e.w + i[0]], t11 = img[(size_t)j[1] * e.w + i[1]];
    auto mix = [&](float c00, float c10, float c01, float c11) {
        return (1 - a) * (1 - b) * c00 + a * (1 - b) * c10 + (1 - a) * b * c01 + a * b * c11;
    };
    return mk3(mix(t00.x, t10.x, t01.x, t11.x), mix(t00.y, t10.y, t01.y, t11.y), mix(t00.z, t10.z, t01.z, t11.z));
}
// EnvironmentLight::L (Light.cu:180-199); getLe(r) = L(r.dir)
__device__ __forceinline__ v3 env_L(const EnvView &e, v3 r)
{
    v3 k = normalize(r);
    float phi = 0.0f;
    float theta = m_acosf(k.y);
    float sint = m_sinf(theta);
    if (sint > 0.001f) {
        phi = m_atan2f(r.x, r.z);
        if (r.z < 0.0f)
            phi = (float)((double)phi + kPiD);
        if (phi < 0.0f)
            phi = (float)((double)phi + (double)2.0f * kPiD);
    }
    return env_tex2d(e, phi * kEnvInvPi * 0.5f, theta * kEnvInvPi);
}
// EnvironmentLight::PDF (Light.cu:157-173)
__device__ __forceinline__ float env_pdf(const EnvView &e, v3 wi)
{
    v3 k = normalize(wi);
    float theta = m_acosf(k.y);
    float sint = m_sinf(theta);
    float phi = m_atan2f(k.x, k.z);
    if (sint < 0.001f)
        return 0.0f;
    if (wi.z < 0.0f)
        phi = (float)((double)phi + kPiD);
    if (phi < 0.0f)
        phi = (float)((double)phi + (double)2.0f * kPiD);
    return (float)((double)env_dist_pdf(e, phi * 0.5f * kEnvInvPi, theta * kEnvInvPi) /
                   ((double)2.0f * kPiD * kPiD * (double)sint));
}

// ------------------------------------------------------------------ mesh textures
// The texture object of Texture.cu:165-184 in software (DESIGN H13): normalised coordinates, clamp
// addressing, linear filtering within a level (the tex2D convention of H11 with edge clamping), linear
// between the two levels nearest lambda = log2(rho), rho from the gradients scaled to level 0's texels.
__device__ __forceinline__ float4 tex_fetch(const float4 *img, int w, int h, float x, float y)
{
    float xb = x * (float)w - 0.5f, yb = y * (float)h - 0.5f;
    if (!(xb > -1.0f)) xb = -1.0f;                                     // clamp addressing (and NaN)
    if (xb > (float)w) xb = (float)w;
    if (!(yb > -1.0f)) yb = -1.0f;
    if (yb > (float)h) yb = (float)h;
    float fi = floorf(xb), fj = floorf(yb);
    float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f, b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    int i0 = (int)fi, j0 = (int)fj;
    int i[2] = {min(max(i0, 0), w - 1), min(max(i0 + 1, 0), w - 1)};
    int j[2] = {min(max(j0, 0), h - 1), min(max(j0 + 1, 0), h - 1)};
    const float4 t00 = img[(size_t)j[0] * w + i[0]], t10 = img[(size_t)j[0] * w + i[1]];
    const float4 t01 = img[(size_t)j[1] * w + i[0]], t11 = img[(size_t)j[1] * w + i[1]];
    auto mix = [&](float c00, float c10, float c01, float c11) {
        return (1 - a) * (1 - b) * c00 + a * (1 - b) * c10 + (1 - a) * b * c01 + a * b * c11;
    };
    float4 r;
    r.x = mix(t00.x, t10.x, t01.x, t11.x); r.y = mix(t00.y, t10.y, t01.y, t11.y);
    r.z = mix(t00.z, t10.z, t01.z, t11.z); r.w = mix(t00.w, t10.w, t01.w, t11.w);
    return r;
}
// tex2DGrad<float4>(Texture, x, y, dPdx, dPdy) of Model.cu:309
__device__ __forceinline__ v3 tex_grad(const TexView &tv, const TexRec &T, float x, float y, float dxx, float dxy,
                                       float dyx, float dyy)
{
    const float ax = dxx * (float)T.w, ay = dxy * (float)T.h, bx = dyx * (float)T.w, by = dyy * (float)T.h;
    const float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
    float lam = 0.0f;
    if (rho > 1.0f)
        lam = fminf((float)log2((double)rho), (float)(T.levels - 1));
    const float lq = floorf(lam * 256.0f + 0.5f) / 256.0f;
    int d = (int)floorf(lq);
    float wgt = lq - (float)d;
    if (d >= T.levels - 1) {
        d = T.levels - 1;
        wgt = 0.0f;
    }
    const float4 *base = (const float4 *)tv.data;
    const float4 c0 = tex_fetch(base + T.off[d], max(1, T.w >> d), max(1, T.h >> d), x, y);
    if (wgt == 0.0f)
        return mk3(c0.x, c0.y, c0.z);
    const float4 c1 = tex_fetch(base + T.off[d + 1], max(1, T.w >> (d + 1)), max(1, T.h >> (d + 1)), x, y);
    return mk3((1 - wgt) * c0.x + wgt * c1.x, (1 - wgt) * c0.y + wgt * c1.y, (1 - wgt) * c0.z + wgt * c1.z);
}
// Camera::getxy (Camera.cu:26-42)
__device__ __forceinline__ void cam_getxy(const CameraRec &c, v3 poso, float &rx, float &ry)
{
    const float step = 0.25f;
    const v3 dir = normalize(sub(poso, c.pos));
    rx = 0.0f; ry = 0.0f;
    if (dot(dir, c.front) == 0.0f)
        return;
    const float aspect = c.aspect;
    const float t = c.n / dot(dir, c.front);
    const v3 rpos = scale(dir, t);
    rx = dot(rpos, c.right) * c.res_x / step / aspect;
    ry = dot(rpos, c.up) * c.res_y / step;
}
// Model.cu:304-309 for mesh triangle T (uv = its vText), hit by ray r: Triangle::anyPoint (:60-88),
// IntersectRecord::getdxdy (IntersectionRecord.cuh:60-80), tex2DGrad on the mesh's image.
__device__ __forceinline__ v3 mesh_texture_color(const TexView &tv, const TexRec &X, const CameraRec &cam, const TriRec &T,
                                                 const TriUv &uv, const Ray &r)
{
    const v3 &o = r.o, &d = r.d;
    const float dnd = T.nx * d.x + T.ny * d.y + T.nz * d.z;
    const float t = (-(o.x * T.nx + o.y * T.ny + o.z * T.nz) + T.dn) / dnd;
    const v3 Pos = ray_pos(r, t);                                                     // rec.pos, Model.cu:129
    const v3 ta = mk3(T.ax, T.ay, T.az), tb = mk3(T.bx, T.by, T.bz), tc = mk3(T.cx, T.cy, T.cz);
    const float s2 = length(cross(sub(Pos, tc), sub(Pos, ta)));
    const float s3 = length(cross(sub(Pos, tc), sub(Pos, tb)));
    float m1 = s3 / T.S, m2 = s2 / T.S, m3 = 1.0f - m1 - m2;                          // :67
    const float U = uv.u0 * m1 + uv.u1 * m2 + uv.u2 * m3, V = uv.v0 * m1 + uv.v1 * m2 + uv.v2 * m3;   // :68 = rec.uv
    if (m3 > m2) {                                                                    // :69-84 (double literals)
        if (m3 > m1) {
            m1 = (float)((double)m1 + 0.01); m2 = (float)((double)m2 + 0.01); m3 = (float)((double)m3 - 0.02);
        } else {
            m2 = (float)((double)m2 + 0.01); m3 = (float)((double)m3 + 0.01); m1 = (float)((double)m1 - 0.02);
        }
    } else {
        if (m1 > m2) {
            m2 = (float)((double)m2 + 0.01); m3 = (float)((double)m3 + 0.01); m1 = (float)((double)m1 - 0.02);
        } else {
            m1 = (float)((double)m1 + 0.01); m3 = (float)((double)m3 + 0.01); m2 = (float)((double)m2 - 0.02);
        }
    }
    const float aU = uv.u0 * m1 + uv.u1 * m2 + uv.u2 * m3, aV = uv.v0 * m1 + uv.v1 * m2 + uv.v2 * m3;   // :86
    const v3 anyPos = add(scale(sub(Pos, ta), m1), scale(sub(Pos, tb), m2));          // :87
    const float du = aU - U, dv = aV - V;                                             // getdxdy :64
    float x1, y1, x2, y2;
    cam_getxy(cam, Pos, x1, y1);
    cam_getxy(cam, anyPos, x2, y2);
    const float dx = x2 - x1, dy = y2 - y1;
    float rx = du / dx, ry = dv / dx, rz = du / dy, rw = dv / dy;                     // :69-72
    if (du == 0.0f)                                                                   // :74-75 (:76 assigns: never taken)
        rz = rx = 0.0f;
    return tex_grad(tv, X, U, V, fabsf(rx), fabsf(ry), fabsf(rz), fabsf(rw));         // Model.cu:309
}

// black(x) = length(x) < 0.001f (Scene.cu:3-5) and the `length(f) < 0.0001f` of PathTracer.cu:77, without the
// square root: sqrtf is correctly rounded and monotone, so sqrtf(y) < c holds exactly for the y below
// the smallest float whose root reaches c -- 0x1.0c6f7ap-20 for c = 0.001f, 0x1.5798ecp-27 for
// c = 0.0001f (found by bisection over the float bit patterns; tests/test_oracle_cpu.py checks both).
constexpr float kLenSqBelow1e3 = 0x1.0c6f7ap-20f;
constexpr float kLenSqBelow1e4 = 0x1.5798ecp-27f;
__device__ __forceinline__ bool is_black(v3 x) { return dot(x, x) < kLenSqBelow1e3; }
__device__ __forceinline__ bool len_below_1e4(v3 x) { return dot(x, x) < kLenSqBelow1e4; }

} // namespace rt
