// rt_kernels.hip -- gfx950 kernels of the per-pixel render path and their launchers.
//
// k_path renders one progressive sample of a range of pixels in ONE launch: a wave owns a
// chunk of 128 pixels (two per lane) and carries each pixel's whole path in registers --
//   jitter + Camera::generateRay                       kernel.cu:65-73
//   closest hit                                        Scene/Scene.cu:8-67 -> Model/Model.cu:91-137, 349-410
//   per light: illumination, BRDF term, shadow ray     Scene/Scene.cu:69-124, 205-251; Light/Light.cu:7-14
//   BRDF sampling, roulette, next bounce               core/PathTracer.cu:25-90
//   running mean into the framebuffer                  kernel.cu:65-66,75,77
// so the only HBM traffic per pixel-sample is the framebuffer texel, the RNG counter and
// the id / t buffers; rays, hits and shadow rays never leave the register file.  Lanes
// whose path has ended (miss, absorbed) idle until the chunk's last path ends; the chunk's
// 16x8-pixel footprint keeps that loss small, and rays towards ONE light from neighbouring
// pixels traverse the tree together, which is what the wave-uniform traversal wants.
//
// k_intersect is Scene::hit on a caller's ray queue (rt_scene_hit), same traversal.
#include "rt_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "rt_device.h"

namespace rt {

constexpr int kBlock = 256;
constexpr float kHitMax = kInf - 0.1f;      // Scene.cu:66

__device__ __forceinline__ bool scene_ishit(float t) { return t > 0.001f && t < kHitMax; }   // Scene.cu:66

// ------------------------------------------------------------------ closest hit
// Scene::hit (Scene.cu:8-67): the closest of all primitives the reference's scan would
// accept -- min over (t, reference index) -- found without visiting most of them.
//
// A wave traverses for 128 rays at once (one packed pair per lane).  The triangle side is
// wave-uniform: box groups (96 B) and triangle records (64 B) come through the scalar data
// path and sit in SGPRs as operands of packed fp32 VALU ops.  Traversal of the implicit
// 4-ary box tree is stackless: (level, group index) in SGPRs plus one VGPR whose lane l
// holds level l's pending children; a group is expanded when any ray of the wave may
// still hit below it, nearest child first.  node_test() only ever discards (ray,
// triangle) pairs Triangle::hit would reject, so the result is the scan's.
// CULL = false visits every triangle (validation mode, rt_set_culling(ctx, 0)).
//
// Because the result is a min over (t, index), it does not depend on who finds what
// first.  That is what lets the four waves of a block share ONE chunk when the previous
// sample found it heavy (COOP): every wave holds the chunk's rays, walks the top of the
// tree, and descends only into the level-kSplitLevel subtrees a hash assigns to it;
// hits are merged with a 64-bit ds_min on (t bits, index + 1) per ray in LDS, which every
// wave re-reads at each expansion so that all of them prune against the common best.
// A single wave retires a heavy chunk at about a microsecond per node group, and a kernel
// cannot end before its longest chunk does.

// Wave-uniform, read-only scene records are read through the constant address space: that
// pins them to the scalar data path (s_load into SGPRs) whatever the compiler can or cannot
// prove about stores elsewhere in the kernel.  (Without it the path kernel's traversal
// fell back to per-lane global_load of the same address -- 24 VGPRs and six TA
// instructions per node group.)  The scene is written only by rt_upload_scene's copies,
// which complete before any launch; the scalar cache is invalidated at kernel start.
template <class T>
__device__ __forceinline__ T load_uniform(const T *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const T __attribute__((address_space(4))) *cptr;
    return *(cptr)(unsigned long long)p;
#else
    return *p;                                   // host pass of the single-source compile
#endif
}

#ifdef RT_TRAV_STATS
// Debug build only (make EXTRA=-DRT_TRAV_STATS): traversal counters, summed over all waves.
// [0] traversals  [1] node groups expanded  [2] triangle records tested
// [3] triangle tests in which some lane went past the window pre-test
__device__ unsigned long long g_trav_stats[8];
extern "C" void rt_debug_trav_stats(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_stats), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), z, sizeof(z));
    }
}
#define RT_STAT(i, n) do { if ((threadIdx.x & 63u) == 0) atomicAdd(&g_trav_stats[i], (unsigned long long)(n)); } while (0)
#else
#define RT_STAT(i, n) do { } while (0)
#endif
#ifdef RT_TRAV_LOG
// Debug build only (make EXTRA=-DRT_TRAV_LOG): per-chunk timeline of the launches whose p0
// equals g_log_p0 (one stream's pixel range)
struct ChunkLog { unsigned long long t0, t1; unsigned cost, coop; };
__device__ ChunkLog g_log[1 << 17];
__device__ unsigned g_log_p0 = 0xffffffffu;
extern "C" void rt_debug_log(void *out, unsigned p0, unsigned n)
{
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_log), sizeof(ChunkLog) * n);
    hipMemcpyToSymbol(HIP_SYMBOL(g_log_p0), &p0, sizeof(p0));
}
#endif

#ifdef RT_TICKWAIT
// Diagnostic build (tools/build_variant.sh tick -DRT_TICKWAIT, tools/tick_report.py): how long k_batch's waves wait for
// dispenser tickets.  Summed over the waves of all launches since the last reset, in s_memtime ticks:
// 0 wave lifetime, 1 wait for tickets drawn one unit ahead, 2 wait for tickets drawn on the spot (a stage or band
// change), 3 / 4 their numbers, 5 / 6 / 7 time in a unit's phases P + H, S and in its epilogue, 8 units.
// 16 + 4 k ..: the same three phase sums and the units by where the unit came from, k = 0 a listed pixel, 1 an early
// unit, 2 the wave's own band, 3 a band it helps, 4 the late units, 5 / 6 / 7 k = 2 / 3 / 4 again for the wave's
// first four units only.
__device__ unsigned long long g_tick[48];
extern "C" void rt_debug_tick(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tick), sizeof(unsigned long long) * 48);
    if (reset) {
        unsigned long long z[48] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_tick), z, sizeof(z));
    }
}
#define RT_TICK(x) x
#else
#define RT_TICK(x)
#endif
#ifdef RT_STAMPS
// Diagnostic build only (tools/build_stamps.sh): where a wave's cycles go.  s_memtime stamps
// around the segments of the batch kernels, summed over all waves into g_stamps[kernel][segment]
// (shader cycles; every stamp costs the wave a scalar-memory round trip, so totals run ~10 % high).
//  0 wave lifetime   1 dispenser   2 unit prologue (loads until ready)   3 shading between traversals
//  4 stash save/load 5 traverse    6 node-group load wait   7 node tests + ordering
//  8 triangle load wait   9 triangle tests   10 unit tail (feedback)   11 units   12 traversals
// row 1 (segments 16..): 16 prologue loads until arrived   17 ray generation (Philox, camera ray, pair set-up)
//  18 hit record + BRDF-sample verdict after a camera traversal   19 prediction check / chain walk, records to LDS
//  20 (phase S) record loads   21 light sampling + shadow pair set-up   22 blocked verdict, radiance
//  23 epilogue: shares to LDS   24 epilogue: running mean chain + texel store
__device__ unsigned long long g_stamps[4][16];
// per wave of the last k_batch launch: begin, end (100 MHz counter), units it took, XCD
__device__ unsigned long long g_wave_log[8][32768];
extern "C" void rt_debug_wave_log(unsigned long long *out)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_log), sizeof(unsigned long long) * 8 * 32768);
}
extern "C" void rt_debug_stamps(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
    if (reset) {
        unsigned long long z[64] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    }
}
struct Stamps {
    unsigned long long a[32];
    unsigned long long last, unit_t0, unit_max, hard_units, hard_passes;
    __device__ __forceinline__ void start()
    {
        for (int i = 0; i < 32; i++) a[i] = 0;
        unit_max = hard_units = hard_passes = 0;
        last = unit_t0 = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void unit_begin() { unit_t0 = __builtin_amdgcn_s_memtime(); }
    // the longest unit of this wave: cycles in the high word, (expansions + triangle records) << 20 | unit below
    __device__ __forceinline__ void unit_end(uint32_t unit, uint32_t cost)
    {
        const unsigned long long d = __builtin_amdgcn_s_memtime() - unit_t0;
        const unsigned long long key = (d << 32) | ((unsigned long long)(cost > 4095u ? 4095u : cost) << 20) | (unit & 0xfffffu);
        unit_max = key > unit_max ? key : unit_max;
        // histogram of unit costs by power of two, and the cycles spent in each class: g_stamps[3][b], g_stamps[2+...]
        if ((threadIdx.x & 63u) == 0) {
            const int b = cost ? 32 - __builtin_clz(cost) : 0;
            atomicAdd(&g_stamps[3][b > 15 ? 15 : b], 1ull);
            atomicAdd(&g_stamps[0][b > 15 ? 15 : b], d);         // (row 0 holds no kernel's stamps)
        }
    }
    // close the segment that began at the previous mark and charge it to `seg`
    __device__ __forceinline__ void mark(int seg)
    {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        a[seg] += t - last;
        last = t;
    }
    __device__ __forceinline__ void flush(int kernel)
    {
        if ((threadIdx.x & 63u) == 0) {
            for (int i = 0; i < 13; i++)
                if (a[i]) atomicAdd(&g_stamps[kernel][i], a[i]);
            for (int i = 16; i < 32; i++)
                if (a[i]) atomicAdd(&g_stamps[1][i - 16], a[i]);
            atomicMax(&g_stamps[kernel][13], unit_max);          // the longest single unit of any wave
            atomicAdd(&g_stamps[kernel][14], hard_units);
            atomicAdd(&g_stamps[kernel][15], hard_passes);
        }
    }
};
#define RT_MARK(sp, seg) (sp).mark(seg)
#define RT_COUNT(sp, seg) ((sp).a[seg]++)
#define RT_STAMP(x) x
template <class T> __device__ __forceinline__ void stamp_arrived(const T &v)
{
    // the load must have landed before the stamp that follows: use its first and last dword
    const uint32_t *w = (const uint32_t *)&v;
    asm volatile("" :: "s"(w[0]), "s"(w[sizeof(T) / 4 - 1]));
}
__device__ __forceinline__ void stamp_arrived_v(uint32_t v) { asm volatile("" :: "v"(v)); }
#else
struct Stamps { };
#define RT_MARK(sp, seg) do { } while (0)
#define RT_COUNT(sp, seg) do { } while (0)
#define RT_STAMP(x)
template <class T> __device__ __forceinline__ void stamp_arrived(const T &) { }
__device__ __forceinline__ void stamp_arrived_v(uint32_t) { }
#endif

constexpr unsigned long long kHeavyFactor = 8;   // heavy = this many times the mean chunk cost (4 and 16 measured slower)
constexpr int kSplitLevel = 2;       // cooperative chunks: subtrees of 4^(kSplitLevel+1) triangles

__device__ __forceinline__ unsigned long long pack_hit(float t, int32_t id)
{
    return ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)(id + 1);
}

// Wave-uniform side of a traversal.
struct TravArgs {
    TreeView tv;
    const TriRec *tris;
    const SphereRec *spheres;
    int n_tris, n_spheres;
    int sorted_planes;               // 0: always the generic node test (RT_SORTED=0, validation)
    const TriLightRec *tlights;      // traverse<.., LIGHTS = true> only
    int n_tlights;
};
struct P4 { uint32_t v[4]; };        // one axis' planes of a node group's four boxes (float bits)

__device__ __forceinline__ void test_triangle(const TriRec *__restrict__ tris, uint32_t node,
                                              RayPair &r, uint32_t lanebits, int j, Stamps &sp)
{
    RT_MARK(sp, 7);
    const TriRec T = load_uniform(&tris[node]);
    stamp_arrived(T);
    RT_MARK(sp, 8);
    bool cx = (lanebits >> (2 * j)) & 1u, cy = (lanebits >> (2 * j + 1)) & 1u;
    RT_STAT(2, 1);
    if (__ballot(cx | cy)) {
        tri_test_pair(T, r, cx, cy);
        // shadow rays (occlusion form): a hit safely below the threshold ends the ray --
        // nothing can pass its node tests any more
        if (r.best.x < r.stop.x) r.bestp.x = -1.0f;
        if (r.best.y < r.stop.y) r.bestp.y = -1.0f;
    }
    RT_MARK(sp, 9);
}

// The six plane loads of a node group with the octant's offsets as the instructions' own SGPR offset
// operand (the compiler adds each offset to the 64-bit base first: two scalar adds per load, twelve per
// group; measured -2 % of the step time on C2, C3 and C4).  Loads only; the wait for them is in the block.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_planes(const NodeGroup *gp, const uint32_t off_n[3], const uint32_t off_f[3],
                                            P4 &nx, P4 &ny, P4 &nz, P4 &fx, P4 &fy, P4 &fz)
{
    u32x4 a, b, c, d, e, f;
    const unsigned long long addr = (unsigned long long)gp;
    const unsigned long long base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr) |
                                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(addr >> 32)) << 32;
    asm volatile("s_load_dwordx4 %0, %6, %7\n\t"
                 "s_load_dwordx4 %1, %6, %8\n\t"
                 "s_load_dwordx4 %2, %6, %9\n\t"
                 "s_load_dwordx4 %3, %6, %10\n\t"
                 "s_load_dwordx4 %4, %6, %11\n\t"
                 "s_load_dwordx4 %5, %6, %12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(f)
                 : "s"(base), "s"(off_n[0]), "s"(off_n[1]), "s"(off_n[2]), "s"(off_f[0]), "s"(off_f[1]), "s"(off_f[2])
                 : "memory");
    nx = P4{{a.x, a.y, a.z, a.w}}; ny = P4{{b.x, b.y, b.z, b.w}}; nz = P4{{c.x, c.y, c.z, c.w}};
    fx = P4{{d.x, d.y, d.z, d.w}}; fy = P4{{e.x, e.y, e.z, e.w}}; fz = P4{{f.x, f.y, f.z, f.w}};
}

// Per-level traversal state lives in ONE VGPR: lane l holds level l's word
// (bits 0-7: the passing children still to visit, nearest first, 2 bits each;
// bits 8-10: how many), read with v_readlane at a wave-uniform lane index and
// written with a lane-id select.  Level offsets sit in a second VGPR the same way, so
// descending costs no memory access besides the node block itself.
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ uint32_t lane_set(uint32_t v, int l, uint32_t x)
{
    return (threadIdx.x & 63u) == (uint32_t)l ? x : v;     // v_cmp + v_cndmask
}
__device__ __forceinline__ void cswap(uint32_t &a, uint32_t &b)
{
    uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}

// Loads a lane's two rays into the packed pair.  live[h] = false parks the ray far away
// with an empty t window: it passes no node test and no triangle test.  `best` is the
// start value of the running closest t (IntersectRecord(): INF), `stop` the occlusion
// threshold (-1: none).
__device__ __forceinline__ void pair_set(RayPair &r, const v3 o[2], const v3 d[2],
                                         const float best[2], const float stop[2],
                                         const bool live[2], float E)
{
    v3 oo[2], dd[2];
    float b[2], s[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        oo[h] = live[h] ? o[h] : mk3(0.0f, 1e18f, 0.0f);
        dd[h] = live[h] ? d[h] : mk3(1.0f, 0.0f, 0.0f);
        b[h] = live[h] ? best[h] : 0.0f;
        s[h] = live[h] ? stop[h] : -1.0f;
    }
    r.ox = f2{oo[0].x, oo[1].x}; r.oy = f2{oo[0].y, oo[1].y}; r.oz = f2{oo[0].z, oo[1].z};
    r.dx = f2{dd[0].x, dd[1].x}; r.dy = f2{dd[0].y, dd[1].y}; r.dz = f2{dd[0].z, dd[1].z};
    // rays whose origin lies outside the culling extent (or is not finite) are never culled
    bool wild[2];
#pragma unroll
    for (int h = 0; h < 2; h++)
        wild[h] = live[h] & !((fabsf(oo[h].x) <= E) & (fabsf(oo[h].y) <= E) & (fabsf(oo[h].z) <= E) &
                              (fabsf(dd[h].x) <= 2.0f) & (fabsf(dd[h].y) <= 2.0f) & (fabsf(dd[h].z) <= 2.0f));
    r.ix = f2{wild[0] ? 0.0f : slab_inv(dd[0].x), wild[1] ? 0.0f : slab_inv(dd[1].x)};
    r.iy = f2{wild[0] ? 0.0f : slab_inv(dd[0].y), wild[1] ? 0.0f : slab_inv(dd[1].y)};
    r.iz = f2{wild[0] ? 0.0f : slab_inv(dd[0].z), wild[1] ? 0.0f : slab_inv(dd[1].z)};
    r.px = f2{wild[0] ? 0.0f : oo[0].x * r.ix.x, wild[1] ? 0.0f : oo[1].x * r.ix.y};
    r.py = f2{wild[0] ? 0.0f : oo[0].y * r.iy.x, wild[1] ? 0.0f : oo[1].y * r.iy.y};
    r.pz = f2{wild[0] ? 0.0f : oo[0].z * r.iz.x, wild[1] ? 0.0f : oo[1].z * r.iz.y};
    r.best = f2{b[0], b[1]};
    r.bestp = r.best * kBestInfl;
    r.stop = f2{s[0], s[1]};
    r.idx = -1; r.idy = -1;
    // Do the wave's culled rays share an octant?  (Never-culled and parked rays do not vote: both
    // planes of an axis give them the same parameter.)
    uint32_t oct = 8u;
    const float inv[3][2] = {{r.ix.x, r.ix.y}, {r.iy.x, r.iy.y}, {r.iz.x, r.iz.y}};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const bool v0 = live[0] & !wild[0], v1 = live[1] & !wild[1];
        const bool n0 = (__float_as_uint(inv[a][0]) >> 31) != 0u, n1 = (__float_as_uint(inv[a][1]) >> 31) != 0u;
        const unsigned long long neg = __ballot(v0 & n0) | __ballot(v1 & n1);
        const unsigned long long pos = __ballot(v0 & !n0) | __ballot(v1 & !n1);
        if (neg && pos) oct = 0u;
        if (neg) oct |= 1u << a;
    }
    r.oct = (oct & 8u) ? oct : 0u;
}

// What Scene::hit scans after the triangles (Scene.cu:31-64), shared by both traversals: the quadrics, then
// (LIGHTS) the triangle lights, each with strict < against what stands; before them the closest TRIANGLE
// is noted -- the mesh whose texture colour the record carries (Model.cu:301-310).
template <bool LIGHTS>
__device__ __forceinline__ void after_triangles(const TravArgs &A, RayPair &r)
{
    if (LIGHTS) {
        r.tidx = r.idx;
        r.tidy = r.idy;
    }
    // quadrics come last (Scene.cu:31-37), strict < against the triangles' result
    for (int s = 0; s < A.n_spheres; s++) {
        const SphereRec S = load_uniform(&A.spheres[s]);
        float t;
        if (sphere_hit(S, mk3(r.ox.x, r.oy.x, r.oz.x), mk3(r.dx.x, r.dy.x, r.dz.x), r.best.x, t)) {
            r.best.x = t;
            r.idx = A.n_tris + s;
        }
        if (sphere_hit(S, mk3(r.ox.y, r.oy.y, r.oz.y), mk3(r.dx.y, r.dy.y, r.dz.y), r.best.y, t)) {
            r.best.y = t;
            r.idy = A.n_tris + s;
        }
    }
    // triangle lights after every model (Scene.cu:43-58), strict < again; a hit reports the
    // light's id n_tris + n_quads + k (what rec.lightidx = k, rec.isLight say in the reference)
    if (LIGHTS) {
        for (int k = 0; k < A.n_tlights; k++) {
            const TriRec T = load_uniform(&A.tlights[k].T);
            float t;
            if (tri_hit(T, mk3(r.ox.x, r.oy.x, r.oz.x), mk3(r.dx.x, r.dy.x, r.dz.x), r.best.x, t)) {
                r.best.x = t;
                r.idx = T.orig;
            }
            if (tri_hit(T, mk3(r.ox.y, r.oy.y, r.oz.y), mk3(r.dx.y, r.dy.y, r.dz.y), r.best.y, t)) {
                r.best.y = t;
                r.idy = T.orig;
            }
        }
    }
}

// Scene::hit for the pair in `r` (Scene.cu:8-67): on return r.best / r.idx, r.idy hold the
// closest accepted t and its reference index per ray (-1 = none; occlusion rays: any hit
// below r.stop).  `cost` accumulates node groups expanded + triangles tested.
// lvloff: lane l holds the first group of level l.
// COOP: called by all four waves of the block with identical rays; s_best is the block's
// LDS merge array (128 entries).  Every wave returns the block's common answer.
template <bool CULL, bool COOP, bool ORDER = true, bool LIGHTS = false, bool OWN_BASES = false>
__device__ __forceinline__ void traverse(const TravArgs &A, uint32_t lvloff, RayPair &r,
                                         uint32_t lane, uint32_t wib,
                                         unsigned long long *s_best, uint32_t &cost, Stamps &sp)
{
    const TreeView &tv = A.tv;
    const bool sorted_ok = A.sorted_planes != 0;
    // OWN_BASES: the two base pointers of the loops below as values of their own.  Left as parts of the argument
    // block they sit in registers eight dwords at a time, and a spill of one restores all eight inside the loop
    // (k_batch's five-wave build: eight v_readlane per triangle record for one 64-bit base, C5 -2.6 %).  The
    // four-wave build allocates better without (C3 +1.6 %), so k_batch asks for it by its build.
    const TriRec *tris = A.tris;
    const NodeGroup *groups = tv.groups;
    if (OWN_BASES)
        asm volatile("" : "+s"(tris), "+s"(groups));
    RT_STAT(0, 1);
    RT_COUNT(sp, 12);
    unsigned long long pub0 = 0, pub1 = 0;       // what this wave last saw or wrote per ray
    if (COOP) {
        pub0 = pack_hit(r.best.x, -1);
        pub1 = pack_hit(r.best.y, -1);
        __syncthreads();                         // the previous traversal's readers are done
        if (wib == 0) {
            s_best[lane] = pub0;
            s_best[64 + lane] = pub1;
        }
        __syncthreads();
    }
    // take over what the other waves have found (COOP): min over (t, index)
    auto merge_in = [&]() {
        unsigned long long vx = s_best[lane], vy = s_best[64 + lane];
        if (vx < pack_hit(r.best.x, r.idx)) {
            r.best.x = __uint_as_float((uint32_t)(vx >> 32));
            r.idx = (int32_t)(uint32_t)vx - 1;
            r.bestp.x = r.best.x < r.stop.x ? -1.0f : r.best.x * kBestInfl;
        }
        if (vy < pack_hit(r.best.y, r.idy)) {
            r.best.y = __uint_as_float((uint32_t)(vy >> 32));
            r.idy = (int32_t)(uint32_t)vy - 1;
            r.bestp.y = r.best.y < r.stop.y ? -1.0f : r.best.y * kBestInfl;
        }
        pub0 = vx < pub0 ? vx : pub0;
        pub1 = vy < pub1 ? vy : pub1;
    };
    // publish this wave's improvements (COOP)
    auto merge_out = [&]() {
        unsigned long long mx = pack_hit(r.best.x, r.idx), my = pack_hit(r.best.y, r.idy);
        if (mx < pub0) { atomicMin(&s_best[lane], mx); pub0 = mx; }
        if (my < pub1) { atomicMin(&s_best[64 + lane], my); pub1 = my; }
    };

    if (CULL) {
        int l = tv.top;
        uint32_t g = 0;
        uint32_t state = 0;                  // lane l: pending children of level l
        uint32_t lanebits = 0;               // level 0: which (triangle, ray) pairs passed
        bool expand = true;
        const bool split = COOP && tv.top > kSplitLevel;
        // Every culled ray of the wave in one octant (the usual case: the rays of a unit leave one
        // point or go to one light): the node groups are read plane by plane, entry side and exit
        // side (rt_math.h NodeGroup: lo[axis][4] at axis * 16, hi[axis][4] at 48 + axis * 16).
        const uint32_t oct = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.oct);
        const bool sorted = (oct & 8u) != 0u && sorted_ok;
        // a dead box (lo = +inf, hi = -inf) shows on its x entry plane: lo.x = +inf, or hi.x = -inf
        // for rays running towards -x; no other box has either
        const uint32_t dead_key = 0x7f800000u | (oct & 1u) << 31;
        uint32_t off_n[3], off_f[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const bool neg = ((oct >> a) & 1u) != 0u;
            off_n[a] = (neg ? 48u : 0u) + 16u * (uint32_t)a;
            off_f[a] = (neg ? 0u : 48u) + 16u * (uint32_t)a;
        }
        for (;;) {
            if (expand) {
                cost++;
                RT_STAT(1, 1);
                if (COOP) merge_in();
                RT_MARK(sp, 7);
                const NodeGroup *gp = &groups[lane_get(lvloff, l) + g];
                uint32_t key[4], n_pass = 0, lb = 0, tmask = 0;
                // One child: `test` fills the two rays' verdicts and entry parameters and says
                // whether the box is dead (lo = +inf, hi = -inf: padding, or a triangle
                // Model.cu:99 always rejects -- dead for every ray, including the unculled ones).
                auto child = [&](int j, bool px, bool py, float ex, float ey, bool dead) {
                    const unsigned long long b = __ballot(px) | __ballot(py);   // the compares' own lane masks
                    key[j] = 0xffffffffu;
                    bool take = b && !dead;
                    if (take && split && l == kSplitLevel + 1) {
                        // the subtree below this child belongs to one wave of the block
                        uint32_t node = g * kTreeArity + (uint32_t)j;
                        take = ((node * 2654435761u) >> 30) == wib;
                    }
                    if (!ORDER) {
                        // occlusion rays visit in index order: the state word is just the mask of
                        // the children to visit
                        tmask |= (uint32_t)take << j;
                    } else if (take) {
                        // visit order: entry parameter of the first passing lane (a coherent
                        // wave's rays agree on it closely enough; the smaller of the lane's
                        // passing rays, compared as integers: entry parameters are >= +0); the
                        // child index rides in the two low mantissa bits.  ORDER = false
                        // (occlusion rays: any hit below the threshold ends the ray, so
                        // nearest-first buys nothing): index order.
                        if (ORDER) {
                            const uint32_t ux = px ? __float_as_uint(ex) : 0x7f000000u;
                            const uint32_t uy = py ? __float_as_uint(ey) : 0x7f000000u;
                            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)min(ux, uy), __builtin_ctzll(b));
                            key[j] = (e & ~3u) | (uint32_t)j;
                        } else {
                            key[j] = (uint32_t)j;
                        }
                        n_pass++;
                    }
                    if (l == 0)                  // leaves: which (triangle, ray) pairs passed, per lane
                        lb |= ((uint32_t)px << (2 * j)) | ((uint32_t)py << (2 * j + 1));
                };
                if (sorted) {
                    // the planes already ordered for the wave's octant: entry side, exit side
                    P4 nx, ny, nz, fx, fy, fz;
                    load_planes(gp, off_n, off_f, nx, ny, nz, fx, fy, fz);
                    stamp_arrived(fz);
                    RT_MARK(sp, 6);
#pragma unroll
                    for (int j = 0; j < kTreeArity; j++) {
                        bool px, py;
                        float ex, ey;
                        node_test_sorted(r, __uint_as_float(nx.v[j]), __uint_as_float(ny.v[j]), __uint_as_float(nz.v[j]),
                                         __uint_as_float(fx.v[j]), __uint_as_float(fy.v[j]), __uint_as_float(fz.v[j]),
                                         px, py, ex, ey);
                        child(j, px, py, ex, ey, nx.v[j] == dead_key);
                    }
                } else {
                    const NodeGroup G = load_uniform(gp);
                    stamp_arrived(G);
                    RT_MARK(sp, 6);
#pragma unroll
                    for (int j = 0; j < kTreeArity; j++) {
                        bool px, py;
                        float ex, ey;
                        node_test(r, G.lo[0][j], G.lo[1][j], G.lo[2][j], G.hi[0][j], G.hi[1][j],
                                  G.hi[2][j], px, py, ex, ey);
                        child(j, px, py, ex, ey, __float_as_uint(G.lo[0][j]) == 0x7f800000u);
                    }
                }
                uint32_t word;
                if (!ORDER) {
                    word = tmask;
                } else {
                    cswap(key[0], key[1]); cswap(key[2], key[3]);
                    cswap(key[0], key[2]); cswap(key[1], key[3]);
                    cswap(key[1], key[2]);
                    word = (key[0] & 3u) | (key[1] & 3u) << 2 | (key[2] & 3u) << 4 | (key[3] & 3u) << 6 | n_pass << 8;
                }
                state = lane_set(state, l, word);
                if (l == 0)
                    lanebits = lb;
                expand = false;
            }
            // ORDER: bits 0-7 the children still to visit, nearest first, 2 bits each, bits 8-10 how
            // many; else: bit j = child j still to visit
            uint32_t word = lane_get(state, l);
            if ((ORDER ? word >> 8 : word) == 0) {
                if (l == tv.top)
                    break;
                l++;
                g >>= 2;
                continue;
            }
            int j;
            if (ORDER) {
                j = (int)(word & 3u);
                state = lane_set(state, l, ((word & 0xffu) >> 2) | ((word >> 8) - 1u) << 8);
            } else {
                j = __builtin_ctz(word);
                state = lane_set(state, l, word & (word - 1u));
            }
            uint32_t node = g * kTreeArity + (uint32_t)j;
            if (l == 0) {
                cost++;
                test_triangle(tris, node, r, lanebits, j, sp);
                if (COOP) merge_out();
            } else {
                l--;
                g = node;
                expand = true;
            }
        }
    } else {
        for (uint32_t node = 0; node < (uint32_t)tv.n_groups0 * kTreeArity; node++)
            test_triangle(tris, node, r, 0xffffffffu, 0, sp);
    }
    RT_MARK(sp, 7);
    if (COOP) {
        __syncthreads();
        merge_in();                              // the block's final answer
    }
    after_triangles<LIGHTS>(A, r);
}

// The same query with the traversal PER LANE: every lane walks the tree for its own pair of rays
// (children still to visit: four bits per level in a 64-bit word; node groups and triangle records
// through vector loads).  The wave-wide traversal above costs what the UNION of the wave's rays
// visits -- next to nothing more than one ray for the camera rays of neighbouring pixels or shadow
// rays towards one light, but most of the tree for the rays a surface scatters: bounce rays (and the
// shadow rays cast where they land) were 0.3 Grays/s against 18 for the camera rays.  Here a lane pays
// for its own two rays only (and for its neighbours' divergence).  Same node test, same triangle
// test, same tie rule (H8) as above, so the same result; children are visited in index order.
// s_lvl: the tree's level offsets in LDS (a lane indexes them by its own level).
template <bool CULL, bool LIGHTS = false>
__device__ __forceinline__ void traverse_lane(const TravArgs &A, const uint32_t *s_lvl, RayPair &r, bool live_x,
                                              bool live_y, uint32_t &cost)
{
    const TreeView &tv = A.tv;
    uint32_t steps = 0;
    if (CULL) {
        int level = tv.top;
        uint32_t g = 0;
        unsigned long long pend = 0;
        auto expand = [&](int lvl, uint32_t grp) -> uint32_t {
            const float4 *G = (const float4 *)(tv.groups + s_lvl[lvl] + grp);
            const float4 l0 = G[0], l1 = G[1], l2 = G[2], h0 = G[3], h1 = G[4], h2 = G[5];
            uint32_t m = 0;
            bool px, py;
            float ex, ey;
            node_test(r, l0.x, l1.x, l2.x, h0.x, h1.x, h2.x, px, py, ex, ey);
            m |= ((px | py) & !(l0.x > h0.x)) ? 1u : 0u;
            node_test(r, l0.y, l1.y, l2.y, h0.y, h1.y, h2.y, px, py, ex, ey);
            m |= ((px | py) & !(l0.y > h0.y)) ? 2u : 0u;
            node_test(r, l0.z, l1.z, l2.z, h0.z, h1.z, h2.z, px, py, ex, ey);
            m |= ((px | py) & !(l0.z > h0.z)) ? 4u : 0u;
            node_test(r, l0.w, l1.w, l2.w, h0.w, h1.w, h2.w, px, py, ex, ey);
            m |= ((px | py) & !(l0.w > h0.w)) ? 8u : 0u;
            return m;
        };
        bool run = live_x | live_y;
        if (run)
            pend = (unsigned long long)expand(level, 0u) << (4 * level);
        while (run) {
            const uint32_t m = (uint32_t)(pend >> (4 * level)) & 15u;
            if (m == 0u) {                       // this group is done: back to its parent's
                level++;
                g >>= 2;
                run = level <= tv.top;
                continue;
            }
            const int j = __builtin_ctz(m);
            pend &= ~(1ull << (4 * level + j));
            const uint32_t node = g * kTreeArity + (uint32_t)j;
            steps++;
            if (level == 0) {
                const float4 *tp = (const float4 *)(A.tris + node);
                const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
                TriRec T;
                T.ax = t0.x; T.ay = t0.y; T.az = t0.z; T.dn = t0.w;
                T.bx = t1.x; T.by = t1.y; T.bz = t1.z; T.S = t1.w;
                T.cx = t2.x; T.cy = t2.y; T.cz = t2.z; T.orig = __float_as_int(t2.w);
                T.nx = t3.x; T.ny = t3.y; T.nz = t3.z; T.qmax = t3.w;
                tri_test_pair(T, r, live_x, live_y);
                if (r.best.x < r.stop.x) r.bestp.x = -1.0f;      // occlusion form: the ray is over
                if (r.best.y < r.stop.y) r.bestp.y = -1.0f;
            } else {
                level--;
                g = node;
                pend |= (unsigned long long)expand(level, g) << (4 * level);
            }
        }
    } else {
        for (uint32_t node = 0; node < (uint32_t)tv.n_groups0 * kTreeArity; node++) {
            const TriRec T = load_uniform(&A.tris[node]);
            tri_test_pair(T, r, live_x, live_y);
        }
    }
    // the wave's cost: its longest lane
    for (int o = 32; o > 0; o >>= 1)
        steps = max(steps, (uint32_t)__shfl_xor((int)steps, o));
    cost += steps;
    after_triangles<LIGHTS>(A, r);
}

// ------------------------------------------------------------------ hit record
__device__ __forceinline__ void fill_hit(const DevView &v, const Ray &r, int32_t id, Hit &h)
{
    if (id < v.n_tris)
        fill_tri_hit(v.tris[v.tri_pos_of[id]], v.tri_cold[id], v.models[v.tri_cold[id].model], r, h);
    else
        fill_sphere_hit(v.spheres[id - v.n_tris], r, h);
}

// kernel.cu:65-66,75,77: running mean over progressive samples, alpha = 1.
__device__ __forceinline__ void finalize_path(const DevView &v, uint32_t p, v3 res, int cnt)
{
    float frac1 = 1.0f / cnt;
    float frac2 = frac1 * (cnt - 1);
    float4 prev = v.fb[p];
    float4 out;
    out.x = frac2 * prev.x + frac1 * res.x;
    out.y = frac2 * prev.y + frac1 * res.y;
    out.z = frac2 * prev.z + frac1 * res.z;
    out.w = 1.0f;
    v.fb[p] = out;
}

// (x, y) of global pixel index pix = y * W + x (kernel.cu:46,63-64) without the generic 32-bit division (some
// thirty-five instructions, twice per camera ray): the quotient from a float estimate, put right by one step either
// way -- exact for every image whose sides fit 16 bits; anything larger takes the division.
__device__ __forceinline__ void pixel_xy(uint32_t pix, int width, int height, int &x, int &y)
{
    const uint32_t W = (uint32_t)width;
    if ((uint32_t)(width | height) < 65536u) {
        uint32_t q = (uint32_t)((float)pix * __builtin_amdgcn_rcpf((float)W));       // within one of pix / W
        int r = (int)(pix - q * W);
        if (r < 0) { q -= 1u; r += (int)W; }
        if (r >= (int)W) { q += 1u; r -= (int)W; }
        x = r; y = (int)q;
    } else {
        x = (int)(pix % W); y = (int)(pix / W);
    }
}

// One delta light against one hit: PointLight::lightIllumi (Light/Light.cu:7-14) or
// DirectionalLight::lightIllumi (Light/Light.cu:21-33), then Scene::evaluateDirectLight up
// to the shadow-ray test (Scene.cu:212-224).  L is wave-uniform, so the type branch is too.
// wo_y = dot(-r.dir, normal): all Material::f reads of the outgoing direction (Object.cu:107-110,
// material.cu:52-57).
struct LightSample { Ray sr; v3 contrib; float thr; bool cast; };
__device__ __forceinline__ LightSample sample_point_light(const Hit &h, const ModelRec &M,
                                                         float wo_y, const LightRec &L)
{
    LightSample s;
    v3 lp = mk3(L.pos[0], L.pos[1], L.pos[2]), illum = mk3(L.illum[0], L.illum[1], L.illum[2]);
    Ray wi;
    v3 color;
    if (L.type == kLightPoint) {
        wi = make_ray(lp, sub(h.pos, lp));                                  // Light.cu:9
        v3 dd = sub(h.pos, lp);
        color = divs(illum, fmaxf(0.001f, dot(dd, dd)));                    // Light.cu:13
    } else {
        wi = make_ray(sub(h.pos, scale(lp, 10000.0f)), lp);                 // Light.cu:25
        color = illum;                                                      // Light.cu:27
    }
    v3 lpos = wi.o;                                                         // Scene.cu:213
    s.sr = spawn_ray(h, neg(wi.d));                                         // Scene.cu:218
    v3 f = scale(rec_f_y(h, M, wo_y, s.sr.d), fabsf(dot(h.normal, s.sr.d)));   // Scene.cu:222
    s.cast = !is_black(f);                                                  // Scene.cu:224
    s.thr = length(sub(s.sr.o, lpos)) - 0.01f;                              // Scene.cu:227 rhs
    // Scene.cu:233-241 with pdf_light = 1 (Light.cu:12), isDelta = true
    s.contrib = is_black(color) ? mk3(0.0f, 0.0f, 0.0f) : divs(mul(color, f), 1.0f);
    return s;
}

// ------------------------------------------------------------------ the path kernel
struct PathArgs {
    DevView v;
    TravArgs T;
    uint32_t p0, n;                  // this launch renders local pixels [p0, p0 + n)
    int max_depth;                   // MAX_DEPTH of PathTracer.cu:3-5
    bool occlusion;                  // shadow rays may stop at the first hit below the threshold
    Ctl *ctl;
    ChunkFeedback fb;
    uint32_t heavy_x4;               // heavy = cost >= heavy_x4 / 4 times the mean chunk cost
    bool regen;                      // general kernel: dead slots take new pixels at once (see general_wave)
    uint32_t n_samples;              // k_path: progressive samples cnt, cnt + 1, .. a chunk renders in this launch
    bool lane_traversal;             // k_path: bounce rays walk the tree lane by lane (traverse_lane)
    BounceQueue bq;                  // k_path<.., EMIT>, k_bounce
};
// Per-launch values every chunk shares (kept apart from the kernel argument block, which
// stays read-only in the constant address space).
struct ChunkEnv {
    uint32_t lvloff;                 // lane l: first group of level l
    uint32_t heavy_cost;             // cost from which a chunk counts as heavy (feedback)
    uint32_t mean_cost;              // mean chunk cost of the previous launch (0: unknown)
    bool feedback;
    const uint32_t *s_lvl;           // the tree's level offsets in LDS (traverse_lane), or null
};

// Feedback of one finished chunk (called by one lane): its class for the next launch.
__device__ __forceinline__ void record_cost(const ChunkFeedback &fb, const ChunkEnv &env,
                                            uint32_t chunk, uint32_t cost)
{
    const bool heavy = cost >= env.heavy_cost;
    const bool early_a = !heavy && env.mean_cost && cost >= 4u * env.mean_cost;
    const bool early_b = !heavy && !early_a && env.mean_cost && cost >= 2u * env.mean_cost;
    fb.flag_w[chunk] = heavy ? 1 : (early_a | early_b) ? 2 : 0;
    atomicAdd(&fb.cnt_w->total, (unsigned long long)cost);
    if (heavy)
        fb.heavy_w[atomicAdd(&fb.cnt_w->n_heavy, 1u)] = chunk;
    else if (early_a)
        fb.early_w[atomicAdd(&fb.cnt_w->n_early_a, 1u)] = chunk;
    else if (early_b)
        fb.early_w[fb.n_slots - 1u - atomicAdd(&fb.cnt_w->n_early_b, 1u)] = chunk;
}

// What one ray of a lane carries from one traversal to the next.  A traversal wants every
// VGPR it can get (occupancy hides its scalar-load latency), so this state is parked in LDS
// for its duration: one dword column per thread, kSlots rows (conflict-free).
// DEPTH0 (max_depth == 0): beta = 1, res = 0, and nothing samples the BRDF's direction.
template <bool DEPTH0>
struct RayState {
    uint32_t draw;                   // next Philox draw of the pixel's stream
    float wo_y;                      // dot(-r.dir, normal) of the current hit
    v3 pos, normal;                  // IntersectRecord of the current hit
    int model;
    v3 le;                           // direct light gathered at the current hit, in light order
    v3 contrib;                      // unshadowed contribution of the light under test
    float thr;                       // its Scene.cu:227 threshold |o - l| - 0.01
    v3 tangent, beta, res;           // !DEPTH0 only
    static constexpr int kSlots = DEPTH0 ? 16 : 25;

    __device__ __forceinline__ void save(float *st) const
    {
        int k = 0;
        auto put = [&](float x) { st[(k++) * kBlock] = x; };
        put(__uint_as_float(draw)); put(wo_y);
        put(pos.x); put(pos.y); put(pos.z); put(normal.x); put(normal.y); put(normal.z);
        put(__int_as_float(model));
        put(le.x); put(le.y); put(le.z); put(contrib.x); put(contrib.y); put(contrib.z); put(thr);
        if (!DEPTH0) {
            put(tangent.x); put(tangent.y); put(tangent.z);
            put(beta.x); put(beta.y); put(beta.z); put(res.x); put(res.y); put(res.z);
        }
    }
    __device__ __forceinline__ void load(const float *st)
    {
        int k = 0;
        auto get = [&]() { return st[(k++) * kBlock]; };
        draw = __float_as_uint(get()); wo_y = get();
        pos.x = get(); pos.y = get(); pos.z = get(); normal.x = get(); normal.y = get(); normal.z = get();
        model = __float_as_int(get());
        le.x = get(); le.y = get(); le.z = get(); contrib.x = get(); contrib.y = get(); contrib.z = get();
        thr = get();
        if (!DEPTH0) {
            tangent.x = get(); tangent.y = get(); tangent.z = get();
            beta.x = get(); beta.y = get(); beta.z = get(); res.x = get(); res.y = get(); res.z = get();
        }
    }
    __device__ __forceinline__ Hit hit() const
    {
        Hit h;
        h.pos = pos; h.normal = normal;
        h.tangent = DEPTH0 ? mk3(0.0f, 0.0f, 0.0f) : tangent;
        h.color = mk3(-1.0f, -1.0f, -1.0f);      // H5: unbound mesh texture / quadrics leave rec.color
        h.model = model;
        return h;
    }
};

// One chunk = the pixels [chunk * 128, chunk * 128 + 128) of the launch's range, two per
// lane (h = 0, 1: 64 apart, i.e. two neighbouring 8x8 blocks of the image).  The closest
// hit of a bounce and the shadow rays towards each light go through the SAME traversal
// call: `j` says which it is (-1: closest hit, else the light).  A shadow ray towards
// light j is cast only for lanes whose BRDF term is not black (Scene.cu:224); lights for
// which no lane of the chunk casts are skipped without a traversal.  Without quadrics in
// the scene the shadow ray's closest hit itself is not needed, only
// `|hit - o| < |o - l| - 0.01` (Scene.cu:227): hits beyond thr + m cannot matter and the
// ray stops at the first hit below thr - m, m covering the difference between t and the
// reference's |fl(o + t d) - o|; inside the band the true closest hit decides, exactly as
// the reference computes it.
// COOP: all four waves of the block run the chunk with identical registers and share only
// the traversals; wave 0 writes.  st: this thread's column of the block's LDS stash.
// Between the two slots of a lane in the shading code: keeps the scheduler from interleaving them (their
// live ranges then overlap and the three-wave build spills: 117 -> 14 registers, C2 26.7 -> 29.4 Grays/s;
// no effect in k_batch, which does not spill, and none on the general kernel's); -DRT_SCHED_FENCE_OFF to compare.
#ifdef RT_SCHED_FENCE_OFF
#define RT_SCHED_FENCE() ((void)0)
#else
#define RT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <bool CULL, bool COOP, bool DEPTH0, bool EMIT = false>
__device__ __forceinline__ void path_chunk(const PathArgs &A, const ChunkEnv &env, uint32_t chunk,
                                           uint32_t lane, uint32_t wib,
                                           unsigned long long *s_best, uint32_t *s_cost, float *st,
                                           int cnt, uint32_t &cost_sum)
{
    using State = RayState<DEPTH0>;
    const DevView &v = A.v;
    const bool writer = !COOP || wib == 0;
    const int L = v.n_lights;
    const float E = A.T.tv.extent;
    uint32_t cost = 0, n_shadow = 0, n_bounce = 0;
#ifdef RT_TRAV_LOG
    const unsigned long long tlog0 = (unsigned long long)wall_clock64();
#endif
    if (COOP && threadIdx.x == 0)
        *s_cost = 0;                             // ordered before the adds by traverse()'s barriers

    auto pixel_of = [&](int h) { return A.p0 + min(chunk * kPathChunk + (uint32_t)h * 64u + lane, A.n - 1u); };
    bool mine[2], alive[2], ishit[2] = {false, false}, cast[2] = {false, false};
    bool emitted[2] = {false, false};        // EMIT: the path goes on in k_bounce, which writes its pixel's counter
    State S[2];
    const float inf2[2] = {kInf, kInf}, none2[2] = {-1.0f, -1.0f};
    RayPair w;
    {
        v3 ro[2], rd[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            mine[h] = chunk * kPathChunk + (uint32_t)h * 64u + lane < A.n;
            alive[h] = mine[h];
            const uint32_t p = pixel_of(h), pix = v.pixmap[p];
            // kernel.cu:65-73 with the H4 convention: the jitter is (.x, .y) of one
            // curand_uniform4 of the pixel's own stream
            float u[4];
            const uint32_t draw = v.rng_ctr[p];
            pixel_uniform4(pix, draw, u);
            int x, y;
            pixel_xy(pix, v.width, v.height, x, y);
            Ray r = camera_ray(v.cam, x - v.width / 2 + u[0], y - v.height / 2 + u[1]);   // kernel.cu:73
            ro[h] = r.o; rd[h] = r.d;
            S[h] = State{};
            S[h].draw = draw + 1u;
            S[h].beta = mk3(1.0f, 1.0f, 1.0f);   // PathTracer.cu:19-20
            S[h].res = mk3(0.0f, 0.0f, 0.0f);
        }
        pair_set(w, ro, rd, inf2, none2, alive, E);
    }

    int bounce = 0, j = -1;
    for (;;) {
        S[0].save(st);
        S[1].save(st + State::kSlots * kBlock);
        asm volatile("" ::: "memory");
        Stamps sp0;
        // (two copies of the traversal: shadow rays of the occlusion form visit children in index
        // order and keep them as a mask -- cheaper bookkeeping per node group; C2 +1.9 %)
        // (and a third: rays a surface scattered, and the shadow rays cast where they land, walk the tree
        // lane by lane -- see traverse_lane; in the cooperative form every wave then does the same work)
        if (!DEPTH0 && !EMIT && bounce > 0 && A.lane_traversal)
            traverse_lane<CULL>(A.T, env.s_lvl, w, j < 0 ? alive[0] : cast[0], j < 0 ? alive[1] : cast[1], cost);
        else if (j >= 0 && A.occlusion)
            traverse<CULL, COOP, false>(A.T, env.lvloff, w, lane, wib, s_best, cost, sp0);
        else
            traverse<CULL, COOP>(A.T, env.lvloff, w, lane, wib, s_best, cost, sp0);
        asm volatile("" ::: "memory");
        S[0].load(st);
        S[1].load(st + State::kSlots * kBlock);
        if (j < 0) {
            // ---- closest hit of this bounce: PathTracer.cu:27-64
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float t = h ? w.best.y : w.best.x;
                const int32_t id = h ? w.idy : w.idx;
                ishit[h] = alive[h] && scene_ishit(t) && id >= 0;
                S[h].le = mk3(0.0f, 0.0f, 0.0f);
                if (alive[h]) {
                    const uint32_t p = pixel_of(h);
                    if (bounce == 0 && writer) {             // id / t buffers of the primary hit
                        v.first_id[p] = id;
                        v.first_t[p] = t;
                    }
                    // draws of this bounce: tmp (PathTracer.cu:27) + one per light (Scene.cu:87)
                    S[h].draw += 1u + (ishit[h] ? (uint32_t)L : 0u);
                    if (!ishit[h]) {                         // PathTracer.cu:64 -> return res
                        if (writer) finalize_path(v, p, S[h].res, cnt);
                        alive[h] = false;
                    }
                }
                if (ishit[h]) {
                    Ray r;
                    r.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
                    r.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
                    Hit H;
                    fill_hit(v, r, id, H);
                    S[h].pos = H.pos; S[h].normal = H.normal; S[h].tangent = H.tangent;
                    S[h].model = H.model;
                    S[h].wo_y = local_y(H, neg(r.d));        // Object.cu:107-110 on wo = -r.dir
                }
                RT_SCHED_FENCE();
            }
        } else {
            // ---- verdict of the shadow rays towards light j: Scene.cu:225-241
            // blocked = |light_rec.pos - r.o| < |r.o - lpos| - 0.01
            // (H3: `lightidx == idx` restated as never true for point lights)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if (!cast[h])
                    continue;
                const float best = h ? w.best.y : w.best.x;
                const int32_t sid = h ? w.idy : w.idx;
                bool blocked = false;
                if (scene_ishit(best) && sid >= 0) {
                    Ray sr;
                    sr.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
                    sr.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
                    v3 hp = sid < v.n_tris ? ray_pos(sr, best)
                                           : hit_pos_sphere(v.spheres[sid - v.n_tris], sr, best);
                    blocked = length(sub(hp, sr.o)) < S[h].thr;
                }
                if (!blocked)                                // res += ... in light order (Scene.cu:120)
                    S[h].le = add(S[h].le, S[h].contrib);
            }
        }

        // ---- the next light some lane of the chunk casts a shadow ray towards
        bool more = false;
        if (__ballot(ishit[0] | ishit[1])) {
            for (j = j + 1; j < L; j++) {
                const LightRec light = load_uniform(&v.lights[j]);
                v3 so[2], sd[2];
                float sb[2], ss[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    sb[h] = kInf; ss[h] = -1.0f;
                    if (ishit[h]) {
                        LightSample s = sample_point_light(S[h].hit(), v.models[S[h].model], S[h].wo_y, light);
                        cast[h] = s.cast;
                        so[h] = s.sr.o; sd[h] = s.sr.d;
                        // Scene.cu:227 rhs; a negative one (hit within 0.01 of the light)
                        // can never be undercut
                        S[h].thr = fmaxf(s.thr, 0.0f);
                        S[h].contrib = s.contrib;
                        if (A.occlusion) {
                            float m = 1e-4f * (fabsf(S[h].thr) + fabsf(so[h].x) + fabsf(so[h].y) + fabsf(so[h].z)) + 1e-6f;
                            sb[h] = fminf(kInf, S[h].thr + m);
                            ss[h] = S[h].thr - m;
                        }
                    }
                    RT_SCHED_FENCE();
                }
                const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                if (c0 | c1) {
                    n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                    pair_set(w, so, sd, sb, ss, cast, E);
                    more = true;
                    break;
                }
            }
        }
        if (more)
            continue;

        // ---- end of the bounce: PathTracer.cu:73-88
        v3 ro[2], rd[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            ro[h] = mk3(0.0f, 0.0f, 0.0f); rd[h] = mk3(0.0f, 0.0f, 0.0f);
            if (!ishit[h])
                continue;
            const uint32_t p = pixel_of(h);
            S[h].res = add(S[h].res, mul(S[h].beta, S[h].le));           // PathTracer.cu:73
            const ModelRec &M = v.models[S[h].model];
            const Hit H = S[h].hit();
            float u[4];
            // tmp of PathTracer.cu:27: drawn before this bounce's light draws
            pixel_uniform4(v.pixmap[p], S[h].draw - 1u - (uint32_t)L, u);
            v3 wi;
            float pdf;
            v3 f = rec_sample_f_y(H, M, S[h].wo_y, wi, pdf, u[0], u[1]);   // PathTracer.cu:76
            bool done = (pdf < 0.0001f || len_below_1e4(f));             // PathTracer.cu:77
            if (!done) {
                S[h].beta = mul(S[h].beta, divs(scale(f, fabsf(dot(wi, H.normal))), pdf));   // :79
                Ray nr = spawn_ray(H, wi);                               // :80
                ro[h] = nr.o; rd[h] = nr.d;
                if (bounce > 3) {                                        // :82-88
                    if (u[2] < 0.5f)
                        done = true;
                    else
                        S[h].beta = divs(S[h].beta, 1.0f - 0.5f);
                }
            }
            if (!done && (DEPTH0 || bounce + 1 > A.max_depth)) {
                // Next iteration: one draw (:27), a trace whose result nothing reads
                // (Lambertian, no environment light), then the break of :64.
                S[h].draw += 1u;
                done = true;
            }
            if (done) {
                if (writer) finalize_path(v, p, S[h].res, cnt);
                alive[h] = false;
            }
            ishit[h] = false;
            RT_SCHED_FENCE();
        }
        if (EMIT) {
            // every path that goes on becomes a job of k_bounce (bounce is 0 here: the loop ends below)
            const unsigned long long e0 = __ballot(alive[0]), e1 = __ballot(alive[1]);
            if (e0 | e1) {
                const uint32_t ne = (uint32_t)__popcll(e0) + (uint32_t)__popcll(e1);
                uint32_t base = 0;
                if (writer && lane == 0)
                    base = atomicAdd(&A.bq.ctr[0], ne);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                n_bounce += ne;
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (!alive[h])
                        continue;
                    const uint32_t idx = base + (h ? (uint32_t)__popcll(e0) : 0u) + (uint32_t)__popcll((h ? e1 : e0) & below);
                    if (writer && idx < A.bq.cap) {
                        float4 *J = (float4 *)&A.bq.jobs[idx];
                        J[0] = make_float4(ro[h].x, ro[h].y, ro[h].z, rd[h].x);
                        J[1] = make_float4(rd[h].y, rd[h].z, S[h].beta.x, S[h].beta.y);
                        J[2] = make_float4(S[h].beta.z, S[h].res.x, S[h].res.y, S[h].res.z);
                        J[3] = make_float4(__uint_as_float(S[h].draw), __uint_as_float(pixel_of(h)), 0.0f, 0.0f);
                    }
                    emitted[h] = true;
                    alive[h] = false;
                }
            }
        }
        const unsigned long long a0 = __ballot(alive[0]), a1 = __ballot(alive[1]);
        if (DEPTH0 || !(a0 | a1))
            break;
        n_bounce += (uint32_t)__popcll(a0) + (uint32_t)__popcll(a1);
        bounce++;
        j = -1;
        pair_set(w, ro, rd, inf2, none2, alive, E);
    }

    if (writer) {
#pragma unroll
        for (int h = 0; h < 2; h++)
            if (mine[h] && !emitted[h])
                v.rng_ctr[pixel_of(h)] = S[h].draw;
    }
    if (COOP) {
        if (lane == 0) atomicAdd(s_cost, cost);
        __syncthreads();
        cost = *s_cost;
        cost_sum += cost;
        if (wib != 0)
            return;
    } else {
        cost_sum += cost;
    }
#ifdef RT_TRAV_LOG
    if (lane == 0 && g_log_p0 == A.p0 && chunk < (1u << 17))
        g_log[chunk] = ChunkLog{tlog0, (unsigned long long)wall_clock64(), cost, COOP ? 1u : 0u};
#endif
    if (lane == 0) {
        if (n_shadow) atomicAdd(&A.ctl->shadow_total, (unsigned long long)n_shadow);
        if (n_bounce) atomicAdd(&A.ctl->bounce_total, (unsigned long long)n_bounce);
        if (COOP) atomicAdd(&A.ctl->coop_total, 1ull);
    }
}

// A chunk's run of A.n_samples progressive samples on one wave (or block, COOP): sample cnt + 1 of a
// pixel starts from the counter and the running mean sample cnt left, both written and read back by
// the same lane.  Launches over deep trees (feedback, cooperative chunks: other waves of the block
// read what wave 0 wrote) keep n_samples = 1; shallow ones (C1, C2) render a whole rt_render call in
// one launch -- no launch gaps and no tail per sample, which is most of a sample's time on a small
// partition.  The chunk's feedback is recorded once, with the run's summed cost.
template <bool CULL, bool COOP, bool DEPTH0, bool EMIT = false>
__device__ __forceinline__ void path_run(const PathArgs &A, const ChunkEnv &env, uint32_t chunk, uint32_t lane,
                                         uint32_t wib, unsigned long long *s_best, uint32_t *s_cost, float *st)
{
    uint32_t cost = 0;
    for (uint32_t s = 0; s < A.n_samples; s++)
        path_chunk<CULL, COOP, DEPTH0, EMIT>(A, env, chunk, lane, wib, s_best, s_cost, st, A.v.cnt + (int)s, cost);
    if (lane == 0 && (!COOP || wib == 0) && env.feedback)
        record_cost(A.fb, env, chunk, cost);
}

// ------------------------------------------------------------------ the general path
// The same chunk of 128 pixels for scenes beyond the bench configurations' Lambertian models under
// delta lights: TriangleLights and every BRDF of material.cu.  What changes
// against path_chunk:
//   * every ray's closest hit also tests the triangle lights, after the models (traverse<.., LIGHTS>);
//   * a hit on a light adds its emission for camera rays and rays off specular surfaces
//     (PathTracer.cu:33-52); a light's surface is shaded like any other (its own Lambertian
//     Material, the unnormalised normal of Light.cu:72,97);
//   * Scene::sampleAllLight (Scene.cu:69-124) goes on after the delta lights with one MIS
//     estimate per triangle light (Scene.cu:205-287, isDelta = false): a shadow ray towards a
//     point sampled on the light (phase A) and a ray along the BRDF's own sample (phase B), both
//     closest-hit queries whose verdict needs `lightidx == idx`;
//   * Material::f / sample_f / PDF dispatch over Lambertian, GGX, Cook_Torrance, Oren_Nayar and
//     Fresnel (Object.cu:3-104); specular surfaces skip sampleAllLight (PathTracer.cu:67-74), and
//     the ray that leaves one at the depth limit is still traced when there is a light it can land
//     on (PathTracer.cu:31-52 run before the break of :64);
//   * no occlusion shortcut and no LDS stash: this is the feature path, kept plain.  The bench
//     configurations never come here.
// Paths of one wave end at very different bounces (most camera rays leave the scene, a few bounce
// to the depth limit), and an iteration costs the same however few of its 128 slots are alive.  So
// the wave is persistent and, over a shallow tree (A.regen), REGENERATES paths: a slot whose path
// has ended takes the next pixel of the wave's current chunk (a new chunk from the dispenser when
// that is used up) and starts its camera ray in the same iteration in which its neighbours trace
// their bounce rays; the bounce number is per slot.  Which slot renders a pixel does not matter to
// the result: the RNG stream and the running mean are the pixel's own.  Over a deeper tree a wave's
// traversal costs what the union of its rays visits, and camera rays from two distant chunks visit
// twice as much as those of one (measured: -25 % on the 410-triangle test scenes), so there the wave
// takes a whole new chunk only when all of its slots are dead.  The wave leaves when the dispensers
// are empty and all of its slots are dead -- every path ends after at most max_depth + 2 iterations.
// Chunks come from one dispenser per XCD, each over its own eighth of the range first (a band of the
// image, hence a part of the tree its L2 can hold), the others' when that is used up.
struct GeneralState {
    int bounce;                      // PathTracer.cu:25 `bounces` of this slot's path
    uint32_t draw0;                  // Philox draw of this bounce's tmp (PathTracer.cu:27); light q draws draw0 + 1 + q
    v3 wo;                           // -r.dir of the ray that found the hit
    Hit H;
    v3 le, cur, contrib;             // sampleAllLight's res; the current light's res; the term under test
    float thr;
    bool go_on;                      // the current area light passed `pdf_light >= 0.001` (Scene.cu:233)
    bool from_specular;              // the ray being traced left a specular surface (specular_bounce)
    bool last;                       // ... at the depth limit: only its emission is still read
    v3 wi;                           // PathTracer.cu:18 `wi`: what Cook_Torrance::sample_f reads on entry (H9)
    v3 beta, res;
};

template <bool CULL>
__device__ __forceinline__ void general_wave(const PathArgs &A, const ChunkEnv &env, uint32_t *__restrict__ work,
                                             uint32_t n_chunks, uint32_t lane)
{
    const DevView &v = A.v;
    const int Ld = v.n_lights, La = v.n_tlights, Ne = v.env.n, L = Ld + La + Ne;
    const int n_models_prims = v.n_tris + v.n_spheres;       // first light id
    const float E = A.T.tv.extent;
    uint32_t cost = 0, n_shadow = 0, n_bounce = 0;
    Stamps sp0;
    uint32_t pl[2] = {A.p0, A.p0};                           // local pixel of the slot's path
    auto pixel_of = [&](int h) { return pl[h]; };
    bool alive[2] = {false, false}, lit[2] = {false, false}, ishit[2] = {false, false}, cast[2] = {false, false};
    GeneralState S[2];
    uint32_t draw[2] = {0u, 0u};
    uint32_t q_next = 0u, q_end = 0u;                        // wave-uniform: pixels of the current chunk not handed out yet
    bool exhausted = false;
    // (regenerating waves share one dispenser: no tree to keep in an L2, and image bands of unequal cost
    // would only leave XCDs waiting for one another -- measured on the reference's demo scene)
    const uint32_t per = A.regen ? n_chunks : (n_chunks + 7u) / 8u, xcd = A.regen ? 0u : blockIdx.x & 7u;
    uint32_t stage = 0u;                                     // dispenser (xcd + stage) & 7 is the one in use
    auto next_chunk = [&]() -> uint32_t {                    // wave-uniform; n_chunks: none left
        for (; stage < 8u; stage++) {
            const uint32_t y = (xcd + stage) & 7u, base = y * per;
            const uint32_t size = base < n_chunks ? min(per, n_chunks - base) : 0u;
            if (!size)
                continue;
            uint32_t d = 0u;
            if (lane == 0)
                d = atomicAdd(&work[y], 1u);
            d = __builtin_amdgcn_readfirstlane(d);
            if (d < size)
                return base + d;
        }
        return n_chunks;
    };
    const float inf2[2] = {kInf, kInf}, none2[2] = {-1.0f, -1.0f};
    RayPair w;
    bool scattered = false;          // wave-uniform: no slot of the wave is on its camera ray any more
    auto trace = [&](const v3 o[2], const v3 d[2], const bool live[2]) {
        pair_set(w, o, d, inf2, none2, live, E);
        if (scattered && A.lane_traversal)       // rays a surface scattered, and what is cast where they land
            traverse_lane<CULL, true>(A.T, env.s_lvl, w, live[0], live[1], cost);
        else
            traverse<CULL, false, true, true>(A.T, env.lvloff, w, lane, 0u, nullptr, cost, sp0);
    };
    auto ray_of = [&](int h) {
        Ray r;
        r.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
        r.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
        return r;
    };
    // rec.pos of the closest hit of the ray just traced (Scene.cu:227 reads it)
    auto hit_pos = [&](const Ray &sr, float best, int32_t sid) {
        return (sid >= v.n_tris && sid < n_models_prims) ? hit_pos_sphere(v.spheres[sid - v.n_tris], sr, best)
                                                         : ray_pos(sr, best);
    };
    v3 ro[2] = {mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)}, rd[2] = {mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
    S[0] = GeneralState{}; S[1] = GeneralState{};
    auto end_path = [&](int h) {                             // kernel.cu:75-77, and the stream's position for the next sample
        finalize_path(v, pl[h], S[h].res, v.cnt);
        v.rng_ctr[pl[h]] = draw[h];
        alive[h] = false;
    };
    for (;;) {
        // ---- dead slots take the next pixels (kernel.cu:50-74 up to the call of pathTracer)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            unsigned long long dead = __ballot(!alive[h]);
            if (!A.regen && (__ballot(alive[0]) | __ballot(alive[1])) && q_next == q_end)
                break;                                       // a whole new chunk, once every slot is dead
            while (dead && !exhausted) {
                if (q_next == q_end) {
                    const uint32_t d = next_chunk();
                    if (d >= n_chunks) {
                        exhausted = true;
                        break;
                    }
                    q_next = d * kPathChunk;
                    q_end = min(q_next + kPathChunk, A.n);
                }
                const uint32_t rank = (uint32_t)__popcll(dead & ((1ull << lane) - 1ull));
                const uint32_t avail = q_end - q_next;
                if (!alive[h] && rank < avail) {
                    pl[h] = A.p0 + q_next + rank;
                    const uint32_t p = pl[h], pix = v.pixmap[p];
                    float u[4];
                    draw[h] = v.rng_ctr[p];
                    pixel_uniform4(pix, draw[h], u);         // kernel.cu:67 (H4)
                    draw[h] += 1u;
                    int x, y;
            pixel_xy(pix, v.width, v.height, x, y);
                    Ray r = camera_ray(v.cam, x - v.width / 2 + u[0], y - v.height / 2 + u[1]);   // kernel.cu:73
                    ro[h] = r.o; rd[h] = r.d;
                    S[h] = GeneralState{};
                    S[h].beta = mk3(1.0f, 1.0f, 1.0f);
                    S[h].res = mk3(0.0f, 0.0f, 0.0f);
                    S[h].wi = mk3(0.0f, 0.0f, 0.0f);         // H9
                    alive[h] = true;
                }
                q_next += min((uint32_t)__popcll(dead), avail);
                dead = __ballot(!alive[h]);
            }
        }
        if (!(__ballot(alive[0]) | __ballot(alive[1])))
            break;                                           // nothing left to hand out, every path ended
        scattered = !(__ballot(alive[0] && S[0].bounce == 0) | __ballot(alive[1] && S[1].bounce == 0));
        // ---- closest hit of this bounce: PathTracer.cu:27-64
        trace(ro, rd, alive);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float t = h ? w.best.y : w.best.x;
            const int32_t id = h ? w.idy : w.idx;
            ishit[h] = alive[h] && scene_ishit(t) && id >= 0;
            lit[h] = false;
            S[h].le = mk3(0.0f, 0.0f, 0.0f);
            if (!alive[h])
                continue;
            const uint32_t p = pixel_of(h);
            const int bounce = S[h].bounce;
            if (bounce == 0) {
                v.first_id[p] = id;
                v.first_t[p] = t;
            }
            const Ray r = ray_of(h);
            const bool on_light = ishit[h] && id >= n_models_prims;
            if (on_light && (bounce == 0 || S[h].from_specular))             // PathTracer.cu:33-47
                S[h].res = add(S[h].res, mul(S[h].beta, light_L(v.tlights[id - n_models_prims], neg(r.d))));
            if (!ishit[h] && Ne && (bounce == 0 || S[h].from_specular))      // PathTracer.cu:53-61
                S[h].res = add(S[h].res, mul(S[h].beta, env_L(v.env, r.d)));
            if (S[h].last) {                                 // traced for its emission only: the break of :64
                ishit[h] = false;
                end_path(h);
                continue;
            }
            S[h].draw0 = draw[h];
            draw[h] += 1u;                                   // tmp (PathTracer.cu:27)
            if (!ishit[h]) {                                 // PathTracer.cu:64
                end_path(h);
                continue;
            }
            if (on_light)
                fill_light_hit(v.tlights[id - n_models_prims], r, t, S[h].H);
            else
                fill_hit(v, r, id, S[h].H);
            if (v.tex.n) {
                // Mesh::hit's texture lookup (Model.cu:301-310): the record carries the colour of the last
                // mesh that updated it -- the mesh of the closest triangle -- also when a quadric or a
                // light in front of it is what the ray hits (nothing resets rec.color, Model.cu:396-406)
                const int32_t tid = h ? w.tidy : w.tidx;
                if (tid >= v.tex.n_tri_models) {
                    const int32_t tx = v.tex.mesh_tex[v.tri_cold[tid].model - v.tex.n_tri_models];
                    if (tx >= 0)
                        S[h].H.color = mesh_texture_color(v.tex, v.tex.recs[tx], v.cam, v.tris[v.tri_pos_of[tid]], v.tex.uv[tid], r);
                }
            }
            S[h].wo = neg(r.d);
            lit[h] = !(v.models[S[h].H.model].mat_type & kMatSpecularBit);   // PathTracer.cu:67-70
            if (lit[h])
                draw[h] += (uint32_t)L;                      // one draw per light (Scene.cu:87)
        }
        if (__ballot(lit[0] | lit[1])) {
            // ---- Scene::sampleAllLight: delta lights (Scene.cu:93-102)
            for (int j = 0; j < Ld; j++) {
                const LightRec light = load_uniform(&v.lights[j]);
                const v3 lp = mk3(light.pos[0], light.pos[1], light.pos[2]),
                         illum = mk3(light.illum[0], light.illum[1], light.illum[2]);
                v3 so[2], sd[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    if (!lit[h])
                        continue;
                    const Hit &H = S[h].H;
                    Ray wi;
                    v3 color;
                    if (light.type == kLightPoint) {
                        wi = make_ray(lp, sub(H.pos, lp));                            // Light.cu:9
                        v3 dd = sub(H.pos, lp);
                        color = divs(illum, fmaxf(0.001f, dot(dd, dd)));              // Light.cu:13
                    } else {
                        wi = make_ray(sub(H.pos, scale(lp, 10000.0f)), lp);           // Light.cu:25
                        color = illum;                                                // Light.cu:27
                    }
                    const Ray sr = spawn_ray(H, neg(wi.d));                           // Scene.cu:218
                    const v3 f = scale(brdf_f(H, v.models[H.model], v.mats[H.model], S[h].wo, sr.d),
                                       fabsf(dot(H.normal, sr.d)));                   // Scene.cu:222
                    cast[h] = !is_black(f);                                           // Scene.cu:224
                    so[h] = sr.o; sd[h] = sr.d;
                    S[h].thr = length(sub(sr.o, wi.o)) - 0.01f;                       // Scene.cu:227 rhs
                    S[h].contrib = is_black(color) ? mk3(0.0f, 0.0f, 0.0f) : divs(mul(color, f), 1.0f);   // :236-240
                }
                const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                if (!(c0 | c1))
                    continue;
                n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                trace(so, sd, cast);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (!cast[h])
                        continue;
                    const float best = h ? w.best.y : w.best.x;
                    const int32_t sid = h ? w.idy : w.idx;
                    bool blocked = false;                    // H3: `lightidx == idx` never true for delta lights
                    if (scene_ishit(best) && sid >= 0) {
                        const Ray sr = ray_of(h);
                        blocked = length(sub(hit_pos(sr, best, sid), sr.o)) < S[h].thr;   // Scene.cu:227
                    }
                    if (!blocked)
                        S[h].le = add(S[h].le, S[h].contrib);
                }
            }
            // ---- triangle lights: Scene.cu:103-108 -> evaluateDirectLight(.., idx = k, isDelta = false)
            for (int k = 0; k < La; k++) {
                const TriLightRec &Lr = v.tlights[k];
                const v3 p0 = mk3(Lr.T.ax, Lr.T.ay, Lr.T.az), p1 = mk3(Lr.T.bx, Lr.T.by, Lr.T.bz),
                         p2 = mk3(Lr.T.cx, Lr.T.cy, Lr.T.cz);
                v3 so[2], sd[2];
                // phase A: TriangleLight::lightIllumi (Light.cu:47-65), shadow ray, light term
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    S[h].cur = mk3(0.0f, 0.0f, 0.0f);
                    S[h].contrib = mk3(0.0f, 0.0f, 0.0f);
                    S[h].go_on = false;
                    if (!lit[h])
                        continue;
                    float u[4];
                    pixel_uniform4(v.pixmap[pixel_of(h)], S[h].draw0 + 1u + (uint32_t)(Ld + k), u);   // Scene.cu:87-89
                    const Hit &H = S[h].H;
                    const ModelRec &M = v.models[H.model];
                    const MatRec &P = v.mats[H.model];
                    const float sq_x = sqrtf(u[0]);                                   // Light.cu:49
                    v3 ts = mk3(1 - sq_x, u[1] * sq_x, 0.0f);
                    ts.z = 1.0f - ts.x - ts.y;
                    const v3 lp = add(add(scale(p0, ts.x), scale(p1, ts.y)), scale(p2, ts.z));   // Light.cu:44
                    const Ray wi = make_ray(lp, sub(H.pos, lp));                      // Light.cu:56
                    const float pdf_light = light_pdf(Lr, length(sub(H.pos, lp)), wi.d);   // Light.cu:58-61
                    const v3 color = light_L(Lr, wi.d);                               // Light.cu:64
                    const Ray sr = spawn_ray(H, neg(wi.d));                           // Scene.cu:218
                    const v3 f = scale(brdf_f(H, M, P, S[h].wo, sr.d), fabsf(dot(H.normal, sr.d)));   // :222
                    cast[h] = !is_black(f);                                           // :224
                    so[h] = sr.o; sd[h] = sr.d;
                    S[h].thr = length(sub(sr.o, lp)) - 0.01f;                         // :227 rhs
                    S[h].go_on = !(pdf_light < 0.001f);                               // :233
                    if (S[h].go_on && cast[h] && !is_black(color)) {                  // :236
                        const float pdf_surface = brdf_pdf(H, M, P, S[h].wo, sr.d);   // :244
                        float weight = pdf_light * pdf_light / (pdf_light * pdf_light + pdf_surface * pdf_surface);   // :245
                        weight /= pdf_light;                                          // :246
                        S[h].contrib = scale(mul(color, f), weight);                  // :247
                    }
                }
                {
                    const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                    if (c0 | c1) {
                        n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                        trace(so, sd, cast);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (!cast[h])
                                continue;
                            const float best = h ? w.best.y : w.best.x;
                            const int32_t sid = h ? w.idy : w.idx;
                            bool blocked = false;
                            if (scene_ishit(best) && sid >= 0 && sid != n_models_prims + k) {   // :227 lightidx == idx
                                const Ray sr = ray_of(h);
                                blocked = length(sub(hit_pos(sr, best, sid), sr.o)) < S[h].thr;
                            }
                            if (!blocked)
                                S[h].cur = S[h].contrib;                              // :249 (BLACK when :236 failed)
                        }
                    }
                }
                // phase B: the BRDF's own sample (Scene.cu:253-284)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    if (!lit[h] || !S[h].go_on)
                        continue;
                    float u[4];
                    pixel_uniform4(v.pixmap[pixel_of(h)], S[h].draw0 + 1u + (uint32_t)(Ld + k), u);
                    const Hit &H = S[h].H;
                    v3 wi = mk3(0.0f, 0.0f, 0.0f);                                    // :253 uninitialised: H9
                    float pdf_surface = 0.0f;
                    v3 f = brdf_sample_f(H, v.models[H.model], v.mats[H.model], S[h].wo, wi, pdf_surface, u[2], u[3]);   // :258
                    f = scale(f, fabsf(dot(H.normal, wi)));                           // :259
                    const Ray r = spawn_ray(H, wi);                                   // :260
                    if (is_black(f) || !(pdf_surface > 0.001f))                       // :264
                        continue;
                    const float pdf = light_pdf(Lr, kInf, wi);                        // :268 with rec.t = INF (:262)
                    if (fabsf(pdf) < 0.01f)                                           // :269
                        continue;
                    const float weight = pdf_surface * pdf_surface / (pdf_surface * pdf_surface + pdf * pdf);   // :271
                    S[h].contrib = divs(scale(mul(light_L(Lr, neg(wi)), f), weight), pdf_surface);   // :277,282
                    cast[h] = true;
                    so[h] = r.o; sd[h] = r.d;
                }
                {
                    const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                    if (c0 | c1) {
                        n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                        trace(so, sd, cast);
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (!cast[h])
                                continue;
                            const float best = h ? w.best.y : w.best.x;
                            const int32_t sid = h ? w.idy : w.idx;
                            if (scene_ishit(best) && sid == n_models_prims + k)       // :275
                                S[h].cur = add(S[h].cur, S[h].contrib);               // :282 (else l = getLe = BLACK)
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
                    if (lit[h])
                        S[h].le = add(S[h].le, S[h].cur);                             // Scene.cu:120
            }
            // ---- the environment light: Scene.cu:109-114 -> evaluateDirectLight(.., idx = La, isDelta = false),
            // EnvironmentLight::lightIllumi (Light.cu:137-155).  H10: lpos = (0, 0, 0).  Its second half
            // (Scene.cu:253-284) needs no ray: rec.lightidx never equals idx, so l = getLe(r) whatever the ray hits.
            if (Ne) {
                v3 so[2], sd[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    S[h].cur = mk3(0.0f, 0.0f, 0.0f);
                    S[h].contrib = mk3(0.0f, 0.0f, 0.0f);
                    S[h].go_on = false;
                    if (!lit[h])
                        continue;
                    float u[4];
                    pixel_uniform4(v.pixmap[pixel_of(h)], S[h].draw0 + 1u + (uint32_t)(Ld + La), u);
                    const Hit &H = S[h].H;
                    const ModelRec &M = v.models[H.model];
                    const MatRec &P = v.mats[H.model];
                    float d0, d1, pdf_light;
                    env_dist_sample(v.env, u[0], u[1], d0, d1, pdf_light);             // Light.cu:139
                    const float theta = (float)((double)d1 * kPiD), phi = (float)((double)(2.0f * d0) * kPiD);   // :140
                    const float sint = m_sinf(theta);                                  // :141
                    const Ray wi = make_ray(mk3(0.0f, 0.0f, 0.0f),                     // :142 (H10)
                                            mk3(m_sinf(theta) * m_sinf(phi), m_cosf(theta), m_sinf(theta) * m_cosf(phi)));
                    v3 color = mk3(0.0f, 0.0f, 0.0f);
                    if (!(pdf_light < 0.001f) && !(sint < 0.001f)) {                   // :143-147
                        pdf_light = (float)((double)pdf_light / ((double)(2.0f * sint) * kPiD * kPiD));   // :149
                        color = env_tex2d(v.env, 0.5f * phi * kEnvInvPi, theta * kEnvInvPi);   // :153
                    }
                    const Ray sr = spawn_ray(H, neg(wi.d));                            // Scene.cu:218
                    const v3 f = scale(brdf_f(H, M, P, S[h].wo, sr.d), fabsf(dot(H.normal, sr.d)));   // :222
                    cast[h] = !is_black(f);                                            // :224
                    so[h] = sr.o; sd[h] = sr.d;
                    S[h].thr = length(sub(sr.o, wi.o)) - 0.01f;                        // :227 rhs, lpos = 0
                    S[h].go_on = !(pdf_light < 0.001f);                                // :233
                    if (S[h].go_on && cast[h] && !is_black(color)) {                   // :236
                        const float pdf_surface = brdf_pdf(H, M, P, S[h].wo, sr.d);    // :244
                        float weight = pdf_light * pdf_light / (pdf_light * pdf_light + pdf_surface * pdf_surface);
                        weight /= pdf_light;
                        S[h].contrib = scale(mul(color, f), weight);                   // :247
                    }
                }
                const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                if (c0 | c1) {
                    n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                    trace(so, sd, cast);
                }
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (!lit[h])
                        continue;
                    if (cast[h]) {
                        const float best = h ? w.best.y : w.best.x;
                        const int32_t sid = h ? w.idy : w.idx;
                        bool blocked = false;
                        if (scene_ishit(best) && sid >= 0) {                           // (a triangle light's index never equals idx)
                            const Ray sr = ray_of(h);
                            blocked = length(sub(hit_pos(sr, best, sid), sr.o)) < S[h].thr;
                        }
                        if (!blocked)
                            S[h].cur = S[h].contrib;
                    }
                    if (S[h].go_on) {                                                  // Scene.cu:253-284
                        float u[4];
                        pixel_uniform4(v.pixmap[pixel_of(h)], S[h].draw0 + 1u + (uint32_t)(Ld + La), u);
                        const Hit &H = S[h].H;
                        v3 wi = mk3(0.0f, 0.0f, 0.0f);                                 // H9
                        float pdf_surface = 0.0f;
                        v3 f = brdf_sample_f(H, v.models[H.model], v.mats[H.model], S[h].wo, wi, pdf_surface, u[2], u[3]);
                        f = scale(f, fabsf(dot(H.normal, wi)));
                        const Ray r = spawn_ray(H, wi);
                        if (!is_black(f) && pdf_surface > 0.001f) {
                            const float pdf = env_pdf(v.env, wi);                      // :268
                            if (!(fabsf(pdf) < 0.01f)) {                               // :269
                                const float weight = pdf_surface * pdf_surface / (pdf_surface * pdf_surface + pdf * pdf);
                                S[h].cur = add(S[h].cur, divs(scale(mul(env_L(v.env, r.d), f), weight), pdf_surface));   // :280-282
                            }
                        }
                    }
                    S[h].le = add(S[h].le, S[h].cur);                                  // Scene.cu:120
                }
            }
        }

        // ---- end of the bounce: PathTracer.cu:73-88
#pragma unroll
        for (int h = 0; h < 2; h++) {
            ro[h] = mk3(0.0f, 0.0f, 0.0f); rd[h] = mk3(0.0f, 0.0f, 0.0f);
            if (!ishit[h])
                continue;
            const uint32_t p = pixel_of(h);
            const int bounce = S[h].bounce;
            if (lit[h])
                S[h].res = add(S[h].res, mul(S[h].beta, S[h].le));       // PathTracer.cu:73
            const Hit &H = S[h].H;
            float u[4];
            pixel_uniform4(v.pixmap[p], S[h].draw0, u);                  // tmp of PathTracer.cu:27
            float pdf = 0.0f;
            v3 f = brdf_sample_f(H, v.models[H.model], v.mats[H.model], S[h].wo, S[h].wi, pdf, u[0], u[1]);   // :76
            const v3 wi = S[h].wi;
            bool done = (pdf < 0.0001f || len_below_1e4(f));             // PathTracer.cu:77
            if (!done) {
                S[h].beta = mul(S[h].beta, divs(scale(f, fabsf(dot(wi, H.normal))), pdf));   // :79
                Ray nr = spawn_ray(H, wi);                               // :80
                ro[h] = nr.o; rd[h] = nr.d;
                if (bounce > 3) {                                        // :82-88
                    if (u[2] < 0.5f)
                        done = true;
                    else
                        S[h].beta = divs(S[h].beta, 1.0f - 0.5f);
                }
            }
            S[h].from_specular = !lit[h];                                // :67
            if (!done && bounce + 1 > A.max_depth) {
                // next iteration: one draw (:27) and a trace (:31) that only the emission test of
                // :33-52 can read -- off a specular surface, with a light to land on -- then the
                // break of :64
                draw[h] += 1u;
                if (S[h].from_specular && (La > 0 || Ne))
                    S[h].last = true;
                else
                    done = true;
            }
            if (done)
                end_path(h);
            else
                S[h].bounce = bounce + 1;
            ishit[h] = false;
            lit[h] = false;
        }
        n_bounce += (uint32_t)__popcll(__ballot(alive[0])) + (uint32_t)__popcll(__ballot(alive[1]));
    }
    if (lane == 0) {
        if (n_shadow) atomicAdd(&A.ctl->shadow_total, (unsigned long long)n_shadow);
        if (n_bounce) atomicAdd(&A.ctl->bounce_total, (unsigned long long)n_bounce);
    }
}

// Over a deep tree a chunk costs anything from a handful of node tests (rays into empty
// space) to thousands, so waves draw chunks from a dispenser (`dynamic`); over a shallow
// one every chunk is cheap, the same-address atomic would be the bottleneck, and waves
// stride the range instead.  With feedback (ChunkFeedback) the chunks the previous sample
// found heavy go first, one per BLOCK (cooperative form); the rest follow, one per wave.
// Every wave leaves as soon as its next chunk lies past the end.
// WAVES: waves per SIMD the register allocation aims at (2: no spills; 3, 4: spills, all of
// them in the per-bounce shading code, none inside the traversal loops).  More resident
// waves hide the traversal's scalar-load latency, fewer run each chunk faster; which wins
// depends on how many chunks a launch has (launch_path).
// The dispatch every chunked kernel here shares.  solo(chunk) runs one chunk on the calling
// wave, coop(chunk) runs it on the whole block (cooperative form; needs feedback).
template <bool CULL, class Solo, class Coop>
__device__ __forceinline__ void run_chunks(uint32_t n_chunks, const ChunkFeedback &fb,
                                           uint32_t heavy_x4, uint32_t *__restrict__ work,
                                           uint32_t *__restrict__ work_next, int dynamic,
                                           ChunkEnv &env, uint32_t *s_chunk, Solo solo, Coop coop)
{
    const uint32_t lane = threadIdx.x & 63u;
    env.feedback = dynamic && fb.flag_w != nullptr;
    env.heavy_cost = 0xffffffffu;
    env.mean_cost = 0u;
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)kWorkSlots)
        work_next[threadIdx.x] = 0u;     // the next launch's dispensers (stream-ordered after us)
    if (blockIdx.x == 0 && threadIdx.x == 0 && fb.cnt_z) {
        FbCounts z{};                    // and the feedback set the next launch writes
        *fb.cnt_z = z;
    }
    uint32_t n_heavy = 0, n_early_a = 0, n_early_b = 0;
    if (env.feedback) {
        const FbCounts cr = *fb.cnt_r;
        n_heavy = cr.n_heavy; n_early_a = cr.n_early_a; n_early_b = cr.n_early_b;
        if (cr.total) {
            env.mean_cost = (uint32_t)std::max<unsigned long long>(cr.total / std::max(n_chunks, 1u), 12ull);
            env.heavy_cost = (uint32_t)std::max<unsigned long long>(
                (unsigned long long)heavy_x4 * cr.total / (4ull * std::max(n_chunks, 1u)), 96ull);
        }
    }
    const bool ordered = CULL && (n_heavy | n_early_a | n_early_b);

    // phase 1: the previous launch's heavy chunks, one per block
    if (CULL && n_heavy) {
        for (;;) {
            if (threadIdx.x == 0)
                *s_chunk = atomicAdd(&work[kWorkHeavy], 1u);
            __syncthreads();
            const uint32_t d = *s_chunk;
            __syncthreads();
            if (d >= n_heavy)
                break;
            coop(fb.heavy_r[d]);
        }
    }
    // phase 2, one chunk per wave at a time: first the previous launch's early chunks,
    // costliest class first (stage 0), then everything else (stages 1..8).  Workgroups go
    // to the XCDs round-robin (blockIdx % 8), and every XCD has its own L2: each XCD works
    // through its own eighth of the range first -- a band of the image, hence a part of the
    // tree its L2 can hold -- and helps with the others' only when that is used up.
    if (dynamic) {
        const uint32_t per = (n_chunks + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u;
        const uint32_t n_early = n_early_a + n_early_b;
        for (uint32_t stage = n_early ? 0u : 1u; stage <= 8u; stage++) {
            const uint32_t y = (xcd + stage - 1u) & 7u;              // stages 1..8
            const uint32_t base = y * per;
            const uint32_t size = stage == 0u ? n_early : base < n_chunks ? min(per, n_chunks - base) : 0u;
            uint32_t *counter = &work[stage == 0u ? (uint32_t)kWorkEarly : y];
            for (;;) {
                uint32_t d = size;
                if (size) {
                    if (lane == 0)
                        d = atomicAdd(counter, 1u);
                    d = __builtin_amdgcn_readfirstlane(d);
                }
                if (d >= size)
                    break;
                uint32_t chunk = base + d;
                if (stage == 0u)
                    chunk = d < n_early_a ? fb.early_r[d] : fb.early_r[fb.n_slots - 1u - (d - n_early_a)];
                else if (ordered) {
                    const uint8_t fl = fb.flag_r[chunk];
                    if ((fl == 1 && n_heavy) || (fl == 2 && n_early))
                        continue;                // went out with the heavy / early ones
                }
                solo(chunk);
            }
        }
    } else {
        const uint32_t n_waves = (gridDim.x * kBlock) >> 6;
        for (uint32_t chunk = (blockIdx.x * kBlock + threadIdx.x) >> 6; chunk < n_chunks; chunk += n_waves)
            solo(chunk);
    }
}

template <bool CULL, bool DEPTH0, int WAVES, bool EMIT = false>
__global__ __launch_bounds__(kBlock, WAVES) void k_path(const PathArgs A, uint32_t *__restrict__ work,
                                                 uint32_t *__restrict__ work_next, int dynamic)
{
    __shared__ unsigned long long s_best[128];
    __shared__ uint32_t s_cost, s_chunk;
    __shared__ float s_stash[2 * RayState<DEPTH0>::kSlots][kBlock];
    __shared__ uint32_t s_lvl[kMaxLevels];
    float *st = &s_stash[0][threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const uint32_t n_chunks = (A.n + kPathChunk - 1u) / kPathChunk;
    ChunkEnv env;
    env.lvloff = A.T.tv.level_off[lane & (kMaxLevels - 1)];
    env.s_lvl = s_lvl;
    if (!DEPTH0) {
        if (threadIdx.x < (uint32_t)kMaxLevels)
            s_lvl[threadIdx.x] = A.T.tv.level_off[threadIdx.x];
        __syncthreads();
    }
    if (EMIT && blockIdx.x == 0 && threadIdx.x < 2u)
        A.bq.ctr_next[threadIdx.x] = 0u;                     // the next sample's counters (stream-ordered after us and our k_bounce)
    run_chunks<CULL>(
        n_chunks, A.fb, A.heavy_x4, work, work_next, dynamic, env, &s_chunk,
        [&](uint32_t chunk) { path_run<CULL, false, DEPTH0, EMIT>(A, env, chunk, lane, wib, nullptr, nullptr, st); },
        [&](uint32_t chunk) { path_run<CULL, true, DEPTH0, EMIT>(A, env, chunk, lane, wib, s_best, &s_cost, st); });
}

// The paths k_path<.., EMIT> handed on, from their first bounce ray to their end (PathTracer.cu:22-88 from the second
// iteration on).  Scattered rays share nothing, so everything here is per lane: the traversal (traverse_lane), the
// bounce a lane is at, and its end -- a lane (two slots each) whose path is over takes the next job of the queue at
// the next bounce boundary, so the waves stay full however differently the paths end (in the one-kernel form a
// wave carried its 128 pixels until the last of them had ended: at the third bounce one slot in seven was alive).
// Same arithmetic, same order per path as path_chunk; the pixel's running mean and counter are written at the
// path's end, once.
template <bool CULL>
__global__ __launch_bounds__(kBlock, 3) void k_bounce(const PathArgs A)
{
    using State = RayState<false>;
    __shared__ float s_stash[2 * State::kSlots][kBlock];
    __shared__ uint32_t s_lvl[kMaxLevels];
    float *st = &s_stash[0][threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < (uint32_t)kMaxLevels)
        s_lvl[threadIdx.x] = A.T.tv.level_off[threadIdx.x];
    __syncthreads();
    const DevView &v = A.v;
    const int L = v.n_lights;
    const float E = A.T.tv.extent;
    const uint32_t n_jobs = min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.bq.ctr[0]), A.bq.cap);
    const float inf2[2] = {kInf, kInf}, none2[2] = {-1.0f, -1.0f};
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t cost = 0, n_shadow = 0, n_bounce = 0;
    bool alive[2] = {false, false}, more_jobs = n_jobs != 0u;
    uint32_t pl[2] = {0u, 0u};
    int bnc[2] = {0, 0};
    State S[2];
    S[0] = State{}; S[1] = State{};
    v3 ro[2] = {mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)}, rd[2] = {mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
    RayPair w;
    for (;;) {
        // ---- free slots take jobs
        const unsigned long long d0 = __ballot(!alive[0]), d1 = A.bq.pairs ? __ballot(!alive[1]) : 0ull;
        if (more_jobs && (d0 | d1)) {
            const uint32_t nd = (uint32_t)__popcll(d0) + (uint32_t)__popcll(d1);
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&A.bq.ctr[1], nd);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base + nd >= n_jobs)
                more_jobs = false;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t idx = base + (h ? (uint32_t)__popcll(d0) : 0u) + (uint32_t)__popcll((h ? d1 : d0) & below);
                if (!alive[h] && (h == 0 || A.bq.pairs) && idx < n_jobs) {
                    const float4 *J = (const float4 *)&A.bq.jobs[idx];
                    const float4 j0 = J[0], j1 = J[1], j2 = J[2], j3 = J[3];
                    ro[h] = mk3(j0.x, j0.y, j0.z); rd[h] = mk3(j0.w, j1.x, j1.y);
                    S[h].beta = mk3(j1.z, j1.w, j2.x); S[h].res = mk3(j2.y, j2.z, j2.w);
                    S[h].draw = __float_as_uint(j3.x);
                    pl[h] = __float_as_uint(j3.y);
                    bnc[h] = 1;
                    alive[h] = true;
                }
            }
        }
        if (!__ballot(alive[0] | alive[1]))
            break;
        // ---- closest hit of this bounce: PathTracer.cu:27-64
        pair_set(w, ro, rd, inf2, none2, alive, E);
        S[0].save(st);
        S[1].save(st + State::kSlots * kBlock);
        asm volatile("" ::: "memory");
        traverse_lane<CULL>(A.T, s_lvl, w, alive[0], alive[1], cost);
        asm volatile("" ::: "memory");
        S[0].load(st);
        S[1].load(st + State::kSlots * kBlock);
        bool ishit[2], cast[2] = {false, false};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float t = h ? w.best.y : w.best.x;
            const int32_t id = h ? w.idy : w.idx;
            ishit[h] = alive[h] && scene_ishit(t) && id >= 0;
            S[h].le = mk3(0.0f, 0.0f, 0.0f);
            if (alive[h]) {
                // draws of this bounce: tmp (PathTracer.cu:27) + one per light (Scene.cu:87)
                S[h].draw += 1u + (ishit[h] ? (uint32_t)L : 0u);
                if (!ishit[h]) {                             // PathTracer.cu:64 -> return res
                    finalize_path(v, pl[h], S[h].res, v.cnt);
                    v.rng_ctr[pl[h]] = S[h].draw;
                    alive[h] = false;
                }
            }
            if (ishit[h]) {
                Ray r;
                r.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
                r.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
                Hit H;
                fill_hit(v, r, id, H);
                S[h].pos = H.pos; S[h].normal = H.normal; S[h].tangent = H.tangent;
                S[h].model = H.model;
                S[h].wo_y = local_y(H, neg(r.d));            // Object.cu:107-110 on wo = -r.dir
            }
            RT_SCHED_FENCE();
        }
        // ---- the lights, in order: Scene.cu:84-122, shadow rays Scene.cu:225-241
        if (__ballot(ishit[0] | ishit[1])) {
            for (int j = 0; j < L; j++) {
                const LightRec light = load_uniform(&v.lights[j]);
                v3 so[2], sd[2];
                float sb[2], ss[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    cast[h] = false;
                    so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
                    sb[h] = kInf; ss[h] = -1.0f;
                    if (ishit[h]) {
                        LightSample s = sample_point_light(S[h].hit(), v.models[S[h].model], S[h].wo_y, light);
                        cast[h] = s.cast;
                        so[h] = s.sr.o; sd[h] = s.sr.d;
                        S[h].thr = fmaxf(s.thr, 0.0f);
                        S[h].contrib = s.contrib;
                        if (A.occlusion) {
                            float m = 1e-4f * (fabsf(S[h].thr) + fabsf(so[h].x) + fabsf(so[h].y) + fabsf(so[h].z)) + 1e-6f;
                            sb[h] = fminf(kInf, S[h].thr + m);
                            ss[h] = S[h].thr - m;
                        }
                    }
                    RT_SCHED_FENCE();
                }
                const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
                if (!(c0 | c1))
                    continue;
                n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
                pair_set(w, so, sd, sb, ss, cast, E);
                S[0].save(st);
                S[1].save(st + State::kSlots * kBlock);
                asm volatile("" ::: "memory");
                traverse_lane<CULL>(A.T, s_lvl, w, cast[0], cast[1], cost);
                asm volatile("" ::: "memory");
                S[0].load(st);
                S[1].load(st + State::kSlots * kBlock);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (!cast[h])
                        continue;
                    const float best = h ? w.best.y : w.best.x;
                    const int32_t sid = h ? w.idy : w.idx;
                    bool blocked = false;
                    if (scene_ishit(best) && sid >= 0) {
                        Ray sr;
                        sr.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
                        sr.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
                        v3 hp = sid < v.n_tris ? ray_pos(sr, best)
                                               : hit_pos_sphere(v.spheres[sid - v.n_tris], sr, best);
                        blocked = length(sub(hp, sr.o)) < S[h].thr;
                    }
                    if (!blocked)                            // res += ... in light order (Scene.cu:120)
                        S[h].le = add(S[h].le, S[h].contrib);
                }
            }
        }
        // ---- end of the bounce: PathTracer.cu:73-88
#pragma unroll
        for (int h = 0; h < 2; h++) {
            ro[h] = mk3(0.0f, 0.0f, 0.0f); rd[h] = mk3(0.0f, 0.0f, 0.0f);
            if (!ishit[h])
                continue;
            const uint32_t p = pl[h];
            S[h].res = add(S[h].res, mul(S[h].beta, S[h].le));           // PathTracer.cu:73
            const ModelRec &M = v.models[S[h].model];
            const Hit H = S[h].hit();
            float u[4];
            pixel_uniform4(v.pixmap[p], S[h].draw - 1u - (uint32_t)L, u);  // tmp of PathTracer.cu:27
            v3 wi;
            float pdf;
            v3 f = rec_sample_f_y(H, M, S[h].wo_y, wi, pdf, u[0], u[1]);   // PathTracer.cu:76
            bool done = (pdf < 0.0001f || len_below_1e4(f));             // PathTracer.cu:77
            if (!done) {
                S[h].beta = mul(S[h].beta, divs(scale(f, fabsf(dot(wi, H.normal))), pdf));   // :79
                Ray nr = spawn_ray(H, wi);                               // :80
                ro[h] = nr.o; rd[h] = nr.d;
                if (bnc[h] > 3) {                                        // :82-88
                    if (u[2] < 0.5f)
                        done = true;
                    else
                        S[h].beta = divs(S[h].beta, 1.0f - 0.5f);
                }
            }
            if (!done && bnc[h] + 1 > A.max_depth) {
                // Next iteration: one draw (:27), a trace whose result nothing reads
                // (Lambertian, no environment light), then the break of :64.
                S[h].draw += 1u;
                done = true;
            }
            if (done) {
                finalize_path(v, p, S[h].res, v.cnt);
                v.rng_ctr[p] = S[h].draw;
                alive[h] = false;
            } else {
                bnc[h]++;
                n_bounce++;
            }
            RT_SCHED_FENCE();
        }
    }
    // (n_bounce is per lane here)
    for (int o = 32; o > 0; o >>= 1)
        n_bounce += (uint32_t)__shfl_xor((int)n_bounce, o);
    if (lane == 0) {
        if (n_shadow) atomicAdd(&A.ctl->shadow_total, (unsigned long long)n_shadow);
        if (n_bounce) atomicAdd(&A.ctl->bounce_total, (unsigned long long)n_bounce);
    }
}

// WAVES per SIMD the register allocation aims at: 1 keeps the whole state in registers, 2 spills
// ~230 of them but hides twice the traversal's load latency -- faster wherever there is a tree to
// traverse (materials scene 3.8 -> 3.0 ms, area 19.2 -> 11.5), slower with the environment light's
// table walks in the shading code (demo scene with it 1.11 -> 1.34): launch_path picks.
template <bool CULL, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_path_general(const PathArgs A, uint32_t *__restrict__ work,
                                                         uint32_t *__restrict__ work_next, int dynamic)
{
    (void)dynamic;                   // always the dispenser: a wave asks for pixels whenever slots fall free
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = (A.n + kPathChunk - 1u) / kPathChunk;
    __shared__ uint32_t s_lvl[kMaxLevels];
    if (threadIdx.x < (uint32_t)kMaxLevels)
        s_lvl[threadIdx.x] = A.T.tv.level_off[threadIdx.x];
    __syncthreads();
    ChunkEnv env{};
    env.lvloff = A.T.tv.level_off[lane & (kMaxLevels - 1)];
    env.s_lvl = s_lvl;
    env.heavy_cost = 0xffffffffu;
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)kWorkSlots)
        work_next[threadIdx.x] = 0u;     // the next launch's dispensers (stream-ordered after us)
    general_wave<CULL>(A, env, work, n_chunks, lane);
}

// k_bounce as a per-lane STATE MACHINE (RT_BOUNCE_QUEUE=2).  In k_bounce every traversal call ends when its slowest
// lane does, and walks of scattered rays differ by an order of magnitude in length: 11 % of the lanes are active on
// average.  Here the loop's body is ONE step of a walk (a node group or a triangle record), for every lane that is
// walking; a lane whose walk is over waits, and when a quarter of the wave's live lanes wait (or nobody walks) those
// lanes alone go through what follows a walk -- the closest hit's record and the first light, a shadow ray's verdict
// and the next light, the end of the bounce and the next ray, or the end of the path and the next job -- and start
// their next walk.  One path per lane; its state is parked in the lane's LDS column while it walks.  Same device
// functions and the same order per path as path_chunk (lights are read per lane here: lanes are at different ones).
template <bool CULL>
__global__ __launch_bounds__(kBlock, 3) void k_bounce_sm(const PathArgs A)
{
    using State = RayState<false>;
    __shared__ float s_stash[State::kSlots][kBlock];
    __shared__ uint32_t s_lvl[kMaxLevels];
    float *st = &s_stash[0][threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < (uint32_t)kMaxLevels)
        s_lvl[threadIdx.x] = A.T.tv.level_off[threadIdx.x];
    __syncthreads();
    const DevView &v = A.v;
    const TreeView &tv = A.T.tv;
    const int L = v.n_lights;
    const float E = tv.extent;
    const uint32_t n_jobs = min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.bq.ctr[0]), A.bq.cap);
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t n_shadow = 0, n_bounce = 0;
    bool alive = false, walking = false, fresh = false, more_jobs = n_jobs != 0u;
    uint32_t pl = 0;
    int bnc = 0, ph = -1;                                    // ph: -1 the walk is a closest hit, else the shadow ray towards light ph
    int level = 0;
    uint32_t g = 0;
    unsigned long long pend = 0;
    v3 jo = mk3(0.0f, 0.0f, 0.0f), jd = mk3(1.0f, 0.0f, 0.0f);   // a fresh job's ray
    State S;
    S = State{};
    RayPair w;
    {
        const v3 z[2] = {jo, jo}, x[2] = {jd, jd};
        const float b[2] = {0.0f, 0.0f}, sn[2] = {-1.0f, -1.0f};
        const bool no[2] = {false, false};
        pair_set(w, z, x, b, sn, no, E);
    }
    auto expand = [&](int lvl, uint32_t grp) -> uint32_t {
        const float4 *G = (const float4 *)(tv.groups + s_lvl[lvl] + grp);
        const float4 l0 = G[0], l1 = G[1], l2 = G[2], h0 = G[3], h1 = G[4], h2 = G[5];
        uint32_t m = 0;
        bool px, py;
        float ex, ey;
        node_test(w, l0.x, l1.x, l2.x, h0.x, h1.x, h2.x, px, py, ex, ey);
        m |= ((px | py) & !(l0.x > h0.x)) ? 1u : 0u;
        node_test(w, l0.y, l1.y, l2.y, h0.y, h1.y, h2.y, px, py, ex, ey);
        m |= ((px | py) & !(l0.y > h0.y)) ? 2u : 0u;
        node_test(w, l0.z, l1.z, l2.z, h0.z, h1.z, h2.z, px, py, ex, ey);
        m |= ((px | py) & !(l0.z > h0.z)) ? 4u : 0u;
        node_test(w, l0.w, l1.w, l2.w, h0.w, h1.w, h2.w, px, py, ex, ey);
        m |= ((px | py) & !(l0.w > h0.w)) ? 8u : 0u;
        return m;
    };
    for (;;) {
        // ---- who can go on only through the code below?
        const unsigned long long walk_m = __ballot(walking), live_m = __ballot(alive);
        const unsigned long long ready_m = __ballot(!walking && (alive || more_jobs));
        if (!(walk_m | ready_m))
            break;
        const uint32_t n_ready = (uint32_t)__popcll(ready_m);
        if (n_ready && (!walk_m || 4u * n_ready >= (uint32_t)__popcll(live_m) + 4u)) {
            // free lanes take jobs
            const unsigned long long dead_m = __ballot(!alive);
            if (more_jobs && dead_m) {
                const uint32_t nd = (uint32_t)__popcll(dead_m);
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&A.bq.ctr[1], nd);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base + nd >= n_jobs)
                    more_jobs = false;
                const uint32_t idx = base + (uint32_t)__popcll(dead_m & below);
                if (!alive && idx < n_jobs) {
                    const float4 *J = (const float4 *)&A.bq.jobs[idx];
                    const float4 j0 = J[0], j1 = J[1], j2 = J[2], j3 = J[3];
                    jo = mk3(j0.x, j0.y, j0.z); jd = mk3(j0.w, j1.x, j1.y);
                    S = State{};
                    S.beta = mk3(j1.z, j1.w, j2.x); S.res = mk3(j2.y, j2.z, j2.w);
                    S.draw = __float_as_uint(j3.x);
                    pl = __float_as_uint(j3.y);
                    bnc = 1;
                    alive = true;
                    fresh = true;
                    S.save(st);
                }
            }
            if (!walking && alive) {
                bool go = false;                             // a new walk starts: ray (no, nd), window (nb, ns)
                v3 no = jo, nd = jd;
                float nb = kInf, ns = -1.0f;
                S.load(st);
                if (fresh) {                                 // the job's own ray: closest hit of bounce 1
                    fresh = false;
                    ph = -1;
                    go = true;
                } else {
                    after_triangles<false>(A.T, w);          // the quadrics (Scene.cu:31-37)
                    if (ph < 0) {
                        // ---- closest hit of this bounce: PathTracer.cu:27-64
                        const float t = w.best.x;
                        const int32_t id = w.idx;
                        const bool ishit = scene_ishit(t) && id >= 0;
                        S.le = mk3(0.0f, 0.0f, 0.0f);
                        S.draw += 1u + (ishit ? (uint32_t)L : 0u);
                        if (!ishit) {                        // PathTracer.cu:64 -> return res
                            finalize_path(v, pl, S.res, v.cnt);
                            v.rng_ctr[pl] = S.draw;
                            alive = false;
                        } else {
                            Ray r;
                            r.o = mk3(w.ox.x, w.oy.x, w.oz.x);
                            r.d = mk3(w.dx.x, w.dy.x, w.dz.x);
                            Hit H;
                            fill_hit(v, r, id, H);
                            S.pos = H.pos; S.normal = H.normal; S.tangent = H.tangent;
                            S.model = H.model;
                            S.wo_y = local_y(H, neg(r.d));
                            ph = 0;
                        }
                    } else {
                        // ---- verdict of the shadow ray towards light ph: Scene.cu:225-241
                        const float best = w.best.x;
                        const int32_t sid = w.idx;
                        bool blocked = false;
                        if (scene_ishit(best) && sid >= 0) {
                            Ray sr;
                            sr.o = mk3(w.ox.x, w.oy.x, w.oz.x);
                            sr.d = mk3(w.dx.x, w.dy.x, w.dz.x);
                            v3 hp = sid < v.n_tris ? ray_pos(sr, best)
                                                   : hit_pos_sphere(v.spheres[sid - v.n_tris], sr, best);
                            blocked = length(sub(hp, sr.o)) < S.thr;
                        }
                        if (!blocked)
                            S.le = add(S.le, S.contrib);
                        ph++;
                    }
                    if (alive) {
                        // ---- the next light this path casts a shadow ray towards (Scene.cu:84-122)
                        for (; ph < L; ph++) {
                            const LightRec light = v.lights[ph];
                            LightSample ls = sample_point_light(S.hit(), v.models[S.model], S.wo_y, light);
                            S.thr = fmaxf(ls.thr, 0.0f);
                            S.contrib = ls.contrib;
                            if (ls.cast) {
                                no = ls.sr.o; nd = ls.sr.d;
                                if (A.occlusion) {
                                    float m = 1e-4f * (fabsf(S.thr) + fabsf(no.x) + fabsf(no.y) + fabsf(no.z)) + 1e-6f;
                                    nb = fminf(kInf, S.thr + m);
                                    ns = S.thr - m;
                                }
                                n_shadow++;
                                go = true;
                                break;
                            }
                        }
                        if (!go) {
                            // ---- end of the bounce: PathTracer.cu:73-88
                            S.res = add(S.res, mul(S.beta, S.le));
                            const ModelRec &M = v.models[S.model];
                            const Hit H = S.hit();
                            float u[4];
                            pixel_uniform4(v.pixmap[pl], S.draw - 1u - (uint32_t)L, u);
                            v3 wi;
                            float pdf;
                            v3 f = rec_sample_f_y(H, M, S.wo_y, wi, pdf, u[0], u[1]);
                            bool done = (pdf < 0.0001f || len_below_1e4(f));
                            if (!done) {
                                S.beta = mul(S.beta, divs(scale(f, fabsf(dot(wi, H.normal))), pdf));
                                Ray nr = spawn_ray(H, wi);
                                no = nr.o; nd = nr.d;
                                if (bnc > 3) {
                                    if (u[2] < 0.5f)
                                        done = true;
                                    else
                                        S.beta = divs(S.beta, 1.0f - 0.5f);
                                }
                            }
                            if (!done && bnc + 1 > A.max_depth) {
                                S.draw += 1u;
                                done = true;
                            }
                            if (done) {
                                finalize_path(v, pl, S.res, v.cnt);
                                v.rng_ctr[pl] = S.draw;
                                alive = false;
                            } else {
                                bnc++;
                                n_bounce++;
                                ph = -1;
                                go = true;
                            }
                        }
                    }
                }
                if (go) {
                    S.save(st);
                    const v3 oo[2] = {no, no}, dd[2] = {nd, nd};
                    const float bb[2] = {nb, 0.0f}, ss[2] = {ns, -1.0f};
                    const bool lv[2] = {true, false};
                    pair_set(w, oo, dd, bb, ss, lv, E);
                    level = tv.top;
                    g = 0;
                    pend = (unsigned long long)expand(level, 0u) << (4 * level);
                    walking = true;
                }
            }
        }
        // ---- one step of the walks under way (traverse_lane's loop body, one KIND of step per pass: lanes that expand a
        // node group and lanes that test a triangle run different code, and a pass that did both would run each
        // with the other's lanes idle; the kind more lanes wait for goes, the others keep their place)
        uint32_t m = 0;
        if (walking) {
            m = (uint32_t)(pend >> (4 * level)) & 15u;
            while (m == 0u && walking) {                     // groups that are done: back to their parents' (cheap)
                level++;
                g >>= 2;
                walking = level <= tv.top;
                m = walking ? (uint32_t)(pend >> (4 * level)) & 15u : 0u;
            }
        }
        const bool want_tri = walking && level == 0, want_node = walking && level != 0;
        const unsigned long long tri_m = __ballot(want_tri), node_m = __ballot(want_node);
        if (tri_m | node_m) {
            const bool do_tri = __popcll(tri_m) > __popcll(node_m);
            if (do_tri ? want_tri : want_node) {
                const int j = __builtin_ctz(m);
                pend &= ~(1ull << (4 * level + j));
                const uint32_t node = g * kTreeArity + (uint32_t)j;
                if (do_tri) {
                    const float4 *tp = (const float4 *)(A.T.tris + node);
                    const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
                    TriRec T;
                    T.ax = t0.x; T.ay = t0.y; T.az = t0.z; T.dn = t0.w;
                    T.bx = t1.x; T.by = t1.y; T.bz = t1.z; T.S = t1.w;
                    T.cx = t2.x; T.cy = t2.y; T.cz = t2.z; T.orig = __float_as_int(t2.w);
                    T.nx = t3.x; T.ny = t3.y; T.nz = t3.z; T.qmax = t3.w;
                    tri_test_pair(T, w, true, false);
                    if (w.best.x < w.stop.x) w.bestp.x = -1.0f;  // occlusion form: the ray is over
                } else {
                    level--;
                    g = node;
                    pend |= (unsigned long long)expand(level, g) << (4 * level);
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        n_bounce += (uint32_t)__shfl_xor((int)n_bounce, o);
        n_shadow += (uint32_t)__shfl_xor((int)n_shadow, o);
    }
    if (lane == 0) {
        if (n_shadow) atomicAdd(&A.ctl->shadow_total, (unsigned long long)n_shadow);
        if (n_bounce) atomicAdd(&A.ctl->bounce_total, (unsigned long long)n_bounce);
    }
}

// ------------------------------------------------------------------ depth 0: whole batches of samples
// With max_depth = 0 (primary hit + direct light, the benchmark mode) the samples of a pixel
// depend on one another only through the closest hit of the camera ray: the pixel's Philox
// counter advances by 1 (jitter) + 1 (PathTracer.cu:27) + one per light if the ray hit
// (Scene.cu:87) + 1 if the BRDF sample is usable (PathTracer.cu:77, the discarded trace of the
// next iteration).  The shadow rays -- most of the work -- do not feed back into it, and the
// count itself rarely changes from one sample of a pixel to the next.  So rt_render(n, 0) over a
// deep tree works in batches of S samples (16, 32 or 64), and a wave takes a UNIT of px pixels x ss
// samples = 128 rays (ss = 2^ss_log2 >= S: 8 x 16, 4 x 32, 2 x 64) through three phases (k_batch,
// fused_unit below):
//   P  camera rays of all samples at once from PREDICTED counters: every sample's counter is
//      predicted from its pixel's expected draws per sample (round 0: what the pixel's last sample
//      did; round 1: the same, or what the mispredicted sample did if that was the pixel's first),
//      all are traced together, and the prediction is checked pixel by pixel: the samples up to
//      and including the first whose draw count differs were traced from the counter the
//      sequential order gives them and are final.
//   H  pixels still open after two rounds flip between outcomes from sample to sample
//      (silhouettes, ~1 % of the pixels).  Their next 8 samples can only start at the counters
//      base + 2 a + (2 + lights) b + (3 + lights) c, a + b + c < 8 -- 120 candidates -- so the wave
//      traces ALL of them for the one pixel at once and then walks the chain through the table:
//      eight samples per round, so two rounds settle a batch of 16, four one of 32.
//   S  lights, shadow rays, radiance of every sample, and the pixels' running mean in sample
//      order (kernel.cu:65-66,75,77).
// Every record was computed from the exact counter of the reference's order, so the result is
// bit-identical to sample-after-sample; only the chain of spp dependent traversals is gone: no
// pixel needs more than 2 + ss / 8 (Multi-GPU).
struct BatchArgs {
    DevView v;
    TravArgs T;
    uint32_t p0, n;                  // this launch renders local pixels [p0, p0 + n)
    uint32_t n_samples;              // samples in this batch
    uint32_t ss_log2;                // log2 of the samples a wave holds (batch_ss_log2)
    int cnt0;                        // samples accumulated before this batch
    bool occlusion;
    bool task_prio;                  // listed pixel tasks run at raised wave priority (RT_TASK_PRIO=0: off)
    uint32_t early_x4;               // units of >= early_x4 / 4 times the mean unit cost go out early in the next launch (0: off)
    uint32_t late_x16;               // units of <= late_x16 / 16 of the mean unit cost are kept for last in the next launch (0: off)
    uint32_t band_units;             // units per XCD band (a multiple of 64; the last bands may be short or empty)
    bool draw_early_ok;              // the next unit's ticket may be drawn at the unit's start while the dispenser is far from its end
    uint32_t grab, grab_late;        // units per ticket: of a band's dispenser, of the late units'
    bool band_help;                  // a wave out of work helps the band with the most estimated work left (else: cyclic order)
    Ctl *ctl;
    BatchFeedback fb;                // cnt_w == nullptr: no feedback
};

// What a lane carries across a traversal is parked in LDS (slot i of a lane: st[i * kBlock]),
// as in the path kernel: the closest-hit traversal needs the register file to itself, and what
// the compiler would otherwise spill goes to scratch memory and from there to HBM.
// Is the BRDF sample of PathTracer.cu:76 usable, i.e. !(pdf < 0.0001f || length(f) < 0.0001f) (:77)?
// At depth 0 nothing else of rec_sample_f is read.  For a Lambertian surface pdf = wl.y / pi with
// wl.y = sqrt(max(0, 1 - dx^2 - dy^2)), (dx, dy) = r (sin, cos)(theta) (sampleHemi, material.cu:28-50), and
// f = colour / pi if wo.y > 0 and wl.y > 0, else 0.  While r^2 < 0.9999 the sample lies well inside
// the unit disk -- wl.y > 0.0099, pdf > 0.003 whatever the last bits of sin / cos are -- so the verdict
// is that of `wo.y > 0 && |colour / pi| >= 0.0001`; only samples on the rim (5e-5 of them) need
// sampleHemi's own arithmetic, double-precision sin / cos included.  Same verdict either way.
__device__ __forceinline__ bool brdf_sample_usable(const Hit &H, const ModelRec &M, float wo_y, float sx, float sy)
{
    if (M.mat_type == kMatLambertian) {
        const float rx = 2.0f * sx - 1.0f, ry = 2.0f * sy - 1.0f;
        const float r = fabsf(rx) > fabsf(ry) ? rx : ry;
        if (r * r < 0.9999f && !(rx == 0.0f && ry == 0.0f)) {
            v3 c = H.color;
            if (c.x < 0.0f)
                c = mk3(M.color[0], M.color[1], M.color[2]);
            return wo_y > 0.0f && !len_below_1e4(scale(c, kInvPi));
        }
    }
    v3 wi;
    float pdf;
    v3 f = rec_sample_f_y(H, M, wo_y, wi, pdf, sx, sy);
    return !(pdf < 0.0001f || len_below_1e4(f));
}

constexpr int kPrimSlots = 12;       // 0-1 pixel, 2-3 counter, 4 live bits; 5-10 the caller's
struct PrimaryTrace {
    uint32_t delta[2];               // draws the sample consumes
    float rec[2][7];
    int model[2];
    int32_t id[2];
    float t[2];
};

template <bool CULL, bool OWN_BASES>
__device__ __forceinline__ void trace_primary(const BatchArgs &A, const ChunkEnv &env, uint32_t lane,
                                              uint32_t pix[2], uint32_t ctr[2], bool live[2],
                                              PrimaryTrace &o, uint32_t &cost, uint32_t *st, Stamps &sp)
{
    const DevView &v = A.v;
    const uint32_t L = (uint32_t)v.n_lights;
    const float inf2[2] = {kInf, kInf}, none2[2] = {-1.0f, -1.0f};
    RayPair w;
    {
        v3 ro[2], rd[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // kernel.cu:65-73 with the H4 convention: the jitter is (.x, .y) of one
            // curand_uniform4 of the pixel's own stream
            float u[4];
            pixel_uniform4(pix[h], ctr[h], u);
            int x, y;
            pixel_xy(pix[h], v.width, v.height, x, y);
            Ray r = camera_ray(v.cam, x - v.width / 2 + u[0], y - v.height / 2 + u[1]);   // kernel.cu:73
            ro[h] = r.o; rd[h] = r.d;
        }
        pair_set(w, ro, rd, inf2, none2, live, A.T.tv.extent);
    }
    st[0 * kBlock] = pix[0]; st[1 * kBlock] = pix[1];
    st[2 * kBlock] = ctr[0]; st[3 * kBlock] = ctr[1];
    st[4 * kBlock] = (live[0] ? 1u : 0u) | (live[1] ? 2u : 0u);
    asm volatile("" ::: "memory");
    RT_MARK(sp, 17);
    traverse<CULL, false, true, false, OWN_BASES>(A.T, env.lvloff, w, lane, 0u, nullptr, cost, sp);
    asm volatile("" ::: "memory");
    pix[0] = st[0 * kBlock]; pix[1] = st[1 * kBlock];
    ctr[0] = st[2 * kBlock]; ctr[1] = st[3 * kBlock];
    live[0] = (st[4 * kBlock] & 1u) != 0u; live[1] = (st[4 * kBlock] & 2u) != 0u;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 7; k++)
            o.rec[h][k] = 0.0f;
        o.delta[h] = 0u;
        o.model[h] = -1;                                     // miss: PathTracer.cu:64 -> return res
        o.t[h] = h ? w.best.y : w.best.x;
        o.id[h] = h ? w.idy : w.idx;
        if (!live[h])
            continue;
        const bool ishit = scene_ishit(o.t[h]) && o.id[h] >= 0;
        // the jitter + tmp (PathTracer.cu:27) + one per light (Scene.cu:87)
        o.delta[h] = 2u + (ishit ? L : 0u);
        if (ishit) {
            Ray r;
            r.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
            r.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
            Hit H;
            fill_hit(v, r, o.id[h], H);
            const float wo_y = local_y(H, neg(r.d));         // Object.cu:107-110 on wo = -r.dir
            // PathTracer.cu:76-77: only whether the BRDF sample is usable matters at depth 0
            float u[4];
            pixel_uniform4(pix[h], ctr[h] + 1u, u);
            if (brdf_sample_usable(H, v.models[H.model], wo_y, u[0], u[1]))
                o.delta[h] += 1u;                            // the next iteration's draw (:27) before its break (:64)
            o.rec[h][0] = H.pos.x; o.rec[h][1] = H.pos.y; o.rec[h][2] = H.pos.z;
            o.rec[h][3] = H.normal.x; o.rec[h][4] = H.normal.y; o.rec[h][5] = H.normal.z;
            o.rec[h][6] = wo_y;
            o.model[h] = H.model;
        }
    }
    RT_MARK(sp, 18);
}

// Candidate r of the table: level k = samples before it, of which b hits without and c hits with
// a usable BRDF sample.  Levels start at k (k + 1) (k + 2) / 6; inside a level, t = b + c,
// index t (t + 1) / 2 + c.
constexpr uint32_t kHardLevels = 8;                          // 120 candidates <= 128 rays of a wave
__device__ __forceinline__ uint32_t hard_level_off(uint32_t k) { return k * (k + 1u) * (k + 2u) / 6u; }

// Candidate r of the table: k samples before it, of which b hits without and c hits with a
// usable BRDF sample (a = k - b - c misses).  k = kHardLevels: past the table.
__device__ __forceinline__ void hard_candidate(uint32_t r, uint32_t &k, uint32_t &a, uint32_t &b, uint32_t &c)
{
    k = 0;
    while (k + 1u < kHardLevels && hard_level_off(k + 1u) <= r) k++;
    const uint32_t rem = r - hard_level_off(k);
    uint32_t t = 0;
    while ((t + 1u) * (t + 2u) / 2u <= rem) t++;
    c = rem - t * (t + 1u) / 2u;
    b = t - c;
    a = k - min(t, k);
    if (r >= hard_level_off(kHardLevels)) k = kHardLevels;
}

// ------------------------------------------------------------------ depth 0, one kernel per batch
// k_batch takes a unit (px pixels x ss samples = 128 rays of one wave) through phases P, H and S
// without leaving the wave.  The hit records never reach memory: they
// sit in LDS (8 dwords per ray, row-major by field, one column per lane) from the moment a
// sample's camera ray is final until its shadow rays are done; a record traced by ANOTHER lane
// (the enumeration walks one pixel's candidates on all lanes) is written straight into the
// column of the lane that owns the sample.  What the wave waits for besides arithmetic is taken
// off its critical path: the next unit's ticket is drawn from the dispenser while this unit is
// traced, the framebuffer texels are requested before the shadow rays and used after them, and
// the running mean reads the samples' radiance from LDS (one owner lane per pixel, in sample
// order: kernel.cu:65-66,75,77).
constexpr uint32_t kAllPixels = 0xffffffffu;
constexpr int kRecRows = 8;                  // pos (3), normal (3), dot(-dir, normal), model
constexpr int kFuseTmpRows = 14;             // phase P: trace_primary's 5 + 8 of the unit; phase S: 7 per ray
constexpr int kFuseRows = 2 * kRecRows + kFuseTmpRows;
static_assert(kFuseTmpRows >= kPrimSlots + 1, "phase P rows");

// Record of slot r (lane r & 63, half r >> 6) of this wave; col0 = column of the wave's lane 0.
__device__ __forceinline__ void put_record(uint32_t *col0, uint32_t r, const PrimaryTrace &o, int h)
{
    uint32_t *c = col0 + (r & 63u) + (r >> 6) * (uint32_t)(kRecRows * kBlock);
#pragma unroll
    for (int k = 0; k < 7; k++)
        c[k * kBlock] = __float_as_uint(o.rec[h][k]);
    c[7 * kBlock] = (uint32_t)o.model[h];
}
// What the last sample of a batch leaves behind per pixel: id / t buffers, the stream's counter
// and the draws it consumed (what the next batch predicts from).
__device__ __forceinline__ void write_last(const BatchArgs &A, const PrimaryTrace &o, int h,
                                           uint32_t p, uint32_t ctr)
{
    A.v.first_id[p] = o.id[h];
    A.v.first_t[p] = o.t[h];
    A.v.rng_ctr[p] = ctr + o.delta[h];
    A.v.last_draws[p] = o.delta[h];
}

// A wave's DESK: what it knows about the dispensers, the lists and its own sums lives in LDS while a unit is under
// way, one row of words per wave, and is read back afterwards -- a unit needs every register there is, and two dozen
// wave-uniform values kept alive across it were what the compiler spilled (first the SGPRs into VGPR lanes, then
// VGPRs to scratch, and the scratch of 4096 waves no longer fit the L2s).
enum : int {
    kDkStage, kDkBias, kDkSize, kDkNun, kDkGrab, kDkBand, kDkHelped, kDkLateHelped, kDkK, kDkD, kDkListKind, kDkListOff,
    kDkCtr,                          // slot of the dispenser the next ticket comes from
    kDkDraw,                         // 0: this unit draws no ticket, 1: behind the shadow rays, 2: at its start
    kDkInBand,                       // the unit's cost counts for its band's sum
    kDkBandCostLo, kDkBandCostHi, kDkCostLo, kDkCostHi,
    kDkHeavy, kDkEarly, kDkLate,     // the launch's cost thresholds (feedback)
    kDkTick1, kDkTick2, kDkTickN1, kDkTickN2,   // RT_TICKWAIT only
    kDkTickA, kDkTickB, kDkTickC, kDkTickU,     // ... cycles in a unit's phases P + H, S, epilogue; units
#ifdef RT_TICKWAIT
    kDkTickBy = 32,                              // ... the same by where the unit came from: 8 x (A, B, C, units)
    kDkTickFrom = 64,                            // where this unit came from
    kDkWords = 72
#else
    kDkWords = 32
#endif
};

template <bool CULL, bool OWN_BASES>
__device__ __forceinline__ void fused_unit(const BatchArgs &A, const ChunkEnv &env, uint32_t unit,
                                           uint32_t only_px, uint32_t lane, uint32_t *col, uint32_t *col0, uint32_t *dk,
                                           uint32_t *__restrict__ work, uint32_t &ticket, Stamps &sp)
{
    // only_px = kAllPixels: the whole unit; else only its pixel `only_px` (a listed unit, split by pixel).
    const DevView &v = A.v;
    const uint32_t L = (uint32_t)v.n_lights;
    const float E = A.T.tv.extent;
    const uint32_t px_log2 = 7u - A.ss_log2, px_mask = (1u << px_log2) - 1u, ss = 1u << A.ss_log2;
    const uint32_t p_end = A.p0 + A.n;
    // (col0: the column of this wave's lane 0 -- the same for every lane, and handed in as such: a scalar register)
    uint32_t *const tmp = col + 2 * kRecRows * kBlock;       // this lane's phase rows
    uint32_t *const tmp0 = col0 + 2 * kRecRows * kBlock;
    uint32_t cost = 0;
    RT_MARK(sp, 1);
    RT_COUNT(sp, 11);
    RT_STAMP(sp.unit_begin());
    RT_TICK(const unsigned long long tk0 = __builtin_amdgcn_s_memtime());

    // ---------------- phase P: the camera rays' closest hits, records to LDS
    uint32_t pix[2], base_c[2], base_s[2], dhat[2];
    bool todo[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t r = (uint32_t)h * 64u + lane;
        const uint32_t pl = A.p0 + (unit << px_log2) + (r & px_mask);
        todo[h] = pl < p_end && (r >> px_log2) < A.n_samples && (only_px == kAllPixels || (r & px_mask) == only_px);
        const uint32_t p = min(pl, p_end - 1u);
        pix[h] = v.pixmap[p];
        base_c[h] = v.rng_ctr[p];
        base_s[h] = 0u;
        const uint32_t last = v.last_draws[p];
        dhat[h] = last ? last : 3u + L;
        col[(h * kRecRows + 7) * kBlock] = 0xffffffffu;     // no record yet: model = -1
        RT_STAMP(stamp_arrived_v(pix[h]); stamp_arrived_v(base_c[h]); stamp_arrived_v(dhat[h]));
    }
    RT_MARK(sp, 16);
    // The NEXT unit's ticket.  A draw takes microseconds to come back (the counters live on the memory side), so
    // while its dispenser still has plenty to hand out a wave may draw now, behind the prologue's loads
    // (RT_DRAW_EARLY=1), and has the answer long before it needs it.  Near the dispenser's end that would leave
    // every wave with two units to do when it runs dry -- the one it is in and the one it holds a ticket for -- and
    // the launch's tail two units long (measured on an eighth of C3: waves ended over 600 us of 2400); there, and by
    // default everywhere, the draw waits until just before the epilogue (below).
    if (lane == 0 && dk[kDkDraw] == 2u)
        ticket = atomicAdd(&work[dk[kDkCtr]], 1u);

    // Phases P and H share ONE call site of the closest-hit traversal (one copy of its loop in the
    // instruction cache).  mode 0, 1: the two prediction rounds; mode 2: the pixels still open
    // after them, one at a time, eight samples per pass.
    unsigned long long stride = 0ull;                        // bits at the multiples of px: one pixel's samples in a lane mask
    for (uint32_t i = 0; i < 64u; i += 1u << px_log2) stride |= 1ull << i;
    int mode = 0;
    unsigned long long open[2] = {0ull, 0ull};               // lanes that speak for an open pixel
    uint32_t e_c = 0, e_s = A.n_samples, qq = 0, p1 = 0, pix1 = 0;   // the open pixel under enumeration (wave-uniform)
    for (;;) {
        if (mode < 2 && !__ballot(todo[0] | todo[1]))
            break;                                           // every sample is final
        if (mode == 2 && e_s >= A.n_samples) {               // on to the next open pixel
            if (!(open[0] | open[1]))
                break;
            const int hh = open[0] ? 0 : 1;
            const uint32_t sl = (uint32_t)__builtin_ctzll(open[hh]);
            open[hh] &= open[hh] - 1ull;
            qq = ((uint32_t)hh * 64u + sl) & px_mask;
            p1 = min(A.p0 + (unit << px_log2) + qq, p_end - 1u);
            // wave-uniform LDS reads of what the speaking lane parked
            e_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)tmp0[sl + (5 + hh) * kBlock]);
            e_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)tmp0[sl + (7 + hh) * kBlock]);
            pix1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.pixmap[p1]);
            RT_STAMP(sp.hard_units++);
            if (e_s >= A.n_samples)
                continue;
        }
        RT_STAMP(if (mode == 2) sp.hard_passes++);
        const uint32_t e_n = min(A.n_samples - e_s, kHardLevels);   // mode 2: samples this pass settles
        uint32_t tp[2], ctr[2];
        bool live[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (mode < 2) {
                const uint32_t smp = ((uint32_t)h * 64u + lane) >> px_log2;
                ctr[h] = base_c[h] + (smp - base_s[h]) * dhat[h];
                tmp[(5 + h) * kBlock] = base_c[h];
                tmp[(7 + h) * kBlock] = base_s[h];
                tmp[(9 + h) * kBlock] = dhat[h];
                tp[h] = pix[h];
                live[h] = todo[h];
            } else {
                // candidate r of the table: k samples before it, b hits without and c hits with
                // a usable BRDF sample among them
                uint32_t k, a, b, c;
                hard_candidate((uint32_t)h * 64u + lane, k, a, b, c);
                live[h] = k < e_n;
                ctr[h] = e_c + 2u * a + (2u + L) * b + (3u + L) * c;
                tp[h] = pix1;
            }
        }
        // rows 5-8 of every lane's column (the pixels' bookkeeping) survive this: trace_primary parks in rows 0-4
        PrimaryTrace o;
        trace_primary<CULL, OWN_BASES>(A, env, lane, tp, ctr, live, o, cost, tmp, sp);
        uint32_t ln = lane;
        asm volatile("" : "+v"(ln));                         // lane-derived values are recomputed, not kept across the traversal
        if (mode < 2) {
            uint32_t q[2], smp[2], p[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t r = (uint32_t)h * 64u + ln;
                q[h] = r & px_mask;
                smp[h] = r >> px_log2;
                p[h] = min(A.p0 + (unit << px_log2) + q[h], p_end - 1u);
                pix[h] = tp[h];
                todo[h] = live[h];
                base_c[h] = tmp[(5 + h) * kBlock];
                base_s[h] = tmp[(7 + h) * kBlock];
                dhat[h] = tmp[(9 + h) * kBlock];
            }
            // first sample of each pixel whose draw count is not the expected one
            const unsigned long long m0 = __ballot(todo[0] && o.delta[0] != dhat[0]);
            const unsigned long long m1 = __ballot(todo[1] && o.delta[1] != dhat[1]);
            uint32_t first[2], src[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                first[h] = ss;
                if (px_log2 < 6u) {                          // sample k of pixel q sits at bit (k << px_log2) + q
                    const unsigned long long w0 = (m0 >> q[h]) & stride, w1 = (m1 >> q[h]) & stride;
                    if (w0)      first[h] = (uint32_t)__builtin_ctzll(w0) >> px_log2;
                    else if (w1) first[h] = (64u + (uint32_t)__builtin_ctzll(w1)) >> px_log2;
                } else {
                    for (uint32_t k = ss; k-- > 0u;) {       // at most two samples per pixel
                        const uint32_t r = (k << px_log2) + q[h];
                        const bool bit = (((r < 64u ? m0 : m1) >> (r & 63u)) & 1ull) != 0ull;
                        first[h] = bit ? k : first[h];
                    }
                }
                src[h] = (min(first[h], ss - 1u) << px_log2) + q[h];
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // its draw count, from the lane that holds it
                const uint32_t lo = (uint32_t)__shfl((int)o.delta[0], (int)(src[h] & 63u));
                const uint32_t hi = (uint32_t)__shfl((int)o.delta[1], (int)(src[h] & 63u));
                const uint32_t dj = src[h] < 64u ? lo : hi;
                if (todo[h] && smp[h] <= first[h]) {         // traced from the right counter: final
                    put_record(col0, (uint32_t)h * 64u + ln, o, h);
                    if (smp[h] + 1u == A.n_samples)
                        write_last(A, o, h, p[h], ctr[h]);
                    todo[h] = false;
                }
                if (first[h] < ss) {                         // the pixel goes on from behind sample `first`
                    // ... expecting what it expected before, unless its very first open sample
                    // was the odd one (then the pixel as a whole has changed: expect that instead)
                    const bool changed = first[h] == base_s[h];
                    base_c[h] += (first[h] - base_s[h]) * dhat[h] + dj;
                    base_s[h] = first[h] + 1u;
                    dhat[h] = changed ? dj : dhat[h];
                }
            }
            if (mode == 1) {
                // Pixels still open flip between outcomes from sample to sample.  The lane of a
                // pixel's first open sample speaks for it; its counter and sample index go to
                // LDS, where the enumeration reads them (wave-uniform).
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    open[h] = __ballot(todo[h] && smp[h] == base_s[h]);
                    tmp[(5 + h) * kBlock] = base_c[h];
                    tmp[(7 + h) * kBlock] = base_s[h];
                }
                e_s = A.n_samples;                           // nothing under enumeration yet
            }
            mode++;
        } else {
            // which candidates lie on the chain: walk it through the table (wave-uniform)
            const unsigned long long u0 = __ballot(live[0] && o.delta[0] == 3u + L);
            const unsigned long long u1 = __ballot(live[1] && o.delta[1] == 3u + L);
            const unsigned long long b0 = __ballot(live[0] && o.delta[0] == 2u + L);
            const unsigned long long b1 = __ballot(live[1] && o.delta[1] == 2u + L);
            unsigned long long c0 = 0ull, c1 = 0ull;
            uint32_t b = 0, c = 0, c_new = e_c;
            for (uint32_t k = 0; k < e_n; k++) {
                const uint32_t t = b + c;
                const uint32_t r = hard_level_off(k) + t * (t + 1u) / 2u + c;
                const unsigned long long bit = 1ull << (r & 63u);
                if (r < 64u) c0 |= bit; else c1 |= bit;
                if (((r < 64u ? u0 : u1) & bit) != 0ull)      { c_new += 3u + L; c++; }
                else if (((r < 64u ? b0 : b1) & bit) != 0ull) { c_new += 2u + L; b++; }
                else                                          { c_new += 2u; }
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
                if ((((h ? c1 : c0) >> ln) & 1ull) != 0ull) {
                    uint32_t k, ca, cb, cc;
                    hard_candidate((uint32_t)h * 64u + ln, k, ca, cb, cc);
                    const uint32_t smp = e_s + k;
                    put_record(col0, (smp << px_log2) + qq, o, h);   // into the column of the lane that owns the sample
                    if (smp + 1u == A.n_samples)
                        write_last(A, o, h, p1, ctr[h]);
                }
            e_s += e_n;
            e_c = c_new;
        }
        RT_MARK(sp, 19);
    }
    // every record of the unit is final and in LDS; lanes read their own column from here on
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    RT_TICK(const unsigned long long tk1 = __builtin_amdgcn_s_memtime());
    // ---------------- phase S: lights, shadow rays, the pixels' running mean
    uint32_t ln = lane;
    asm volatile("" : "+v"(ln));
    const uint32_t own_pl = A.p0 + (unit << px_log2) + ln;
    // the framebuffer texels: requested now, used after the shadow rays (who reads what: the epilogue
    // below; one register set for all unit shapes, .x alone in the channel-parallel form)
    float4 prev0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (px_log2 >= 5u) {
        if (ln <= px_mask && own_pl < p_end && (only_px == kAllPixels || ln == only_px)) prev0 = v.fb[own_pl];
    } else {
        const uint32_t q = ln & px_mask, ch = ln >> px_log2;
        const uint32_t pl = A.p0 + (unit << px_log2) + q;
        if (ch < 3u && pl < p_end && (only_px == kAllPixels || q == only_px)) prev0.x = ((const float *)&v.fb[pl])[ch];
    }
    struct RS { v3 pos, normal; float wo_y; int model; v3 le, contrib; float thr; } S[2];
    bool hit[2], cast[2] = {false, false};
    auto load_rec = [&](int h) {
        const uint32_t *c = col + h * kRecRows * kBlock;
        S[h].pos = mk3(__uint_as_float(c[0]), __uint_as_float(c[kBlock]), __uint_as_float(c[2 * kBlock]));
        S[h].normal = mk3(__uint_as_float(c[3 * kBlock]), __uint_as_float(c[4 * kBlock]), __uint_as_float(c[5 * kBlock]));
        S[h].wo_y = __uint_as_float(c[6 * kBlock]);
        S[h].model = (int)c[7 * kBlock];
    };
    auto as_hit = [&](int h) {
        Hit H;
        H.pos = S[h].pos; H.normal = S[h].normal;
        H.tangent = mk3(0.0f, 0.0f, 0.0f);
        H.color = mk3(-1.0f, -1.0f, -1.0f);                  // H5: unbound mesh texture / quadrics leave rec.color
        H.model = S[h].model;
        return H;
    };
    uint32_t n_shadow = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        load_rec(h);
        hit[h] = S[h].model >= 0;
        S[h].le = mk3(0.0f, 0.0f, 0.0f);
        S[h].contrib = mk3(0.0f, 0.0f, 0.0f);
        S[h].thr = 0.0f;
    }
    RT_MARK(sp, 20);
    RayPair w;
    const uint32_t n_lights = __ballot(hit[0] | hit[1]) ? L : 0u;   // nobody hit: nothing to light
    for (uint32_t j = 0; j < n_lights; j++) {
        // one delta light against the hits: Scene.cu:83-120 -> evaluateDirectLight :205-224
        const LightRec light = load_uniform(&v.lights[j]);
        v3 so[2], sd[2];
        float sb[2], st2[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            cast[h] = false;
            so[h] = mk3(0.0f, 0.0f, 0.0f); sd[h] = mk3(0.0f, 0.0f, 0.0f);
            sb[h] = kInf; st2[h] = -1.0f;
            if (hit[h]) {
                LightSample ls = sample_point_light(as_hit(h), v.models[S[h].model], S[h].wo_y, light);
                cast[h] = ls.cast;
                so[h] = ls.sr.o; sd[h] = ls.sr.d;
                S[h].thr = fmaxf(ls.thr, 0.0f);              // Scene.cu:227 rhs
                S[h].contrib = ls.contrib;
                if (A.occlusion) {
                    float m = 1e-4f * (fabsf(S[h].thr) + fabsf(so[h].x) + fabsf(so[h].y) + fabsf(so[h].z)) + 1e-6f;
                    sb[h] = fminf(kInf, S[h].thr + m);
                    st2[h] = S[h].thr - m;
                }
            }
        }
        const unsigned long long c0 = __ballot(cast[0]), c1 = __ballot(cast[1]);
        if (!(c0 | c1))
            continue;                                        // nobody casts towards this light
        n_shadow += (uint32_t)__popcll(c0) + (uint32_t)__popcll(c1);
        pair_set(w, so, sd, sb, st2, cast, E);
        RT_MARK(sp, 21);
#pragma unroll
        for (int h = 0; h < 2; h++) {                        // the record itself stays where it is
            float *t7 = (float *)tmp + h * 7 * kBlock;
            t7[0] = S[h].le.x; t7[kBlock] = S[h].le.y; t7[2 * kBlock] = S[h].le.z;
            t7[3 * kBlock] = S[h].contrib.x; t7[4 * kBlock] = S[h].contrib.y; t7[5 * kBlock] = S[h].contrib.z;
            t7[6 * kBlock] = S[h].thr;
        }
        asm volatile("" ::: "memory");
        RT_MARK(sp, 4);
        traverse<CULL, false, false, false, OWN_BASES>(A.T, env.lvloff, w, lane, 0u, nullptr, cost, sp);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const float *t7 = (const float *)tmp + h * 7 * kBlock;
            S[h].le = mk3(t7[0], t7[kBlock], t7[2 * kBlock]);
            S[h].contrib = mk3(t7[3 * kBlock], t7[4 * kBlock], t7[5 * kBlock]);
            S[h].thr = t7[6 * kBlock];
            load_rec(h);
            hit[h] = S[h].model >= 0;
        }
        RT_MARK(sp, 4);
        // Scene.cu:225-241: blocked = |light_rec.pos - r.o| < |r.o - lpos| - 0.01
        // (H3: `lightidx == idx` restated as never true for delta lights)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const bool did = h ? (((c1 >> lane) & 1ull) != 0ull) : (((c0 >> lane) & 1ull) != 0ull);
            if (!did)
                continue;
            const float best = h ? w.best.y : w.best.x;
            const int32_t sid = h ? w.idy : w.idx;
            bool blocked = false;
            if (scene_ishit(best) && sid >= 0) {
                Ray sr;
                sr.o = h ? mk3(w.ox.y, w.oy.y, w.oz.y) : mk3(w.ox.x, w.oy.x, w.oz.x);
                sr.d = h ? mk3(w.dx.y, w.dy.y, w.dz.y) : mk3(w.dx.x, w.dy.x, w.dz.x);
                // occlusion form: a hit safely below the threshold blocks (the margin the ray was
                // cast with covers the difference between t and |fl(o + t d) - o|); only inside the
                // band does the reference's own comparison decide
                const float m = 1e-4f * (fabsf(S[h].thr) + fabsf(sr.o.x) + fabsf(sr.o.y) + fabsf(sr.o.z)) + 1e-6f;
                const bool sure = A.occlusion && best < S[h].thr - m;
                blocked = sure;
                if (__ballot(!sure)) {
                    if (!sure) {
                        v3 hp = sid < v.n_tris ? ray_pos(sr, best)
                                               : hit_pos_sphere(v.spheres[sid - v.n_tris], sr, best);
                        blocked = length(sub(hp, sr.o)) < S[h].thr;
                    }
                }
            }
            if (!blocked)                                    // res += ... in light order (Scene.cu:120)
                S[h].le = add(S[h].le, S[h].contrib);
        }
        RT_MARK(sp, 22);
    }
    RT_TICK(const unsigned long long tk2 = __builtin_amdgcn_s_memtime());
    if (lane == 0 && dk[kDkDraw] == 1u)                      // the next unit's ticket, late form: back behind the epilogue
        ticket = atomicAdd(&work[dk[kDkCtr]], 1u);
    RT_MARK(sp, 21);
    // PathTracer.cu:73 with beta = 1, res = 0; a miss returns 0 (:64).  Every sample's share of the
    // running mean px = frac2 * prev + frac1 * c (kernel.cu:65-66,75) goes to LDS: its term
    // frac1 * c, which does not depend on the samples before it, and its frac2.
    float *const ft = (float *)tmp;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const v3 res = hit[h] ? add(mk3(0.0f, 0.0f, 0.0f), mul(mk3(1.0f, 1.0f, 1.0f), S[h].le)) : mk3(0.0f, 0.0f, 0.0f);
        const uint32_t smp = ((uint32_t)h * 64u + ln) >> px_log2;
        const int cnt = A.cnt0 + (int)smp + 1;
        const float frac1 = 1.0f / cnt, frac2 = frac1 * (cnt - 1);
        ft[(4 * h + 0) * kBlock] = frac1 * res.x; ft[(4 * h + 1) * kBlock] = frac1 * res.y;
        ft[(4 * h + 2) * kBlock] = frac1 * res.z; ft[(4 * h + 3) * kBlock] = frac2;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    RT_MARK(sp, 23);
    if (px_log2 >= 5u) {
        // few samples per pixel (batches of up to 4): lane q owns pixel q (and q + 64), all channels
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t q = (uint32_t)h * 64u + ln;       // second pass only when a wave holds 128 pixels
            const uint32_t pl = A.p0 + (unit << px_log2) + q;
            if (q <= px_mask && pl < p_end && (only_px == kAllPixels || q == only_px)) {
                const float *const f0 = (const float *)tmp0;
                float4 prev = h ? v.fb[pl] : prev0;           // (a wave of 128 pixels: the second texel only now)
                for (uint32_t smp = 0; smp < A.n_samples; smp++) {
                    const uint32_t r = (smp << px_log2) + q;
                    const float *c = f0 + (r & 63u) + (r >> 6) * (4u * kBlock);
                    const float frac2 = c[3 * kBlock];
                    prev.x = frac2 * prev.x + c[0];
                    prev.y = frac2 * prev.y + c[kBlock];
                    prev.z = frac2 * prev.z + c[2 * kBlock];
                    prev.w = 1.0f;
                }
                v.fb[pl] = prev;
            }
        }
    } else {
        // lane ch * px + q runs channel ch of pixel q through the pixel's samples, in order, each
        // from the column of the lane that traced it
        const uint32_t q = ln & px_mask, ch = ln >> px_log2;
        const uint32_t pl = A.p0 + (unit << px_log2) + q;
        if (ch < 3u && pl < p_end && (only_px == kAllPixels || q == only_px)) {
            float prev = prev0.x;
            const uint32_t half = min(A.n_samples, 64u >> px_log2);          // samples held by lanes' first rays
            const uint32_t stride = 1u << px_log2, fo = (3u - ch) * kBlock;
            // the chain itself is sequential (every step rounds: kernel.cu:75); its operands are not -- eight
            // samples' shares are read from LDS at a time, so the chain waits for LDS once per eight steps
            auto chain = [&](const float *c, uint32_t n) {
                uint32_t i = 0;
                for (; i + 8u <= n; i += 8u) {
                    float t[8], f[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) { t[k] = c[0]; f[k] = c[fo]; c += stride; }
#pragma unroll
                    for (int k = 0; k < 8; k++) prev = f[k] * prev + t[k];
                }
                for (; i < n; i++, c += stride)
                    prev = c[fo] * prev + c[0];
            };
            chain((const float *)tmp0 + q + ch * kBlock, half);
            chain((const float *)tmp0 + q + (4u + ch) * kBlock, A.n_samples - half);
            float *texel = (float *)&v.fb[pl];
            texel[ch] = prev;
            if (ch == 0u)
                texel[3] = 1.0f;
        }
    }
    RT_MARK(sp, 24);
    if (lane == 0 && n_shadow)
        atomicAdd(&A.ctl->shadow_total, (unsigned long long)n_shadow);
    // feedback for the next launch: what this unit cost, and whether it should go out first, by pixel
    // (a pixel task costs a quarter of the lanes but much less than its unit: listed again from a quarter on)
    {   // (every lane writes the same words)
        const uint32_t lo = dk[kDkCostLo], nlo = lo + cost;
        dk[kDkCostLo] = nlo;
        if (nlo < lo) dk[kDkCostHi] = dk[kDkCostHi] + 1u;
        if (dk[kDkInBand]) {
            const uint32_t blo = dk[kDkBandCostLo], nblo = blo + cost;
            dk[kDkBandCostLo] = nblo;
            if (nblo < blo) dk[kDkBandCostHi] = dk[kDkBandCostHi] + 1u;
        }
    }
    const uint32_t heavy_cost = dk[kDkHeavy], early_cost = dk[kDkEarly], late_cost = dk[kDkLate];
    if (A.fb.cnt_w && lane == 0 && cost >= (only_px == kAllPixels ? heavy_cost : heavy_cost >> 2)) {
        if (atomicExch(&A.fb.flag_w[unit], 1u) == 0u) {
            const uint32_t slot = atomicAdd(&A.fb.cnt_w->n_heavy, 1u);
            if (slot < A.fb.n_slots)
                A.fb.heavy_w[slot] = unit;
        }
    } else if (A.fb.cnt_w && lane == 0 && (only_px == kAllPixels ? cost : cost << px_log2) >= early_cost) {
        // costly, but not worth splitting (a pixel task speaks for its unit by its own cost times the unit's pixels:
        // a unit whose pixels all fall below the re-listing threshold must not drop back into band order): out
        // early next time, whole (class a: from twice the threshold on)
        if (atomicExch(&A.fb.flag_w[unit], 2u) == 0u) {
            const bool a = (only_px == kAllPixels ? cost : cost << px_log2) >= 2u * early_cost;
            const uint32_t slot = atomicAdd(a ? &A.fb.cnt_w->n_early_a : &A.fb.cnt_w->n_early_b, 1u);
            if (slot < A.fb.n_slots)                         // (a + b <= units <= n_slots: every unit is listed at most once)
                A.fb.early_w[a ? slot : A.fb.n_slots - 1u - slot] = unit;
        }
    } else if (A.fb.cnt_w && lane == 0 && only_px == kAllPixels && cost <= late_cost) {
        // next to nothing (its camera rays miss everything): kept for the launch's last stretch next time.  A flag
        // only -- a third of an image's units are like this, and a list would put every one of them through ONE
        // counter (measured: 4096 waves on one counter, 460 us per unit instead of 20)
        A.fb.flag_w[unit] = 3u;
    }
    // the next unit's phase P writes these rows: every lane's reads above must be done first
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    RT_TICK(const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
            dk[kDkTickA] = dk[kDkTickA] + (uint32_t)(tk1 - tk0); dk[kDkTickB] = dk[kDkTickB] + (uint32_t)(tk2 - tk1);
            dk[kDkTickC] = dk[kDkTickC] + (uint32_t)(tk3 - tk2); dk[kDkTickU] = dk[kDkTickU] + 1u;
            { uint32_t *const by = dk + kDkTickBy + 4u * dk[kDkTickFrom];
              by[0] = by[0] + (uint32_t)(tk1 - tk0); by[1] = by[1] + (uint32_t)(tk2 - tk1); by[2] = by[2] + (uint32_t)(tk3 - tk2); by[3] = by[3] + 1u; });
    RT_MARK(sp, 10);
    RT_STAMP(sp.unit_end(unit, cost));
}

// Before every k_batch launch: the previous launch's per-unit flags as lists.  Band y (units [y * band_units,
// (y + 1) * band_units), band_units a multiple of 64) gets its ordinary units -- flag 0 -- in `norm`, the units kept
// for last -- flag 3 -- go to `late`; listed and early units have their lists already.  A wave takes 64 units (one
// band) and appends with one atomic per class; order inside a list is by arrival, which keeps neighbours
// together in runs of 64 -- what the bands' locality needs.  k_batch itself cannot build these: a third of an
// image's units are "late", and a list appended to by every one of them goes through ONE counter (measured:
// 460 us per unit instead of 20).  The list header was zeroed by the launch before (k_batch, `lists_next`); the
// same pass zeroes the feedback set after the one this launch writes (n_slots flags and the counts).
__global__ __launch_bounds__(256) void k_classify(const uint32_t *__restrict__ flags, uint32_t flags_valid, uint32_t n_units,
                                                   uint32_t band_units, uint32_t *__restrict__ norm, uint32_t *__restrict__ late,
                                                   BatchLists *__restrict__ lists, uint32_t *__restrict__ flag_z,
                                                   FbCounts *__restrict__ cnt_z, uint32_t n_slots)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t u0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;                    // wave-uniform
    const uint32_t u = u0 + lane;
    if (flag_z && u < n_slots)
        flag_z[u] = 0u;
    if (cnt_z && blockIdx.x == 0 && threadIdx.x < sizeof(FbCounts) / sizeof(uint32_t))
        ((uint32_t *)cnt_z)[threadIdx.x] = 0u;
    if (u0 >= n_units)
        return;
    const uint32_t f = u < n_units ? (flags_valid ? flags[u] : 0u) : 0xffu;
    const unsigned long long m_n = __ballot(f == 0u), m_l = __ballot(f == 3u);
    const uint32_t y = u0 / band_units;
    uint32_t b_n = 0, b_l = 0;
    if (lane == 0) {
        if (m_n) b_n = atomicAdd(&lists->n_norm[y], (uint32_t)__popcll(m_n));
        if (m_l) b_l = atomicAdd(&lists->n_late, (uint32_t)__popcll(m_l));
    }
    b_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_n);
    b_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)b_l);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (f == 0u) norm[(size_t)y * band_units + b_n + (uint32_t)__popcll(m_n & below)] = u;
    if (f == 3u) late[b_l + (uint32_t)__popcll(m_l & below)] = u;
}

// One kernel per batch and pixel range.  The grid is what is resident at once; waves draw units
// from one dispenser per XCD (run_chunks' phase 2: every XCD first works through its own eighth
// of the range -- a band of the image, hence a part of the tree its L2 can hold -- then helps with
// the others'), one ticket ahead.
// WAVES per SIMD the register allocation aims at: four (128 VGPRs, 8 spilled) or five (96 VGPRs, ~50
// spilled, 5 x 30 KB of LDS per CU; six would not fit the LDS).  launch_batch picks, and says why.
template <bool CULL, int WAVES>
__global__ __launch_bounds__(kBlock, WAVES) void k_batch(const BatchArgs A, uint32_t *__restrict__ work,
                                                     uint32_t *__restrict__ work_next)
{
    __shared__ uint32_t s_lds[kFuseRows][kBlock];
    const uint32_t lane = threadIdx.x & 63u;
    // the wave's first column and its desk: wave-uniform, so from a scalar register (derived from threadIdx.x alone
    // they sit in vector registers that get spilled, and their reloads are what waits for a ticket draw: DESIGN.md 4.1)
    // (four-wave build: C3 -1.7 %, C4 -1.4 % of the step; the five-wave build spills more with it, 77 -> 99 registers,
    // and C5 loses 0.7 %: it keeps the vector form)
    const uint32_t wave_in_block = WAVES >= 5 ? threadIdx.x >> 6 : (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *col = WAVES >= 5 ? &s_lds[0][threadIdx.x] : &s_lds[0][wave_in_block * 64u] + lane;
    uint32_t *const col0 = col - lane;
    ChunkEnv env;
    env.lvloff = A.T.tv.level_off[lane & (kMaxLevels - 1)];
    env.feedback = false; env.heavy_cost = 0xffffffffu; env.mean_cost = 0u;
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)kWorkSlots)
        work_next[threadIdx.x] = 0u;                         // the next launch's dispensers (stream-ordered after us)
    if (blockIdx.x == 0 && threadIdx.x < sizeof(BatchLists) / sizeof(uint32_t))
        ((uint32_t *)A.fb.lists_next)[threadIdx.x] = 0u;     // and its list header, for its k_classify
    Stamps sp;
#ifdef RT_STAMPS
    sp.start();
    const unsigned long long t_begin = sp.last;
    unsigned long long t_tasks_end = 0, n_tasks_done = 0, t_tasks_sum = 0, t_task_max = 0;
    const unsigned long long w_begin = wall_clock64();       // (s_memtime is per XCD; the 100 MHz counter is the chip's)
    unsigned long long w_last_start = 0, w_last_unit = 0, w_last_stage = 0, w_last_cost = 0;
#endif
    const uint32_t px_log2 = 7u - A.ss_log2, px = 1u << px_log2;
    const uint32_t n_units = (A.n + px - 1u) / px;
    __shared__ uint32_t s_desk[kBlock / 64u][kDkWords];
    uint32_t *const dk = s_desk[wave_in_block];            // (plain LDS accesses, fenced by compiler barriers: a volatile
                                                             // pointer loses its address space and turns into flat accesses)
    auto desk = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)dk[i]); };
    // What the previous launch of this shape found costly: those units first, one pixel per task.
    uint32_t n_listed = 0, n_early_a = 0, n_early = 0, heavy_cost = 0xffffffffu, early_cost = 0xffffffffu, late_cost = 0u;
    const BatchLists *const LS = A.fb.lists;
    const bool fb_valid = A.fb.cnt_w && A.fb.flags_valid;    // the lists and flags are the previous launch's, of this shape
    if (A.fb.cnt_w) {
        const FbCounts cr = *A.fb.cnt_r;
        if (fb_valid && cr.total) {
            n_listed = min(cr.n_heavy, A.fb.n_slots);
            n_early_a = min(cr.n_early_a, A.fb.n_slots);
            n_early = n_early_a + min(cr.n_early_b, A.fb.n_slots - n_early_a);
            late_cost = (uint32_t)std::min<unsigned long long>((unsigned long long)A.late_x16 * cr.total / (16ull * std::max(n_units, 1u)), 0x3fffffffull);
            if (A.early_x4)                                  // (never below 24 node groups + triangle records)
                early_cost = (uint32_t)std::min<unsigned long long>(
                    std::max((unsigned long long)A.early_x4 * cr.total / (4ull * std::max(n_units, 1u)), 24ull), 0x3fffffffull);
            // A unit holds the launch up when it is a good part of what one wave does in the whole
            // launch: listed from a quarter of a resident wave's share on, and never below eight
            // times the mean unit cost (splitting costs lanes) or 96 node groups + triangle records
            const unsigned long long share = cr.total / ((unsigned long long)gridDim.x * (kBlock / 64u));
            const unsigned long long mean8 = 8ull * cr.total / std::max(n_units, 1u);
            heavy_cost = (uint32_t)std::min<unsigned long long>(std::max(std::max(share / 4ull, mean8), 96ull), 0x7fffffffull);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            A.fb.cnt_w->pad = A.ss_log2 + 1u;                // (n_heavy and total were zeroed by the host)
    }
    // Stage -1: the lists -- the listed units' pixels, then the early units, whole, costliest class first (one
    // counter for the whole chip); stage 0: this XCD's band of ordinary units; stage 1: the other XCDs' bands,
    // whichever has the most work left; stage 7: the late units, in eight slices; 8: done.  The bands' and the late
    // units' lists are k_classify's.  One ticket ahead throughout.
    // The lists are ONE sequence of n_tasks + n_early entries with one counter for the whole chip -- and a counter
    // answers some 4096 draws in a quarter of a millisecond, so no wave draws from it to find out that the lists
    // are not for it: wave w of the launch takes entry w without asking, waves beyond the lists go straight to
    // their band, and only a wave that has done an entry draws the next (the counter starts behind the waves'
    // own entries: `bias`).
    struct Disp {                                            // the desk's dispatch words, in registers between two units only
        int stage;
        uint32_t bias, size, nun, grab, band, helped, late_helped, k, d, list_kind, list_off, ctr;
    };
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t n_waves = gridDim.x * (kBlock / 64u), wave_id = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);
    auto n_tasks_of = [&]() { return min((uint32_t)__builtin_amdgcn_readfirstlane((int)(fb_valid && A.fb.cnt_r->total ? A.fb.cnt_r->n_heavy : 0u)), A.fb.n_slots) << px_log2; };
    auto enter_list = [&](Disp &D, uint32_t n) {
        D.nun = n; D.grab = 1u; D.size = n; D.ctr = (uint32_t)kWorkHeavy; D.list_kind = 0u; D.list_off = 0u;
    };
    // what the units this wave took from a band cost: added to the band's sum when the wave leaves it (the next
    // launch's helpers go by these sums)
    auto flush_band = [&](uint32_t band) {
        const unsigned long long bc = ((unsigned long long)desk(kDkBandCostHi) << 32) | desk(kDkBandCostLo);
        if (A.fb.cnt_w && lane == 0 && bc)
            atomicAdd(&A.fb.cnt_w->band[band], bc);
        dk[kDkBandCostLo] = 0u; dk[kDkBandCostHi] = 0u;
    };
    auto enter_band = [&](Disp &D, uint32_t y) {
        flush_band(D.band);
        D.band = y;
        D.list_kind = 1u; D.list_off = y * A.band_units;
        D.nun = (uint32_t)__builtin_amdgcn_readfirstlane((int)LS->n_norm[y]);
        D.grab = A.grab;
        D.size = (D.nun + D.grab - 1u) / D.grab;
        D.ctr = y;
    };
    auto enter_late = [&](Disp &D, uint32_t y) {
        flush_band(D.band);
        const uint32_t n_late = (uint32_t)__builtin_amdgcn_readfirstlane((int)LS->n_late), late_per = (n_late + 7u) / 8u;
        D.list_kind = 2u; D.list_off = y * late_per;
        D.nun = y * late_per < n_late ? min(late_per, n_late - y * late_per) : 0u;
        D.grab = A.grab_late;
        D.size = (D.nun + D.grab - 1u) / D.grab;
        D.ctr = (uint32_t)kWorkLate + y;
    };
    // The band with the most work left, by units still to hand out x what its units cost last time (wave-uniform;
    // -1: every band is used up).  The bands of an image cost very different amounts -- sky against silhouette --,
    // and helpers that go round in a fixed order reach the costliest band last.
    auto pick = [&]() -> int {
        const uint32_t y = lane & 7u, bsz = LS->n_norm[y];
        // what a unit of band y cost on average in the previous launch of this shape, x 16 (>= 1)
        uint32_t band_w = 16u;
        if (fb_valid && A.fb.cnt_r->total && bsz)
            band_w = (uint32_t)std::min<unsigned long long>(16ull * A.fb.cnt_r->band[y] / bsz + 1ull, 0xffffffull);
        const uint32_t tickets = (bsz + A.grab - 1u) / A.grab;
        const uint32_t c = __hip_atomic_load(&work[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t rem = tickets - min(c, tickets);
        unsigned long long key = rem ? (((unsigned long long)rem * band_w) << 3) | y : 0ull;
        for (int o = 1; o < 8; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), o);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            key = other > key ? other : key;
        }
        const uint32_t klo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)key);
        const uint32_t khi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(key >> 32));
        return (klo | khi) ? (int)(klo & 7u) : -1;
    };
    // The slice of the late units with the most tickets left (-1: none): going round the eight slices in order
    // costs a draw on the spot per slice, 6 us each at the very end of a wave's life.
    auto pick_late = [&]() -> int {
        const uint32_t y = lane & 7u, n_late = LS->n_late, late_per = (n_late + 7u) / 8u;
        const uint32_t nun = y * late_per < n_late ? min(late_per, n_late - y * late_per) : 0u;
        const uint32_t tickets = (nun + A.grab_late - 1u) / A.grab_late;
        const uint32_t c = __hip_atomic_load(&work[kWorkLate + y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t key = tickets > c ? ((tickets - c) << 3) | y : 0u;
        for (int o = 1; o < 8; o <<= 1)
            key = max(key, (uint32_t)__shfl_xor((int)key, o));
        key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
        return key ? (int)(key & 7u) : -1;
    };
    uint32_t ticket = 0;
    RT_TICK(const unsigned long long tick_begin = __builtin_amdgcn_s_memtime());
    {
        const uint32_t n_lists = (n_listed << px_log2) + n_early;
        Disp D = {};
        D.band = xcd;
        dk[kDkBandCostLo] = 0u; dk[kDkBandCostHi] = 0u; dk[kDkCostLo] = 0u; dk[kDkCostHi] = 0u;
        dk[kDkHeavy] = heavy_cost; dk[kDkEarly] = early_cost; dk[kDkLate] = late_cost;
        RT_TICK(dk[kDkTick1] = 0u; dk[kDkTick2] = 0u; dk[kDkTickN1] = 0u; dk[kDkTickN2] = 0u; dk[kDkTickA] = 0u; dk[kDkTickB] = 0u; dk[kDkTickC] = 0u; dk[kDkTickU] = 0u;
                for (int i = kDkTickBy; i < kDkWords; i++) dk[i] = 0u);
        D.stage = wave_id < n_lists ? -1 : 0;
        if (D.stage == -1) {
            enter_list(D, n_lists);
            ticket = wave_id;                                // its own entry of the lists, no draw
        } else {
            enter_band(D, xcd);
            if (lane == 0 && D.size)
                ticket = atomicAdd(&work[D.ctr], 1u);
        }
        D.k = D.grab;                                        // (no ticket in hand yet: the first pass of the loop takes `ticket`)
        dk[kDkStage] = (uint32_t)D.stage; dk[kDkBias] = 0u; dk[kDkSize] = D.size; dk[kDkNun] = D.nun; dk[kDkGrab] = D.grab;
        dk[kDkBand] = D.band; dk[kDkHelped] = 0u; dk[kDkLateHelped] = 0u; dk[kDkK] = D.k; dk[kDkD] = 0u;
        dk[kDkListKind] = D.list_kind; dk[kDkListOff] = D.list_off; dk[kDkCtr] = D.ctr;
    }
    for (;;) {
        asm volatile("" ::: "memory");                      // the desk is read afresh after every unit
        Disp D;
        D.stage = (int)desk(kDkStage); D.bias = desk(kDkBias); D.size = desk(kDkSize); D.nun = desk(kDkNun);
        D.grab = desk(kDkGrab); D.band = desk(kDkBand); D.helped = desk(kDkHelped); D.late_helped = desk(kDkLateHelped);
        D.k = desk(kDkK); D.d = desk(kDkD); D.list_kind = desk(kDkListKind); D.list_off = desk(kDkListOff); D.ctr = desk(kDkCtr);
        const uint32_t n_tasks = D.stage == -1 ? n_tasks_of() : 0u;   // (two loads from memory: only while on the lists)
        if (D.k >= D.grab || D.d * D.grab + D.k >= D.nun) {  // this ticket is used up: the one drawn meanwhile
            D.k = 0;
#ifdef RT_TICKWAIT
            {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                asm volatile("" ::: "memory");
                uint32_t tk = (uint32_t)__builtin_amdgcn_readfirstlane(ticket);
                asm volatile("" : "+s"(tk));
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                if (D.size) { dk[kDkTick1] = desk(kDkTick1) + (uint32_t)(t1 - t0); dk[kDkTickN1] = desk(kDkTickN1) + 1u; }
            }
#endif
            D.d = D.size ? (uint32_t)__builtin_amdgcn_readfirstlane(ticket) + D.bias : 0u;
            if (D.stage == -1)
                D.bias = min(n_waves, D.size);               // every later ticket of the lists is a drawn one
            while (D.d >= D.size) {                          // used up: on to the next stage
                D.bias = 0;
                if (D.stage < 0) {
                    D.stage = 0;
                    enter_band(D, xcd);
                } else if (D.stage == 7) {                    // this slice of the late units is used up: the fullest
                    const int y = ++D.late_helped < 16u ? pick_late() : -1;
                    if (y < 0) {
                        D.stage = 8;
                        break;
                    }
                    enter_late(D, (uint32_t)y);
                } else {
                    const int y = A.band_help ? pick() : ++D.helped < 8u ? (int)((xcd + D.helped) & 7u) : -1;
                    if (y >= 0) {
                        D.stage = 1;
                        enter_band(D, (uint32_t)y);
                    } else {                                 // every band is used up: what was kept for last
                        D.stage = 7;
                        enter_late(D, xcd);
                    }
                }
                D.d = D.size;
                if (D.size) {
                    RT_TICK(const unsigned long long ts0 = __builtin_amdgcn_s_memtime());
                    if (lane == 0)
                        ticket = atomicAdd(&work[D.ctr], 1u);
                    D.d = (uint32_t)__builtin_amdgcn_readfirstlane(ticket);
                    RT_TICK(asm volatile("" : "+s"(D.d)); dk[kDkTick2] = desk(kDkTick2) + (uint32_t)(__builtin_amdgcn_s_memtime() - ts0); dk[kDkTickN2] = desk(kDkTickN2) + 1u);
                }
            }
            if (D.stage >= 8)
                break;
        }
        const uint32_t idx = D.d * D.grab + D.k;
        D.k++;
        uint32_t unit, only_px = kAllPixels;
        if (D.stage == -1 && idx < n_tasks) {
            // (a listed id is a unit of this shape: the min only keeps a damaged list harmless --
            // unit n_units has no pixels)
            unit = min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.fb.heavy_r[idx >> px_log2]), n_units);
            only_px = idx & (px - 1u);
        } else if (D.stage == -1) {
            const uint32_t n_ea = min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.fb.cnt_r->n_early_a), A.fb.n_slots);
            const uint32_t e = idx - n_tasks, slot = e < n_ea ? e : A.fb.n_slots - 1u - (e - n_ea);
            unit = min((uint32_t)__builtin_amdgcn_readfirstlane((int)A.fb.early_r[slot]), n_units);
        } else {
            const uint32_t *const list = D.list_kind == 1u ? A.fb.norm : A.fb.late;
            unit = min((uint32_t)__builtin_amdgcn_readfirstlane((int)list[D.list_off + idx]), n_units);
        }
        // the ticket after this one is drawn inside the last unit of this one
        // (early form only while the dispenser has plenty left: more than two tickets for every wave of the launch)
        const bool last_of_ticket = D.k >= D.grab || idx + 1u >= D.nun;
        const bool draw_early = A.draw_early_ok && D.size - D.d > 2u * n_waves;
        // A listed pixel is the longest dependent chain of the launch (a dozen traversals one after another), and a
        // launch over a small partition is as long as that chain: its wave issues ahead of the three it shares the SIMD with.
        const bool prio = D.stage == -1 && idx < n_tasks && A.task_prio;
        dk[kDkStage] = (uint32_t)D.stage; dk[kDkBias] = D.bias; dk[kDkSize] = D.size; dk[kDkNun] = D.nun; dk[kDkGrab] = D.grab;
        dk[kDkBand] = D.band; dk[kDkHelped] = D.helped; dk[kDkLateHelped] = D.late_helped; dk[kDkK] = D.k; dk[kDkD] = D.d;
        dk[kDkListKind] = D.list_kind; dk[kDkListOff] = D.list_off; dk[kDkCtr] = D.ctr;
        dk[kDkDraw] = last_of_ticket ? (draw_early ? 2u : 1u) : 0u;
        dk[kDkInBand] = (D.stage == 0 || D.stage == 1) ? 1u : 0u;
        RT_TICK({ const uint32_t from = D.stage == -1 ? (idx < n_tasks ? 0u : 1u) : D.stage == 0 ? 2u : D.stage == 1 ? 3u : 4u;
                  dk[kDkTickFrom] = from >= 2u && desk(kDkTickU) < 4u ? from + 3u : from; });
        asm volatile("" ::: "memory");
        if (prio)
            __builtin_amdgcn_s_setprio(3);
#ifdef RT_STAMPS
        const unsigned long long t_u0 = __builtin_amdgcn_s_memtime();
        w_last_start = wall_clock64();
        w_last_unit = unit; w_last_stage = (unsigned long long)(D.stage + 2);
        const unsigned long long cost_before = desk(kDkCostLo);
        const bool is_task = D.stage == -1 && idx < n_tasks;
#endif
        fused_unit<CULL, WAVES >= 5>(A, env, unit, only_px, lane, col, col0, dk, work, ticket, sp);
        if (prio)
            __builtin_amdgcn_s_setprio(0);
#ifdef RT_STAMPS
        w_last_cost = (uint32_t)(desk(kDkCostLo) - (uint32_t)cost_before);
        if (is_task) {
            const unsigned long long t_u1 = __builtin_amdgcn_s_memtime();
            t_tasks_end = wall_clock64();
            n_tasks_done++;
            t_tasks_sum += t_u1 - t_u0;
            t_task_max = t_u1 - t_u0 > t_task_max ? t_u1 - t_u0 : t_task_max;
        }
#endif
    }
    flush_band(desk(kDkBand));
#ifdef RT_TICKWAIT
    if (lane == 0) {
        atomicAdd(&g_tick[0], __builtin_amdgcn_s_memtime() - tick_begin);
        atomicAdd(&g_tick[1], (unsigned long long)desk(kDkTick1)); atomicAdd(&g_tick[2], (unsigned long long)desk(kDkTick2));
        atomicAdd(&g_tick[3], (unsigned long long)desk(kDkTickN1)); atomicAdd(&g_tick[4], (unsigned long long)desk(kDkTickN2));
        atomicAdd(&g_tick[5], (unsigned long long)desk(kDkTickA)); atomicAdd(&g_tick[6], (unsigned long long)desk(kDkTickB));
        atomicAdd(&g_tick[7], (unsigned long long)desk(kDkTickC)); atomicAdd(&g_tick[8], (unsigned long long)desk(kDkTickU));
        for (int i = 0; i < 32; i++)
            if (dk[kDkTickBy + i]) atomicAdd(&g_tick[16 + i], (unsigned long long)dk[kDkTickBy + i]);
    }
#endif
    {
        const unsigned long long cost_sum = ((unsigned long long)desk(kDkCostHi) << 32) | desk(kDkCostLo);
        if (A.fb.cnt_w && lane == 0 && cost_sum)
            atomicAdd(&A.fb.cnt_w->total, cost_sum);
    }
#ifdef RT_STAMPS
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    sp.a[0] = t_end - t_begin;
    sp.flush(2);
    if (lane == 0) {
        // the launch's timeline: first wave start (as its complement), last listed task's end, last wave's end
        const uint32_t wid = (blockIdx.x * kBlock + threadIdx.x) >> 6;
        if (wid < 32768u) {
            g_wave_log[0][wid] = w_begin;
            g_wave_log[1][wid] = wall_clock64();
            g_wave_log[2][wid] = sp.a[11];
            g_wave_log[3][wid] = (unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) | (unsigned long long)(blockIdx.x & 7u) << 8 | n_tasks_done << 16;
            g_wave_log[4][wid] = w_last_start;
            g_wave_log[5][wid] = w_last_unit | w_last_stage << 32;
            g_wave_log[6][wid] = w_last_cost;
        }
        atomicMax(&g_stamps[1][12], ~w_begin);
        atomicMax(&g_stamps[1][14], (unsigned long long)wall_clock64());
        if (t_tasks_end) atomicMax(&g_stamps[1][13], t_tasks_end);
        if (n_tasks_done) { atomicAdd(&g_stamps[1][15], n_tasks_done); atomicAdd(&g_stamps[1][11], t_tasks_sum); atomicMax(&g_stamps[1][10], t_task_max); }
    }
#endif
}

// ------------------------------------------------------------------ Scene::hit on a ray queue
template <bool CULL>
__global__ __launch_bounds__(kBlock) void k_intersect(const TravArgs T, RayQueue q, uint32_t count,
                                                      float *__restrict__ out_t,
                                                      int32_t *__restrict__ out_id,
                                                      uint32_t *__restrict__ work,
                                                      uint32_t *__restrict__ work_next, int dynamic, int lane_mode)
{
    __shared__ uint32_t s_lvl[kMaxLevels];
    if (threadIdx.x < (uint32_t)kMaxLevels)
        s_lvl[threadIdx.x] = T.tv.level_off[threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = (count + kPathChunk - 1u) / kPathChunk;
    const uint32_t lvloff = T.tv.level_off[lane & (kMaxLevels - 1)];
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)kWorkSlots)
        work_next[threadIdx.x] = 0u;
    const uint32_t n_waves = (gridDim.x * kBlock) >> 6;
    uint32_t next_chunk = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    for (;;) {
        uint32_t chunk = 0;
        if (dynamic) {
            if (lane == 0)
                chunk = atomicAdd(&work[0], 1u);
            chunk = __builtin_amdgcn_readfirstlane(chunk);
        } else {
            chunk = next_chunk;
            next_chunk += n_waves;
        }
        if (chunk >= n_chunks)
            break;
        v3 o[2], d[2];
        bool live[2];
        uint32_t i[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            i[h] = chunk * kPathChunk + (uint32_t)h * 64u + lane;
            live[h] = i[h] < count;
            o[h] = mk3(0.0f, 0.0f, 0.0f); d[h] = mk3(0.0f, 0.0f, 0.0f);
            if (live[h]) {
                o[h] = mk3(q.ox[i[h]], q.oy[i[h]], q.oz[i[h]]);
                d[h] = mk3(q.dx[i[h]], q.dy[i[h]], q.dz[i[h]]);
            }
        }
        const float inf2[2] = {kInf, kInf}, none2[2] = {-1.0f, -1.0f};
        RayPair w;
        uint32_t cost = 0;
        pair_set(w, o, d, inf2, none2, live, T.tv.extent);
        Stamps sp0;
        // A caller's rays may share an origin or a target (camera rays, shadow rays: the wave-wide walk) or
        // nothing at all (the per-lane walk, traverse_lane).  lane_mode 2 decides per chunk: the wave's rays
        // count as coherent when their directions lie within 0.35 of each other per component and their
        // origins within 1/64 of the culling extent (w.oct bit 3 -- all in one octant -- is not enough:
        // rays across an octant still visit most of the tree between them).
        bool per_lane = lane_mode == 1;
        if (CULL && lane_mode == 2) {
            float lo[6], hi[6];
            const float vv[2][6] = {{o[0].x, o[0].y, o[0].z, d[0].x, d[0].y, d[0].z}, {o[1].x, o[1].y, o[1].z, d[1].x, d[1].y, d[1].z}};
#pragma unroll
            for (int k = 0; k < 6; k++) {
                lo[k] = fminf(live[0] ? vv[0][k] : 3.0e38f, live[1] ? vv[1][k] : 3.0e38f);
                hi[k] = fmaxf(live[0] ? vv[0][k] : -3.0e38f, live[1] ? vv[1][k] : -3.0e38f);
                for (int x = 32; x > 0; x >>= 1) {
                    lo[k] = fminf(lo[k], __shfl_xor(lo[k], x));
                    hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], x));
                }
            }
            const float eo = T.tv.extent * (1.0f / 64.0f);
            const bool coherent = (hi[0] - lo[0] <= eo) & (hi[1] - lo[1] <= eo) & (hi[2] - lo[2] <= eo) &
                                  (hi[3] - lo[3] <= 0.35f) & (hi[4] - lo[4] <= 0.35f) & (hi[5] - lo[5] <= 0.35f);
            per_lane = !coherent;
        }
        if (CULL && per_lane && T.tv.top >= 3)
            traverse_lane<CULL, true>(T, s_lvl, w, live[0], live[1], cost);
        else
            traverse<CULL, false, true, true>(T, lvloff, w, lane, 0u, nullptr, cost, sp0);   // triangle lights included (Scene.cu:43-58)
        if (live[0]) { out_t[i[0]] = w.best.x; out_id[i[0]] = w.idx; }
        if (live[1]) { out_t[i[1]] = w.best.y; out_id[i[1]] = w.idy; }
    }
}

// ------------------------------------------------------------------ readback helpers
__global__ void k_scatter_image(const uint32_t *__restrict__ pixmap, uint32_t n,
                                const float4 *__restrict__ fb, const int32_t *__restrict__ first_id,
                                const float *__restrict__ first_t, float4 *rgba, int32_t *prim,
                                float *t)
{
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    uint32_t pix = pixmap[p];
    if (rgba) rgba[pix] = fb[p];
    if (prim) prim[pix] = first_id[p];
    if (t) t[pix] = first_t[p];
}

// rt_reset: every per-pixel array of the partition in one launch (what re-running initCUDA does, kernel.cu:193-256)
__global__ void k_reset_pixels(uint32_t *__restrict__ rng_ctr, uint32_t *__restrict__ last_draws, float4 *__restrict__ fb,
                               int32_t *__restrict__ first_id, float *__restrict__ first_t, uint32_t n)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    rng_ctr[p] = 0u;
    last_draws[p] = 0u;
    fb[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    first_id[p] = -1;
    first_t[p] = 0.0f;
}

__global__ void k_fill_i32(int32_t *a, size_t n, int32_t value)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        a[i] = value;
}

// Scene::hit on caller rays: the Ray constructor's normalisation (Ray.cu:2), then
// the same closest-hit kernel; this kernel only turns t into the returned bool.
__global__ void k_prepare_rays(RayQueue q, const float *__restrict__ org,
                               const float *__restrict__ dir, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    Ray r = make_ray(mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]),
                     mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    q.ox[i] = r.o.x; q.oy[i] = r.o.y; q.oz[i] = r.o.z;
    q.dx[i] = r.d.x; q.dy[i] = r.d.y; q.dz[i] = r.d.z;
}
__global__ void k_ishit(const float *__restrict__ t, int32_t *ishit, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        ishit[i] = scene_ishit(t[i]) ? 1 : 0;
}

// d_mipmap (Texture.cu:14-45), one thread per texel of the new level: four bilinear taps of the level
// above at the texel's corners in normalised coordinates (:30-33), averaged (:36), clamped to 1 (:37).
// px, py = the `1.0 / float(imageW)` of :19-20, rounded on the host.
__global__ void k_mip_level(const float4 *__restrict__ src, int sw, int sh, float4 *__restrict__ dst, int dw, int dh,
                            float px, float py)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)dw * (uint32_t)dh)
        return;
    const uint32_t x = i % (uint32_t)dw, y = i / (uint32_t)dw;
    const float4 a = tex_fetch(src, sw, sh, (float)(x + 0u) * px, (float)(y + 0u) * py);
    const float4 b = tex_fetch(src, sw, sh, (float)(x + 1u) * px, (float)(y + 0u) * py);
    const float4 c = tex_fetch(src, sw, sh, (float)(x + 1u) * px, (float)(y + 1u) * py);
    const float4 d = tex_fetch(src, sw, sh, (float)(x + 0u) * px, (float)(y + 1u) * py);
    float4 o;
    o.x = fminf((a.x + b.x + c.x + d.x) / 4.0f, 1.0f);
    o.y = fminf((a.y + b.y + c.y + d.y) / 4.0f, 1.0f);
    o.z = fminf((a.z + b.z + c.z + d.z) / 4.0f, 1.0f);
    o.w = fminf((a.w + b.w + c.w + d.w) / 4.0f, 1.0f);
    dst[i] = o;
}

// ------------------------------------------------------------------ post-process
// HDRKernel (PostProcess/PostProcess.cuh:29-56): ACES filmic curve + saturate.
__device__ __forceinline__ float aces1(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return fminf(1.0f, fmaxf(0.0f, (x * (a * x + b)) / (x * (c * x + d) + e)));
}
__global__ void k_tonemap(const float4 *__restrict__ in, float4 *__restrict__ out, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float4 c = in[i];
    out[i] = make_float4(aces1(c.x), aces1(c.y), aces1(c.z), 1.0f);
}
// filterKernel (PostProcess.cuh:59-104): 3x3 window gathered x-major, the
// reference's exchange sort by luma (descending), element count/2.
__global__ void k_median3x3(const float4 *__restrict__ in, float4 *__restrict__ out, int width,
                            int height)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height)
        return;
    float4 temp[9];
    float ty[9];
    int count = 0;
    for (int i = x - 1; i <= x + 1; i++)
        for (int j = y - 1; j <= y + 1; j++)
            if (i >= 0 && i < width && j >= 0 && j < height) {
                float4 c = in[(size_t)j * width + i];
                temp[count] = c;
                ty[count++] = 0.299f * c.x + 0.587f * c.y + 0.114f * c.z;
            }
    for (int i = 0; i < count; i++)
        for (int j = 0; j < i; j++)
            if (ty[i] > ty[j]) {
                float sf = ty[i];
                float4 s4 = temp[i];
                ty[i] = ty[j];
                temp[i] = temp[j];
                ty[j] = sf;
                temp[j] = s4;
            }
    out[(size_t)y * width + x] = temp[count / 2];
}


// ------------------------------------------------------------------ launchers
static TravArgs trav_args(const DevView &v)
{
    TravArgs T{};
    T.tv = v.tree; T.tris = v.tris; T.spheres = v.spheres;
    T.n_tris = v.n_tris; T.n_spheres = v.n_spheres;
    T.sorted_planes = v.sorted_planes;
    T.tlights = v.tlights; T.n_tlights = v.n_tlights;
    return T;
}

// Blocks of four waves, one 128-pixel chunk per wave at a time; at most max_blocks.
static uint32_t grid_for_chunks(uint32_t n, int max_blocks)
{
    const uint32_t per_block = (kBlock / 64) * kPathChunk;
    uint32_t g = (n + per_block - 1) / per_block;
    return std::min<uint32_t>(std::max<uint32_t>(g, 1u), (uint32_t)std::max(max_blocks, 1));
}

Knobs read_knobs()
{
    Knobs k;
    auto geti = [](const char *name, int &dst) { if (const char *e = std::getenv(name)) dst = std::atoi(e); };
    geti("RT_LANE_TRAVERSAL", k.lane_traversal);
    geti("RT_BOUNCE_WAVES", k.bounce_waves);
    geti("RT_BOUNCE_QUEUE", k.bounce_queue);
    geti("RT_BOUNCE_PAIRS", k.bounce_pairs);
    geti("RT_HEAVY_X4", k.heavy_x4);
    geti("RT_GENERAL_REGEN", k.general_regen);
    geti("RT_GENERAL_WAVES", k.general_waves);
    geti("RT_UNITS_PER_BLOCK", k.units_per_block);
    geti("RT_BATCH_WAVES", k.batch_waves);
    geti("RT_INTERSECT_LANE", k.intersect_lane);
    geti("RT_TASK_PRIO", k.task_prio);
    geti("RT_EARLY_X4", k.early_x4);
    geti("RT_BAND_HELP", k.band_help);
    geti("RT_LATE_X16", k.late_x16);
    geti("RT_GRAB", k.grab);
    geti("RT_DRAW_EARLY", k.draw_early);
    geti("RT_GRAB_LATE", k.grab_late);
    return k;
}

bool path_renders_runs(const DevView &v)
{
    return !v.general && !(v.cull && v.tree.top >= 3);
}

bool launch_path(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, int max_depth,
                 int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                 const ChunkFeedback *fb, uint32_t n_samples, const Knobs &knobs, const BounceQueue *bq)
{
    if (n == 0)
        return false;
    PathArgs A{};
    A.v = v;
    A.T = trav_args(v);
    A.p0 = p0; A.n = n;
    A.max_depth = max_depth;
    A.occlusion = v.n_spheres == 0;
    A.ctl = ctl;
    A.n_samples = path_renders_runs(v) ? std::max(n_samples, 1u) : 1u;
    A.lane_traversal = knobs.lane_traversal != 0 && v.cull && v.tree.top >= 3;   // >= 256 triangles
    // (runs keep the strided chunk assignment of shallow scenes: handing chunks out on demand was slower
    // at every run length, C2 3.53 -> 3.67..4.72 ms)
    const bool deep = v.cull && v.tree.top >= 3;         // >= 256 triangles
    const int dynamic = deep;
    if (fb && deep) A.fb = *fb;
    const bool d0 = max_depth == 0;
    // Waves per SIMD the register allocation aims at.  Depth 0 with culling reaches this kernel
    // only over a shallow tree (deeper ones take the batch pipeline, rt_context.hip): cheap,
    // uniform chunks, three waves per SIMD measured best (C2: +17 % over two).  With bounces three as
    // well, spills and all (depth 6: C2 3.89 -> 3.12 ms per sample, C3 6.2 -> 6.0, C4 11.1 -> 10.3;
    // RT_BOUNCE_WAVES=2 for the two-wave build).  The grid is what is resident at once.
    const uint32_t n_chunks = (n + kPathChunk - 1) / kPathChunk;
    const bool bounce3 = !d0 && v.cull && knobs.bounce_waves != 2;
    const int waves = (d0 && v.cull) || bounce3 ? 3 : 2;
    const uint32_t g = grid_for_chunks(n, max_blocks * waves);
    // Cooperative form from which multiple of the mean chunk cost on: a launch with many
    // chunks per resident wave is throughput-bound, and sharing a chunk among four waves
    // costs throughput (8x); one with few is as long as its costliest chunk, which the
    // cooperative form shortens (measured on 1/8 partitions of C3 and C4: 2x and 4x best).
    const uint32_t slots = (uint32_t)std::max(max_blocks, 1) * (uint32_t)waves * (kBlock / 64);
    A.heavy_x4 = n_chunks <= slots ? 8u : n_chunks <= 3u * slots ? 16u : 4u * (uint32_t)kHeavyFactor;
    if (A.lane_traversal)
        A.heavy_x4 *= 2u;            // the cooperative form shares camera and first shadow rays only: later (C3 3.03 -> 2.97 ms, C4 4.54 -> 4.24)
    if (knobs.heavy_x4 > 0) A.heavy_x4 = (uint32_t)knobs.heavy_x4;   // diagnostics
    if (v.general) {
        // triangle lights, non-Lambertian materials or textures: the general kernel -- persistent waves
        // (one block per CU is resident), no chunk feedback
        A.fb = ChunkFeedback{};
        A.occlusion = false;
        const int regen = knobs.general_regen;
        A.regen = regen < 0 ? v.tree.top < 3 : regen != 0;   // < 256 triangles: coherence of a chunk's camera rays buys nothing
        const int forced = knobs.general_waves;
        const int gw = forced ? std::min(2, std::max(1, forced)) : v.env.n ? 1 : 2;
        const uint32_t ga = grid_for_chunks(n, max_blocks * gw);
        if (!v.cull)      k_path_general<false, 1><<<ga, kBlock, 0, s>>>(A, work, work_next, dynamic);
        else if (gw == 1) k_path_general<true, 1><<<ga, kBlock, 0, s>>>(A, work, work_next, dynamic);
        else              k_path_general<true, 2><<<ga, kBlock, 0, s>>>(A, work, work_next, dynamic);
        return false;
    }
    if (v.cull && bounce3 && deep && A.lane_traversal && knobs.bounce_queue && bq && bq->jobs && bq->cap >= n) {
        // bounces over a deep tree: camera rays, direct light and the BRDF sample in chunks (coherent rays), the
        // rest of every path from a queue, lane by lane (k_bounce)
        A.bq = *bq;
        A.bq.pairs = knobs.bounce_pairs;
        k_path<true, false, 3, true><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
        if (knobs.bounce_queue >= 2) k_bounce_sm<true><<<(uint32_t)std::max(max_blocks, 1) * 3u, kBlock, 0, s>>>(A);
        else                         k_bounce<true><<<(uint32_t)std::max(max_blocks, 1) * 3u, kBlock, 0, s>>>(A);
        return true;
    }
    if (v.cull) {
        if (bounce3)  k_path<true, false, 3><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
        else if (!d0) k_path<true, false, 2><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
        else     k_path<true, true, 3><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
    } else {
        if (d0) k_path<false, true, 2><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
        else    k_path<false, false, 2><<<g, kBlock, 0, s>>>(A, work, work_next, dynamic);
    }
    return false;
}

// Samples a wave holds: the smallest power of two that takes the whole batch (<= 64).
static uint32_t batch_ss_log2(uint32_t n_samples)
{
    uint32_t l = 0;
    while (l < 6u && (1u << l) < n_samples) l++;
    return l;
}

// Blocks of a launch over `units` wave-sized units: at least units_per_block / 4 units per wave, at
// most the resident capacity.  (RT_UNITS_PER_BLOCK overrides, for tuning.)
static uint32_t batch_grid(uint32_t units, uint32_t cap, const Knobs &knobs)
{
    const uint32_t per_block = (uint32_t)std::max(1, knobs.units_per_block);
    return std::min<uint32_t>(std::max<uint32_t>(units / per_block, 1u), cap);
}

// Units per XCD band: an eighth of the range, rounded up to whole waves of k_classify.
uint32_t batch_band_units(uint32_t units)
{
    return (((units + 7u) / 8u) + 63u) / 64u * 64u;
}

static BatchArgs batch_args(const DevView &v, uint32_t p0, uint32_t n, uint32_t n_samples, int cnt0, Ctl *ctl)
{
    BatchArgs A{};
    A.v = v;
    A.T = trav_args(v);
    A.p0 = p0; A.n = n; A.n_samples = n_samples;
    A.ss_log2 = batch_ss_log2(n_samples);
    A.cnt0 = cnt0;
    A.occlusion = v.n_spheres == 0;
    A.ctl = ctl;
    return A;
}

constexpr size_t kLastLevelCacheBytes = 256u << 20;

void launch_batch(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, uint32_t n_samples,
                  int cnt0, int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                  const BatchFeedback *fb, const Knobs &knobs)
{
    if (n == 0 || n_samples == 0)
        return;
    BatchArgs A = batch_args(v, p0, n, n_samples, cnt0, ctl);
    A.fb = *fb;                                          // (the lists are always there; cnt_w == nullptr: no feedback)
    const uint32_t px = kPathChunk >> A.ss_log2;
    const uint32_t units = (n + px - 1) / px;
    // Five waves per SIMD only over a scene that does not fit the last-level cache (256 MB): there the
    // traversal waits for HBM and the fifth wave pays (C5 416 -> 390 ms per 16 samples).  Elsewhere it buys
    // 1.5-2 % (C3 14.1 -> 13.9 ms, C4 57.5 -> 56.6) for 4.6 GB instead of 0.27 GB of HBM traffic per launch
    // -- its spilled registers' scratch (43 MB over the chip) no longer fits the L2s -- and a small
    // partition, which is as long as its longest unit, gets slower (an eighth of C3 2.36 -> 2.42 ms).
    const size_t scene_bytes = ((size_t)v.tree.level_off[v.tree.top] + 1u) * sizeof(NodeGroup) + (size_t)v.n_tris * sizeof(TriRec);
    // (Since the ray pairs' arithmetic is written out per ray the five-wave build spills 39 registers, not 77, and would
    // also pay over C4 -- 115 MB, beyond the L2s: whole frame 197.3 -> 193.0 ms, an eighth +2 % -- but its scratch, 52 MB
    // over the chip, does not fit the L2s: 32 GB of HBM traffic per launch instead of 2.0.  Not taken for 2 %.)
    const int waves = knobs.batch_waves ? (knobs.batch_waves >= 5 ? 5 : 4) : scene_bytes > kLastLevelCacheBytes ? 5 : 4;
    const uint32_t g = batch_grid(units, (uint32_t)std::max(max_blocks, 1) * (uint32_t)waves, knobs);
    A.task_prio = knobs.task_prio != 0;
    A.early_x4 = (uint32_t)std::max(0, knobs.early_x4);
    A.band_help = knobs.band_help != 0;
    A.late_x16 = (uint32_t)std::max(0, knobs.late_x16);
    // Units per ticket of a band's dispenser.  A draw costs the unit that makes it 2-5 us (the epilogue's first reload of
    // a spilled register waits for it: DESIGN.md 4.1), a ticket of several units makes the launch's tail that many units
    // long.  By default (RT_GRAB=0) several only where a wave has many short units to do -- the four-wave build at more
    // than 192 units per wave (C3 whole: 5 per ticket, -2 % of the step; a half, a quarter or an eighth of it: 1), and
    // only once the lists are in use (from the third launch of a shape: before, costly units are met in band order
    // and a ticket of several makes the tail long -- C3's first two launches 14.9 -> 16.5-17.7 ms);
    // C4's units are long enough not to feel the draw, C5's (five-wave build) so long that the tail costs more.
    const uint32_t units_per_wave = units / std::max(1u, g * (kBlock / 64u));
    A.grab = knobs.grab > 0 ? (uint32_t)std::min(64, knobs.grab)
                            : (waves == 5 || units_per_wave < 192u || !A.fb.settled ? 1u : std::min(8u, units_per_wave / 48u));
    A.grab_late = (uint32_t)std::min(64, std::max(1, knobs.grab_late));
    A.band_units = batch_band_units(units);
    A.draw_early_ok = knobs.draw_early != 0;
    // the previous launch's flags as this launch's lists (stream-ordered: that launch is over, this one not begun)
    const uint32_t n_cls = std::max(units, A.fb.flag_z ? A.fb.n_slots : 0u);
    k_classify<<<(n_cls + 255u) / 256u, 256, 0, s>>>(A.fb.flag_r, A.fb.flags_valid, units, A.band_units, A.fb.norm_w, A.fb.late_w,
                                                      A.fb.lists_w, A.fb.flag_z, A.fb.cnt_z, A.fb.n_slots);
    // the batch kernels run with culling only (rt_render)
    if (waves == 5) k_batch<true, 5><<<g, kBlock, 0, s>>>(A, work, work_next);
    else            k_batch<true, 4><<<g, kBlock, 0, s>>>(A, work, work_next);
}

void launch_intersect(hipStream_t s, const DevView &v, const RayQueue &q, uint32_t count,
                      float *out_t, int32_t *out_id, int max_blocks, uint32_t *work,
                      uint32_t *work_next, const Knobs &knobs)
{
    if (count == 0)
        return;
    const int dynamic = v.cull && v.tree.top >= 3;
    const uint32_t g = grid_for_chunks(count, max_blocks * 6);
    const int lane_mode = knobs.intersect_lane;              // diagnostics: 0 wave-wide, 1 per lane, 2 (default) per chunk
    if (v.cull) k_intersect<true><<<g, kBlock, 0, s>>>(trav_args(v), q, count, out_t, out_id, work, work_next, dynamic, lane_mode);
    else        k_intersect<false><<<g, kBlock, 0, s>>>(trav_args(v), q, count, out_t, out_id, work, work_next, dynamic, lane_mode);
}

void launch_scatter_image(hipStream_t s, const uint32_t *pixmap, uint32_t n, const float4 *fb,
                          const int32_t *first_id, const float *first_t, float4 *rgba,
                          int32_t *prim, float *t)
{
    if (n)
        k_scatter_image<<<(n + 255) / 256, 256, 0, s>>>(pixmap, n, fb, first_id, first_t, rgba,
                                                        prim, t);
}

void launch_mip_level(hipStream_t s, const float4 *src, int sw, int sh, float4 *dst, int dw, int dh)
{
    const float px = (float)(1.0 / (double)(float)dw), py = (float)(1.0 / (double)(float)dh);
    const size_t n = (size_t)dw * dh;
    k_mip_level<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(src, sw, sh, dst, dw, dh, px, py);
}
void launch_reset_pixels(hipStream_t s, const DevView &v, uint32_t n)
{
    if (n)
        k_reset_pixels<<<(n + 255) / 256, 256, 0, s>>>(v.rng_ctr, v.last_draws, v.fb, v.first_id, v.first_t, n);
}
void launch_fill_i32(hipStream_t s, int32_t *a, size_t n, int32_t value)
{
    if (n)
        k_fill_i32<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, n, value);
}

void launch_prepare_rays(hipStream_t s, const RayQueue &q, const float *org, const float *dir,
                         uint32_t n)
{
    if (n)
        k_prepare_rays<<<(n + 255) / 256, 256, 0, s>>>(q, org, dir, n);
}

void launch_ishit(hipStream_t s, const float *t, int32_t *ishit, uint32_t n)
{
    if (n)
        k_ishit<<<(n + 255) / 256, 256, 0, s>>>(t, ishit, n);
}

void launch_tonemap(hipStream_t s, const float4 *in, float4 *out, uint32_t n)
{
    if (n)
        k_tonemap<<<(n + 255) / 256, 256, 0, s>>>(in, out, n);
}

void launch_median3x3(hipStream_t s, const float4 *in, float4 *out, int width, int height)
{
    dim3 b(16, 16), g((width + 15) / 16, (height + 15) / 16);
    k_median3x3<<<g, b, 0, s>>>(in, out, width, height);
}

} // namespace rt
