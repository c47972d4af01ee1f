// rt_kernels.hip -- gfx950 kernels of the per-pixel render path and their launchers.
//
// Wavefront organisation (one lane = one ray, 64-wide waves):
//   k_gen        ray generation            kernel.cu:65-73   (jitter + Camera::generateRay)
//   k_intersect  closest hit, brute force  Scene/Scene.cu:8-67 -> Model/Model.cu:91-137, 349-410
//   k_shade      per light: illumination, BRDF, shadow-ray spawn with ballot compaction
//                                          Scene/Scene.cu:69-124, 205-225; Light/Light.cu:7-14
//   k_intersect  (again) on the compacted shadow-ray queue      Scene.cu:225
//   k_finish     blocked test + accumulate, BRDF sampling, roulette, next-ray
//                compaction, framebuffer running mean
//                                          Scene.cu:227-251; core/PathTracer.cu:73-88; kernel.cu:65-77
// All queue lengths live in device memory (Ctl), kernels are grid-stride over
// 64-ray chunks, so a whole sample is enqueued without a host round trip.
#include "rt_kernels.h"

#include <algorithm>

#include "rt_device.h"

namespace rt {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr float kHitMax = kInf - 0.1f;      // Scene.cu:66

__device__ __forceinline__ bool scene_ishit(float t) { return t > 0.001f && t < kHitMax; }   // Scene.cu:66

// ------------------------------------------------------------------ control
__global__ void k_ctl_begin(Ctl *ctl, uint32_t n)
{
    ctl->n_cur = n;
    ctl->n_next = 0;
}
__global__ void k_ctl_next_bounce(Ctl *ctl)
{
    ctl->bounce_total += ctl->n_next;
    ctl->n_cur = ctl->n_next;
    ctl->n_next = 0;
}

// ------------------------------------------------------------------ ray generation
// One lane per local pixel of the batch [p0, p0 + n).  kernel.cu:65-73 with the H4
// convention: the jitter is (.x, .y) of one curand_uniform4 of the pixel's own stream.
__global__ __launch_bounds__(kBlock) void k_gen(DevView v, uint32_t p0, uint32_t n)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
        return;
    uint32_t p = p0 + i;
    uint32_t pix = v.pixmap[p];
    int x = (int)(pix % (uint32_t)v.width), y = (int)(pix / (uint32_t)v.width);
    uint32_t draw = v.rng_ctr[p];
    float u[4];
    pixel_uniform4(pix, draw, u);
    v.rng_ctr[p] = draw + 1;
    Ray r = camera_ray(v.cam, x - v.width / 2 + u[0], y - v.height / 2 + u[1]);   // kernel.cu:73
    RayQueue q = v.rays[0];
    q.ox[i] = r.o.x; q.oy[i] = r.o.y; q.oz[i] = r.o.z;
    q.dx[i] = r.d.x; q.dy[i] = r.d.y; q.dz[i] = r.d.z;
    q.path[i] = p;
    // pathTracer locals (PathTracer.cu:19-20): beta = 1, res = 0
    v.beta[0][p] = 1.0f; v.beta[1][p] = 1.0f; v.beta[2][p] = 1.0f;
    v.res[0][p] = 0.0f; v.res[1][p] = 0.0f; v.res[2][p] = 0.0f;
}

// ------------------------------------------------------------------ closest hit
// Scene::hit (Scene.cu:8-67): the closest of all primitives the reference's scan would
// accept -- min over (t, reference index) -- found without visiting most of them.
//
// A wave owns a chunk of 64 * 2P rays (P packed pairs per lane).  The triangle side is
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

#ifdef RT_TRAV_STATS
struct ChunkLog { unsigned long long t0, t1; unsigned cost, coop; };
__device__ ChunkLog g_log[1 << 17];
__device__ unsigned g_log_on;
extern "C" void rt_debug_log(void *out, unsigned on, unsigned n)
{
    hipDeviceSynchronize();
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_log), sizeof(ChunkLog) * n);
    hipMemcpyToSymbol(HIP_SYMBOL(g_log_on), &on, sizeof(on));
}
#endif

template <int P>
struct WaveRays {
    RayPair r[P];
};

constexpr unsigned long long kHeavyFactor = 8;   // heavy = this many times the mean chunk cost (4 and 16 measured slower)
constexpr int kSplitLevel = 2;       // cooperative chunks: subtrees of 4^(kSplitLevel+1) triangles

__device__ __forceinline__ unsigned long long pack_hit(float t, int32_t id)
{
    return ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)(id + 1);
}

template <int P>
__device__ __forceinline__ void test_triangle(const TriRec *__restrict__ tris, uint32_t node,
                                              WaveRays<P> &w, uint32_t lanebits, int j)
{
    const TriRec T = tris[node];
#pragma unroll
    for (int p = 0; p < P; p++) {
        bool cx = (lanebits >> (j * 2 * P + 2 * p)) & 1u, cy = (lanebits >> (j * 2 * P + 2 * p + 1)) & 1u;
        if (__ballot(cx | cy)) {
            tri_test_pair(T, w.r[p], cx, cy);
            // shadow rays (occlusion form): a hit safely below the threshold ends the ray --
            // nothing can pass its node tests any more
            RayPair &r = w.r[p];
            if (r.best.x < r.stop.x) { r.bestp.x = -1.0f; r.wildx = false; }
            if (r.best.y < r.stop.y) { r.bestp.y = -1.0f; r.wildy = false; }
        }
    }
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

struct IsectArgs {
    TreeView tv;
    const TriRec *tris;
    const SphereRec *spheres;
    int n_tris, n_spheres;
    RayQueue q;
    float *out_t;
    int32_t *out_id;
    float *sh_thr;
    uint32_t stride;
    uint32_t count, chunks_per_slab;
    bool occlusion;
    uint32_t lvloff;                 // lane l: first group of level l
    uint32_t heavy_cost;             // cost from which a chunk counts as heavy (feedback)
    ChunkFeedback fb;
    bool feedback;
};

// One chunk.  COOP: called by all four waves of the block with the same chunk; s_best is
// the block's LDS merge array (64 * 2P entries), s_cost its cost accumulator.
//
// SHADOW = true is the shadow-ray form of the same traversal (Scene.cu:224-227): the queue
// holds n_slabs slabs of `stride` slots, slot j * stride + i = the ray from path i towards
// light j (light-major, so a chunk's rays head for ONE light from neighbouring pixels);
// sh_thr[slot] is the threshold |o - l| - 0.01 of Scene.cu:227, or < 0 for an empty slot, and
// receives the verdict: 1 = blocked, 0 = visible.  Without quadrics in the scene the closest
// hit itself is not needed, only `|hit - o| < thr`: hits beyond thr + m cannot matter and the
// ray stops at the first hit below thr - m, m covering the difference between t and the
// reference's |fl(o + t d) - o|; inside the band the true closest hit decides, exactly as
// the reference computes it.
template <int P, bool CULL, bool SHADOW, bool COOP>
__device__ __forceinline__ void process_chunk(const IsectArgs &A, uint32_t chunk, uint32_t lane,
                                              uint32_t wib, unsigned long long *s_best,
                                              uint32_t *s_cost)
{
    constexpr uint32_t R = 2 * P;
    const TreeView &tv = A.tv;
    const uint32_t count = A.count;
    const uint32_t slab = SHADOW ? chunk / A.chunks_per_slab : 0u;
    const uint32_t base = (chunk - slab * A.chunks_per_slab) * 64u * R;     // path position
    const uint32_t qbase = slab * A.stride + base;                         // queue position
    const RayQueue &q = A.q;
    uint32_t cost = 0;                           // node groups expanded + triangles tested
#ifdef RT_TRAV_STATS
    unsigned long long tlog0 = (unsigned long long)wall_clock64();
#endif
    WaveRays<P> w;
    float thr[R];
    bool live[R];
    bool any_live = false;
#pragma unroll
    for (int k = 0; k < (int)R; k++) {
        live[k] = base + k * 64u + lane < count;
        thr[k] = 0.0f;
        if (SHADOW) {
            thr[k] = live[k] ? A.sh_thr[qbase + k * 64u + lane] : -1.0f;
            live[k] = thr[k] >= 0.0f;
        }
        any_live |= live[k];
    }
    if (SHADOW && !__ballot(any_live)) {         // a chunk of empty slots (e.g. sky pixels)
        if (A.feedback && lane == 0 && (!COOP || wib == 0))
            A.fb.flag_w[chunk] = 0;
        return;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        RayPair &r = w.r[p];
        uint32_t i0 = qbase + (2 * p) * 64u + lane, i1 = i0 + 64u;
        // dead lanes: parked far away with an empty t window
        float o0[3] = {0.0f, 1e18f, 0.0f}, d0[3] = {1.0f, 0.0f, 0.0f};
        float o1[3] = {0.0f, 1e18f, 0.0f}, d1[3] = {1.0f, 0.0f, 0.0f};
        float b0 = 0.0f, b1 = 0.0f;
        if (live[2 * p]) {
            o0[0] = q.ox[i0]; o0[1] = q.oy[i0]; o0[2] = q.oz[i0];
            d0[0] = q.dx[i0]; d0[1] = q.dy[i0]; d0[2] = q.dz[i0];
            b0 = kInf;                                   // IntersectRecord(): t = INF
        }
        if (live[2 * p + 1]) {
            o1[0] = q.ox[i1]; o1[1] = q.oy[i1]; o1[2] = q.oz[i1];
            d1[0] = q.dx[i1]; d1[1] = q.dy[i1]; d1[2] = q.dz[i1];
            b1 = kInf;
        }
        r.stop = f2{-1.0f, -1.0f};
        if (A.occlusion) {
            float m0 = 1e-4f * (fabsf(thr[2 * p]) + fabsf(o0[0]) + fabsf(o0[1]) + fabsf(o0[2])) + 1e-6f;
            float m1 = 1e-4f * (fabsf(thr[2 * p + 1]) + fabsf(o1[0]) + fabsf(o1[1]) + fabsf(o1[2])) + 1e-6f;
            if (live[2 * p]) { b0 = fminf(kInf, thr[2 * p] + m0); r.stop.x = thr[2 * p] - m0; }
            if (live[2 * p + 1]) { b1 = fminf(kInf, thr[2 * p + 1] + m1); r.stop.y = thr[2 * p + 1] - m1; }
        }
        r.ox = f2{o0[0], o1[0]}; r.oy = f2{o0[1], o1[1]}; r.oz = f2{o0[2], o1[2]};
        r.dx = f2{d0[0], d1[0]}; r.dy = f2{d0[1], d1[1]}; r.dz = f2{d0[2], d1[2]};
        r.ix = f2{slab_inv(d0[0]), slab_inv(d1[0])};
        r.iy = f2{slab_inv(d0[1]), slab_inv(d1[1])};
        r.iz = f2{slab_inv(d0[2]), slab_inv(d1[2])};
        r.best = f2{b0, b1};
        r.bestp = r.best * kBestInfl;
        r.idx = -1; r.idy = -1;
        const float E = tv.extent;
        r.wildx = live[2 * p] & !((fabsf(o0[0]) <= E) & (fabsf(o0[1]) <= E) & (fabsf(o0[2]) <= E) &
                                  (fabsf(d0[0]) <= 2.0f) & (fabsf(d0[1]) <= 2.0f) & (fabsf(d0[2]) <= 2.0f));
        r.wildy = live[2 * p + 1] & !((fabsf(o1[0]) <= E) & (fabsf(o1[1]) <= E) & (fabsf(o1[2]) <= E) &
                                      (fabsf(d1[0]) <= 2.0f) & (fabsf(d1[1]) <= 2.0f) & (fabsf(d1[2]) <= 2.0f));
    }

    // cooperative form: the block's merge array starts from every ray's own start value
    unsigned long long pub[R];                   // what this wave last saw or wrote per ray
    if (COOP) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            pub[2 * p] = pack_hit(w.r[p].best.x, -1);
            pub[2 * p + 1] = pack_hit(w.r[p].best.y, -1);
            if (wib == 0) {
                s_best[(2 * p) * 64 + lane] = pub[2 * p];
                s_best[(2 * p + 1) * 64 + lane] = pub[2 * p + 1];
            }
        }
        if (threadIdx.x == 0) *s_cost = 0;
        __syncthreads();
    }
    // take over what the other waves have found (COOP): min over (t, index)
    auto merge_in = [&]() {
#pragma unroll
        for (int p = 0; p < P; p++) {
            RayPair &r = w.r[p];
            unsigned long long vx = s_best[(2 * p) * 64 + lane], vy = s_best[(2 * p + 1) * 64 + lane];
            if (vx < pack_hit(r.best.x, r.idx)) {
                r.best.x = __uint_as_float((uint32_t)(vx >> 32));
                r.idx = (int32_t)(uint32_t)vx - 1;
                r.bestp.x = r.best.x < r.stop.x ? -1.0f : r.best.x * kBestInfl;
                if (r.best.x < r.stop.x) r.wildx = false;
            }
            if (vy < pack_hit(r.best.y, r.idy)) {
                r.best.y = __uint_as_float((uint32_t)(vy >> 32));
                r.idy = (int32_t)(uint32_t)vy - 1;
                r.bestp.y = r.best.y < r.stop.y ? -1.0f : r.best.y * kBestInfl;
                if (r.best.y < r.stop.y) r.wildy = false;
            }
            pub[2 * p] = vx < pub[2 * p] ? vx : pub[2 * p];
            pub[2 * p + 1] = vy < pub[2 * p + 1] ? vy : pub[2 * p + 1];
        }
    };
    // publish this wave's improvements (COOP)
    auto merge_out = [&]() {
#pragma unroll
        for (int p = 0; p < P; p++) {
            RayPair &r = w.r[p];
            unsigned long long mx = pack_hit(r.best.x, r.idx), my = pack_hit(r.best.y, r.idy);
            if (mx < pub[2 * p]) { atomicMin(&s_best[(2 * p) * 64 + lane], mx); pub[2 * p] = mx; }
            if (my < pub[2 * p + 1]) { atomicMin(&s_best[(2 * p + 1) * 64 + lane], my); pub[2 * p + 1] = my; }
        }
    };

    if (CULL) {
        int l = tv.top;
        uint32_t g = 0;
        uint32_t state = 0;                  // lane l: pending children of level l
        uint32_t lanebits = 0;               // level 0: which (triangle, ray) pairs passed
        bool expand = true;
        const bool split = COOP && tv.top > kSplitLevel;
        for (;;) {
            if (expand) {
                cost++;
                if (COOP) merge_in();
                const NodeGroup G = tv.groups[lane_get(A.lvloff, l) + g];
                uint32_t key[4], lb = 0, n_pass = 0;
#pragma unroll
                for (int j = 0; j < kTreeArity; j++) {
                    bool any = false;
                    float near = 3.0e38f;    // smallest entry parameter among this lane's passing rays
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        bool px, py;
                        float ex, ey;
                        node_test(w.r[p], G.lo[0][j], G.lo[1][j], G.lo[2][j], G.hi[0][j], G.hi[1][j],
                                  G.hi[2][j], px, py, ex, ey);
                        any |= px | py;
                        near = fminf(near, fminf(px ? ex : 3.0e38f, py ? ey : 3.0e38f));
                        lb |= ((uint32_t)px << (j * 2 * P + 2 * p)) | ((uint32_t)py << (j * 2 * P + 2 * p + 1));
                    }
                    // lo > hi (padding, or a triangle Model.cu:99 always rejects) is dead
                    // for every ray, including the unculled ones
                    unsigned long long b = __ballot(any);
                    key[j] = 0xffffffffu;
                    bool take = b && !(G.lo[0][j] > G.hi[0][j]);
                    if (take && split && l == kSplitLevel + 1) {
                        // the subtree below this child belongs to one wave of the block
                        uint32_t node = g * kTreeArity + (uint32_t)j;
                        take = ((node * 2654435761u) >> 30) == wib;
                    }
                    if (take) {
                        // visit order: entry parameter of the first passing lane (a coherent
                        // wave's rays agree on it closely enough); the child index rides in
                        // the two low mantissa bits
                        float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                            __builtin_bit_cast(int, near), __builtin_ctzll(b)));
                        key[j] = (__builtin_bit_cast(uint32_t, e) & ~3u) | (uint32_t)j;
                        n_pass++;
                    }
                }
                cswap(key[0], key[1]); cswap(key[2], key[3]);
                cswap(key[0], key[2]); cswap(key[1], key[3]);
                cswap(key[1], key[2]);
                uint32_t word = (key[0] & 3u) | (key[1] & 3u) << 2 | (key[2] & 3u) << 4 |
                                (key[3] & 3u) << 6 | n_pass << 8;
                state = lane_set(state, l, word);
                if (l == 0)
                    lanebits = lb;
                expand = false;
            }
            uint32_t word = lane_get(state, l);
            if ((word >> 8) == 0) {
                if (l == tv.top)
                    break;
                l++;
                g >>= 2;
                continue;
            }
            int j = (int)(word & 3u);
            state = lane_set(state, l, ((word & 0xffu) >> 2) | ((word >> 8) - 1u) << 8);
            uint32_t node = g * kTreeArity + (uint32_t)j;
            if (l == 0) {
                cost++;
                test_triangle<P>(A.tris, node, w, lanebits, j);
                if (COOP) merge_out();
            } else {
                l--;
                g = node;
                expand = true;
            }
        }
    } else {
        for (uint32_t node = 0; node < (uint32_t)tv.n_groups0 * kTreeArity; node++)
            test_triangle<P>(A.tris, node, w, 0xffffffffu, 0);
    }

    if (COOP) {
        if (lane == 0) atomicAdd(s_cost, cost);
        __syncthreads();
        merge_in();                              // the block's final answer
        cost = *s_cost;
        __syncthreads();                         // s_best / s_cost are free for the next chunk
        if (wib != 0)
            return;                              // one wave writes the chunk's results
    }
#ifdef RT_TRAV_STATS
    if (lane == 0 && g_log_on == (SHADOW ? 2u : 1u) && chunk < (1u << 17))
        g_log[chunk] = ChunkLog{tlog0, (unsigned long long)wall_clock64(), cost, COOP ? 1u : 0u};
#endif
    if (A.feedback && lane == 0) {
        const bool heavy = cost >= A.heavy_cost;
        A.fb.flag_w[chunk] = heavy ? 1 : 0;
        atomicAdd(A.fb.total_w, (unsigned long long)cost);
        if (heavy)
            A.fb.heavy_w[atomicAdd(A.fb.n_heavy_w, 1u)] = chunk;
    }
#pragma unroll
    for (int p = 0; p < P; p++) {
        RayPair &r = w.r[p];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t i = qbase + (2 * p + h) * 64u + lane;
            v3 o = mk3(h ? r.ox.y : r.ox.x, h ? r.oy.y : r.oy.x, h ? r.oz.y : r.oz.x);
            v3 d = mk3(h ? r.dx.y : r.dx.x, h ? r.dy.y : r.dy.x, h ? r.dz.y : r.dz.x);
            float best = h ? r.best.y : r.best.x;
            int32_t id = h ? r.idy : r.idx;
            for (int s = 0; s < A.n_spheres; s++) {       // quadrics come last (Scene.cu:31-37)
                float t;
                if (sphere_hit(A.spheres[s], o, d, best, t)) {
                    best = t;
                    id = A.n_tris + s;
                }
            }
            if (!live[2 * p + h])
                continue;
            if (SHADOW) {
                // Scene.cu:225-227: blocked = |light_rec.pos - r.o| < |r.o - lpos| - 0.01
                // (H3: `lightidx == idx` restated as never true for point lights)
                bool blocked = false;
                if (scene_ishit(best) && id >= 0) {
                    Ray sr; sr.o = o; sr.d = d;
                    v3 hp = id < A.n_tris ? ray_pos(sr, best)
                                          : hit_pos_sphere(A.spheres[id - A.n_tris], sr, best);
                    blocked = length(sub(hp, o)) < thr[2 * p + h];
                }
                A.sh_thr[i] = blocked ? 1.0f : 0.0f;
            } else {
                A.out_t[i] = best;
                A.out_id[i] = id;
            }
        }
    }
}

// Over a deep tree a chunk costs anything from a handful of node tests (rays into empty
// space) to hundreds, so waves draw chunks from a dispenser (`dynamic`); over a shallow one
// every chunk is cheap, the same-address atomic would be the bottleneck, and waves stride
// the queue instead.  With feedback (ChunkFeedback) the chunks the previous sample found
// heavy go first, one per BLOCK (cooperative form); the rest follow, one per wave.  Every
// wave leaves as soon as its next chunk lies past the end.
template <int P, bool CULL, bool SHADOW>
__global__ __launch_bounds__(kBlock) void k_intersect(TreeView tv, const TriRec *__restrict__ tris,
                                                      const SphereRec *__restrict__ spheres,
                                                      int n_tris, int n_spheres, RayQueue q,
                                                      const uint32_t *__restrict__ count_ptr,
                                                      float *__restrict__ out_t,
                                                      int32_t *__restrict__ out_id,
                                                      float *__restrict__ sh_thr, uint32_t stride,
                                                      uint32_t n_slabs,
                                                      uint32_t *__restrict__ work,
                                                      uint32_t *__restrict__ work_next, int dynamic,
                                                      ChunkFeedback fb)
{
    constexpr uint32_t R = 2 * P;
    __shared__ unsigned long long s_best[64 * R];
    __shared__ uint32_t s_cost, s_chunk;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    IsectArgs A;
    A.tv = tv; A.tris = tris; A.spheres = spheres; A.n_tris = n_tris; A.n_spheres = n_spheres;
    A.q = q; A.out_t = out_t; A.out_id = out_id; A.sh_thr = sh_thr; A.stride = stride;
    A.count = *count_ptr;
    A.chunks_per_slab = (A.count + 64u * R - 1u) / (64u * R);
    const uint32_t n_chunks = A.chunks_per_slab * (SHADOW ? n_slabs : 1u);
    A.occlusion = SHADOW && n_spheres == 0;
    A.lvloff = tv.level_off[lane & (kMaxLevels - 1)];
    A.fb = fb;
    A.feedback = dynamic && fb.flag_w != nullptr;
    A.heavy_cost = 0xffffffffu;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        work_next[0] = 0u;               // the next launch's dispensers (stream-ordered after us)
        work_next[1] = 0u;
    }
    uint32_t n_heavy = 0;
    if (A.feedback) {
        n_heavy = *fb.n_heavy_r;
        unsigned long long total = *fb.total_r;
        if (total)
            A.heavy_cost = (uint32_t)std::max<unsigned long long>(kHeavyFactor * total / std::max(n_chunks, 1u), 96ull);
    }

    // phase 1: the previous sample's heavy chunks, one per block
    if (CULL && n_heavy) {
        for (;;) {
            if (threadIdx.x == 0)
                s_chunk = atomicAdd(&work[1], 1u);
            __syncthreads();
            const uint32_t d = s_chunk;
            __syncthreads();
            if (d >= n_heavy)
                break;
            process_chunk<P, CULL, SHADOW, true>(A, fb.heavy_r[d], lane, wib, s_best, &s_cost);
        }
    }

    // phase 2: everything else, one per wave
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
        if (CULL && n_heavy && fb.flag_r[chunk])
            continue;                            // went out with the heavy ones
        process_chunk<P, CULL, SHADOW, false>(A, chunk, lane, wib, nullptr, nullptr);
    }
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
__device__ __forceinline__ void finalize_path(const DevView &v, uint32_t p, v3 res)
{
    float frac1 = 1.0f / v.cnt;
    float frac2 = frac1 * (v.cnt - 1);
    float4 prev = v.fb[p];
    float4 out;
    out.x = frac2 * prev.x + frac1 * res.x;
    out.y = frac2 * prev.y + frac1 * res.y;
    out.z = frac2 * prev.z + frac1 * res.z;
    out.w = 1.0f;
    v.fb[p] = out;
}

// One point light against one hit: PointLight::lightIllumi (Light/Light.cu:7-14)
// and Scene::evaluateDirectLight up to the shadow-ray test (Scene.cu:212-224).
struct LightSample { Ray sr; v3 contrib; float thr; bool cast; };
__device__ __forceinline__ LightSample sample_point_light(const Hit &h, const ModelRec &M,
                                                         const v3 &wo_dir, const LightRec &L)
{
    LightSample s;
    v3 lp = mk3(L.pos[0], L.pos[1], L.pos[2]), illum = mk3(L.illum[0], L.illum[1], L.illum[2]);
    Ray wi = make_ray(lp, sub(h.pos, lp));                                  // Light.cu:9
    v3 dd = sub(h.pos, lp);
    v3 color = divs(illum, fmaxf(0.001f, dot(dd, dd)));                     // Light.cu:13
    v3 lpos = wi.o;                                                         // Scene.cu:213
    s.sr = spawn_ray(h, neg(wi.d));                                         // Scene.cu:218
    v3 f = scale(rec_f(h, M, neg(wo_dir), s.sr.d), fabsf(dot(h.normal, s.sr.d)));   // Scene.cu:222
    s.cast = !is_black(f);                                                  // Scene.cu:224
    s.thr = length(sub(s.sr.o, lpos)) - 0.01f;                              // Scene.cu:227 rhs
    // Scene.cu:233-241 with pdf_light = 1 (Light.cu:12), isDelta = true
    s.contrib = is_black(color) ? mk3(0.0f, 0.0f, 0.0f) : divs(mul(color, f), 1.0f);
    return s;
}

// ------------------------------------------------------------------ shade: spawn shadow rays
// For every active path of the current bounce: draw the bounce's uniform4
// (PathTracer.cu:27), classify hit / miss (PathTracer.cu:64), and for lights
// [l0, l0 + nl) write one shadow ray per (hit, light) whose BRDF term is not black
// (Scene.cu:224) into slot j * stride + i of the light-major shadow queue (i = the
// path's queue position): a slab per light, so a chunk of the shadow pass holds
// neighbouring pixels towards ONE light.  Slots that cast nothing get thr = -1; the
// shadow pass finds them with one load and skips whole chunks of them by ballot.
__global__ __launch_bounds__(kBlock) void k_shade(DevView v, int cur, int bounce, int l0, int nl,
                                                  int n_lights_total, uint32_t stride)
{
    const uint32_t count = v.ctl->n_cur;
    const RayQueue q = v.rays[cur];
    uint32_t n_cast = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i - threadIdx.x < count;
         i += gridDim.x * kBlock) {
        bool valid = i < count;
        uint32_t p = 0;
        Ray r;
        r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 0);
        float t = kInf;
        int32_t id = -1;
        if (valid) {
            p = q.path[i];
            r.o = mk3(q.ox[i], q.oy[i], q.oz[i]);
            r.d = mk3(q.dx[i], q.dy[i], q.dz[i]);
            t = v.hit_t[i];
            id = v.hit_id[i];
        }
        bool ishit = valid && scene_ishit(t) && id >= 0;
        if (valid && l0 == 0) {
            if (bounce == 0) {                       // id / t buffers of the primary hit
                v.first_id[p] = id;
                v.first_t[p] = t;
            }
            // draws of this bounce: tmp (PathTracer.cu:27) + one per light (Scene.cu:87)
            uint32_t draw = v.rng_ctr[p];
            v.bounce_draw[p] = draw;
            v.rng_ctr[p] = draw + 1u + (ishit ? (uint32_t)n_lights_total : 0u);
            v.le[0][p] = 0.0f; v.le[1][p] = 0.0f; v.le[2][p] = 0.0f;
            if (!ishit)                              // PathTracer.cu:64 -> return res
                finalize_path(v, p, mk3(v.res[0][p], v.res[1][p], v.res[2][p]));
        }
        Hit h;
        if (ishit)
            fill_hit(v, r, id, h);
        const ModelRec *M = ishit ? &v.models[h.model] : &v.models[0];
        for (int j = 0; j < nl; j++) {
            const uint32_t e = (uint32_t)j * stride + i;
            bool cast = false;
            if (ishit) {
                LightSample s = sample_point_light(h, *M, r.d, v.lights[l0 + j]);
                cast = s.cast;
                if (cast) {
                    RayQueue sq = v.shadow;
                    sq.ox[e] = s.sr.o.x; sq.oy[e] = s.sr.o.y; sq.oz[e] = s.sr.o.z;
                    sq.dx[e] = s.sr.d.x; sq.dy[e] = s.sr.d.y; sq.dz[e] = s.sr.d.z;
                    // Scene.cu:227 rhs; a negative one (hit within 0.01 of the light) can
                    // never be undercut, and 0 keeps the slot apart from the empty ones
                    v.sh_thr[e] = fmaxf(s.thr, 0.0f);
                    v.sh_contrib[0][e] = s.contrib.x;
                    v.sh_contrib[1][e] = s.contrib.y;
                    v.sh_contrib[2][e] = s.contrib.z;
                }
            }
            if (valid && !cast)
                v.sh_thr[e] = -1.0f;
            n_cast += (uint32_t)__popcll(__ballot(cast));
        }
    }
    // shadow rays actually cast (statistics only): one fire-and-forget add per wave
    if ((threadIdx.x & 63u) == 0 && n_cast)
        atomicAdd(&v.ctl->shadow_total, (unsigned long long)n_cast);
}

// ------------------------------------------------------------------ finish the bounce
// Adds the unblocked contributions in light order (Scene.cu:120, 227-249: the shadow
// pass left 0 = visible / 1 = blocked / -1 = not cast in each slot), then, after the
// last light pass, PathTracer.cu:73-88.
__global__ __launch_bounds__(kBlock) void k_finish(DevView v, int cur, int bounce, int max_depth,
                                                   int l0, int nl, int last_pass, uint32_t stride)
{
    __shared__ uint32_t s_tot[kWavesPerBlock];
    __shared__ uint32_t s_base;
    const uint32_t count = v.ctl->n_cur;
    const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    const RayQueue q = v.rays[cur];
    const RayQueue qn = v.rays[cur ^ 1];
    for (uint32_t group = blockIdx.x; group * kBlock < count; group += gridDim.x) {
        uint32_t i = group * kBlock + threadIdx.x;
        bool valid = i < count;
        uint32_t p = 0;
        Ray r;
        r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 0);
        float t = kInf;
        int32_t id = -1;
        if (valid) {
            p = q.path[i];
            r.o = mk3(q.ox[i], q.oy[i], q.oz[i]);
            r.d = mk3(q.dx[i], q.dy[i], q.dz[i]);
            t = v.hit_t[i];
            id = v.hit_id[i];
        }
        bool ishit = valid && scene_ishit(t) && id >= 0;

        // res += evaluateDirectLight(...) per light, in order (Scene.cu:120)
        v3 le = mk3(0, 0, 0);
        if (ishit) {
            le = mk3(v.le[0][p], v.le[1][p], v.le[2][p]);
            for (int j = 0; j < nl; j++) {
                const uint32_t e = (uint32_t)j * stride + i;
                if (v.sh_thr[e] == 0.0f)                                 // cast and not blocked (Scene.cu:236)
                    le = add(le, mk3(v.sh_contrib[0][e], v.sh_contrib[1][e], v.sh_contrib[2][e]));
            }
        }

        bool survive = false;
        Ray nr;
        nr.o = mk3(0, 0, 0); nr.d = mk3(0, 0, 0);
        if (ishit && !last_pass) {
            v.le[0][p] = le.x; v.le[1][p] = le.y; v.le[2][p] = le.z;
        }
        if (ishit && last_pass) {
            v3 beta = mk3(v.beta[0][p], v.beta[1][p], v.beta[2][p]);
            v3 res = mk3(v.res[0][p], v.res[1][p], v.res[2][p]);
            res = add(res, mul(beta, le));                               // PathTracer.cu:73
            Hit h;
            fill_hit(v, r, id, h);
            const ModelRec &M = v.models[h.model];
            float u[4];
            pixel_uniform4(v.pixmap[p], v.bounce_draw[p], u);            // tmp of PathTracer.cu:27
            v3 wi;
            float pdf;
            v3 f = rec_sample_f(h, M, neg(r.d), wi, pdf, u[0], u[1]);    // PathTracer.cu:76
            bool done = (pdf < 0.0001f || length(f) < 0.0001f);          // PathTracer.cu:77
            if (!done) {
                beta = mul(beta, divs(scale(f, fabsf(dot(wi, h.normal))), pdf));   // :79
                nr = spawn_ray(h, wi);                                   // :80
                if (bounce > 3) {                                        // :82-88
                    if (u[2] < 0.5f)
                        done = true;
                    else
                        beta = divs(beta, 1.0f - 0.5f);
                }
            }
            if (!done && bounce + 1 > max_depth) {
                // Next iteration: one draw (:27), a trace whose result nothing reads
                // (Lambertian, no environment light), then the break of :64.
                v.rng_ctr[p] = v.rng_ctr[p] + 1u;
                done = true;
            }
            if (done) {
                finalize_path(v, p, res);
            } else {
                survive = true;
                v.beta[0][p] = beta.x; v.beta[1][p] = beta.y; v.beta[2][p] = beta.z;
                v.res[0][p] = res.x; v.res[1][p] = res.y; v.res[2][p] = res.z;
            }
        }
        if (!last_pass || bounce >= max_depth)
            continue;                         // uniform: nothing survives the last bounce
        // compaction of the surviving paths into the next bounce's queue
        uint64_t b = __ballot(survive);
        if (lane == 0)
            s_tot[wib] = (uint32_t)__popcll(b);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < kWavesPerBlock; w++)
                tot += s_tot[w];
            s_base = tot ? atomicAdd(&v.ctl->n_next, tot) : 0u;
        }
        __syncthreads();
        uint32_t base = s_base;
        for (uint32_t w = 0; w < wib; w++)
            base += s_tot[w];
        __syncthreads();
        if (survive) {
            uint32_t e = base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
            qn.ox[e] = nr.o.x; qn.oy[e] = nr.o.y; qn.oz[e] = nr.o.z;
            qn.dx[e] = nr.d.x; qn.dy[e] = nr.d.y; qn.dz[e] = nr.d.z;
            qn.path[e] = p;
        }
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
static inline int grid_for(uint32_t n, int max_blocks)
{
    uint32_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if ((int)g > max_blocks) g = (uint32_t)max_blocks;
    return (int)g;
}

void launch_ctl_begin(hipStream_t s, Ctl *ctl, uint32_t n) { k_ctl_begin<<<1, 1, 0, s>>>(ctl, n); }
void launch_ctl_next_bounce(hipStream_t s, Ctl *ctl) { k_ctl_next_bounce<<<1, 1, 0, s>>>(ctl); }

void launch_gen(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n)
{
    k_gen<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(v, p0, n);
}

// Rays per chunk: two packed pairs per lane measured slower than one on every config (a
// wave visits the union of what its rays need, and the longest chunks set the kernel's
// length), so the default is one pair = 128 rays; RT_ISECT_PAIRS=2 keeps the other form.
static bool intersect_wide(const DevView &v) { return v.isect_pairs == 2; }
uint32_t intersect_rays_per_chunk(const DevView &v, uint32_t) { return intersect_wide(v) ? 256u : 128u; }

// One launch of the closest-hit kernel over `capacity` queue slots at most.  Shadow form:
// sh_thr != nullptr, the queue is n_slabs slabs of `stride` slots (see process_chunk).
// work / work_next: this launch's and the next launch's pair of dispenser counters.
void launch_intersect(hipStream_t s, const DevView &v, const RayQueue &q, const uint32_t *count,
                      uint32_t capacity, float *out_t, int32_t *out_id, float *sh_thr,
                      uint32_t stride, uint32_t n_slabs, int max_blocks, uint32_t *work,
                      uint32_t *work_next, const ChunkFeedback *fb)
{
    const bool wide = intersect_wide(v);
    const int dynamic = v.cull && v.tree.top >= 3;       // >= 256 triangles
    const uint32_t rays_per_block = kBlock * (wide ? 4u : 2u);
    unsigned long long gl = ((unsigned long long)capacity + rays_per_block - 1) / rays_per_block;
    if (sh_thr) gl *= n_slabs;
    uint32_t g = (uint32_t)std::min<unsigned long long>(std::max<unsigned long long>(gl, 1ull),
                                                        (unsigned long long)max_blocks);
    ChunkFeedback f{};
    if (fb && dynamic) f = *fb;
#define RT_LAUNCH_INTERSECT(P, CULL, SH)                                                        \
    k_intersect<P, CULL, SH><<<g, kBlock, 0, s>>>(v.tree, v.tris, v.spheres, v.n_tris,          \
                                                  v.n_spheres, q, count, out_t, out_id, sh_thr, \
                                                  stride, n_slabs, work, work_next, dynamic, f)
    if (sh_thr) {
        if (v.cull) { if (wide) RT_LAUNCH_INTERSECT(2, true, true); else RT_LAUNCH_INTERSECT(1, true, true); }
        else        { if (wide) RT_LAUNCH_INTERSECT(2, false, true); else RT_LAUNCH_INTERSECT(1, false, true); }
    } else {
        if (v.cull) { if (wide) RT_LAUNCH_INTERSECT(2, true, false); else RT_LAUNCH_INTERSECT(1, true, false); }
        else        { if (wide) RT_LAUNCH_INTERSECT(2, false, false); else RT_LAUNCH_INTERSECT(1, false, false); }
    }
#undef RT_LAUNCH_INTERSECT
}

void launch_shade(hipStream_t s, const DevView &v, uint32_t capacity, int cur, int bounce, int l0,
                  int nl, int n_lights_total, uint32_t stride, int max_blocks)
{
    k_shade<<<grid_for(capacity, max_blocks), kBlock, 0, s>>>(v, cur, bounce, l0, nl,
                                                              n_lights_total, stride);
}

void launch_finish(hipStream_t s, const DevView &v, uint32_t capacity, int cur, int bounce,
                   int max_depth, int l0, int nl, int last_pass, uint32_t stride, int max_blocks)
{
    k_finish<<<grid_for(capacity, max_blocks), kBlock, 0, s>>>(v, cur, bounce, max_depth, l0, nl,
                                                               last_pass, stride);
}

void launch_scatter_image(hipStream_t s, const uint32_t *pixmap, uint32_t n, const float4 *fb,
                          const int32_t *first_id, const float *first_t, float4 *rgba,
                          int32_t *prim, float *t)
{
    if (n)
        k_scatter_image<<<(n + 255) / 256, 256, 0, s>>>(pixmap, n, fb, first_id, first_t, rgba,
                                                        prim, t);
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
