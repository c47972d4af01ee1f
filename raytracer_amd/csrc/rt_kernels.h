// rt_kernels.h -- device-side views and kernel launchers (internal to the library).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_math.h"

namespace rt {

constexpr int kWorkSlots = 24;

// Per-lane control block, resident in device memory.
struct Ctl {
    // Chunk dispensers: [0..7] waves draw ordinary chunks, one counter per XCD (each over its
    // own eighth of the range, the others' when that is used up); [8] blocks draw the heavy
    // chunks; [9] waves draw the early ones.  Launch k uses set k & 1 and zeroes the other
    // for launch k + 1.
    uint32_t work[2][kWorkSlots];
    unsigned long long shadow_total;   // shadow rays cast since rt_reset_stats
    unsigned long long bounce_total;   // bounce (>= 1) rays traced since rt_reset_stats
    unsigned long long coop_total;     // chunks run in the cooperative (four-wave) form since rt_reset_stats
};

constexpr int kWorkHeavy = 8, kWorkEarly = 9, kWorkLate = 12;   // kWorkLate .. kWorkLate + 7: the late units' slices

// Struct-of-arrays ray queue of rt_scene_hit: lane i of a wave reads element base + i.
struct RayQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz;
};

// Implicit 4-ary box tree over the triangle records (rt_math.h NodeGroup; built by
// rt_upload.cpp).  The children of node i of level l+1 are the four boxes of group i
// of level l: no pointers, no stack.
struct TreeView {
    const NodeGroup *groups;          // all levels back to back
    uint32_t level_off[kMaxLevels];   // first group of level l in `groups`
    int top;               // the top level is one group
    int n_groups0;         // number of 4-triangle groups of level 0
    float extent;          // rays whose origin leaves [-extent, extent]^3 are never culled
};

// Chunk-order feedback of the path kernel (deep trees only).  A chunk's cost is stable
// from one progressive sample to the next (same pixels), so each launch records which
// chunks were costly and the next launch hands those out FIRST: the kernel then ends with
// cheap chunks instead of waiting for a straggler that was dispensed last.  Classes by
// cost relative to the mean chunk cost of the previous launch:
//   heavy (>= 8x): one per BLOCK, cooperative form (four waves share the traversals)
//   early (>= 4x, then >= 2x): one per wave, before everything else
// Three sets rotate by launch number: `r` was written by the previous launch, `w` is
// written now, `z` is zeroed now for the next launch to write.  Only an ordering hint --
// every chunk is processed exactly once.
struct FbCounts {
    uint32_t n_heavy, n_early_a, n_early_b, pad;
    unsigned long long total;            // summed cost of all chunks
    unsigned long long band[8];          // k_batch: summed cost of the units each XCD's band handed out in band order
};
// What k_classify makes of the previous launch's flags, right before a k_batch launch: per band the ordinary units
// (neither listed, early nor late) as a list, and the late units as one list.
struct BatchLists {
    uint32_t n_norm[8];
    uint32_t n_late, pad[7];
};
struct ChunkFeedback {
    const FbCounts *cnt_r;
    const uint32_t *heavy_r;             // chunk ids that were heavy
    const uint32_t *early_r;             // early chunk ids: class a from the front, b from the back
    const uint8_t *flag_r;               // per chunk: 1 = in heavy_r, 2 = in early_r
    FbCounts *cnt_w;
    uint32_t *heavy_w;
    uint32_t *early_w;
    uint8_t *flag_w;
    FbCounts *cnt_z;
    uint32_t n_slots;                    // entries of the heavy / early arrays (= chunks of the range)
};

// Unit-order feedback of k_batch.  A unit's cost (node groups + triangle records of all its
// traversals) is stable from one batch of samples to the next -- same pixels -- and a few units on
// silhouettes cost fifty times the mean: a launch cannot end before its longest unit does.  So every
// launch lists the units that cost >= heavy_cost, and the next launch of the same shape starts with
// them, split by pixel: each pixel of a listed unit is a task of its own (a quarter of the lanes,
// but rays through one pixel walk far less of the tree together than rays along a silhouette), so
// four waves share what one wave would have carried.  Sets rotate by launch (r: written by the
// previous launch, w: written now, z: zeroed by k_classify for the launch after).  Only an ordering
// hint: every pixel-sample is processed exactly once whatever the sets hold.
// Costly units below that threshold ("early": >= early_x4 / 4 times the mean unit cost) are handed out whole, right
// after the listed pixels and before the bands, costliest class first: a launch then ends with cheap units, not with
// a wave that drew a unit twenty times the mean as its last (longest-processing-time-first).  And the cheapest
// units (<= late_x16 / 16 of the mean: camera rays that miss everything) are kept back for the very end, after the
// bands: they touch nothing but the top of the tree, so they lose nothing by leaving band order, and they are the
// fine-grained work that fills the last stretch, when every wave would otherwise finish a mid-sized unit at its own time.
struct BatchFeedback {
    const FbCounts *cnt_r;               // n_heavy, n_early_a / _b, pad = 1 + ss_log2 of the launch that wrote the set, total
    const uint32_t *heavy_r;             // unit ids: handed out first, one pixel per task
    const uint32_t *early_r;             // unit ids: handed out next, whole -- class a (costliest) from the front, b from the back
    const uint32_t *flag_r;              // per unit: 1 = listed in heavy_r, 2 = in early_r, 3 = kept for last (no list: k_classify makes it)
    FbCounts *cnt_w;
    uint32_t *heavy_w;
    uint32_t *early_w;
    uint32_t *flag_w;
    uint32_t n_slots;                    // entries of the arrays
    // made by k_classify for THIS launch (launch_batch runs it first, on the same stream)
    const BatchLists *lists;
    const uint32_t *norm;                // band y's ordinary units: norm[y * band_units ..], lists->n_norm[y] of them
    const uint32_t *late;                // lists->n_late units, handed out when every band is used up
    uint32_t *norm_w, *late_w;           // (the same arrays, for k_classify)
    BatchLists *lists_w;
    BatchLists *lists_next;              // the other of the lane's two: zeroed by k_batch for the next launch's k_classify
    FbCounts *cnt_z;                     // the set after `w`: zeroed by k_classify for the launch that will write it
    uint32_t *flag_z;                    // (nullptr: no feedback)
    uint32_t flags_valid;                // 1: flag_r was written by a launch of this shape; 0: every unit is ordinary
    uint32_t settled;                    // 1: ... and that launch had thresholds of its own to flag by (the third launch of a shape on)
};

// Bounce queue (rt_render with max_depth > 0 over a deep tree).  A sample is two launches: k_path's EMIT build
// carries every pixel through its camera ray, direct light and BRDF sample -- coherent rays, wave-wide
// traversal -- and appends a job per path that goes on; k_bounce's persistent waves then take jobs lane by lane,
// walk the tree per lane, and a lane whose path has ended takes the next job at once.
struct BounceJob {                   // 64 bytes
    float o[3], d[3];                // the ray leaving the first hit (PathTracer.cu:80)
    float beta[3], res[3];           // throughput and radiance so far (PathTracer.cu:73,79)
    uint32_t draw;                   // next Philox draw of the pixel's stream
    uint32_t p;                      // local pixel
    uint32_t pad[2];
};
struct BounceQueue {
    BounceJob *jobs;                 // nullptr: no queue (the one-kernel path)
    uint32_t *ctr;                   // [0] jobs appended (k_path), [1] jobs handed out (k_bounce)
    uint32_t *ctr_next;              // the other pair: zeroed by k_path for the next sample
    uint32_t cap;                    // entries of jobs (>= pixels of the range)
    int pairs;                       // k_bounce: a lane carries two paths (else one: half the walk per traversal)
};

struct DevView {
    int width, height;
    int cnt;                        // 1-based sample index (*cnt of kernel.cu:52-66)
    CameraRec cam;
    // scene (read-only)
    const TriRec *tris;             // tree order
    const uint32_t *tri_pos_of;     // reference traversal index -> position in tris
    const TriCold *tri_cold;        // reference traversal order
    TreeView tree;
    int cull;                       // 0: scan every triangle (validation), 1: use the tree
    int sorted_planes;              // 1: waves whose rays share an octant read the boxes' planes pre-ordered
    const ModelRec *models;
    const MatRec *mats;             // per ModelRec: the BRDF's parameters (general kernel only)
    int general;                    // 1: non-Lambertian materials or triangle lights -> k_path_general
    const SphereRec *spheres;
    const LightRec *lights;         // delta lights (point, then directional)
    const TriLightRec *tlights;     // triangle lights: hit-able (Scene.cu:43-64), sampled with MIS (Scene.cu:242-284)
    EnvView env;                    // the environment light (env.n = 0: none)
    TexView tex;                    // mesh textures (tex.n = 0: none)
    int n_tris, n_spheres, n_lights, n_tlights;
    // persistent per local pixel
    const uint32_t *pixmap;         // local index -> global pixel index y*W + x
    uint32_t *rng_ctr;              // next Philox draw of the pixel's stream
    uint32_t *last_draws;           // draws the pixel's latest depth-0 batch sample consumed (0: none yet); k_batch predicts from it
    float4 *fb;                     // RGBA32F running mean
    int32_t *first_id;
    float *first_t;
};

// Diagnostic knobs of the launchers (A/B measurement only: results never depend on them).  Read from the
// environment ONCE, by rt_create, and kept in the context: a live context does not change behaviour between
// launches, and no launch calls getenv.
struct Knobs {
    int lane_traversal = 1;      // RT_LANE_TRAVERSAL=0: bounce rays through the wave-wide traversal
    int bounce_waves = 3;        // RT_BOUNCE_WAVES=2: the two-wave build of the bounce kernel
    int bounce_pairs = 0;        // RT_BOUNCE_PAIRS=1: k_bounce's lanes carry two paths each
    int bounce_queue = 0;        // RT_BOUNCE_QUEUE=1: bounces over a deep tree as k_path<.., EMIT> + k_bounce (bit-identical; measured slower
                                 // than k_path carrying every path to its end: C3 depth 12 3.5 against 2.9 ms per sample)
    int heavy_x4 = 0;            // RT_HEAVY_X4: cooperative form from this many quarters of the mean chunk cost on (0: chosen per launch)
    int general_regen = -1;      // RT_GENERAL_REGEN: 0 / 1 force path regeneration of the general kernel off / on
    int general_waves = 0;       // RT_GENERAL_WAVES: 1 / 2 force the build of the general kernel
    int units_per_block = 16;    // RT_UNITS_PER_BLOCK: grid sizing of k_batch (96 left small launches on a fraction of the chip: an eighth of
                                 // C3, 8 samples 0.96 -> 0.75 ms, 1 sample 1.2 -> 0.43; 4 is no better; large launches fill the chip either way)
    int batch_waves = 0;         // RT_BATCH_WAVES: 4 / 5 force the build of k_batch
    int intersect_lane = 2;      // RT_INTERSECT_LANE: 0 wave-wide, 1 per lane, 2 per chunk (rt_scene_hit)
    int task_prio = 1;           // RT_TASK_PRIO=0: listed pixel tasks of k_batch at ordinary wave priority
    int early_x4 = 20;           // RT_EARLY_X4: k_batch hands units of >= this many quarters of the mean unit cost out early (0: off;
                                 // 8 measured 13 % slower on C3 -- too many units out of band order --, 16..48 alike)
    int draw_early = 0;          // RT_DRAW_EARLY=1: the next unit's ticket is drawn at the unit's start while its dispenser is far from
                                 // its end (measured: no difference, C3 13.67 / C4 55.2 ms either way, so the simpler form is the default)
    int grab = 0;                // RT_GRAB: units a wave takes per ticket of a band's dispenser (0: by the launch's units per wave)
    int grab_late = 4;           // RT_GRAB_LATE: ... of the late units' dispenser
    int late_x16 = 4;            // RT_LATE_X16: k_batch keeps units of <= this many sixteenths of the mean unit cost for last (0: off)
    int band_help = 1;           // RT_BAND_HELP=0: an XCD that has used its band up helps the others' in fixed cyclic order
};
Knobs read_knobs();

// Progressive sample(s) of local pixels [p0, p0 + n): the whole path of every pixel
// (kernel.cu:50-79 -> pathTracer) in ONE launch.  work / work_next: this launch's and the
// next launch's pair of dispenser counters; fb may be null (no feedback).
// n_samples > 1: every chunk renders that many consecutive samples (v.cnt, v.cnt + 1, ..) in this one
// launch -- only where path_renders_runs(v) says so (shallow trees or the scan, no general scene);
// elsewhere the caller launches once per sample.
// -> true if the launch took the bounce queue (its counter pair is then used up: the caller alternates)
bool launch_path(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, int max_depth,
                 int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                 const ChunkFeedback *fb, uint32_t n_samples, const Knobs &knobs, const BounceQueue *bq = nullptr);
bool path_renders_runs(const DevView &v);
constexpr uint32_t kPathChunk = 128;   // pixels per chunk of the path kernel (two per lane)

// Depth 0 over a deep tree, one batch of n_samples samples of local pixels [p0, p0 + n) in ONE
// kernel: camera rays from predicted counters, the pixels the prediction fails on, shadow rays
// and the running mean without leaving the wave (k_batch).
// fb is never null: its lists / norm / late arrays are filled by k_classify inside the launch (norm: 8 bands of
// batch_band_units(units) entries, late: `units` entries); fb->cnt_w == nullptr switches the feedback off.
uint32_t batch_band_units(uint32_t units);
void launch_batch(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, uint32_t n_samples,
                  int cnt0, int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                  const BatchFeedback *fb, const Knobs &knobs);

// Scene::hit on a queue of caller rays (rt_scene_hit).
void launch_intersect(hipStream_t s, const DevView &v, const RayQueue &q, uint32_t count,
                      float *out_t, int32_t *out_id, int max_blocks, uint32_t *work,
                      uint32_t *work_next, const Knobs &knobs);
void launch_scatter_image(hipStream_t s, const uint32_t *pixmap, uint32_t n, const float4 *fb,
                          const int32_t *first_id, const float *first_t, float4 *rgba,
                          int32_t *prim, float *t);
void launch_fill_i32(hipStream_t s, int32_t *a, size_t n, int32_t value);
// zero framebuffer, RNG counters, draw counts, t buffer; id buffer = -1 (rt_reset) for local pixels [0, n)
void launch_reset_pixels(hipStream_t s, const DevView &v, uint32_t n);
// d_mipmap (Texture.cu:14-45): level dst (dw x dh) of a mip chain from the level above it (sw x sh)
void launch_mip_level(hipStream_t s, const float4 *src, int sw, int sh, float4 *dst, int dw, int dh);
void launch_prepare_rays(hipStream_t s, const RayQueue &q, const float *org, const float *dir,
                         uint32_t n);
void launch_ishit(hipStream_t s, const float *t, int32_t *ishit, uint32_t n);
void launch_tonemap(hipStream_t s, const float4 *in, float4 *out, uint32_t n);
void launch_median3x3(hipStream_t s, const float4 *in, float4 *out, int width, int height);

} // namespace rt
