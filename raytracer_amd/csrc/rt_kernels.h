// rt_kernels.h -- device-side views and kernel launchers (internal to the library).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_math.h"

namespace rt {

constexpr int kWorkSlots = 12;

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

constexpr int kWorkHeavy = 8, kWorkEarly = 9;

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
// previous launch, w: written now, z: zeroed by the host for the launch after).  Only an ordering
// hint: every pixel-sample is processed exactly once whatever the sets hold.
struct BatchFeedback {
    const FbCounts *cnt_r;               // n_heavy, pad = 1 + ss_log2 of the launch that wrote the set, total
    const uint32_t *heavy_r;             // unit ids
    const uint32_t *flag_r;              // per unit: != 0 = listed in heavy_r
    FbCounts *cnt_w;
    uint32_t *heavy_w;
    uint32_t *flag_w;
    uint32_t n_slots;                    // entries of the arrays
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

// Progressive sample(s) of local pixels [p0, p0 + n): the whole path of every pixel
// (kernel.cu:50-79 -> pathTracer) in ONE launch.  work / work_next: this launch's and the
// next launch's pair of dispenser counters; fb may be null (no feedback).
// n_samples > 1: every chunk renders that many consecutive samples (v.cnt, v.cnt + 1, ..) in this one
// launch -- only where path_renders_runs(v) says so (shallow trees or the scan, no general scene);
// elsewhere the caller launches once per sample.
void launch_path(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, int max_depth,
                 int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                 const ChunkFeedback *fb, uint32_t n_samples);
bool path_renders_runs(const DevView &v);
constexpr uint32_t kPathChunk = 128;   // pixels per chunk of the path kernel (two per lane)

// Depth 0 over a deep tree, one batch of n_samples samples of local pixels [p0, p0 + n) in ONE
// kernel: camera rays from predicted counters, the pixels the prediction fails on, shadow rays
// and the running mean without leaving the wave (k_batch).
void launch_batch(hipStream_t s, const DevView &v, uint32_t p0, uint32_t n, uint32_t n_samples,
                  int cnt0, int max_blocks, Ctl *ctl, uint32_t *work, uint32_t *work_next,
                  const BatchFeedback *fb);

// Scene::hit on a queue of caller rays (rt_scene_hit).
void launch_intersect(hipStream_t s, const DevView &v, const RayQueue &q, uint32_t count,
                      float *out_t, int32_t *out_id, int max_blocks, uint32_t *work,
                      uint32_t *work_next);
void launch_scatter_image(hipStream_t s, const uint32_t *pixmap, uint32_t n, const float4 *fb,
                          const int32_t *first_id, const float *first_t, float4 *rgba,
                          int32_t *prim, float *t);
void launch_fill_i32(hipStream_t s, int32_t *a, size_t n, int32_t value);
// d_mipmap (Texture.cu:14-45): level dst (dw x dh) of a mip chain from the level above it (sw x sh)
void launch_mip_level(hipStream_t s, const float4 *src, int sw, int sh, float4 *dst, int dw, int dh);
void launch_prepare_rays(hipStream_t s, const RayQueue &q, const float *org, const float *dir,
                         uint32_t n);
void launch_ishit(hipStream_t s, const float *t, int32_t *ishit, uint32_t n);
void launch_tonemap(hipStream_t s, const float4 *in, float4 *out, uint32_t n);
void launch_median3x3(hipStream_t s, const float4 *in, float4 *out, int width, int height);

} // namespace rt
