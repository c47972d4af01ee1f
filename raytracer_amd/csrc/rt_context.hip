// rt_context.hip -- implementation of the C ABI (include/rt_abi.h): context, device
// memory, scene/camera upload, the per-sample launch sequence, readback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_kernels.h"
#include "rt_upload.h"

using namespace rt;

namespace {

constexpr uint32_t kWholeBatchPixels = 160000; // from here on k_batch takes the partition in one launch (rt_render)
constexpr uint32_t kMinLanePixels = 1u << 16;   // below this a second lane is not worth its launches

thread_local std::string g_err;              // for calls that have no context yet

// ---- scratch of rt_scene_hit_noalloc (the allocation-free Scene::hit, see there) ----
// The struct-of-arrays queue of normalised rays (what rt_scene_hit allocates per call) and the
// closest-hit kernel's chunk dispensers, in a static device array: every GPU the library runs
// on holds its own copy, at its own address, until the process ends.  In host code `scratch`
// is only a handle for hipGetSymbolAddress: rt_create looks the device address up once per
// device (with that device current) and everything below uses that pointer.
constexpr size_t kNoAllocMaxRays = RT_SCENE_HIT_NOALLOC_MAX_RAYS;
constexpr size_t kScratchQueueBytes = kNoAllocMaxRays * 6 * sizeof(float);
constexpr size_t kScratchCtlBytes = 256;     // two sets of kWorkSlots dispenser counters
static_assert(2 * kWorkSlots * sizeof(uint32_t) <= kScratchCtlBytes, "dispensers do not fit");
static __device__ __align__(256) unsigned char scratch[kScratchQueueBytes + kScratchCtlBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices];   // per device: device address of `scratch`, nullptr = not looked up yet

// With `device` current: its address of `scratch` (looked up on first use).
hipError_t scratch_address(int device, unsigned char **out)
{
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess)
            return e;
        g_scratch_ptr[device] = (unsigned char *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

} // namespace

// A lane renders a fixed, contiguous range of this context's pixels on its own HIP stream
// with its own dispenser counters.  Lanes never touch the same pixel, so they need no
// ordering among themselves; their kernels overlap, which is what fills the SIMDs a
// traversal kernel leaves idle while its few heaviest chunks finish.
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint32_t lo = 0, hi = 0;             // local pixel range [lo, hi)
    Ctl *ctl = nullptr;
    uint32_t seq = 0;                    // launches that used this lane's dispensers so far
    // chunk-order feedback (rt_kernels.h ChunkFeedback): three sets rotating by launch
    uint32_t *heavy[3] = {nullptr, nullptr, nullptr};
    uint32_t *early[3] = {nullptr, nullptr, nullptr};
    uint8_t *flag[3] = {nullptr, nullptr, nullptr};
    FbCounts *cnt = nullptr;             // [3]
    uint32_t fb_slots = 0;
    uint32_t fb_seq = 0;                 // path launches so far (set rotation)
    BounceJob *bjobs = nullptr;          // bounce queue (rt_kernels.h BounceQueue): one entry per pixel of the range
    uint32_t *bq_ctr = nullptr;          // two pairs of counters, alternating by launch
    uint32_t bq_cap = 0, bq_seq = 0;
    // unit-order feedback of k_batch (rt_kernels.h BatchFeedback): three sets rotating by launch
    uint32_t *bheavy[3] = {nullptr, nullptr, nullptr};
    uint32_t *bearly[3] = {nullptr, nullptr, nullptr};
    uint32_t *bnorm = nullptr, *blate = nullptr;   // k_classify's lists for the launch at hand (8 x band_units, units)
    BatchLists *blists = nullptr;                  // two: a launch's k_batch zeroes the other for the next launch's k_classify
    uint32_t bl_seq = 0;
    uint32_t bfb_shape[3] = {0, 0, 0};              // 1 + ss_log2 of the launch that wrote each set (0: none)
    uint32_t bfb_run = 0;                           // launches of one shape in a row so far
    uint32_t *bflag[3] = {nullptr, nullptr, nullptr};
    FbCounts *bcnt = nullptr;            // [3]
    uint32_t bfb_slots = 0;
    uint32_t bfb_seq = 0;                // k_batch launches so far
};

struct rt_context {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocs;          // everything freed by rt_destroy
    std::vector<void *> scene_allocs;    // freed on re-upload
    std::vector<void *> part_allocs;     // freed on re-partition

    int rank = 0, world = 1, tile = 64;
    uint32_t n_local = 0;
    std::vector<uint32_t> h_pixmap;
    std::vector<Lane> lanes;             // streams/events live as long as the context
    int n_lanes = 1;                     // lanes in use for the current partition
    int lanes_wanted = 2;                // 2 measured best for k_path (C2); more only add launches
    bool lanes_forced = false;           // RT_LANES given: k_batch keeps to the lanes too
    int lanes_used = 1;                  // by the last rt_render

    bool have_scene = false, have_cam = false;
    int n_lights = 0;
    uint64_t n_prims = 0;
    float geom_extent = 1.0f;            // max |coordinate| of the uploaded geometry
    int max_blocks = 256;                // blocks per wave-per-SIMD of occupancy (= CUs)

    DevView v{};                         // scene + per-pixel state
    int cnt = 0;                         // samples accumulated (kernel.cu:52-54)

    // depth 0 over a deep tree (k_batch, rt_kernels.hip): samples per batch
    int batch_wanted = 0;                // RT_BATCH: samples per k_batch launch (0: chosen by the call's sample count)
    uint64_t batch_launches = 0;         // launches of k_batch
    int batch_min_top = 3;               // k_batch from this tree height on (RT_BATCH_MIN_TOP)
    int path_run_wanted = 0;             // samples per k_path launch over shallow trees (RT_PATH_RUN; 0: 4)
    bool feedback_debug = false;
    bool batch_feedback = true;          // costly units of one launch go out first, by pixel, in the next (RT_FEEDBACK=0: off)
    Knobs knobs;                         // the launchers' diagnostic knobs, read once by rt_create
    bool force_general = false;          // RT_FORCE_GENERAL=1: every scene through the general kernel (diagnostics)

    // stats
    uint64_t samples = 0, primary = 0, launches = 0, timed_launches = 0;
    double path_ms = 0.0, render_ms = 0.0;
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;

    unsigned char *scratch = nullptr;    // this device's address of the static scratch array (rt_scene_hit_noalloc)

    // readback staging (full image)
    float4 *img_rgba = nullptr, *img_tmp = nullptr;
    int32_t *img_prim = nullptr;
    float *img_t = nullptr;
};

namespace {

int fail(rt_context *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    g_err = msg;
    return code;
}

#define RT_HIP(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), RT_ERR_DEVICE,                                                    \
                        std::string(#call) + ": " + hipGetErrorString(e_));                    \
    } while (0)

template <class T>
int dalloc(rt_context *c, std::vector<void *> &owner, T *&p, size_t n)
{
    p = nullptr;
    if (n == 0) n = 1;
    void *q = nullptr;
    RT_HIP(c, hipMalloc(&q, n * sizeof(T)));
    owner.push_back(q);
    p = (T *)q;
    return RT_OK;
}

void free_all(std::vector<void *> &owner)
{
    for (void *p : owner) (void)hipFree(p);
    owner.clear();
}

int alloc_queue(rt_context *c, std::vector<void *> &owner, RayQueue &q, size_t n)
{
    int rc;
    if ((rc = dalloc(c, owner, q.ox, n))) return rc;
    if ((rc = dalloc(c, owner, q.oy, n))) return rc;
    if ((rc = dalloc(c, owner, q.oz, n))) return rc;
    if ((rc = dalloc(c, owner, q.dx, n))) return rc;
    if ((rc = dalloc(c, owner, q.dy, n))) return rc;
    if ((rc = dalloc(c, owner, q.dz, n))) return rc;
    return RT_OK;
}

// Pixel-tile partition: tiles row-major over the image,
// tile_id % world == rank; inside a tile, 8x8 pixel blocks so that one wave
// (64 consecutive local indices) covers a compact square of the image.
void partition_pixels(int W, int H, int rank, int world, int T, std::vector<uint32_t> &out)
{
    const int ntx = (W + T - 1) / T, nty = (H + T - 1) / T;
    out.clear();
    for (int ty = 0; ty < nty; ty++)
        for (int tx = 0; tx < ntx; tx++) {
            int tile_id = ty * ntx + tx;
            if (tile_id % world != rank)
                continue;
            for (int by = 0; by < T; by += 8)
                for (int bx = 0; bx < T; bx += 8)
                    for (int iy = 0; iy < 8; iy++)
                        for (int ix = 0; ix < 8; ix++) {
                            int x = tx * T + bx + ix, y = ty * T + by + iy;
                            if (bx + ix < T && by + iy < T && x < W && y < H)
                                out.push_back((uint32_t)(y * W + x));
                        }
        }
}

void build_pixmap(rt_context *c)
{
    partition_pixels(c->width, c->height, c->rank, c->world, c->tile, c->h_pixmap);
    c->n_local = (uint32_t)c->h_pixmap.size();
}

// Feedback sets of a lane: one entry per 128-pixel chunk of its range.
int alloc_feedback(rt_context *c, Lane &L)
{
    const size_t chunks = ((size_t)(L.hi - L.lo) + kPathChunk - 1) / kPathChunk;
    int rc;
    for (int s = 0; s < 3; s++) {
        if ((rc = dalloc(c, c->part_allocs, L.heavy[s], chunks))) return rc;
        if ((rc = dalloc(c, c->part_allocs, L.early[s], chunks))) return rc;
        if ((rc = dalloc(c, c->part_allocs, L.flag[s], chunks))) return rc;
        RT_HIP(c, hipMemset(L.flag[s], 0, std::max<size_t>(chunks, 1)));
    }
    if ((rc = dalloc(c, c->part_allocs, L.cnt, 3))) return rc;
    RT_HIP(c, hipMemset(L.cnt, 0, 3 * sizeof(FbCounts)));
    L.fb_slots = (uint32_t)std::max<size_t>(chunks, 1);
    L.fb_seq = 0;
    if ((rc = dalloc(c, c->part_allocs, L.bjobs, (size_t)(L.hi - L.lo) + 1))) return rc;
    if ((rc = dalloc(c, c->part_allocs, L.bq_ctr, 4))) return rc;
    RT_HIP(c, hipMemset(L.bq_ctr, 0, 4 * sizeof(uint32_t)));
    L.bq_cap = L.hi - L.lo;
    L.bq_seq = 0;
    // k_batch: one entry per unit of the smallest shape (2 pixels x 64 samples)
    // (lane 0 may be given the whole partition: rt_render)
    const size_t units = ((size_t)(&L == &c->lanes[0] ? c->n_local : L.hi - L.lo) + 1) / 2 + 1;
    for (int s = 0; s < 3; s++) {
        if ((rc = dalloc(c, c->part_allocs, L.bheavy[s], units))) return rc;
        if ((rc = dalloc(c, c->part_allocs, L.bearly[s], units))) return rc;
        if ((rc = dalloc(c, c->part_allocs, L.bflag[s], units))) return rc;
        RT_HIP(c, hipMemset(L.bflag[s], 0, units * sizeof(uint32_t)));
    }
    if ((rc = dalloc(c, c->part_allocs, L.bcnt, 3))) return rc;
    RT_HIP(c, hipMemset(L.bcnt, 0, 3 * sizeof(FbCounts)));
    // (units of the smallest shape bound every shape's count; a band is rounded up to 64 units)
    if ((rc = dalloc(c, c->part_allocs, L.bnorm, units + 8 * 64))) return rc;
    if ((rc = dalloc(c, c->part_allocs, L.blate, units))) return rc;
    if ((rc = dalloc(c, c->part_allocs, L.blists, 2))) return rc;
    RT_HIP(c, hipMemset(L.blists, 0, 2 * sizeof(BatchLists)));
    L.bl_seq = 0;
    L.bfb_shape[0] = L.bfb_shape[1] = L.bfb_shape[2] = 0;
    L.bfb_run = 0;
    L.bfb_slots = (uint32_t)units;
    L.bfb_seq = 0;
    return RT_OK;
}

int setup_partition(rt_context *c)
{
    free_all(c->part_allocs);
    build_pixmap(c);
    const size_t n = c->n_local;
    DevView &v = c->v;
    int rc;
    uint32_t *pixmap = nullptr;
    if ((rc = dalloc(c, c->part_allocs, pixmap, n))) return rc;
    if (n)
        RT_HIP(c, hipMemcpy(pixmap, c->h_pixmap.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    v.pixmap = pixmap;
    if ((rc = dalloc(c, c->part_allocs, v.rng_ctr, n))) return rc;
    if ((rc = dalloc(c, c->part_allocs, v.last_draws, n))) return rc;
    if ((rc = dalloc(c, c->part_allocs, v.fb, n))) return rc;
    if ((rc = dalloc(c, c->part_allocs, v.first_id, n))) return rc;
    if ((rc = dalloc(c, c->part_allocs, v.first_t, n))) return rc;

    // lanes: contiguous pixel ranges cut at 64x64-tile boundaries (4096 local indices)
    int K = (int)std::min<size_t>((size_t)std::max(1, c->lanes_wanted), std::max<size_t>(1, n / kMinLanePixels));
    K = std::min<int>(K, (int)c->lanes.size());
    c->n_lanes = K;
    for (int k = 0; k < K; k++) {
        Lane &L = c->lanes[k];
        auto cut = [&](int i) -> uint32_t {
            if (i <= 0) return 0u;
            if (i >= K) return (uint32_t)n;
            return (uint32_t)((n * (size_t)i / (size_t)K) / 4096u * 4096u);
        };
        L.lo = cut(k);
        L.hi = cut(k + 1);
        L.seq = 0;
        if ((rc = dalloc(c, c->part_allocs, L.ctl, 1))) return rc;
        RT_HIP(c, hipMemset(L.ctl, 0, sizeof(Ctl)));
        if ((rc = alloc_feedback(c, L))) return rc;
    }
    return RT_OK;
}

int do_reset(rt_context *c)
{
    const size_t n = c->n_local;
    DevView &v = c->v;
    // one launch, stream-ordered before whatever the context's stream does next (rt_render forks its lanes off
    // this stream): a fresh frame costs no host round trip
    launch_reset_pixels(c->stream, v, (uint32_t)n);
    RT_HIP(c, hipGetLastError());
    c->cnt = 0;
    return RT_OK;
}

hipEvent_t next_event(rt_context *c)
{
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess)
            return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

// Sum the recorded (start, stop) pairs; the stream must be idle.
void resolve_events(rt_context *c)
{
    for (size_t k = 0; k + 1 < c->ev_used; k += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_pool[k], c->ev_pool[k + 1]) == hipSuccess) {
            c->path_ms += ms;
            c->timed_launches++;
        }
    }
    c->ev_used = 0;
}

// Feedback sets rotate by launch: read what the previous launch wrote, write the next set,
// zero the one after (nobody touches it during this launch).
ChunkFeedback next_feedback(Lane &L)
{
    ChunkFeedback f{};
    const int r = (int)((L.fb_seq + 2u) % 3u), w = (int)(L.fb_seq % 3u), z = (int)((L.fb_seq + 1u) % 3u);
    f.cnt_r = L.cnt + r; f.heavy_r = L.heavy[r]; f.early_r = L.early[r]; f.flag_r = L.flag[r];
    f.cnt_w = L.cnt + w; f.heavy_w = L.heavy[w]; f.early_w = L.early[w]; f.flag_w = L.flag[w];
    f.cnt_z = L.cnt + z;
    f.n_slots = L.fb_slots;
    L.fb_seq++;
    return f;
}

// Feedback sets of k_batch rotate by launch: read what the previous launch wrote, write the next
// set; the set after that is zeroed by this launch's k_classify for the launch that will write it.
// Called once per launch that does take place (rt_render): the lane's two list headers alternate, and
// each launch's k_batch zeroes the one its successor's k_classify fills.
int next_batch_feedback(rt_context *c, Lane &L, BatchFeedback &f, uint32_t ss_log2)
{
    const int r = (int)((L.bfb_seq + 2u) % 3u), w = (int)(L.bfb_seq % 3u), z = (int)((L.bfb_seq + 1u) % 3u);
    f.cnt_r = L.bcnt + r; f.heavy_r = L.bheavy[r]; f.early_r = L.bearly[r]; f.flag_r = L.bflag[r];
    f.n_slots = L.bfb_slots;
    f.lists = f.lists_w = L.blists + (L.bl_seq & 1u); f.lists_next = L.blists + ((L.bl_seq + 1u) & 1u);
    f.norm = f.norm_w = L.bnorm; f.late = f.late_w = L.blate;
    L.bl_seq++;
    if (!c->batch_feedback) {                // ordering off: every unit is ordinary, nothing is recorded
        f.cnt_w = nullptr; f.heavy_w = nullptr; f.early_w = nullptr; f.flag_w = nullptr;
        f.cnt_z = nullptr; f.flag_z = nullptr;
        f.flags_valid = 0; f.settled = 0;
        return RT_OK;
    }
    f.cnt_w = L.bcnt + w; f.heavy_w = L.bheavy[w]; f.early_w = L.bearly[w]; f.flag_w = L.bflag[w];
    f.cnt_z = L.bcnt + z; f.flag_z = L.bflag[z];
    // the lists of set r are for this launch only if a launch of the same shape (samples per wave) wrote them
    f.flags_valid = L.bfb_shape[r] == ss_log2 + 1u ? 1u : 0u;
    // the first launch of a shape flags nothing (it has no costs to set thresholds by), so the lists count from the third
    L.bfb_run = f.flags_valid ? L.bfb_run + 1u : 0u;
    f.settled = L.bfb_run >= 2u ? 1u : 0u;
    L.bfb_shape[w] = ss_log2 + 1u;
    L.bfb_shape[z] = 0;
    L.bfb_seq++;
    return RT_OK;
}

// One launch of the path kernel on a lane (one progressive sample of its pixel range),
// optionally bracketed by HIP events on the lane's stream.
void timed_path(rt_context *c, Lane &L, const DevView &v, int max_depth, uint32_t n_samples)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        e0 = next_event(c);
        e1 = next_event(c);
    }
    ChunkFeedback f = next_feedback(L);
    BounceQueue bq{};
    bq.jobs = L.bjobs; bq.cap = L.bq_cap;
    bq.ctr = L.bq_ctr + 2u * (L.bq_seq & 1u); bq.ctr_next = L.bq_ctr + 2u * ((L.bq_seq + 1u) & 1u);
    if (e0) (void)hipEventRecord(e0, L.stream);
    if (launch_path(L.stream, v, L.lo, L.hi - L.lo, max_depth, c->max_blocks, L.ctl,
                    L.ctl->work[L.seq & 1u], L.ctl->work[(L.seq + 1u) & 1u], &f, n_samples, c->knobs, &bq))
        L.bq_seq++;
    L.seq++;
    if (e1) (void)hipEventRecord(e1, L.stream);
    c->launches++;
}

// Culling extent: rays whose origin lies inside [-extent, extent]^3 (the camera and
// every hit point of an ordinary scene) are culled; the boxes carry the fp32 slack of
// p = o + t d for coordinates up to that magnitude (rt_upload.cpp).  Other rays are
// tested against every live triangle.
void update_extent(rt_context *c)
{
    c->v.tree.extent = kCullExtent * std::max(c->geom_extent, 1.0f);
}

} // namespace

// =====================================================================================
extern "C" const char *rt_last_error(const rt_context *ctx)
{
    return ctx ? ctx->err.c_str() : g_err.c_str();
}

extern "C" int rt_create(rt_context **out, int device, int width, int height)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID, "ctx out-pointer is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL)
        return fail(nullptr, RT_ERR_INVALID, "bad image size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, RT_ERR_DEVICE, "no HIP device: this library needs a gfx950 GPU");
    if (device < 0 || device >= ndev)
        return fail(nullptr, RT_ERR_INVALID, "device index out of range");
    rt_context *c = new rt_context();
    c->device = device;
    c->width = width;
    c->height = height;
    c->v.width = width;
    c->v.height = height;
    c->v.cull = 1;
    c->v.sorted_planes = 1;
    if (const char *e = std::getenv("RT_SORTED")) c->v.sorted_planes = std::atoi(e) != 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreate(&c->stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_r0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_r1);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) e = scratch_address(device, &c->scratch);   // this device's copy, for rt_scene_hit_noalloc
    if (e != hipSuccess) {
        std::string m = std::string("rt_create: ") + hipGetErrorString(e);
        delete c;
        return fail(nullptr, RT_ERR_DEVICE, m);
    }
    // blocks per wave-per-SIMD of occupancy: one 256-thread block per CU.  launch_path makes
    // the grid what is resident at once; more blocks would only queue behind those.
    c->max_blocks = std::max(1, prop.multiProcessorCount);
    if (const char *e = std::getenv("RT_MAX_BLOCKS")) c->max_blocks = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("RT_LANES")) { c->lanes_wanted = std::max(1, std::min(8, std::atoi(e))); c->lanes_forced = true; }
    int rc = RT_OK;
    c->lanes.resize((size_t)c->lanes_wanted);
    for (Lane &L : c->lanes) {
        if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&L.done, hipEventDisableTiming) != hipSuccess) {
            rc = fail(c, RT_ERR_DEVICE, "rt_create: could not create lane streams");
            break;
        }
    }
    if (const char *e = std::getenv("RT_BATCH")) c->batch_wanted = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_FEEDBACK")) c->batch_feedback = std::atoi(e) != 0;
    if (const char *e = std::getenv("RT_BATCH_MIN_TOP")) c->batch_min_top = std::atoi(e);
    if (const char *e = std::getenv("RT_PATH_RUN")) c->path_run_wanted = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("RT_FEEDBACK_DEBUG")) c->feedback_debug = std::atoi(e) != 0;
    if (const char *e = std::getenv("RT_FORCE_GENERAL")) c->force_general = std::atoi(e) != 0;
    c->knobs = read_knobs();
    if (rc == RT_OK) rc = setup_partition(c);
    if (rc == RT_OK) rc = do_reset(c);
    if (rc != RT_OK) {
        std::string m = c->err;
        rt_destroy(c);
        return fail(nullptr, rc, m);
    }
    *out = c;
    return RT_OK;
}

extern "C" int rt_destroy(rt_context *c)
{
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_all(c->part_allocs);
    free_all(c->scene_allocs);
    free_all(c->allocs);
    for (Lane &L : c->lanes) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev_r0) (void)hipEventDestroy(c->ev_r0);
    if (c->ev_r1) (void)hipEventDestroy(c->ev_r1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

extern "C" int rt_upload_scene(rt_context *c, const rt_scene_desc *d)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(c, hipSetDevice(c->device));
    HostScene hs;
    std::string err;
    int rc = build_host_scene(d, hs, err);
    if (rc != RT_OK) return fail(c, rc, err);
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_all(c->scene_allocs);
    c->have_scene = false;
    DevView &v = c->v;
    TriRec *tris = nullptr; TriCold *cold = nullptr; ModelRec *models = nullptr;
    SphereRec *spheres = nullptr; LightRec *lights = nullptr;
    NodeGroup *nodes = nullptr; uint32_t *pos_of = nullptr; TriLightRec *tlights = nullptr;
    MatRec *mats = nullptr;
    if ((rc = dalloc(c, c->scene_allocs, tlights, hs.tlights.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, mats, hs.mats.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, tris, hs.tris.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, cold, hs.cold.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, pos_of, hs.pos_of.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, nodes, hs.groups.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, models, hs.models.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, spheres, hs.spheres.size()))) return rc;
    if ((rc = dalloc(c, c->scene_allocs, lights, hs.lights.size()))) return rc;
    auto up = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    RT_HIP(c, up(tris, hs.tris.data(), hs.tris.size() * sizeof(TriRec)));
    RT_HIP(c, up(cold, hs.cold.data(), hs.cold.size() * sizeof(TriCold)));
    RT_HIP(c, up(pos_of, hs.pos_of.data(), hs.pos_of.size() * sizeof(uint32_t)));
    RT_HIP(c, up(nodes, hs.groups.data(), hs.groups.size() * sizeof(NodeGroup)));
    RT_HIP(c, up(models, hs.models.data(), hs.models.size() * sizeof(ModelRec)));
    RT_HIP(c, up(spheres, hs.spheres.data(), hs.spheres.size() * sizeof(SphereRec)));
    RT_HIP(c, up(lights, hs.lights.data(), hs.lights.size() * sizeof(LightRec)));
    RT_HIP(c, up(tlights, hs.tlights.data(), hs.tlights.size() * sizeof(TriLightRec)));
    RT_HIP(c, up(mats, hs.mats.data(), hs.mats.size() * sizeof(MatRec)));
    v.tlights = tlights;
    v.n_tlights = (int)hs.tlights.size();
    v.env = EnvView{};
    if (hs.env_n) {
        float4 *eimg = nullptr; float *cv = nullptr, *cc = nullptr, *cf = nullptr, *mv = nullptr, *mc = nullptr;
        if ((rc = dalloc(c, c->scene_allocs, eimg, hs.env_img.size() / 4))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, cv, hs.env_cond_value.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, cc, hs.env_cond_cdf.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, cf, hs.env_cond_funint.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, mv, hs.env_marg_value.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, mc, hs.env_marg_cdf.size()))) return rc;
        RT_HIP(c, up(eimg, hs.env_img.data(), hs.env_img.size() * sizeof(float)));
        RT_HIP(c, up(cv, hs.env_cond_value.data(), hs.env_cond_value.size() * sizeof(float)));
        RT_HIP(c, up(cc, hs.env_cond_cdf.data(), hs.env_cond_cdf.size() * sizeof(float)));
        RT_HIP(c, up(cf, hs.env_cond_funint.data(), hs.env_cond_funint.size() * sizeof(float)));
        RT_HIP(c, up(mv, hs.env_marg_value.data(), hs.env_marg_value.size() * sizeof(float)));
        RT_HIP(c, up(mc, hs.env_marg_cdf.data(), hs.env_marg_cdf.size() * sizeof(float)));
        v.env.img = (const float *)eimg; v.env.cond_value = cv; v.env.cond_cdf = cc; v.env.cond_funint = cf;
        v.env.marg_value = mv; v.env.marg_cdf = mc; v.env.marg_funint = hs.env_marg_funint;
        v.env.w = hs.env_w; v.env.h = hs.env_h; v.env.n = 1;
    }
    v.tex = TexView{};
    if (!hs.textures.empty()) {
        // initImages (Texture.cu:135-186): level 0 from the caller's image (:150-159), the rest of each chain
        // level by level on the device (generateMipMaps, :48-133)
        float4 *tdata = nullptr; TexRec *trecs = nullptr; int32_t *mtex = nullptr; TriUv *uv = nullptr;
        if ((rc = dalloc(c, c->scene_allocs, tdata, hs.tex_texels))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, trecs, hs.textures.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, mtex, hs.mesh_tex.size()))) return rc;
        if ((rc = dalloc(c, c->scene_allocs, uv, hs.uv.size()))) return rc;
        RT_HIP(c, up(trecs, hs.textures.data(), hs.textures.size() * sizeof(TexRec)));
        RT_HIP(c, up(mtex, hs.mesh_tex.data(), hs.mesh_tex.size() * sizeof(int32_t)));
        RT_HIP(c, up(uv, hs.uv.data(), hs.uv.size() * sizeof(TriUv)));
        for (size_t k = 0; k < hs.textures.size(); k++) {
            const TexRec &T = hs.textures[k];
            RT_HIP(c, up(tdata + T.off[0], hs.tex_level0.data() + 4 * (size_t)hs.tex_level0_at[k],
                         (size_t)T.w * T.h * sizeof(float4)));
            int lw = T.w, lh = T.h;
            for (int l = 1; l < T.levels; l++) {
                const int nw = std::max(1, lw / 2), nh = std::max(1, lh / 2);
                launch_mip_level(c->stream, tdata + T.off[l - 1], lw, lh, tdata + T.off[l], nw, nh);
                lw = nw; lh = nh;
            }
        }
        RT_HIP(c, hipGetLastError());
        RT_HIP(c, hipStreamSynchronize(c->stream));
        v.tex.recs = trecs; v.tex.data = (const float *)tdata; v.tex.mesh_tex = mtex; v.tex.uv = uv;
        v.tex.n = (int)hs.textures.size(); v.tex.n_tri_models = hs.n_tri_models;
    }
    v.mats = mats;
    v.general = hs.general ? 1 : 0;
    if (c->force_general) v.general = 1;                     // diagnostics
    v.tris = tris; v.tri_pos_of = pos_of; v.tri_cold = cold; v.models = models;
    v.spheres = spheres; v.lights = lights;
    v.n_tris = (int)hs.cold.size();
    v.n_spheres = (int)hs.spheres.size();
    v.n_lights = (int)hs.lights.size();
    v.tree.groups = nodes;
    for (int l = 0; l < kMaxLevels; l++)
        v.tree.level_off[l] = (uint32_t)hs.level_off[std::min<size_t>((size_t)l, hs.level_off.size() - 1)];
    v.tree.top = hs.top;
    v.tree.n_groups0 = (int)hs.level_cnt[0];
    c->geom_extent = hs.geom_extent;
    update_extent(c);
    c->n_lights = v.n_lights;
    c->n_prims = (uint64_t)hs.cold.size() + hs.spheres.size();   // hs.tris is padded to whole groups
    c->have_scene = true;
    return RT_OK;
}

extern "C" int rt_set_camera(rt_context *c, const rt_camera_desc *cam)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!cam) return fail(c, RT_ERR_INVALID, "camera is NULL");
    if (cam->res_x <= 0 || cam->res_y <= 0) return fail(c, RT_ERR_INVALID, "camera resolution must be positive");
    c->v.cam = make_camera(cam);
    c->have_cam = true;
    update_extent(c);
    return RT_OK;
}

extern "C" int rt_set_partition(rt_context *c, int rank, int world, int tile)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (world < 1 || rank < 0 || rank >= world || tile < 8 || (tile % 8) != 0)
        return fail(c, RT_ERR_INVALID, "need 0 <= rank < world and tile a positive multiple of 8");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->rank = rank; c->world = world; c->tile = tile;
    int rc = setup_partition(c);
    if (rc != RT_OK) return rc;
    return do_reset(c);
}

extern "C" int rt_partition_pixels(int width, int height, int rank, int world, int tile,
                                   int32_t *indices, int64_t *count)
{
    if (width <= 0 || height <= 0 || world < 1 || rank < 0 || rank >= world || tile < 8 ||
        (tile % 8) != 0 || !count)
        return fail(nullptr, RT_ERR_INVALID, "rt_partition_pixels: bad argument");
    std::vector<uint32_t> px;
    partition_pixels(width, height, rank, world, tile, px);
    *count = (int64_t)px.size();
    if (indices)
        std::memcpy(indices, px.data(), px.size() * sizeof(uint32_t));
    return RT_OK;
}

extern "C" int rt_reset(rt_context *c)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(c, hipSetDevice(c->device));
    return do_reset(c);
}

extern "C" int rt_render(rt_context *c, int n_samples, int max_depth)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!c->have_scene || !c->have_cam)
        return fail(c, RT_ERR_STATE, "rt_render needs rt_upload_scene and rt_set_camera first");
    if (n_samples < 0 || max_depth < 0) return fail(c, RT_ERR_INVALID, "n_samples and max_depth must be >= 0");
    RT_HIP(c, hipSetDevice(c->device));
    hipStream_t s0 = c->stream;
    if (max_depth == 0 && c->v.cull && c->v.tree.top >= c->batch_min_top && !c->v.general) {
        // Depth 0 over a deep tree, one kernel per batch of samples and lane (k_batch,
        // rt_kernels.hip): camera rays from predicted counters, shadow rays and the running mean
        // of a unit without leaving the wave.  The lanes' launches follow each other on their own
        // streams, so one lane's tail is filled by the other's next launch.
        // Measured (tools/partition_rehearsal.py, C3 and C4): 64 samples per launch beat 32 and
        // 16 (fewer tails, two-pixel units); one launch over the whole partition beats two lanes
        // from about 160k pixels on (14.8 vs 15.4 ms C3, 59.4 vs 61.7 ms C4), below that the
        // second lane's launches fill the first's tail (2.3 vs 2.7 ms on an eighth of C3).
        const int want = c->batch_wanted ? c->batch_wanted : n_samples >= 64 ? 64 : n_samples >= 32 ? 32 : 16;
        const uint32_t cap = (uint32_t)std::min(std::max(n_samples, 1), want);
        const bool whole = c->n_lanes > 1 && !c->lanes_forced && c->n_local >= kWholeBatchPixels;
        const int nb = whole ? 1 : c->n_lanes;
        c->lanes_used = nb;
        RT_HIP(c, hipEventRecord(c->ev_r0, s0));
        for (int k = 0; k < nb; k++)
            RT_HIP(c, hipStreamWaitEvent(c->lanes[k].stream, c->ev_r0, 0));
        // A failure inside the loop must not leave the lanes forked behind ev_r0 with the sample count, the
        // framebuffer's running mean and the lanes' counters out of step: the loop stops BEFORE the failing
        // batch is counted, the lanes are joined and the stream synchronised below, and then the call fails.
        int rc_fail = RT_OK;
        for (int done = 0; done < n_samples && rc_fail == RT_OK;) {
            const uint32_t S = (uint32_t)std::min<int>((int)cap, n_samples - done);
            // one feedback set per lane, taken before anything of this batch is launched
            BatchFeedback fbs[8] = {};
            uint32_t ss_log2 = 0;                            // samples a wave holds: the smallest power of two >= S (<= 64)
            while (ss_log2 < 6u && (1u << ss_log2) < S) ss_log2++;
            for (int k = 0; k < nb && rc_fail == RT_OK; k++) {
                Lane &ln = c->lanes[k];
                const uint32_t hi = whole ? c->lanes[c->n_lanes - 1].hi : ln.hi;
                if (hi > ln.lo)
                    rc_fail = next_batch_feedback(c, ln, fbs[k], ss_log2);
            }
            if (rc_fail != RT_OK)
                break;
            for (int k = 0; k < nb; k++) {
                Lane &ln = c->lanes[k];
                const uint32_t lo = ln.lo, hi = whole ? c->lanes[c->n_lanes - 1].hi : ln.hi;
                if (hi <= lo) continue;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (c->timing) { e0 = next_event(c); e1 = next_event(c); }
                const BatchFeedback &f = fbs[k];
                if (e0) (void)hipEventRecord(e0, ln.stream);
                launch_batch(ln.stream, c->v, lo, hi - lo, S, c->cnt, c->max_blocks, ln.ctl,
                             ln.ctl->work[ln.seq & 1u], ln.ctl->work[(ln.seq + 1u) & 1u],
                             &f, c->knobs);
                ln.seq++;
                if (e1) (void)hipEventRecord(e1, ln.stream);
                c->launches++;
                c->batch_launches++;
            }
            c->cnt += (int)S;
            done += (int)S;
            c->samples += (uint64_t)c->n_local * S;
            c->primary += (uint64_t)c->n_local * S;
            // event pairs are resolved without disturbing the streams: only when the pool would grow past a bound
            if (c->timing && c->ev_used > 32768) {
                for (int k = 0; k < nb; k++)
                    RT_HIP(c, hipStreamSynchronize(c->lanes[k].stream));
                resolve_events(c);
            }
        }
        for (int k = 0; k < nb; k++) {
            RT_HIP(c, hipEventRecord(c->lanes[k].done, c->lanes[k].stream));
            RT_HIP(c, hipStreamWaitEvent(s0, c->lanes[k].done, 0));
        }
        if (rc_fail != RT_OK) {
            (void)hipStreamSynchronize(s0);
            return rc_fail;
        }
    } else {
    c->lanes_used = c->n_lanes;
    RT_HIP(c, hipEventRecord(c->ev_r0, s0));                 // fork: lanes start after the main stream
    for (int k = 0; k < c->n_lanes; k++)
        RT_HIP(c, hipStreamWaitEvent(c->lanes[k].stream, c->ev_r0, 0));
    // over a shallow tree (or in scan mode) one launch renders a run of samples (path_run); elsewhere one each
    // (runs of 4 measured best on C2: whole frame 3.62 -> 3.53 ms per 16 samples, an eighth of it 0.87 -> 0.78;
    // longer runs lose more to the uneven shares of the strided chunk assignment than they save in launches)
    const int run_cap = !path_renders_runs(c->v) ? 1 : c->path_run_wanted ? c->path_run_wanted : 4;
    for (int smp = 0; smp < n_samples;) {
        const int run = std::min(run_cap, n_samples - smp);
        DevView v = c->v;
        v.cnt = c->cnt + 1;
        c->cnt += run;
        smp += run;
        for (int k = 0; k < c->n_lanes; k++) {
            Lane &ln = c->lanes[k];
            if (ln.hi > ln.lo) {
                timed_path(c, ln, v, max_depth, (uint32_t)run);
                c->primary += (uint64_t)(ln.hi - ln.lo) * (uint64_t)run;
            }
        }
        c->samples += (uint64_t)c->n_local * (uint64_t)run;
        // event pairs are resolved without disturbing the streams: only when the pool
        // would grow past a bound, or at the end of the call (which synchronises anyway)
        if (c->timing && c->ev_used > 32768) {
            for (int k = 0; k < c->n_lanes; k++)
                RT_HIP(c, hipStreamSynchronize(c->lanes[k].stream));
            resolve_events(c);
        }
    }
    for (int k = 0; k < c->n_lanes; k++) {                   // join
        RT_HIP(c, hipEventRecord(c->lanes[k].done, c->lanes[k].stream));
        RT_HIP(c, hipStreamWaitEvent(s0, c->lanes[k].done, 0));
    }
    }
    RT_HIP(c, hipEventRecord(c->ev_r1, s0));
    RT_HIP(c, hipStreamSynchronize(s0));          // cudaStreamSynchronize, kernel.cu:123
    RT_HIP(c, hipGetLastError());
    resolve_events(c);
    if (c->feedback_debug) {                                 // RT_FEEDBACK_DEBUG=1: what the last k_batch launches listed
        for (int k = 0; k < c->n_lanes; k++) {
            Lane &ln = c->lanes[k];
            if (!ln.bcnt || !ln.bfb_seq) continue;
            FbCounts h{};
            RT_HIP(c, hipMemcpy(&h, ln.bcnt + (ln.bfb_seq + 2u) % 3u, sizeof h, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "rt: lane %d launch %u listed %u units, early %u + %u, cost total %llu, shape %u\n", k,
                         ln.bfb_seq, h.n_heavy, h.n_early_a, h.n_early_b, (unsigned long long)h.total, h.pad);
        }
    }
    float ms = 0.0f;
    RT_HIP(c, hipEventElapsedTime(&ms, c->ev_r0, c->ev_r1));
    c->render_ms += ms;
    return RT_OK;
}

static int ensure_image_staging(rt_context *c)
{
    if (c->img_rgba) return RT_OK;
    const size_t n = (size_t)c->width * c->height;
    int rc;
    if ((rc = dalloc(c, c->allocs, c->img_rgba, n))) return rc;
    if ((rc = dalloc(c, c->allocs, c->img_tmp, n))) return rc;
    if ((rc = dalloc(c, c->allocs, c->img_prim, n))) return rc;
    if ((rc = dalloc(c, c->allocs, c->img_t, n))) return rc;
    return RT_OK;
}

static int scatter_full_image(rt_context *c)
{
    int rc = ensure_image_staging(c);
    if (rc) return rc;
    const size_t n = (size_t)c->width * c->height;
    hipStream_t s = c->stream;
    RT_HIP(c, hipMemsetAsync(c->img_rgba, 0, n * sizeof(float4), s));
    RT_HIP(c, hipMemsetAsync(c->img_t, 0, n * sizeof(float), s));
    launch_fill_i32(s, c->img_prim, n, -2);
    launch_scatter_image(s, c->v.pixmap, c->n_local, c->v.fb, c->v.first_id, c->v.first_t,
                         c->img_rgba, c->img_prim, c->img_t);
    RT_HIP(c, hipStreamSynchronize(s));
    return RT_OK;
}

extern "C" int rt_read_framebuffer(rt_context *c, float *rgba, int32_t *prim_id, float *t)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(c, hipSetDevice(c->device));
    int rc = scatter_full_image(c);
    if (rc) return rc;
    const size_t n = (size_t)c->width * c->height;
    if (rgba) RT_HIP(c, hipMemcpy(rgba, c->img_rgba, n * sizeof(float4), hipMemcpyDeviceToHost));
    if (prim_id) RT_HIP(c, hipMemcpy(prim_id, c->img_prim, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (t) RT_HIP(c, hipMemcpy(t, c->img_t, n * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_local_pixel_count(rt_context *c, int64_t *count)
{
    if (!c || !count) return fail(c, RT_ERR_INVALID, "NULL argument");
    *count = (int64_t)c->n_local;
    return RT_OK;
}

extern "C" int rt_local_pixel_indices(rt_context *c, int32_t *indices)
{
    if (!c || !indices) return fail(c, RT_ERR_INVALID, "NULL argument");
    std::memcpy(indices, c->h_pixmap.data(), c->h_pixmap.size() * sizeof(uint32_t));
    return RT_OK;
}

extern "C" int rt_export_local_rgba(rt_context *c, void *device_dst)
{
    if (!c || !device_dst) return fail(c, RT_ERR_INVALID, "NULL argument");
    RT_HIP(c, hipSetDevice(c->device));
    if (c->n_local)
        RT_HIP(c, hipMemcpyAsync(device_dst, c->v.fb, (size_t)c->n_local * sizeof(float4),
                                 hipMemcpyDeviceToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

extern "C" int rt_scene_hit(rt_context *c, int n, const float *org, const float *dir, float *t,
                            int32_t *prim_id, int32_t *ishit)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!c->have_scene) return fail(c, RT_ERR_STATE, "rt_scene_hit needs rt_upload_scene first");
    if (n < 0 || (n > 0 && (!org || !dir))) return fail(c, RT_ERR_INVALID, "bad ray arrays");
    if (n == 0) return RT_OK;
    RT_HIP(c, hipSetDevice(c->device));
    std::vector<void *> tmp;
    RayQueue q{};
    float *d_org = nullptr, *d_dir = nullptr, *d_t = nullptr;
    int32_t *d_id = nullptr, *d_hit = nullptr;
    int rc = alloc_queue(c, tmp, q, (size_t)n);
    if (!rc) rc = dalloc(c, tmp, d_org, (size_t)n * 3);
    if (!rc) rc = dalloc(c, tmp, d_dir, (size_t)n * 3);
    if (!rc) rc = dalloc(c, tmp, d_t, (size_t)n);
    if (!rc) rc = dalloc(c, tmp, d_id, (size_t)n);
    if (!rc) rc = dalloc(c, tmp, d_hit, (size_t)n);
    hipError_t e = hipSuccess;
    if (!rc) {
        uint32_t un = (uint32_t)n;
        hipStream_t s = c->stream;
        e = hipMemcpyAsync(d_org, org, (size_t)n * 12, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_dir, dir, (size_t)n * 12, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            launch_prepare_rays(s, q, d_org, d_dir, un);
            Lane &L0 = c->lanes[0];
            launch_intersect(s, c->v, q, un, d_t, d_id, c->max_blocks, L0.ctl->work[L0.seq & 1u],
                             L0.ctl->work[(L0.seq + 1u) & 1u], c->knobs);
            L0.seq++;
            launch_ishit(s, d_t, d_hit, un);
            e = hipStreamSynchronize(s);
        }
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess && t) e = hipMemcpy(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && prim_id) e = hipMemcpy(prim_id, d_id, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && ishit) e = hipMemcpy(ishit, d_hit, (size_t)n * 4, hipMemcpyDeviceToHost);
    }
    free_all(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, std::string("rt_scene_hit: ") + hipGetErrorString(e));
    return RT_OK;
}

// Scene::hit for callers that cannot allocate device memory (rt_abi.h).  The same three kernels
// as rt_scene_hit -- normalise, closest hit, bool -- on the context's stream; the ray queue and the
// dispensers live in this device's static scratch array, inputs and outputs are the caller's
// device buffers.  No hipMalloc / hipFree / allocating routine on this path.
extern "C" int rt_scene_hit_noalloc(rt_context *c, int n, const float *d_org, const float *d_dir,
                                    float *d_t, int32_t *d_prim_id, int32_t *d_ishit)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (!c->have_scene) return fail(c, RT_ERR_STATE, "rt_scene_hit_noalloc needs rt_upload_scene first");
    if (n < 0) return fail(c, RT_ERR_INVALID, "rt_scene_hit_noalloc: n must be >= 0");
    if ((size_t)n > kNoAllocMaxRays)
        return fail(c, RT_ERR_INVALID, "rt_scene_hit_noalloc: n exceeds RT_SCENE_HIT_NOALLOC_MAX_RAYS (" +
                                           std::to_string(kNoAllocMaxRays) + " rays per call)");
    if (n == 0) return RT_OK;
    if (!d_org || !d_dir || !d_t || !d_prim_id || !d_ishit)
        return fail(c, RT_ERR_INVALID, "rt_scene_hit_noalloc: device buffers must not be NULL");
    if (!c->scratch) return fail(c, RT_ERR_STATE, "rt_scene_hit_noalloc: no scratch address for this device");
    RT_HIP(c, hipSetDevice(c->device));
    const uint32_t un = (uint32_t)n;
    hipStream_t s = c->stream;
    float *qf = (float *)c->scratch;
    RayQueue q{qf, qf + kNoAllocMaxRays, qf + 2 * kNoAllocMaxRays, qf + 3 * kNoAllocMaxRays,
               qf + 4 * kNoAllocMaxRays, qf + 5 * kNoAllocMaxRays};
    uint32_t *work = (uint32_t *)(c->scratch + kScratchQueueBytes);
    // the array keeps the last call's contents: the dispensers must start at zero
    RT_HIP(c, hipMemsetAsync(work, 0, kScratchCtlBytes, s));
    launch_prepare_rays(s, q, d_org, d_dir, un);
    launch_intersect(s, c->v, q, un, d_t, d_prim_id, c->max_blocks, work, work + kWorkSlots, c->knobs);
    launch_ishit(s, d_t, d_ishit, un);
    RT_HIP(c, hipStreamSynchronize(s));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_postprocess(rt_context *c, float *hdr_rgba, float *filtered_rgba)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    if (c->world != 1)
        return fail(c, RT_ERR_STATE, "rt_postprocess needs the whole image on one context (world = 1)");
    RT_HIP(c, hipSetDevice(c->device));
    int rc = scatter_full_image(c);
    if (rc) return rc;
    const size_t n = (size_t)c->width * c->height;
    hipStream_t s = c->stream;
    // render -> HDR -> filter (kernel.cu:249-251): raw in img_rgba, HDR in img_tmp
    launch_tonemap(s, c->img_rgba, c->img_tmp, (uint32_t)n);
    RT_HIP(c, hipStreamSynchronize(s));
    if (hdr_rgba) RT_HIP(c, hipMemcpy(hdr_rgba, c->img_tmp, n * sizeof(float4), hipMemcpyDeviceToHost));
    if (filtered_rgba) {
        launch_median3x3(s, c->img_tmp, c->img_rgba, c->width, c->height);
        RT_HIP(c, hipStreamSynchronize(s));
        RT_HIP(c, hipMemcpy(filtered_rgba, c->img_rgba, n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_get_stats(rt_context *c, rt_stats *st)
{
    if (!c || !st) return fail(c, RT_ERR_INVALID, "NULL argument");
    RT_HIP(c, hipSetDevice(c->device));
    unsigned long long shadow_total = 0, bounce_total = 0, coop_total = 0;
    std::vector<Ctl *> ctls;
    for (int k = 0; k < c->n_lanes; k++) ctls.push_back(c->lanes[k].ctl);
    for (Ctl *p : ctls) {
        Ctl h;
        RT_HIP(c, hipMemcpy(&h, p, sizeof(Ctl), hipMemcpyDeviceToHost));
        shadow_total += h.shadow_total;
        bounce_total += h.bounce_total;
        coop_total += h.coop_total;
    }
    st->samples = c->samples;
    st->primary_rays = c->primary;
    st->shadow_rays = shadow_total;
    st->bounce_rays = bounce_total;
    st->path_launches = c->timing ? c->timed_launches : c->launches;
    st->path_ms = c->path_ms;
    st->render_ms = c->render_ms;
    st->n_prims = c->n_prims;
    st->streams = (uint64_t)c->lanes_used;
    st->coop_chunks = coop_total;
    st->batch_launches = c->batch_launches;
    st->batch_listed = 0;
    st->batch_early = 0;
    for (int k = 0; k < c->n_lanes; k++) {
        const Lane &ln = c->lanes[k];
        if (!ln.bcnt || !ln.bfb_seq) continue;
        FbCounts h{};
        RT_HIP(c, hipMemcpy(&h, ln.bcnt + (ln.bfb_seq + 2u) % 3u, sizeof h, hipMemcpyDeviceToHost));
        st->batch_listed += std::min(h.n_heavy, ln.bfb_slots);
        st->batch_early += std::min(h.n_early_a + h.n_early_b, ln.bfb_slots);
    }
    return RT_OK;
}

extern "C" int rt_reset_stats(rt_context *c)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->n_lanes; k++) {
        RT_HIP(c, hipMemset(c->lanes[k].ctl, 0, sizeof(Ctl)));
        c->lanes[k].seq = 0;
    }
    c->batch_launches = 0;
    c->samples = c->primary = c->launches = c->timed_launches = 0;
    c->path_ms = c->render_ms = 0.0;
    return RT_OK;
}

extern "C" int rt_host_bounds(const rt_scene_desc *d, float *bounds, int32_t *position,
                              float *geom_extent)
{
    HostScene hs;
    std::string err;
    int rc = build_host_scene(d, hs, err);
    if (rc != RT_OK) return fail(nullptr, rc, err);
    const size_t n = hs.cold.size();
    for (size_t g = 0; g < n; g++) {
        const uint32_t pos = hs.pos_of[g];
        const NodeGroup &G = hs.groups[hs.level_off[0] + pos / kTreeArity];
        if (bounds)
            for (int a = 0; a < 3; a++) {
                bounds[6 * g + a] = G.lo[a][pos % kTreeArity];
                bounds[6 * g + 3 + a] = G.hi[a][pos % kTreeArity];
            }
        if (position) position[g] = (int32_t)pos;
    }
    if (geom_extent) *geom_extent = hs.geom_extent;
    return RT_OK;
}

extern "C" int rt_set_culling(rt_context *c, int enabled)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    c->v.cull = enabled != 0;
    return RT_OK;
}

extern "C" int rt_set_timing(rt_context *c, int enabled)
{
    if (!c) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    c->timing = enabled != 0;
    return RT_OK;
}
