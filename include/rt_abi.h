/*
 * rt_abi.h -- C ABI of the MI355X-native per-pixel render path.
 *
 * The reference (UI233/RayTracer) has no plugin / FFI layer; the seam this
 * library replaces is the set of host-visible calls its driver makes
 * (SURVEY.md section 8(b)).  Each entry point below cites the reference
 * interface it stands in for (paths relative to src/RayTracer/).
 *
 * Conventions: plain pointers and sizes; every call returns an int status
 * (RT_OK = 0) and never throws; host arrays are caller-owned and copied; calls
 * are blocking; one host thread per context.  The reference keeps only the last
 * cudaError_t of each routine and returns a bool (Scene/Scene.cu:132,201;
 * kernel.cu:202,255); here the first failing step aborts the call, and
 * rt_last_error() describes it.
 *
 * The library needs a gfx950 device; there is no CPU fallback.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID 1      /* bad argument                                  */
#define RT_ERR_UNSUPPORTED 2  /* scene feature outside the hot path's scope    */
#define RT_ERR_DEVICE 3       /* HIP runtime error (no device, OOM, launch)    */
#define RT_ERR_STATE 4        /* call out of order (e.g. render before upload) */
#define RT_ERR_IO 5           /* file could not be read / written              */

typedef struct rt_context rt_context;

/* The 11 arguments of
 *   bool Scene::initializeScene(int light_size[], int model_size[], PointLight*,
 *        DirectionalLight*, TriangleLight*, EnvironmentLight*, Triangle*, Mesh*,
 *        Quadratic*, int material_type[], Material*)        (Scene/Scene.cuh:16-17)
 * flattened to POD arrays.  Triangles are in the reference's traversal order
 * (Scene/Scene.cu:13-41): the model_size[0] standalone Triangle models, then each
 * Mesh's meshTable.  Models are numbered triangles, meshes, quadrics -- the order
 * initializeScene walks material_type[] / Material[] (Scene/Scene.cu:143-149).
 * Point and directional lights (the delta lights, Scene.cu:93-102) and triangle lights
 * (hit-able area lights: Scene.cu:43-64, emission core/PathTracer.cu:33-52, the MIS branch
 * Scene.cu:242-284) are supported; a scene with triangle lights renders through the general
 * path kernel, not the depth-0 batch kernel; so does one with an environment light
 * (Light.cu:137-234, PiecewiseConstant.cu; at most one, as Scene::getEnvironmentLight reads), and
 * one with mesh textures (Model.cu:60-88,304-309, Texture/Texture.cu; software tex2DGrad, DESIGN H13).
 * A light hit reports prim_id = n_tris + n_quads + its index. */
typedef struct rt_scene_desc {
    int32_t n_tri_models;          /* model_size[0]                               */
    int32_t n_meshes;              /* model_size[1]                               */
    int32_t n_quads;               /* model_size[2]                               */
    int32_t n_tris;                /* n_tri_models + sum(mesh_count)              */
    const float *tri_pos;          /* [n_tris][3][3] Triangle::pos (Model.cuh:76) */
    const float *tri_nrm;          /* [n_tris][3][3] Triangle::normal (:77)       */
    const int32_t *tri_model;      /* [n_tris] owning model index                 */
    const int32_t *mesh_first;     /* [n_meshes] first triangle of Mesh j         */
    const int32_t *mesh_count;     /* [n_meshes] Mesh::number (Model.cuh:97)      */
    const float *model_xf;         /* [n_models][16] Model::transformation, row-major (Model.cuh:57) */
    const int32_t *model_mat_type; /* [n_models] material::MATERIAL_TYPE (Object.cuh:13-21): LAMBERTIAN 1,
                                      FRESNEL 7, Cook 9, GGX 17, Oren_Nayar 25 */
    const float *model_color;      /* [n_models][3] the BRDF's color (material.cuh:39,62,81,103,162) */
    const float *quad_coeff;       /* [n_quads][3] Quadratic::coefficient (Model.cuh:116) */
    const int32_t *quad_type;      /* [n_quads] quadraticType, Sphere = 0 (Model.cuh:100-103) */
    int32_t n_point_lights;        /* light_size[0]                               */
    const float *pl_pos;           /* [n_point_lights][3] PointLight::pos (Light.cuh:49) */
    const float *pl_illum;         /* [n_point_lights][3] PointLight::illum       */
    int32_t n_dir_lights;          /* light_size[1]                               */
    const float *dl_dir;           /* [n_dir_lights][3] DirectionalLight(direction, ..) argument;
                                      normalised as its constructor does (Light.cu:21) */
    const float *dl_illum;         /* [n_dir_lights][3] DirectionalLight::illum (Light.cuh:57) */
    int32_t n_tri_lights;          /* light_size[2]                               */
    const float *tl_pos;           /* [n_tri_lights][3][3] TriangleLight(posa, posb, posc, ..) (Light.cuh:71) */
    const float *tl_illum;         /* [n_tri_lights][3] light_color               */
    const int32_t *tl_two_side;    /* [n_tri_lights] two (Light.cuh:71,102)       */
    const int32_t *tl_mat_type;    /* [n_tri_lights] material_type[n_models + k]: the light's own
                                      Material (Scene.cu:151-153); LAMBERTIAN only (Light.cu:115-125) */
    const float *tl_color;         /* [n_tri_lights][3] its Lambertian::color     */
    const float *model_param;      /* [n_models][2] (roughness, eta): the second constructor argument of
                                      GGX / Cook_Torrance (material.cuh:48,88) and of Fresnel (:112);
                                      may be NULL when every model is LAMBERTIAN or Oren_Nayar */
    int32_t n_env_lights;          /* light_size[3]: 0 or 1                        */
    int32_t env_width, env_height;
    const float *env_image;        /* [env_height][env_width][4] floats, as stbi_loadf(.., 4) hands them to
                                      EnvironmentLight(texture, w, h) (kernel.cu:271-272, Light.cuh:108) */
    const float *tri_uv;           /* [n_tris][3][2] Triangle::vText (Model.cuh:75), what Mesh::readFile takes
                                      from the OBJ's vt lines (Model.cu:227-229); NULL = zeros */
    int32_t n_textures;            /* images bound with Mesh::initTexture(data, w, h) (Model.cu:315-319) */
    const int32_t *mesh_texture;   /* [n_meshes] index of the image Mesh j bound, -1 = none: its hits keep the
                                      BRDF's own colour (hazard H5) */
    const int32_t *tex_size;       /* [n_textures][2] width, height */
    const float *tex_data;         /* the images back to back, each [height][width][4] floats; the library
                                      builds the mip chain of Texture.cu:48-133 at upload */
} rt_scene_desc;

/* Camera(pos, lookat, fov, near, far, int2 res, up)          (Camera/Camera.cuh:11).
 * The reference's driver passes res = (WIDTH/2, HEIGHT/2)    (kernel.cu:228-229). */
typedef struct rt_camera_desc {
    float pos[3];
    float lookat[3];
    float fov, near_, far_;
    int32_t res_x, res_y;
    float up[3];
} rt_camera_desc;

typedef struct rt_stats {
    uint64_t samples;        /* pixel-samples rendered since rt_reset_stats         */
    uint64_t primary_rays;   /* camera rays                                         */
    uint64_t shadow_rays;    /* shadow rays actually cast (Scene.cu:224-225)        */
    uint64_t bounce_rays;    /* rays of bounces >= 1 whose hit is used              */
    uint64_t path_launches;  /* launches of the dominant kernel: k_batch (max_depth 0 over a deep tree: one
                                per batch of samples and stream) or k_path (one per sample and stream) */
    double path_ms;          /* their summed HIP-event duration (timing mode only)  */
    double render_ms;        /* HIP-event duration of whole rt_render calls         */
    uint64_t n_prims;        /* triangles + quadrics of the uploaded scene          */
    uint64_t streams;        /* HIP streams the partition's pixel ranges render on   */
    uint64_t coop_chunks;    /* k_path chunks run in the cooperative form (four waves sharing the
                                traversals of one chunk; max_depth > 0 over a deep tree)        */
    uint64_t batch_launches; /* how many of the path_launches were k_batch                           */
    uint64_t batch_listed;   /* units the most recent k_batch launches listed as costly: the next     */
                             /* launch starts with them, one pixel per wave (ordering only)           */
    uint64_t batch_early;    /* units they listed to go out early, whole, before the bands (ordering only) */
} rt_stats;

/* Replaces the surface/graph set-up of initCUDA (kernel.cu:193-256): creates the
 * framebuffer (RGBA32F, kernel.cu:215-219), the per-pixel RNG streams
 * (`initial`, kernel.cu:44-48: seed = idx + 332, subsequence = idx = y*W + x)
 * and the frame counter (kernel.cu:242-243) on HIP device `device`. */
int rt_create(rt_context **ctx, int device, int width, int height);
int rt_destroy(rt_context *ctx);

/* Scene::initializeScene (Scene/Scene.cu:126-202) + the cudaMemcpy of the Scene
 * object (kernel.cu:311-312).  Builds the struct-of-arrays device scene. */
int rt_upload_scene(rt_context *ctx, const rt_scene_desc *scene);

/* Camera constructor + cudaMemcpyToSymbol(globalCam, ...) (kernel.cu:228-229,239). */
int rt_set_camera(rt_context *ctx, const rt_camera_desc *cam);

/* Pixel-tile data parallelism (no counterpart in the single-GPU reference;
 * SURVEY.md section 8(e)): this context renders the tile x tile pixel tiles whose
 * index (row-major over the image) is congruent to `rank` modulo `world`.
 * Default: rank 0 of 1, tile 64.  Resets the accumulation. */
int rt_set_partition(rt_context *ctx, int rank, int world, int tile);

/* The partition as a pure host computation (no device needed): the global pixel
 * indices (y*W + x) rank `rank` of `world` owns, in the order its local buffers use.
 * indices may be NULL to query the count only. */
int rt_partition_pixels(int width, int height, int rank, int world, int tile, int32_t *indices,
                        int64_t *count);

/* Zero the framebuffer, the frame counter and restart every RNG stream
 * (what re-running initCUDA does in the reference). */
int rt_reset(rt_context *ctx);

/* `n_samples` replays of the captured graph (cudaGraphLaunch + StreamSynchronize,
 * kernel.cu:122-123), i.e. n more progressive samples per pixel through
 * `debug` -> pathTracer (kernel.cu:50-79, core/PathTracer.cu:11-92).
 * max_depth is the reference's MAX_DEPTH macro (PathTracer.cu:3-5) made a
 * run-time argument; 0 = primary hit + direct light with shadow rays. */
int rt_render(rt_context *ctx, int n_samples, int max_depth);

/* Read back the accumulated image (the reference has no readback: its image
 * leaves through GL interop, kernel.cu:203-213).  Full-image, row-major, host
 * buffers; any may be NULL.  Pixels outside this context's partition read
 * rgba = 0, prim_id = -2, t = 0.
 *   rgba    [H][W][4] running mean, alpha 1   (surf2Dwrite, kernel.cu:77)
 *   prim_id [H][W]    primary-hit primitive of the last sample, -1 = miss
 *   t       [H][W]    its rec.t (INF = 1e6 on a miss) */
int rt_read_framebuffer(rt_context *ctx, float *rgba, int32_t *prim_id, float *t);

/* Partition-local pixels for the multi-GPU gather: count, their global pixel
 * indices (host int32[count]) and a device-to-device copy of their RGBA values
 * ([count][4] floats) into a caller-owned DEVICE buffer, in the same order. */
int rt_local_pixel_count(rt_context *ctx, int64_t *count);
int rt_local_pixel_indices(rt_context *ctx, int32_t *indices);
int rt_export_local_rgba(rt_context *ctx, void *device_dst);

/* Scene::hit on caller-supplied rays (Scene/Scene.cu:8-67); dir is normalised
 * as the Ray constructor does (Ray/Ray.cu:2).  Host buffers: org/dir [n][3];
 * t [n] (INF on a miss), prim_id [n] (-1), ishit [n] (the bool it returns). */
int rt_scene_hit(rt_context *ctx, int n, const float *org, const float *dir, float *t,
                 int32_t *prim_id, int32_t *ishit);

/* rt_scene_hit for callers that cannot allocate device memory.  It has no counterpart in the
 * reference; it is for tight serving loops and harnesses that forbid hipMalloc / hipFree.  It
 * allocates nothing: org / dir ([n][3]) and t / prim_id / ishit ([n]) are caller-owned DEVICE
 * buffers, and what rt_scene_hit allocates per call (the queue of normalised rays, the kernel's
 * chunk dispensers) lives in a static device array of RT_SCENE_HIT_NOALLOC_MAX_RAYS * 24 + 256
 * bytes that every GPU the library runs on holds until the process ends.  n above the maximum is
 * refused with RT_ERR_INVALID.  Results are rt_scene_hit's, bit for bit.  All calls on one GPU
 * (from any context) share that GPU's array: calls that overlap on one GPU race.
 * Ordering: the call launches on the context's own stream and synchronises it before it returns
 * (blocking, like every call here).  It takes no caller stream or event, so the work that
 * produces d_org / d_dir must be COMPLETE before the call (synchronise the producing stream
 * first), and it must not be called while a stream is being captured into a HIP graph: it
 * synchronises, and its launches are not on the capturing stream. */
#define RT_SCENE_HIT_NOALLOC_MAX_RAYS 1048576
int rt_scene_hit_noalloc(rt_context *ctx, int n, const float *d_org, const float *d_dir,
                         float *d_t, int32_t *d_prim_id, int32_t *d_ishit);

/* Post-process stage of the reference's graph (kernel.cu:250-251):
 * HDRKernel = ACES filmic tone map (PostProcess/PostProcess.cuh:29-56) and
 * filterKernel = 3x3 luminance median (PostProcess.cuh:59-104), applied to the
 * accumulated image of a full-image (world = 1) context.  Host output
 * [H][W][4]; either may be NULL. */
int rt_postprocess(rt_context *ctx, float *hdr_rgba, float *filtered_rgba);

/* No counterpart in the reference, whose Scene::hit is a plain scan ("Brute force
 * method", Scene/Scene.cu:7).  enabled = 1 (default): the closest-hit kernel skips
 * groups of triangles whose conservative bounding box no ray of the wave can hit;
 * the result is the scan's, bit for bit.  enabled = 0: every ray is tested
 * against every triangle -- the validation mode the parity tests compare against. */
int rt_set_culling(rt_context *ctx, int enabled);

/* Host-only view of what rt_upload_scene derives for the culling (no device needed; for
 * tests and tools): per triangle, in the reference's traversal order, the conservative
 * box of its acceptance region bounds[n_tris][6] = (lo.xyz, hi.xyz) and its position in
 * the device order; geom_extent = max |coordinate| of the geometry.  Any pointer may be NULL. */
int rt_host_bounds(const rt_scene_desc *scene, float *bounds, int32_t *position,
                   float *geom_extent);

int rt_get_stats(rt_context *ctx, rt_stats *stats);
int rt_reset_stats(rt_context *ctx);
/* 1: bracket every launch of the path kernel with HIP events on its own stream. */
int rt_set_timing(rt_context *ctx, int enabled);

const char *rt_last_error(const rt_context *ctx);

/* ---- host-side surface that stays as in the reference ---------------------- */

/* bool Mesh::readFile(const char *path)  (Model/Model.cu:139-284): OBJ with
 * v / vt / vn / f (a/b/c or a//c, triangles only, 1-based, no negative indices).
 * Allocates pos/nrm [n][3][3] and uv [n][3][2] with malloc; free with rt_obj_free. */
int rt_obj_load(const char *path, float **pos, float **nrm, float **uv, int32_t *n_tris);
void rt_obj_free(float *p);

/* Image output.  This snapshot of the reference displays through GL only
 * (.gitignore:2,43-47 shows other revisions wrote .ppm/.png); PPM (8-bit, after
 * saturate) and PFM (float) writers stand in for "image output". */
int rt_write_ppm(const char *path, int width, int height, const float *rgba);
int rt_write_pfm(const char *path, int width, int height, const float *rgba);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
