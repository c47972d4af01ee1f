// rt_upload.h -- host-side construction of the device scene records.
#pragma once

#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_math.h"

namespace rt {

struct HostScene {
    std::vector<TriRec> tris;          // tree order, padded to a multiple of kTreeArity
    std::vector<uint32_t> pos_of;      // reference traversal index -> position in `tris`
    std::vector<TriCold> cold;         // reference traversal order
    // Implicit 4-ary box tree over `tris`: level 0 = one box per triangle, level l+1 =
    // one per four boxes of level l, stored four to a NodeGroup; levels are padded
    // with dead boxes; the top level is a single group.
    std::vector<NodeGroup> groups;     // all levels back to back
    std::vector<size_t> level_off;     // first group of level l in `groups`
    std::vector<uint32_t> level_cnt;   // number of groups of level l
    int top = 0;                       // index of the top level
    float geom_extent = 1.0f;          // max |coordinate| of the geometry
    std::vector<ModelRec> models;
    std::vector<SphereRec> spheres;
    std::vector<LightRec> lights;
    // EnvironmentLight tables (EnvView): rows back to back
    int env_w = 0, env_h = 0, env_n = 0;
    std::vector<float> env_img, env_cond_value, env_cond_cdf, env_cond_funint, env_marg_value, env_marg_cdf;
    float env_marg_funint = 0.0f;
    std::vector<MatRec> mats;            // one per ModelRec
    bool general = false;                // non-Lambertian materials or triangle lights
    std::vector<TriLightRec> tlights;    // triangle lights (their Materials: models[n_models + k])
    // mesh textures: level 0 of every image back to back and where each chain will lie (TexRec); the
    // other levels are generated on the device (k_mip_level), as the reference does (Texture.cu:48-133)
    std::vector<TexRec> textures;
    std::vector<float> tex_level0;       // [sum w h][4]
    std::vector<uint32_t> tex_level0_at; // float4 offset of each image inside tex_level0
    size_t tex_texels = 0;               // float4 texels of all chains together
    std::vector<int32_t> mesh_tex;       // [n_meshes]
    std::vector<TriUv> uv;               // [n_tris], only with textures
    int n_tri_models = 0;
};

// Scene::initializeScene (Scene/Scene.cu:126-202) on the host: validates the flat
// description and evaluates the ray-independent parts of Triangle::hit /
// Quadratic::hit with the reference's operation order.  Returns an RT_* status.
int build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err);

// Camera::Camera (Camera/Camera.cu:4-14).
CameraRec make_camera(const rt_camera_desc *c);

} // namespace rt
