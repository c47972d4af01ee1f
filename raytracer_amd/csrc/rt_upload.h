// rt_upload.h -- host-side construction of the device scene records.
#pragma once

#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_math.h"

namespace rt {

struct HostScene {
    std::vector<TriRec> tris;
    std::vector<TriCold> cold;
    std::vector<ModelRec> models;
    std::vector<SphereRec> spheres;
    std::vector<LightRec> lights;
};

// Scene::initializeScene (Scene/Scene.cu:126-202) on the host: validates the flat
// description and evaluates the ray-independent parts of Triangle::hit /
// Quadratic::hit with the reference's operation order.  Returns an RT_* status.
int build_host_scene(const rt_scene_desc *d, HostScene &out, std::string &err);

// Camera::Camera (Camera/Camera.cu:4-14).
CameraRec make_camera(const rt_camera_desc *c);

} // namespace rt
