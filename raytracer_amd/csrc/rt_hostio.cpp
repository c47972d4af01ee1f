// rt_hostio.cpp -- the host-side surface that stays as in the reference: the OBJ
// model loader (Mesh::readFile, Model/Model.cu:139-284) and image output.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace {

struct F3 { float x, y, z; };
struct F2 { float x, y; };
struct Idx { int v, a, b; bool double_slash; int fields; };

// One face token "a/b/c", "a//c", "a/b" or "a" -> indices as the reference reads them
// (Model.cu:176-197): '/' becomes ' ', a doubled separator means "vertex//normal".
bool parse_face_token(const std::string &tok, Idx &out)
{
    std::string s = tok;
    out.double_slash = s.find("//") != std::string::npos;
    for (auto &c : s)
        if (c == '/') c = ' ';
    std::istringstream is(s);
    int vals[3] = {0, 0, 0};
    int n = 0;
    while (n < 3 && (is >> vals[n]))
        n++;
    out.fields = n;
    out.v = vals[0] - 1;
    out.a = vals[1] - 1;
    out.b = vals[2] - 1;
    return n >= 1;
}

} // namespace

extern "C" int rt_obj_load(const char *path, float **pos_out, float **nrm_out, float **uv_out,
                           int32_t *n_tris_out)
{
    if (!path || !pos_out || !nrm_out || !n_tris_out)
        return RT_ERR_INVALID;
    std::ifstream file(path);
    if (!file)                                        // Model.cu:146-148
        return RT_ERR_IO;
    std::vector<F3> positions, normals;
    std::vector<F2> texcoords;
    std::vector<std::vector<Idx>> faces;
    std::string line;
    while (std::getline(file, line)) {                // Model.cu:150-206
        if (line.compare(0, 2, "vt") == 0) {
            std::istringstream s(line.substr(2));
            F2 v{0, 0};
            s >> v.x; s >> v.y;
            texcoords.push_back(v);
        } else if (line.compare(0, 2, "vn") == 0) {
            std::istringstream s(line.substr(2));
            F3 v{0, 0, 0};
            s >> v.x; s >> v.y; s >> v.z;
            normals.push_back(v);
        } else if (line.compare(0, 1, "v") == 0) {
            std::istringstream s(line.substr(1));
            F3 v{0, 0, 0};
            s >> v.x; s >> v.y; s >> v.z;
            positions.push_back(v);
        } else if (line.compare(0, 1, "f") == 0) {
            std::vector<Idx> face;
            std::istringstream toks(line.substr(1));
            std::string tok;
            while (toks >> tok) {
                Idx ix;
                if (!parse_face_token(tok, ix))
                    return RT_ERR_INVALID;
                face.push_back(ix);
            }
            faces.push_back(face);
        }
    }
    if (faces.empty())                                // Model.cu:207-208
        return RT_ERR_INVALID;

    const size_t n = faces.size();
    float *pos = (float *)std::malloc(n * 9 * sizeof(float));
    float *nrm = (float *)std::malloc(n * 9 * sizeof(float));
    float *uv = (float *)std::malloc(n * 6 * sizeof(float));
    if (!pos || !nrm || !uv) {
        std::free(pos); std::free(nrm); std::free(uv);
        return RT_ERR_DEVICE;
    }
    std::memset(uv, 0, n * 6 * sizeof(float));
    int status = RT_OK;
    const bool has_vt = !texcoords.empty();
    for (size_t f = 0; f < n && status == RT_OK; f++) {
        const auto &face = faces[f];
        // The reference indexes fixed V[3]/N[3]/T[3] arrays by face arity (Model.cu:219-221,
        // 262-263): anything but a triangle is out of bounds there, an error here.
        if (face.size() != 3) { status = RT_ERR_UNSUPPORTED; break; }
        for (int k = 0; k < 3; k++) {
            const Idx &ix = face[k];
            int iv = ix.v, in, it = -1;
            if (has_vt) {                             // Model.cu:221-245
                if (!ix.double_slash && ix.fields == 3) { in = ix.b; it = ix.a; }
                else { in = ix.a; }                   // "a//c": the normal index sits second
            } else {                                  // Model.cu:263-273
                in = ix.a;
            }
            // H6: faces must carry normals; the reference reads uninitialised memory otherwise
            if (iv < 0 || iv >= (int)positions.size() || in < 0 || in >= (int)normals.size() ||
                (it >= (int)texcoords.size())) {
                status = RT_ERR_INVALID;
                break;
            }
            float *p = pos + f * 9 + 3 * k, *q = nrm + f * 9 + 3 * k;
            p[0] = positions[iv].x; p[1] = positions[iv].y; p[2] = positions[iv].z;
            q[0] = normals[in].x; q[1] = normals[in].y; q[2] = normals[in].z;
            if (it >= 0) { uv[f * 6 + 2 * k] = texcoords[it].x; uv[f * 6 + 2 * k + 1] = texcoords[it].y; }
        }
    }
    if (status != RT_OK) {
        std::free(pos); std::free(nrm); std::free(uv);
        return status;
    }
    *pos_out = pos;
    *nrm_out = nrm;
    if (uv_out) *uv_out = uv; else std::free(uv);
    *n_tris_out = (int32_t)n;                         // Model.cu:282 number = faces.size()
    return RT_OK;
}

extern "C" void rt_obj_free(float *p) { std::free(p); }

extern "C" int rt_write_ppm(const char *path, int width, int height, const float *rgba)
{
    if (!path || !rgba || width <= 0 || height <= 0)
        return RT_ERR_INVALID;
    FILE *f = std::fopen(path, "wb");
    if (!f)
        return RT_ERR_IO;
    std::fprintf(f, "P6\n%d %d\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    // image row 0 is the bottom scan line in the reference's GL display (texture.vert);
    // files are written top-down.
    for (int y = height - 1; y >= 0; y--) {
        for (int x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) {
                float v = rgba[((size_t)y * width + x) * 4 + c];
                v = std::fmin(1.0f, std::fmax(0.0f, v));
                row[(size_t)x * 3 + c] = (unsigned char)(v * 255.0f + 0.5f);
            }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return RT_OK;
}

extern "C" int rt_write_pfm(const char *path, int width, int height, const float *rgba)
{
    if (!path || !rgba || width <= 0 || height <= 0)
        return RT_ERR_INVALID;
    FILE *f = std::fopen(path, "wb");
    if (!f)
        return RT_ERR_IO;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);   // little-endian, bottom-up rows
    std::vector<float> row((size_t)width * 3);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++)
            for (int c = 0; c < 3; c++)
                row[(size_t)x * 3 + c] = rgba[((size_t)y * width + x) * 4 + c];
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
    return RT_OK;
}
