/*
 * rt_oracle.c -- CPU restatement (plain C) of UI233/RayTracer's per-pixel render
 * path.  TEST INFRASTRUCTURE ONLY; see rt_oracle.h for the pinning statement.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/RayTracer/).  Nothing is hoisted or reordered: per-ray
 * vertex transforms, per-ray matrix inverses and by-value copies are restated as
 * the reference performs them, so that this file can check the product's
 * upload-time precomputation bit for bit.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see Makefile).
 * -ffp-contract=off because the surveyed host build of the reference used it
 * (SURVEY.md section 8(c)); rsqrtf(x) is 1.0f/sqrtf(x) as in helper_math.h:62-65.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- vectors */
typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { float v[4][4]; } m4;

static inline f3 mk3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f4 mk4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
/* helper_math.h:349,579,804,814,997 */
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
static inline f3 div3s(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
/* helper_math.h:1248,1252 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot4(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
/* helper_math.h:1420 */
static inline f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* helper_math.h:62-65 (host rsqrtf), :1291, :1309-1318 */
static inline float rsqrt_host(float x) { return 1.0f / sqrtf(x); }
static inline float length3(f3 v) { return sqrtf(dot3(v, v)); }
static inline f3 normalize3(f3 v) { float inv = rsqrt_host(dot3(v, v)); return scale3(v, inv); }
static inline f4 normalize4(f4 v)
{
    float inv = rsqrt_host(dot4(v, v));
    return mk4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

/* ---------------------------------------------------------------- mat4 */
/* Matrix/Matrix.cu:73-83 */
static f4 m4_apply4(const m4 *m, f4 a)
{
    f4 r;
    r.x = m->v[0][0] * a.x + m->v[0][1] * a.y + m->v[0][2] * a.z + m->v[0][3] * a.w;
    r.y = m->v[1][0] * a.x + m->v[1][1] * a.y + m->v[1][2] * a.z + m->v[1][3] * a.w;
    r.z = m->v[2][0] * a.x + m->v[2][1] * a.y + m->v[2][2] * a.z + m->v[2][3] * a.w;
    r.w = m->v[3][0] * a.x + m->v[3][1] * a.y + m->v[3][2] * a.z + m->v[3][3] * a.w;
    return r;
}
/* Matrix/Matrix.cu:85-91 */
static f3 m4_apply3(const m4 *m, f3 a)
{
    f4 r = m4_apply4(m, mk4(a.x, a.y, a.z, 1.0f));
    return mk3(r.x / r.w, r.y / r.w, r.z / r.w);
}
/* Matrix/Matrix.cu:235-243 */
static m4 m4_transpose(const m4 *m)
{
    m4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.v[i][j] = m->v[j][i];
    return r;
}
/* Matrix/Matrix.cu:181-233 (note factor11 repeats factor07's expression) */
static m4 m4_inverse(const m4 *mm)
{
    const float (*m)[4] = mm->v;
    float factor00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float factor01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float factor02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float factor03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float factor04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float factor05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float factor06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float factor07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float factor08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float factor09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float factor10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float factor11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float factor12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float factor13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float factor14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float factor15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float factor16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float factor17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float factor18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];

    m4 I;
    I.v[0][0] = +(m[1][1] * factor00 - m[1][2] * factor01 + m[1][3] * factor02);
    I.v[0][1] = -(m[1][0] * factor00 - m[1][2] * factor03 + m[1][3] * factor04);
    I.v[0][2] = +(m[1][0] * factor01 - m[1][1] * factor03 + m[1][3] * factor05);
    I.v[0][3] = -(m[1][0] * factor02 - m[1][1] * factor04 + m[1][2] * factor05);

    I.v[1][0] = -(m[0][1] * factor00 - m[0][2] * factor01 + m[0][3] * factor02);
    I.v[1][1] = +(m[0][0] * factor00 - m[0][2] * factor03 + m[0][3] * factor04);
    I.v[1][2] = -(m[0][0] * factor01 - m[0][1] * factor03 + m[0][3] * factor05);
    I.v[1][3] = +(m[0][0] * factor02 - m[0][1] * factor04 + m[0][2] * factor05);

    I.v[2][0] = +(m[0][1] * factor06 - m[0][2] * factor07 + m[0][3] * factor08);
    I.v[2][1] = -(m[0][0] * factor06 - m[0][2] * factor09 + m[0][3] * factor10);
    I.v[2][2] = +(m[0][0] * factor11 - m[0][1] * factor09 + m[0][3] * factor12);
    I.v[2][3] = -(m[0][0] * factor08 - m[0][1] * factor10 + m[0][2] * factor12);

    I.v[3][0] = -(m[0][1] * factor13 - m[0][2] * factor14 + m[0][3] * factor15);
    I.v[3][1] = +(m[0][0] * factor13 - m[0][2] * factor16 + m[0][3] * factor17);
    I.v[3][2] = -(m[0][0] * factor14 - m[0][1] * factor16 + m[0][3] * factor18);
    I.v[3][3] = +(m[0][0] * factor15 - m[0][1] * factor17 + m[0][2] * factor18);

    float determinant = +m[0][0] * I.v[0][0] + m[0][1] * I.v[0][1] + m[0][2] * I.v[0][2] +
                        m[0][3] * I.v[0][3];

    /* Inverse *= 1.0f / determinant  (Matrix.cu:43-51, 230) */
    float b = 1.0f / determinant;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            I.v[i][j] *= b;

    return m4_transpose(&I);
}

/* ---------------------------------------------------------------- Ray */
/* Ray/Ray.cuh:13-28, Ray/Ray.cu:2-7: the constructor normalises. */
typedef struct { f3 o, d; } ray_t;
static inline ray_t mk_ray(f3 o, f3 r) { ray_t q; q.o = o; q.d = normalize3(r); return q; }
static inline f3 ray_pos(const ray_t *r, float t) { return add3(r->o, scale3(r->d, t)); }

/* ---------------------------------------------------------------- constants */
#define ORC_INF 1000000.0f            /* Ray/IntersectionRecord.cuh:7-9 */
#define ORC_FLOAT_EPS (1e-2f)         /* Model/Model.cuh:24-26 FLOAT_EPISLON */
#define ORC_INV_PI 0.318309892f       /* Material/material.cu:5 */
#define ORC_PI_OVER4 0.78539816339744830961f /* material.cu:7 */
#define ORC_PI_OVER2 1.57079632679489661923f /* material.cu:8 */
#define ORC_MAT_LAMBERTIAN 1          /* Material/Object.cuh:15 */
#define ORC_MAT_FRESNEL 7             /* Object.cuh:16 */
#define ORC_MAT_COOK 9                /* Object.cuh:17 */
#define ORC_MAT_OREN_NAYAR 25         /* Object.cuh:18 */
#define ORC_MAT_GGX 17                /* Object.cuh:19 */
#define ORC_PI_D 3.14159265358979323846   /* M_PI */
#define ORC_MAT_SPECULAR_BIT 4        /* Material/Object.cuh:10 */

/* ---------------------------------------------------------------- record */
/* Ray/IntersectionRecord.cuh:18-36; constructor defaults :21 */
typedef struct {
    f3 pos, normal, tangent;
    f3 color;
    float t;
    ray_t wo;
    int lightidx;
    int light_type;
    float pdf_light, pdf_surface;
    int material;      /* model index standing in for Material*; n_models + k: triangle light k's */
    int isLight;
    int material_type;
    int prim;          /* not in the reference: primitive index, for the id buffer */
} rec_t;

static void rec_init(rec_t *r)
{
    memset(r, 0, sizeof(*r));
    r->t = ORC_INF;
    r->isLight = 0;
    r->lightidx = -1;
    r->color = mk3(-1.0f, -1.0f, -1.0f);
    r->material = -1;
    r->prim = -1;
}

static m4 load_m4(const float *p)
{
    m4 m;
    memcpy(m.v, p, sizeof(m.v));
    return m;
}

/* ---------------------------------------------------------------- Triangle */
/* Model/Model.cu:413-420 Triangle::area */
static float tri_area(const m4 *M, const f3 pos[3])
{
    f3 rpos[3];
    for (int i = 0; i < 3; i++)
        rpos[i] = m4_apply3(M, pos[i]);
    return 0.5f * length3(cross3(sub3(rpos[2], rpos[0]), sub3(rpos[1], rpos[0])));
}

/* Model/Model.cu:91-137 Triangle::hit.  g = primitive index in traversal order. */
static int tri_hit(const orc_scene *s, int g, const ray_t *r, rec_t *c)
{
    const float *pp = s->tri_pos + 9 * (size_t)g;
    const float *nn = s->tri_nrm + 9 * (size_t)g;
    int model = s->tri_model[g];
    m4 M = load_m4(s->model_xf + 16 * (size_t)model);
    f3 pos[3] = { mk3(pp[0], pp[1], pp[2]), mk3(pp[3], pp[4], pp[5]), mk3(pp[6], pp[7], pp[8]) };
    f3 nrm[3] = { mk3(nn[0], nn[1], nn[2]), mk3(nn[3], nn[4], nn[5]), mk3(nn[6], nn[7], nn[8]) };

    f3 ta = m4_apply3(&M, pos[0]), tb = m4_apply3(&M, pos[1]), tc = m4_apply3(&M, pos[2]); /* :94 */

    float t;
    f3 norma = cross3(sub3(ta, tc), sub3(tb, tc));                 /* :97 */
    float dot_normal_dir = dot3(norma, r->d);                       /* :98 */
    if (fabsf(dot_normal_dir) < ORC_FLOAT_EPS)                      /* :99 */
        return 0;

    t = (-dot3(r->o, norma) + dot3(ta, norma)) / dot_normal_dir;    /* :102 */

    f3 p = ray_pos(r, t);                                           /* :104 */

    float S = 2.0f * tri_area(&M, pos);                             /* :106 */
    float s1 = length3(cross3(sub3(p, ta), sub3(p, tb)));           /* :107 */
    float s2 = length3(cross3(sub3(p, tc), sub3(p, ta)));           /* :108 */
    float s3 = length3(cross3(sub3(p, tc), sub3(p, tb)));           /* :109 */

    if (fabsf(s1 + s2 + s3 - S) > 0.001f)                           /* :113 */
        return 0;

    float m1 = s3 / S, m2 = s2 / S, m3 = 1.0f - m1 - m2;            /* :116 */

    if (t > ORC_FLOAT_EPS && t < c->t) {                            /* :120 */
        c->material = model;                                        /* :122 */
        c->material_type = s->model_mat_type[model];                /* :123 */
        c->t = t;
        c->normal = normalize3(add3(add3(scale3(nrm[0], m1), scale3(nrm[1], m2)),
                                    scale3(nrm[2], m3)));           /* :125 */
        m4 inv = m4_inverse(&M);
        m4 cas = m4_transpose(&inv);                                /* :126 */
        f4 tmp = normalize4(m4_apply4(&cas, mk4(c->normal.x, c->normal.y, c->normal.z, 0.0f)));
        c->normal = mk3(tmp.x, tmp.y, tmp.z);                       /* :127-128 */
        c->pos = ray_pos(r, t);                                     /* :129 */
        c->isLight = 0;
        /* make_float3(0.3, 0.4, -0.5): double literals narrowed to float (:131) */
        c->tangent = normalize3(cross3(c->normal, mk3((float)0.3, (float)0.4, (float)-0.5)));
        c->prim = g;
        return 1;
    }
    return 0;
}

/* ---------------------------------------------------------------- Quadratic */
/* Model/Model.cu:341-346 getRadius (type Sphere = 0) */
static float quad_radius(const orc_scene *s, int q)
{
    if (s->quad_type[q] == 0)
        return (1 / s->quad_coeff[3 * q + 0]);
    return 0.0f;
}

/* Model/Model.cu:349-410 Quadratic::hit (sphere only). */
static int quad_hit(const orc_scene *s, int q, const ray_t *r, rec_t *c)
{
    if (s->quad_type[q] != 0)
        return 0;
    int model = s->n_tri_models + s->n_meshes + q;
    m4 M = load_m4(s->model_xf + 16 * (size_t)model);

    f3 center = mk3(0.0f, 0.0f, 0.0f);
    m4 inv = m4_inverse(&M);                                        /* :352 */
    f3 Lorigin = m4_apply3(&inv, r->o);                             /* :353 */
    f4 L4dir = m4_apply4(&inv, mk4(r->d.x, r->d.y, r->d.z, 0));     /* :355 */
    f3 Ldir = normalize3(mk3(L4dir.x, L4dir.y, L4dir.z));           /* :356 */
    f3 oc = sub3(Lorigin, center);                                  /* :357 */
    float dotOCD = dot3(Ldir, oc);                                  /* :358 */

    if (dotOCD > 0)                                                 /* :360 */
        return 0;

    float dotOC = dot3(oc, oc);
    float rad = quad_radius(s, q);
    float discriminant = dotOCD * dotOCD - dotOC + rad * rad;       /* :364 */
    float t0, t1;
    if (discriminant < 0)
        return 0;
    else if (discriminant < ORC_FLOAT_EPS)
        t0 = t1 = -dotOCD;
    else {
        discriminant = sqrtf(discriminant);
        t0 = -dotOCD - discriminant;
        t1 = -dotOCD + discriminant;
        if (t0 < ORC_FLOAT_EPS)
            t0 = t1;
    }

    f3 pos = add3(scale3(Ldir, t0), Lorigin);                       /* :382 */
    f4 n4 = m4_apply4(&M, mk4(pos.x, pos.y, pos.z, 0.0f));
    f3 normal = normalize3(mk3(n4.x, n4.y, n4.z));                  /* :383 */
    f3 tangent;
    if (pos.x == 0.0f && pos.y == 0.0f) {                           /* :385-386 */
        f4 t4 = m4_apply4(&M, mk4(0.0f, 0.0f, -1.0f, 0.0f));
        tangent = mk3(t4.x, t4.y, t4.z);
    } else {                                                        /* :387-391, UP_VEC :2 */
        f3 up = mk3(0.0f, rad, 0.0f);
        f3 d = sub3(up, pos);
        f4 u4 = m4_apply4(&M, mk4(d.x, d.y, d.z, 0.0f));
        tangent = cross3(normal, cross3(mk3(u4.x, u4.y, u4.z), normal));
    }
    tangent = normalize3(tangent);                                  /* :393 */
    pos = m4_apply3(&M, pos);                                       /* :394 */

    if (t0 > ORC_FLOAT_EPS && t0 < c->t) {                          /* :396 */
        c->material = model;
        c->material_type = s->model_mat_type[model];
        c->t = t0;
        c->pos = pos;
        c->normal = normal;
        c->tangent = normalize3(tangent);                           /* :404 */
        c->isLight = 0;
        c->prim = s->n_tris + q;
        return 1;
    }
    return 0;
}

/* ---------------------------------------------------------------- TriangleLight */
#define ORC_LIGHT_TRIANGLE 2          /* Light/Light.cuh:16-23 light::TRIANGLE_LIGHT */
typedef struct { f3 pos[3], normal, illum; int two_side; } tl_t;
/* Light/Light.cu:35-40: the constructor normalises the member normal. */
static tl_t tl_load(const orc_scene *s, int k)
{
    const float *pp = s->tl_pos + 9 * (size_t)k, *li = s->tl_illum + 3 * (size_t)k;
    tl_t l;
    l.pos[0] = mk3(pp[0], pp[1], pp[2]);
    l.pos[1] = mk3(pp[3], pp[4], pp[5]);
    l.pos[2] = mk3(pp[6], pp[7], pp[8]);
    l.normal = normalize3(cross3(sub3(l.pos[1], l.pos[0]), sub3(l.pos[2], l.pos[0])));
    l.illum = mk3(li[0], li[1], li[2]);
    l.two_side = s->tl_two_side[k] != 0;
    return l;
}
/* Light/Light.cuh:76-79 (0.5 is a double literal: the halving is exact either way) */
static float tl_area(const tl_t *l)
{
    return (float)(0.5 * (double)length3(cross3(sub3(l->pos[2], l->pos[0]), sub3(l->pos[1], l->pos[0]))));
}
/* Light/Light.cuh:81-85 */
static float tl_pdf(const tl_t *l, float rec_t, f3 wi)
{
    if (fabsf(dot3(l->normal, wi)) < 0.001f)
        return 0.0f;
    return rec_t * rec_t / (tl_area(l) * fabsf(dot3(l->normal, wi)));
}
/* Light/Light.cuh:86-91 */
static f3 tl_L(const tl_t *l, f3 r)
{
    if (l->two_side)
        return l->illum;
    return dot3(r, l->normal) > 0 ? l->illum : mk3(0.0f, 0.0f, 0.0f);
}
/* Light/Light.cu:68-105 TriangleLight::hit.  The local `normal` of :72 shadows the member: the
 * record's normal is the UNNORMALISED cross product.  The light's own Material is model index
 * n_models + k here (Scene.cu:151-153 hands it material_type[] / Material[] entry n_models + k). */
static int tl_hit(const orc_scene *s, int k, const ray_t *r, rec_t *rec)
{
    tl_t l = tl_load(s, k);
    float t;
    f3 normal = cross3(sub3(l.pos[0], l.pos[2]), sub3(l.pos[1], l.pos[2]));   /* :72 */
    float dot_normal_dir = dot3(normal, r->d);                      /* :73 */
    if (fabsf(dot_normal_dir) < ORC_FLOAT_EPS)                      /* :74 */
        return 0;
    t = (-dot3(r->o, normal) + dot3(l.pos[0], normal)) / dot_normal_dir;      /* :77 */
    f3 rpos = ray_pos(r, t);                                        /* :79 */
    float S = 2.0f * tl_area(&l);                                   /* :81 */
    float s1 = length3(cross3(sub3(rpos, l.pos[0]), sub3(rpos, l.pos[1])));   /* :82 */
    float s2 = length3(cross3(sub3(rpos, l.pos[2]), sub3(rpos, l.pos[0])));   /* :83 */
    float s3 = length3(cross3(sub3(rpos, l.pos[2]), sub3(rpos, l.pos[1])));   /* :84 */
    if (fabsf(s1 + s2 + s3 - S) > 0.001f)                           /* :87 */
        return 0;
    float m1 = s3 / S, m2 = s2 / S, m3 = 1.0f - m1 - m2;            /* :90 */
    if (t > ORC_FLOAT_EPS && t < rec->t) {                          /* :92 */
        int n_models = s->n_tri_models + s->n_meshes + s->n_quads;
        rec->material = n_models + k;                               /* :94 */
        rec->material_type = s->tl_mat_type[k];                     /* :95 */
        rec->t = t;
        rec->normal = l.two_side ? (dot3(normal, r->d) > 0 ? neg3(normal) : normal) : normal; /* :97 */
        rec->pos = ray_pos(r, t);                                   /* :98 */
        rec->isLight = 1;                                           /* :99 */
        rec->tangent = normalize3(add3(scale3(sub3(l.pos[1], l.pos[0]), m2),
                                       scale3(sub3(l.pos[2], l.pos[0]), m3)));  /* :100 */
        rec->prim = s->n_tris + s->n_quads + k;
        return 1;
    }
    return 0;
}

/* ---------------------------------------------------------------- Scene::hit */
static f3 mesh_texture_color(const orc_scene *s, int mesh, int g, const rec_t *rec);
/* Scene/Scene.cu:8-67 with Mesh::hit (Model/Model.cu:287-313) inlined. */
static int scene_hit(const orc_scene *s, const ray_t *r, rec_t *rec)
{
    for (int j = 0; j < s->n_tri_models; j++)                       /* :17-23 */
        tri_hit(s, j, r, rec);
    for (int j = 0; j < s->n_meshes; j++) {                         /* :24-30 */
        int ishit = 0, idx = -1;
        int first = s->mesh_first[j], number = s->mesh_count[j];
        for (int i = 0; i < number; i++)                            /* Model.cu:291-298 */
            if (tri_hit(s, first + i, r, rec)) {
                ishit = 1;
                idx = first + i;
            }
        if (ishit) {                                                /* Model.cu:301-310 */
            int model = s->n_tri_models + j;
            rec->material = model;
            rec->material_type = s->model_mat_type[model];
            /* :304-309.  H5: a mesh whose Texture was never bound yields (-1,-1,-1). */
            rec->color = mesh_texture_color(s, j, idx, rec);
        }
    }
    for (int j = 0; j < s->n_quads; j++)                            /* :31-37 */
        quad_hit(s, j, r, rec);
    for (int j = 0; j < s->n_tri_lights; j++)                       /* :43-58 */
        if (tl_hit(s, j, r, rec)) {
            rec->lightidx = j;                                      /* :54 */
            rec->light_type = ORC_LIGHT_TRIANGLE;                   /* :55 */
        }
    return rec->t > 0.001f && rec->t < ORC_INF - 0.1f;              /* :66 */
}

/* ---------------------------------------------------------------- spawnRay */
/* Ray/IntersectionRecord.cu:2-32: findFloatUp/Down do not change the value
 * (`*t++` bumps the pointer) except for the zero mappings. */
static float find_float_up(float v)
{
    if (v > 0.0f && v == INFINITY) return v;
    if (v == -0.0f) return +0.0f;
    return v;
}
static float find_float_down(float v)
{
    if (v < 0.0f && v == INFINITY) return v;
    if (v == +0.0f) return -0.0f;
    return v;
}
/* Ray/IntersectionRecord.cu:34-49 */
static f3 offset_from_point(f3 origin, f3 normal, f3 error_bound, f3 d)
{
    f3 n = mk3(fabsf(normal.x), fabsf(normal.y), fabsf(normal.z));
    f3 offset = scale3(normal, dot3(n, error_bound));
    if (dot3(d, normal) < 0)
        offset = neg3(offset);
    f3 po = add3(origin, offset);
    po.x = po.x > 0.0f ? find_float_up(po.x) : find_float_down(po.x);
    po.y = po.y > 0.0f ? find_float_up(po.y) : find_float_down(po.y);
    po.z = po.z > 0.0f ? find_float_up(po.z) : find_float_down(po.z);
    return po;
}
/* Ray/IntersectionRecord.cuh:38-42 */
static ray_t rec_spawn_ray(const rec_t *rec, f3 w)
{
    f3 ro = offset_from_point(rec->pos, rec->normal, mk3(0.01f, 0.01f, 0.01f), w);
    return mk_ray(ro, w);
}

/* ---------------------------------------------------------------- Material */
/* Material/Object.cu:107-115 */
static f3 world2local(f3 normal, f3 tangent, f3 w)
{
    return mk3(dot3(w, tangent), dot3(w, normal), dot3(w, normalize3(cross3(tangent, normal))));
}
static f3 local2world(f3 normal, f3 tangent, f3 l)
{
    return add3(add3(scale3(tangent, l.x), scale3(normal, l.y)),
                scale3(normalize3(cross3(tangent, normal)), l.z));
}

/* sinf/cosf of material.cu:46 are whatever libm the reference is built against
 * (CUDA libdevice on the GPU, glibc as host code): not bit-specified.  The build
 * fixes them as the float rounding of a double-precision Taylor evaluation made
 * only of IEEE double mul/add/sub, so CPU and GPU agree bit for bit.  The
 * argument range in sampleHemi is (-pi/4, 3pi/4].  tests/ checks <= 1 ulp
 * against glibc sinf/cosf over that range. */
static double sin_poly(double x)
{
    double x2 = x * x;
    double p = -7.6471637318198164759e-13;            /* -1/15! */
    p = p * x2 + 1.6059043836821614599e-10;           /*  1/13! */
    p = p * x2 + -2.5052108385441718775e-08;          /* -1/11! */
    p = p * x2 + 2.7557319223985890653e-06;           /*  1/9!  */
    p = p * x2 + -1.9841269841269841270e-04;          /* -1/7!  */
    p = p * x2 + 8.3333333333333333333e-03;           /*  1/5!  */
    p = p * x2 + -1.6666666666666666667e-01;          /* -1/3!  */
    return x + (x * x2) * p;
}
static double cos_poly(double x)
{
    double x2 = x * x;
    double p = 4.7794773323873852974e-14;             /*  1/16! */
    p = p * x2 + -1.1470745597729724714e-11;          /* -1/14! */
    p = p * x2 + 2.0876756987868098979e-09;           /*  1/12! */
    p = p * x2 + -2.7557319223985890653e-07;          /* -1/10! */
    p = p * x2 + 2.4801587301587301587e-05;           /*  1/8!  */
    p = p * x2 + -1.3888888888888888889e-03;          /* -1/6!  */
    p = p * x2 + 4.1666666666666666667e-02;           /*  1/4!  */
    p = p * x2 + -0.5;
    return 1.0 + x2 * p;
}
#define ORC_PI_2_D 1.5707963267948966192
float orc_sinf(float x)
{
    double xd = (double)x;
    if (xd > 0.78539816339744830962)
        return (float)cos_poly(xd - ORC_PI_2_D);      /* sin(x) = cos(x - pi/2) */
    return (float)sin_poly(xd);
}
float orc_cosf(float x)
{
    double xd = (double)x;
    if (xd > 0.78539816339744830962)
        return (float)(-sin_poly(xd - ORC_PI_2_D));   /* cos(x) = -sin(x - pi/2) */
    return (float)cos_poly(xd);
}

/* Material/material.cu:28-50 sampleHemi */
static f3 sample_hemi(f2 sample)
{
    f2 rs;
    rs.x = 2.0f * sample.x - 1.0f;
    rs.y = 2.0f * sample.y - 1.0f;
    if (rs.x == 0.0f && rs.y == 0.0f)
        return mk3(0.0f, 0.0f, 0.0f);
    f2 d;
    float r, theta;
    if (fabsf(rs.x) > fabsf(rs.y)) {
        r = rs.x;
        theta = ORC_PI_OVER4 * (rs.y / rs.x);
    } else {
        r = rs.y;
        theta = ORC_PI_OVER2 - ORC_PI_OVER4 * (rs.x / rs.y);
    }
    d.x = r * orc_sinf(theta);
    d.y = r * orc_cosf(theta);
    float z = sqrtf(fmaxf(0.0f, 1 - d.x * d.x - d.y * d.y));
    return mk3(d.x, z, d.y);
}

/* Transcendentals of the non-Lambertian BRDFs (material.cu:84-254, material.cuh:7-13,108-164):
 * acosf / sinf / cosf / logf / powf are whatever libm the reference is built against -- not
 * bit-specified.  The build fixes each as the float rounding of the double-precision function, on
 * the CPU here and on the GPU alike (glibc's and the device library's double results differ, if at
 * all, far below a float ulp). */
static float m_acosf(float x) { return (float)acos((double)x); }
static float m_sinf(float x) { return (float)sin((double)x); }
static float m_cosf(float x) { return (float)cos((double)x); }
static float m_logf(float x) { return (float)log((double)x); }
static float m_powf(float x, float y) { return (float)pow((double)x, (double)y); }

/* material.cuh:7-13 rough2alpha */
float orc_rough2alpha(float rough)
{
    rough = fmaxf(rough, 1e-3f);
    float x = m_logf(rough);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x +
           0.000640711f * x * x * x * x;
}

/* The BRDF object behind a Material, as its constructor leaves it (material.cuh:32,48,74,88,112). */
typedef struct { int type; f3 color; float a2, m, eta; } brdf_t;
static f3 model_color(const orc_scene *s, int model)
{
    int n_models = s->n_tri_models + s->n_meshes + s->n_quads;
    const float *c = model < n_models ? s->model_color + 3 * (size_t)model
                                      : s->tl_color + 3 * (size_t)(model - n_models);
    return mk3(c[0], c[1], c[2]);
}
static brdf_t brdf_of(const orc_scene *s, int model, int type)
{
    int n_models = s->n_tri_models + s->n_meshes + s->n_quads;
    brdf_t b;
    b.type = type;
    b.color = model_color(s, model);
    float rough = (model < n_models && s->model_param) ? s->model_param[2 * (size_t)model] : 0.0f;
    b.eta = (model < n_models && s->model_param) ? s->model_param[2 * (size_t)model + 1] : 0.0f;
    b.m = orc_rough2alpha(rough);                       /* material.cuh:88 */
    b.a2 = b.m * b.m;                                   /* material.cuh:48,88 */
    return b;
}

/* Material/material.cu:52-57 Lambertian::f */
static f3 lambert_f(f3 wo, f3 wi, f3 c, f3 brdf)
{
    if (c.x < 0.0f)
        c = brdf;
    return (wo.y > 0.0f && wi.y > 0.0f) ? scale3(c, ORC_INV_PI) : mk3(0.0f, 0.0f, 0.0f);
}
/* Material/material.cu:79-82 */
static float lambert_pdf(f3 wo, f3 wi)
{
    (void)wo;
    return wi.y > +0.0f ? wi.y * ORC_INV_PI : 0.0f;
}

/* material.cu:85-90 GGX::SchlickFresnel = material.cuh:96-101 Cook_Torrance::F */
static f3 schlick(f3 r0, float r)
{
    float t = (1.0f - r);
    f3 a = mk3(1.0f - r0.x, 1.0f - r0.y, 1.0f - r0.z);
    a = scale3(scale3(scale3(a, t * t), t * t), t);
    return add3(r0, a);
}
/* material.cu:92-98 GGX::G = :192-198 Cook_Torrance::G */
static float smith_g(float a2, f3 wi, f3 wo)
{
    float A = wo.y * sqrtf(a2 + (1.0f - a2) * wi.y * wi.y);
    float B = wi.y * sqrtf(a2 + (1.0f - a2) * wo.y * wo.y);
    return 2.0f * wo.y * wi.y / (A + B);
}
/* material.cu:100-103 GGX::D (M_PI makes the expression double) */
static float ggx_d(float a2, f3 wh)
{
    float k = wh.y * wh.y * (a2 - 1) + 1;
    return (float)((double)a2 / (ORC_PI_D * (double)k * (double)k));
}
/* material.cu:105-114 */
static f3 ggx_f(const brdf_t *b, f3 wo, f3 wi, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
    float dotwin_won = fmaxf(0.0001f, fabsf(4.0f * wi.y * wo.y));
    f3 wh = normalize3(add3(wo, wi));
    return div3s(scale3(scale3(schlick(c, dot3(wi, wh)), ggx_d(b->a2, wh)), smith_g(b->a2, wi, wo)),
                 dotwin_won);
}
/* material.cu:132-137 */
static float ggx_pdf(const brdf_t *b, f3 wo, f3 wi)
{
    f3 wh = normalize3(add3(wo, wi));
    return wi.y > 0.0f && dot3(wi, wh) > 0.0f ? ggx_d(b->a2, wh) * wh.y * 0.25f / wo.y : 0.0f;
}
/* material.cu:116-130 */
static f3 ggx_sample_f(const brdf_t *b, f3 wo, f3 *wi, float *pdf, f2 sample, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
    float theta = m_acosf(sqrtf((1.0f - sample.x) / ((b->a2 - 1.0f) * sample.x + 1.0f)));
    float phi = (float)((double)sample.y * ORC_PI_D * (double)2.0f);
    f3 wh = mk3(m_sinf(theta) * m_sinf(phi), m_cosf(theta), m_cosf(phi) * m_sinf(theta));
    *wi = sub3(scale3(wh, 2.0f * dot3(wo, wh)), wo);
    *pdf = ggx_pdf(b, wo, *wi);
    return ggx_f(b, wo, *wi, c);
}

static float clampf(float v, float lo, float hi) { return v < lo ? lo : hi < v ? hi : v; }   /* material.cu:11-21 */
/* material.cu:140-170 Oren_Nayar::f (sigma = 0.5; the double literals make A, B, TanBeta double
 * expressions narrowed to float; SinPhiO divides by SinThetaI as written, :154) */
static f3 oren_f(const brdf_t *b, f3 wo, f3 wi, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
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
    return (wo.y > 0.0f && wi.y > 0.0f) ? scale3(scale3(c, ORC_INV_PI), A + B * maxCos * SinAlpha * TanBeta)
                                        : mk3(0.0f, 0.0f, 0.0f);
}

/* material.cu:241-245 Cook_Torrance::D: pow(float, int) promotes to double (C++11), pow(float,
 * float) is powf; ex = 2.718281828f (:9) */
static float cook_d(const brdf_t *b, f3 wo)
{
    float m = b->m;
    float c = (float)(1.0 / ((double)(m * m) * pow((double)wo.y, 4.0)));
    return c * m_powf(2.718281828f, -(m_acosf(wo.y) * m_acosf(wo.y)) / (m * m));
}
/* material.cu:200-208 */
static f3 cook_f(const brdf_t *b, f3 wo, f3 wi, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
    float dotwin_won = fmaxf(0.0001f, fabsf(4.0f * wi.y * wo.y));
    f3 wh = normalize3(add3(wo, wi));
    return div3s(scale3(scale3(schlick(c, dot3(wi, wh)), cook_d(b, wh)), smith_g(b->a2, wi, wo)),
                 dotwin_won);
}
/* material.cu:246-254 */
static float cook_pdf(const brdf_t *b, f3 wo, f3 wi)
{
    if (!(wo.y * wi.y > 0))                                         /* SameHemisphere, :23-25 */
        return 0;
    f3 wh = normalize3(add3(wo, wi));
    if (dot3(wo, wh) == 0.0f)
        return 0.0f;
    return cook_d(b, wh) / (4 * dot3(wo, wh));
}
/* material.cu:209-234.  The half vector sampled at :216-227 is overwritten at :229 by
 * normalize(wo + *wi), with *wi the caller's variable as it comes in (hazard H9: uninitialised in
 * Scene.cu:253 and at the first bounce, PathTracer.cu:18 -- restated as (0, 0, 0) --, the previous
 * bounce's WORLD-space wi afterwards). */
static f3 cook_sample_f(const brdf_t *b, f3 wo, f3 *wi, float *pdf, f2 sample, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
    (void)sample;       /* :216-227 compute a wh nothing reads */
    f3 wh = normalize3(add3(wo, *wi));                              /* :229 */
    *wi = sub3(scale3(wh, 2.0f * dot3(wo, wh)), wo);                /* :230 */
    *pdf = cook_pdf(b, wo, *wi);
    return cook_f(b, wo, *wi, c);
}

/* material.cuh:118-152 Fresnel::sample_f */
static f3 fresnel_sample_f(const brdf_t *b, f3 wo, f3 *wi, float *pdf, f2 sample, f3 c)
{
    if (c.x < 0.0f)
        c = b->color;
    int entering = wo.y > 0.0f;
    float n1 = 1.0f, n2 = b->eta;
    float R = (n1 - n2) / (n1 + n2);
    R *= R;
    float t = fabsf(wo.y);
    float F = R + (t * t) * (t * t) * t * (1.0f - R);
    if (sample.x < F) {
        *pdf = F;
        *wi = mk3(-wo.x, wo.y, -wo.z);
        return div3s(scale3(c, F), fmaxf(fabsf(wi->y), 0.0000001f));
    } else {
        *pdf = 1 - F;
        float etas;
        float sin2theta = fmaxf(0.001f, 1.0f - wo.y * wo.y);
        if (!entering)
            etas = 1.0f / b->eta;
        else
            etas = b->eta;
        if (sin2theta * etas * etas >= 1.0f)
            return mk3(0.0f, 0.0f, 0.0f);
        float costhetaT = sqrtf(1.0f - etas * etas * sin2theta);
        *wi = add3(scale3(wo, -etas), scale3(mk3(0.0f, 1.0f, 0.0f), etas * wo.y - costhetaT));
        *wi = normalize3(*wi);
        return div3s(scale3(c, 1.0f - F), fmaxf(fabsf(wi->y), 0.0000001f));
    }
}

/* Material/Object.cu:3-32 Material::f via IntersectionRecord.cuh:44-47 */
static f3 rec_f(const orc_scene *s, const rec_t *rec, f3 wo, f3 wi)
{
    brdf_t b = brdf_of(s, rec->material, rec->material_type);
    f3 lo = world2local(rec->normal, rec->tangent, wo), li = world2local(rec->normal, rec->tangent, wi);
    switch (rec->material_type) {
    case ORC_MAT_LAMBERTIAN: return lambert_f(lo, li, rec->color, b.color);
    case ORC_MAT_GGX: return ggx_f(&b, lo, li, rec->color);
    case ORC_MAT_COOK: return cook_f(&b, lo, li, rec->color);
    case ORC_MAT_OREN_NAYAR: return oren_f(&b, lo, li, rec->color);
    default: return mk3(0.0f, 0.0f, 0.0f);                          /* FRESNEL: BLACK (:11-14) */
    }
}
/* Material/Object.cu:34-72 Material::sample_f.  *wi is in/out (H9, cook_sample_f). */
static f3 rec_sample_f(const orc_scene *s, const rec_t *rec, f3 wo, f3 *wi, float *pdf, f2 sample)
{
    brdf_t b = brdf_of(s, rec->material, rec->material_type);
    f3 rwo = world2local(rec->normal, rec->tangent, wo);
    f3 res = mk3(0.0f, 0.0f, 0.0f);
    switch (rec->material_type) {
    case ORC_MAT_LAMBERTIAN:
        *wi = sample_hemi(sample);
        *pdf = lambert_pdf(rwo, *wi);
        res = lambert_f(rwo, *wi, rec->color, b.color);
        break;
    case ORC_MAT_FRESNEL: res = fresnel_sample_f(&b, rwo, wi, pdf, sample, rec->color); break;
    case ORC_MAT_GGX: res = ggx_sample_f(&b, rwo, wi, pdf, sample, rec->color); break;
    case ORC_MAT_COOK: res = cook_sample_f(&b, rwo, wi, pdf, sample, rec->color); break;
    case ORC_MAT_OREN_NAYAR:                                        /* material.cu:171-188 */
        *wi = sample_hemi(sample);
        *pdf = lambert_pdf(rwo, *wi);
        res = oren_f(&b, rwo, *wi, rec->color);
        break;
    default: break;                                                 /* (*wi, *pdf untouched) */
    }
    *wi = local2world(rec->normal, rec->tangent, *wi);
    return res;
}
/* Material/Object.cu:74-104 Material::PDF */
static float rec_pdf(const orc_scene *s, const rec_t *rec, f3 wo, f3 wi)
{
    brdf_t b = brdf_of(s, rec->material, rec->material_type);
    f3 rwo = world2local(rec->normal, rec->tangent, wo), rwi = world2local(rec->normal, rec->tangent, wi);
    switch (rec->material_type) {
    case ORC_MAT_LAMBERTIAN: return lambert_pdf(rwo, rwi);
    case ORC_MAT_GGX: return ggx_pdf(&b, rwo, rwi);
    case ORC_MAT_COOK: return cook_pdf(&b, rwo, rwi);
    case ORC_MAT_OREN_NAYAR: return lambert_pdf(rwo, rwi);          /* material.cu:185-188 */
    default: return 0.0f;                                           /* FRESNEL (:83-86) */
    }
}

/* ---------------------------------------------------------------- lights */
static int is_black(f3 x) { return length3(x) < 0.001f; }           /* Scene.cu:3-5 */

/* Scene/Scene.cu:205-251 evaluateDirectLight, delta-light path (isDelta = true), with
 * PointLight::lightIllumi (Light/Light.cu:7-14) or DirectionalLight::lightIllumi
 * (Light/Light.cu:21-33) inlined.  type 0 = point light j, 1 = directional light j.
 * rec is by value. */
static f3 eval_direct_delta(const orc_scene *s, int type, int j, rec_t rec, orc_counts *cnt)
{
    ray_t r;
    f3 color;
    if (type == 0) {
        const float *lp = s->pl_pos + 3 * (size_t)j, *li = s->pl_illum + 3 * (size_t)j;
        f3 lpos_ = mk3(lp[0], lp[1], lp[2]), illum = mk3(li[0], li[1], li[2]);
        r = mk_ray(lpos_, sub3(rec.pos, lpos_));                    /* Light.cu:9 */
        rec.pdf_light = 1.0f;
        f3 dd = sub3(rec.pos, lpos_);
        color = div3s(illum, fmaxf(0.001f, dot3(dd, dd)));          /* Light.cu:13 */
    } else {
        const float *ld = s->dl_dir + 3 * (size_t)j, *li = s->dl_illum + 3 * (size_t)j;
        f3 dir = normalize3(mk3(ld[0], ld[1], ld[2]));              /* Light.cu:21: dir(normalize(direction)) */
        r = mk_ray(sub3(rec.pos, scale3(dir, 10000.0f)), dir);      /* Light.cu:25 */
        rec.pdf_light = 1.0f;
        color = mk3(li[0], li[1], li[2]);                           /* Light.cu:27 */
    }

    f3 lpos = r.o;                                                  /* Scene.cu:213 */
    rec_t light_rec;
    rec_init(&light_rec);
    int blocked = 0;

    r = rec_spawn_ray(&rec, neg3(r.d));                             /* :218 */

    f3 f = scale3(rec_f(s, &rec, neg3(rec.wo.d), r.d), fabsf(dot3(rec.normal, r.d))); /* :222 */
    if (!is_black(f)) {                                             /* :224 */
        cnt->shadow++;
        if (scene_hit(s, &r, &light_rec)) {                         /* :225 */
            /* H3: `idx` is uninitialised for delta lights (Scene.cu:78, set only at :107,111);
             * `light_rec.lightidx == idx` is restated as never true for them, also when the
             * shadow ray's closest hit is a triangle light. */
            blocked = length3(sub3(light_rec.pos, r.o)) < length3(sub3(r.o, lpos)) - 0.01f; /* :227 */
        }
    }

    if (rec.pdf_light < 0.001f)                                     /* :233 */
        return mk3(0.0f, 0.0f, 0.0f);

    f3 res = mk3(0.0f, 0.0f, 0.0f);
    if (!is_black(f) && !blocked && !is_black(color)) {             /* :236 */
        color = div3s(mul3(color, f), rec.pdf_light);               /* :240 isDelta */
        res = color;
    }
    return res;
}

#define POWER_HEURISTIC(x, y) ((x) * (x) / ((x) * (x) + (y) * (y)))   /* Scene/Scene.cuh:5-7 */

/* ---------------------------------------------------------------- EnvironmentLight */
/* PiecewiseConstant/PiecewiseConstant.cu:8-22 Distribution::Distribution */
typedef struct { int n; float *value, *cdf; float funInt; } dist1_t;
static void dist1_make(dist1_t *d, int n, const float *v)
{
    d->n = n;
    d->value = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(d->value, v, sizeof(float) * (size_t)n);
    d->cdf = (float *)malloc(sizeof(float) * (size_t)(n + 1));
    d->cdf[0] = 0.0f;
    for (int i = 1; i <= n; i++)
        d->cdf[i] = d->cdf[i - 1] + d->value[i - 1] / n;
    d->funInt = d->cdf[n];
    for (int i = 1; i <= n; i++)
        d->cdf[i] = d->funInt == 0 ? (float)i / n : d->cdf[i] / d->funInt;
    d->cdf[n] = 1.0f;
}
/* PiecewiseConstant.cu:24-40 Distribution::sample: the search as written (it can stop one
 * interval early), and the return value is offset + fraction, NOT divided by n */
static float dist1_sample(const dist1_t *d, float u, float *pdf, int *offset)
{
    int l = 0, r = d->n;
    int mid;
    while (r > l) {
        mid = l + (r - l) / 2;
        if (d->cdf[mid] > u)
            r = mid - 1;
        else
            l = mid + 1;
    }
    if (r == d->n)
        --r;
    if (r < 0)
        r = 0;                       /* (cannot happen for u > 0: cdf[0] = 0) */
    *offset = r;
    *pdf = d->value[r] / d->funInt;
    return d->cdf[r + 1] == d->cdf[r] ? r : r + (u - d->cdf[r]) / (d->cdf[r + 1] - d->cdf[r]);
}
/* PiecewiseConstant.cu:65-96 Distribution2D::Distribution2D; rgb2y_xyz :3-6 (0.06007 is a double) */
typedef struct { int width, height; dist1_t *cond; dist1_t marg; float funInt; const float *img; } env_t;
static env_t *env_make(const float *img, int w, int h)
{
    env_t *e = (env_t *)malloc(sizeof(env_t));
    e->width = w; e->height = h; e->img = img;
    e->funInt = 0.0f;
    float *v_tmp = (float *)malloc(sizeof(float) * (size_t)h);
    float *ctmp = (float *)malloc(sizeof(float) * (size_t)w);
    e->cond = (dist1_t *)malloc(sizeof(dist1_t) * (size_t)h);
    for (int i = 0; i < h; i++) {
        for (int j = 0; j < w; j++) {
            size_t idx = 4 * ((size_t)i * w + j);
            float tmp = (float)((double)(img[idx] + 4.5906f * img[idx + 1]) + 0.06007 * (double)img[idx + 2]);
            ctmp[j] = tmp;
            e->funInt += tmp;
        }
        dist1_make(&e->cond[i], w, ctmp);
        v_tmp[i] = e->cond[i].funInt;
    }
    e->funInt /= w * h;
    dist1_make(&e->marg, h, v_tmp);
    free(v_tmp);
    free(ctmp);
    return e;
}
static void env_free(env_t *e)
{
    if (!e) return;
    for (int i = 0; i < e->height; i++) { free(e->cond[i].value); free(e->cond[i].cdf); }
    free(e->cond); free(e->marg.value); free(e->marg.cdf); free(e);
}
/* PiecewiseConstant.cu:108-118 */
static f2 env_dist_sample(const env_t *e, f2 u, float *pdf)
{
    float pdfs[2];
    int v;
    float d1 = dist1_sample(&e->marg, u.y, pdfs + 1, &v);
    float d0 = dist1_sample(&e->cond[v], u.x, pdfs, &v);
    *pdf = pdfs[0] * pdfs[1];
    f2 r = { d0, d1 };
    return r;
}
/* PiecewiseConstant.cu:98-106 Distribution2D::PDF.  H12: an index that reaches n (phi = 2 pi,
 * theta = pi after rounding) reads past the arrays in the reference; clamped to n - 1 here. */
static float env_dist_pdf(const env_t *e, f2 sample)
{
    int i1 = (int)(e->width * sample.x);
    int i2 = (int)(e->height * sample.y);
    if (i1 < 0) i1 = 0;
    if (i2 < 0) i2 = 0;
    if (i1 > e->width - 1) i1 = e->width - 1;
    if (i2 > e->height - 1) i2 = e->height - 1;
    return e->cond[i2].value[i1] / e->marg.funInt;
}
/* tex2D<float4>(img, x, y) of Light.cu:153,197 with the texture object of Light.cu:224-232:
 * normalised coordinates, wrap addressing, linear filtering.  H11: restated from the CUDA
 * programming guide's texture-fetching appendix -- xB = x W - 0.5, i = floor(xB), alpha = frac(xB)
 * kept in 9-bit fixed point with 8 fractional bits (rounded to nearest here), texels wrapped. */
static f3 env_tex2d(const env_t *e, float x, float y)
{
    float xb = (x - floorf(x)) * (float)e->width - 0.5f, yb = (y - floorf(y)) * (float)e->height - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f, b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    int i0 = (int)fi, j0 = (int)fj;
    int i[2] = { ((i0 % e->width) + e->width) % e->width, (((i0 + 1) % e->width) + e->width) % e->width };
    int j[2] = { ((j0 % e->height) + e->height) % e->height, (((j0 + 1) % e->height) + e->height) % e->height };
    float out[3];
    for (int c = 0; c < 3; c++) {
        float t00 = e->img[4 * ((size_t)j[0] * e->width + i[0]) + c], t10 = e->img[4 * ((size_t)j[0] * e->width + i[1]) + c];
        float t01 = e->img[4 * ((size_t)j[1] * e->width + i[0]) + c], t11 = e->img[4 * ((size_t)j[1] * e->width + i[1]) + c];
        out[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
    return mk3(out[0], out[1], out[2]);
}
#define ENV_INV_PI 0.3183098861837907f          /* Light/Light.cu:2-4 */
static float m_atan2f(float y, float x) { return (float)atan2((double)y, (double)x); }
/* Light/Light.cu:180-199 EnvironmentLight::L (getLe(r) = L(r.dir), :175-178) */
static f3 env_L(const env_t *e, f3 r)
{
    f3 k = normalize3(r);
    float phi = 0.0f;
    float theta = m_acosf(k.y);
    float sint = m_sinf(theta);
    if (sint > 0.001f) {
        phi = m_atan2f(r.x, r.z);
        if (r.z < 0.0f)
            phi = (float)((double)phi + ORC_PI_D);
        if (phi < 0.0f)
            phi = (float)((double)phi + (double)2.0f * ORC_PI_D);
    }
    return env_tex2d(e, phi * ENV_INV_PI * 0.5f, theta * ENV_INV_PI);
}
/* Light/Light.cu:157-173 EnvironmentLight::PDF */
static float env_pdf(const env_t *e, f3 wi)
{
    f3 k = normalize3(wi);
    float theta = m_acosf(k.y);
    float sint = m_sinf(theta);
    float phi = m_atan2f(k.x, k.z);
    if (sint < 0.001f)
        return 0.0f;
    if (wi.z < 0.0f)
        phi = (float)((double)phi + ORC_PI_D);
    if (phi < 0.0f)
        phi = (float)((double)phi + (double)2.0f * ORC_PI_D);
    f2 sm = { phi * 0.5f * ENV_INV_PI, theta * ENV_INV_PI };
    return (float)((double)env_dist_pdf(e, sm) / ((double)2.0f * ORC_PI_D * ORC_PI_D * (double)sint));
}

/* The tables EnvironmentLight::setUp (Light.cu:201-234) builds once on the host: built at the
 * entry of every API call that renders (env_begin) and dropped at its end (test infrastructure:
 * one call at a time). */
static env_t *g_env_cur = NULL;
static void env_begin(const orc_scene *s)
{
    g_env_cur = (s->n_env_lights > 0 && s->env_image) ? env_make(s->env_image, s->env_width, s->env_height) : NULL;
}
static void env_end(void)
{
    env_free(g_env_cur);
    g_env_cur = NULL;
}
static const env_t *scene_env(const orc_scene *s) { return s->n_env_lights > 0 ? g_env_cur : NULL; }

/* Scene/Scene.cu:205-287 evaluateDirectLight for triangle light k (isDelta = false, idx = k),
 * with TriangleLight::lightIllumi (Light/Light.cu:47-65) inlined.  rec is by value. */
static f3 eval_direct_area(const orc_scene *s, int k, rec_t rec, f2 sample_light, f2 sample_brdf,
                           orc_counts *cnt)
{
    const int idx = k;
    tl_t l = tl_load(s, k);
    /* lightIllumi, Light.cu:49-64 */
    float sq_x = sqrtf(sample_light.x);
    f3 tri_sample = mk3(1 - sq_x, sample_light.y * sq_x, 0);
    tri_sample.z = 1.0f - tri_sample.x - tri_sample.y;
    f3 lp = add3(add3(scale3(l.pos[0], tri_sample.x), scale3(l.pos[1], tri_sample.y)),
                 scale3(l.pos[2], tri_sample.z));                   /* Light.cu:44 */
    ray_t r = mk_ray(lp, sub3(rec.pos, lp));                        /* Light.cu:56 */
    rec.pdf_light = tl_pdf(&l, length3(sub3(rec.pos, lp)), r.d);    /* Light.cu:58-63 */
    f3 color = tl_L(&l, r.d);                                       /* Light.cu:64 */

    f3 lpos = r.o;                                                  /* Scene.cu:213 */
    rec_t light_rec;
    rec_init(&light_rec);
    int blocked = 0;
    r = rec_spawn_ray(&rec, neg3(r.d));                             /* :218 */
    f3 f = scale3(rec_f(s, &rec, neg3(rec.wo.d), r.d), fabsf(dot3(rec.normal, r.d))); /* :222 */
    if (!is_black(f)) {                                             /* :224 */
        cnt->shadow++;
        if (scene_hit(s, &r, &light_rec))                           /* :225-228 */
            blocked = light_rec.lightidx == idx
                          ? 0
                          : (length3(sub3(light_rec.pos, r.o)) < length3(sub3(r.o, lpos)) - 0.01f);
    }
    f3 res = mk3(0.0f, 0.0f, 0.0f);
    if (rec.pdf_light < 0.001f)                                     /* :233 */
        return mk3(0.0f, 0.0f, 0.0f);
    if (!is_black(f) && !blocked && !is_black(color)) {             /* :236 */
        rec.pdf_surface = rec_pdf(s, &rec, neg3(rec.wo.d), r.d);       /* :244 */
        float weight = POWER_HEURISTIC(rec.pdf_light, rec.pdf_surface);   /* :245 */
        weight /= rec.pdf_light;                                    /* :246 */
        color = scale3(mul3(color, f), weight);                     /* :247 */
        res = color;
    }
    f3 wi = mk3(0.0f, 0.0f, 0.0f);                                  /* :253 uninitialised: H9 */
    float weight = 1.0f;
    f = rec_sample_f(s, &rec, neg3(rec.wo.d), &wi, &rec.pdf_surface, sample_brdf);   /* :258 */
    f = scale3(f, fabsf(dot3(rec.normal, wi)));                     /* :259 */
    r = rec_spawn_ray(&rec, wi);                                    /* :260 */
    rec.t = ORC_INF;                                                /* :262 */
    rec.lightidx = -1;                                              /* :263 */
    if (!is_black(f) && rec.pdf_surface > 0.001f) {                 /* :264 */
        /* :266 isSpecular() is false for every material in scope */
        float pdf = tl_pdf(&l, rec.t, wi);                          /* :268, with rec.t = INF */
        if (fabsf(pdf) < 0.01f)                                     /* :269 */
            return res;
        weight = POWER_HEURISTIC(rec.pdf_surface, pdf);             /* :271 */
        f3 lc = mk3(0.0f, 0.0f, 0.0f);
        cnt->shadow++;
        if (scene_hit(s, &r, &rec) && rec.lightidx == idx)          /* :275 */
            lc = tl_L(&l, neg3(wi));                                /* :277 */
        /* else getLe(r) = BLACK (Light.cuh:94) */
        res = add3(res, div3s(scale3(mul3(lc, f), weight), rec.pdf_surface));   /* :282 */
    }
    return res;
}

/* Scene/Scene.cu:205-287 evaluateDirectLight for the environment light (isDelta = false,
 * idx = n_tri_lights + j, Scene.cu:109-114), EnvironmentLight::lightIllumi (Light.cu:137-155)
 * inlined.  H10: `Ray r;` (Scene.cu:207) is default-constructed (Ray.cuh:16) and lightIllumi keeps
 * its origin (Light.cu:142), so lpos (Scene.cu:213) is uninitialised: restated as (0, 0, 0). */
static int scene_hit(const orc_scene *s, const ray_t *r, rec_t *rec);
static f3 eval_direct_env(const orc_scene *s, const env_t *e, rec_t rec, f2 sample_light, f2 sample_brdf,
                          orc_counts *cnt)
{
    const int idx = s->n_tri_lights;                                /* Scene.cu:111, j = 0 */
    f3 color;
    ray_t r;
    {   /* lightIllumi */
        f2 sm = env_dist_sample(e, sample_light, &rec.pdf_light);   /* Light.cu:139 */
        float theta = (float)((double)sm.y * ORC_PI_D), phi = (float)((double)(2.0f * sm.x) * ORC_PI_D);   /* :140 */
        float sint = m_sinf(theta);                                 /* :141 */
        r = mk_ray(mk3(0.0f, 0.0f, 0.0f),                           /* :142, H10 */
                   mk3(m_sinf(theta) * m_sinf(phi), m_cosf(theta), m_sinf(theta) * m_cosf(phi)));
        if (rec.pdf_light < 0.001f)                                 /* :143-147 */
            color = mk3(0.0f, 0.0f, 0.0f);
        else if (sint < 0.001f)
            color = mk3(0.0f, 0.0f, 0.0f);
        else {
            rec.pdf_light = (float)((double)rec.pdf_light / ((double)(2.0f * sint) * ORC_PI_D * ORC_PI_D));   /* :149 */
            color = env_tex2d(e, 0.5f * phi * ENV_INV_PI, theta * ENV_INV_PI);   /* :153 */
        }
    }
    f3 lpos = r.o;                                                  /* Scene.cu:213 */
    rec_t light_rec;
    rec_init(&light_rec);
    int blocked = 0;
    r = rec_spawn_ray(&rec, neg3(r.d));                             /* :218 */
    f3 f = scale3(rec_f(s, &rec, neg3(rec.wo.d), r.d), fabsf(dot3(rec.normal, r.d))); /* :222 */
    if (!is_black(f)) {                                             /* :224 */
        cnt->shadow++;
        if (scene_hit(s, &r, &light_rec))                           /* :225-228; a triangle light's index never equals idx */
            blocked = light_rec.lightidx == idx
                          ? 0
                          : (length3(sub3(light_rec.pos, r.o)) < length3(sub3(r.o, lpos)) - 0.01f);
    }
    f3 res = mk3(0.0f, 0.0f, 0.0f);
    if (rec.pdf_light < 0.001f)                                     /* :233 */
        return mk3(0.0f, 0.0f, 0.0f);
    if (!is_black(f) && !blocked && !is_black(color)) {             /* :236 */
        rec.pdf_surface = rec_pdf(s, &rec, neg3(rec.wo.d), r.d);    /* :244 */
        float weight = POWER_HEURISTIC(rec.pdf_light, rec.pdf_surface);   /* :245 */
        weight /= rec.pdf_light;                                    /* :246 */
        color = scale3(mul3(color, f), weight);                     /* :247 */
        res = color;
    }
    f3 wi = mk3(0.0f, 0.0f, 0.0f);                                  /* :253 uninitialised: H9 */
    float weight = 1.0f;
    f = rec_sample_f(s, &rec, neg3(rec.wo.d), &wi, &rec.pdf_surface, sample_brdf);   /* :258 */
    f = scale3(f, fabsf(dot3(rec.normal, wi)));                     /* :259 */
    r = rec_spawn_ray(&rec, wi);                                    /* :260 */
    if (!is_black(f) && rec.pdf_surface > 0.001f) {                 /* :264 */
        float pdf = env_pdf(e, wi);                                 /* :268 */
        if (fabsf(pdf) < 0.01f)                                     /* :269 */
            return res;
        weight = POWER_HEURISTIC(rec.pdf_surface, pdf);             /* :271 */
        /* :275-280: hit(r, rec) is evaluated, but rec.lightidx can never equal idx (only triangle
         * lights set it, to indices below n_tri_lights), so l = getLe(r) whether or not the ray
         * is blocked; the trace has no other effect and is not restated */
        f3 lc = env_L(e, r.d);
        res = add3(res, div3s(scale3(mul3(lc, f), weight), rec.pdf_surface));   /* :282 */
    }
    return res;
}

/* ---------------------------------------------------------------- Philox */
/* cuRAND Philox4x32-10 (call sites PathTracer.cu:27, Scene.cu:87, kernel.cu:47,67),
 * restated from the published algorithm (Random123; constants as in
 * /opt/rocm/include/rocrand/rocrand_philox4x32_10.h:62-65, round :286-303). */
void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int round = 0; round < 10; round++) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
        uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* curand_init(seed = idx + 332, subsequence = idx, offset = 0) (kernel.cu:47):
 * key = (seed lo, seed hi), counter words 2-3 = subsequence, words 0-1 = draw
 * number.  curand_uniform4 maps each word x to x * 2^-32 + 2^-33 (H7). */
void orc_pixel_uniform4(uint32_t pixel_index, uint32_t draw, float out[4])
{
    uint64_t seed = (uint64_t)pixel_index + 332u;
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t ctr[4] = { draw, 0u, pixel_index, 0u };
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    for (int i = 0; i < 4; i++)
        out[i] = (float)r[i] * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* ---------------------------------------------------------------- Camera */
/* Camera/Camera.cuh:17-23, Camera/Camera.cu:4-14 (pers/raster2world are dead). */
typedef struct { f3 pos, up, front, right; int res_x, res_y; float n; } cam_t;
static cam_t cam_make(const orc_camera_args *a)
{
    cam_t c;
    f3 pos = mk3(a->pos[0], a->pos[1], a->pos[2]);
    f3 lookat = mk3(a->lookat[0], a->lookat[1], a->lookat[2]);
    c.pos = pos;
    c.up = normalize3(mk3(a->up[0], a->up[1], a->up[2]));
    c.front = normalize3(sub3(lookat, pos));
    c.res_x = a->res_x;
    c.res_y = a->res_y;
    c.n = a->near_;
    c.right = normalize3(cross3(c.front, c.up));
    c.up = normalize3(cross3(c.right, c.front));
    return c;
}
/* Camera/Camera.cu:16-24 */
static ray_t cam_generate_ray(const cam_t *c, float x, float y)
{
    const float step = 0.25f;
    float aspect = (float)c->res_x / c->res_y;
    f3 dir = add3(add3(scale3(c->front, c->n),
                       scale3(scale3(scale3(c->right, (float)x / c->res_x), step), aspect)),
                  scale3(scale3(c->up, (float)y / c->res_y), step));
    return mk_ray(c->pos, normalize3(dir));
}

/* Camera/Camera.cu:26-42 getxy (only the texture LOD reads it) */
static f2 cam_getxy(const cam_t *c, f3 poso)
{
    const float step = 0.25f;
    f3 dir = normalize3(sub3(poso, c->pos));
    f2 res = { 0.0f, 0.0f };
    if (dot3(dir, c->front) == 0.0f)
        return res;
    float aspect = (float)c->res_x / c->res_y;
    float t = c->n / dot3(dir, c->front);
    f3 rpos = scale3(dir, t);
    res.x = dot3(rpos, c->right) * c->res_x / step / aspect;
    res.y = dot3(rpos, c->up) * c->res_y / step;
    return res;
}

/* ---------------------------------------------------------------- mesh textures */
/* Texture/Texture.cu:135-186 initImages: a float4 mip-mapped array (levels = getMipMapLevels, :2-13),
 * level 0 = the caller's image, the others by generateMipMaps (:48-133), and a texture object with
 * normalised coordinates, clamp addressing, linear filtering within and between levels (:174-182).
 * H13: the filtering is the texture unit's and is not bit-specified; it is restated from the CUDA
 * programming guide's texture-fetching appendix as H11 does for tex2D -- xB = x N - 0.5, i = floor(xB),
 * alpha = frac(xB) in 8 fractional bits (rounded to nearest), texel indices clamped to the edge -- and
 * tex2DGrad's level from the OpenGL rule: rho = max(|dPdx scaled to texels|, |dPdy scaled to texels|)
 * over level 0's size, lambda = log2(rho) (float rounding of the double function) clamped to
 * [0, levels - 1] and kept in 8 fractional bits, the two nearest levels blended linearly. */
typedef struct { int w, h, levels; float **lvl; int *lw, *lh; } tex_t;
int orc_texture_levels(int w, int h)
{
    size_t sz = (size_t)(w > h ? w : h);                            /* Texture.cu:3 (depth 0) */
    int levels = 0;
    while (sz) {
        sz /= 2;
        levels++;
    }
    return levels;
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static void tex_fetch(const float *img, int w, int h, float x, float y, float out[4])
{
    float xb = x * (float)w - 0.5f, yb = y * (float)h - 0.5f;
    if (!(xb > -1.0f)) xb = -1.0f;                                  /* clamp addressing (and NaN) */
    if (xb > (float)w) xb = (float)w;
    if (!(yb > -1.0f)) yb = -1.0f;
    if (yb > (float)h) yb = (float)h;
    float fi = floorf(xb), fj = floorf(yb);
    float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f, b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    int i0 = (int)fi, j0 = (int)fj;
    int i[2] = { clampi(i0, 0, w - 1), clampi(i0 + 1, 0, w - 1) };
    int j[2] = { clampi(j0, 0, h - 1), clampi(j0 + 1, 0, h - 1) };
    for (int c = 0; c < 4; c++) {
        float t00 = img[4 * ((size_t)j[0] * w + i[0]) + c], t10 = img[4 * ((size_t)j[0] * w + i[1]) + c];
        float t01 = img[4 * ((size_t)j[1] * w + i[0]) + c], t11 = img[4 * ((size_t)j[1] * w + i[1]) + c];
        out[c] = (1 - a) * (1 - b) * t00 + a * (1 - b) * t10 + (1 - a) * b * t01 + a * b * t11;
    }
}
/* Texture.cu:14-45 d_mipmap over one level (:48-133 generateMipMaps): four bilinear taps of the
 * level above at the corners of the texel, averaged and clamped to 1 */
static void tex_downsample(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    float px = (float)(1.0 / (double)(float)dw);                    /* :19 */
    float py = (float)(1.0 / (double)(float)dh);                    /* :20 */
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            float a[4], b[4], c[4], d[4];
            tex_fetch(src, sw, sh, (float)(x + 0) * px, (float)(y + 0) * py, a);   /* :30-33 */
            tex_fetch(src, sw, sh, (float)(x + 1) * px, (float)(y + 0) * py, b);
            tex_fetch(src, sw, sh, (float)(x + 1) * px, (float)(y + 1) * py, c);
            tex_fetch(src, sw, sh, (float)(x + 0) * px, (float)(y + 1) * py, d);
            for (int k = 0; k < 4; k++) {
                float col = a[k] + b[k] + c[k] + d[k];
                col /= 4.0f;                                        /* :36 */
                dst[4 * ((size_t)y * dw + x) + k] = fminf(col, 1.0f);   /* :37 */
            }
        }
}
static void tex_make(tex_t *t, const float *img, int w, int h)
{
    t->w = w; t->h = h;
    t->levels = orc_texture_levels(w, h);                           /* Texture.cu:145 */
    t->lvl = (float **)malloc(sizeof(float *) * (size_t)t->levels);
    t->lw = (int *)malloc(sizeof(int) * (size_t)t->levels);
    t->lh = (int *)malloc(sizeof(int) * (size_t)t->levels);
    t->lvl[0] = (float *)malloc(sizeof(float) * 4 * (size_t)w * h);
    memcpy(t->lvl[0], img, sizeof(float) * 4 * (size_t)w * h);      /* :150-159 */
    t->lw[0] = w; t->lh[0] = h;
    for (int l = 1; l < t->levels; l++) {                           /* :54-58 */
        t->lw[l] = t->lw[l - 1] / 2 > 1 ? t->lw[l - 1] / 2 : 1;
        t->lh[l] = t->lh[l - 1] / 2 > 1 ? t->lh[l - 1] / 2 : 1;
        t->lvl[l] = (float *)malloc(sizeof(float) * 4 * (size_t)t->lw[l] * t->lh[l]);
        tex_downsample(t->lvl[l - 1], t->lw[l - 1], t->lh[l - 1], t->lvl[l], t->lw[l], t->lh[l]);
    }
}
static void tex_drop(tex_t *t)
{
    for (int l = 0; l < t->levels; l++) free(t->lvl[l]);
    free(t->lvl); free(t->lw); free(t->lh);
}
/* tex2DGrad<float4>(Texture, x, y, dPdx, dPdy) (Model.cu:309), H13 */
static f3 tex_grad(const tex_t *t, float x, float y, f2 dpdx, f2 dpdy)
{
    float ax = dpdx.x * (float)t->w, ay = dpdx.y * (float)t->h;
    float bx = dpdy.x * (float)t->w, by = dpdy.y * (float)t->h;
    float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
    float lam = 0.0f;
    if (rho > 1.0f)
        lam = fminf((float)log2((double)rho), (float)(t->levels - 1));
    float lq = floorf(lam * 256.0f + 0.5f) / 256.0f;
    int d = (int)floorf(lq);
    float wgt = lq - (float)d;
    if (d >= t->levels - 1) {
        d = t->levels - 1;
        wgt = 0.0f;
    }
    float c0[4], c1[4];
    tex_fetch(t->lvl[d], t->lw[d], t->lh[d], x, y, c0);
    if (wgt == 0.0f)
        return mk3(c0[0], c0[1], c0[2]);
    tex_fetch(t->lvl[d + 1], t->lw[d + 1], t->lh[d + 1], x, y, c1);
    return mk3((1 - wgt) * c0[0] + wgt * c1[0], (1 - wgt) * c0[1] + wgt * c1[1], (1 - wgt) * c0[2] + wgt * c1[2]);
}
int orc_texture_mip(const float *img, int w, int h, int level, float *out, int *lw, int *lh)
{
    if (!img || w < 1 || h < 1 || level < 0 || level >= orc_texture_levels(w, h))
        return -1;
    tex_t t;
    tex_make(&t, img, w, h);
    if (out) memcpy(out, t.lvl[level], sizeof(float) * 4 * (size_t)t.lw[level] * t.lh[level]);
    if (lw) *lw = t.lw[level];
    if (lh) *lh = t.lh[level];
    tex_drop(&t);
    return 0;
}
int orc_tex2d_grad(const float *img, int w, int h, float u, float v, const float dpdx[2],
                   const float dpdy[2], float rgb[3])
{
    if (!img || w < 1 || h < 1)
        return -1;
    tex_t t;
    tex_make(&t, img, w, h);
    f2 dx = { dpdx[0], dpdx[1] }, dy = { dpdy[0], dpdy[1] };
    f3 c = tex_grad(&t, u, v, dx, dy);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    tex_drop(&t);
    return 0;
}

/* The texture objects Mesh::initTexture makes once on the host and the camera pathTracer hands every
 * record (PathTracer.cu:30, Scene.cu:223): set at the entry of every API call that renders and dropped
 * at its end, like the environment tables (test infrastructure: one call at a time).  No camera
 * (orc_trace_ray): getdxdy returns zeros (IntersectionRecord.cuh:62-63). */
static tex_t *g_tex_cur = NULL;
static int g_tex_n = 0;
static const cam_t *g_cam_cur = NULL;
static void tex_begin(const orc_scene *s, const cam_t *cam)
{
    g_cam_cur = cam;
    g_tex_n = 0;
    g_tex_cur = NULL;
    if (s->n_textures > 0 && s->tex_size && s->tex_data && s->mesh_texture) {
        g_tex_cur = (tex_t *)malloc(sizeof(tex_t) * (size_t)s->n_textures);
        const float *img = s->tex_data;
        for (int k = 0; k < s->n_textures; k++) {
            int w = s->tex_size[2 * k], h = s->tex_size[2 * k + 1];
            tex_make(&g_tex_cur[k], img, w, h);
            img += 4 * (size_t)w * h;
        }
        g_tex_n = s->n_textures;
    }
}
static void tex_end(void)
{
    for (int k = 0; k < g_tex_n; k++) tex_drop(&g_tex_cur[k]);
    free(g_tex_cur);
    g_tex_cur = NULL;
    g_tex_n = 0;
    g_cam_cur = NULL;
}
/* Model.cu:304-309 for triangle g (traversal index) of mesh `mesh`, the last one that updated rec:
 * Triangle::anyPoint (:60-88), IntersectRecord::getdxdy (IntersectionRecord.cuh:60-80), tex2DGrad. */
static f3 mesh_texture_color(const orc_scene *s, int mesh, int g, const rec_t *rec)
{
    if (!g_tex_cur || s->mesh_texture[mesh] < 0 || s->mesh_texture[mesh] >= g_tex_n)
        return mk3(-1.0f, -1.0f, -1.0f);                            /* H5 */
    const tex_t *tex = &g_tex_cur[s->mesh_texture[mesh]];
    const float *pp = s->tri_pos + 9 * (size_t)g;
    m4 M = load_m4(s->model_xf + 16 * (size_t)s->tri_model[g]);
    f3 pos[3] = { mk3(pp[0], pp[1], pp[2]), mk3(pp[3], pp[4], pp[5]), mk3(pp[6], pp[7], pp[8]) };
    f2 vt[3] = { { 0.0f, 0.0f }, { 0.0f, 0.0f }, { 0.0f, 0.0f } };
    if (s->tri_uv)
        for (int i = 0; i < 3; i++) {
            vt[i].x = s->tri_uv[6 * (size_t)g + 2 * i];
            vt[i].y = s->tri_uv[6 * (size_t)g + 2 * i + 1];
        }
    f3 Pos = rec->pos;
    /* anyPoint, Model.cu:61-87 */
    f3 ta = m4_apply3(&M, pos[0]), tb = m4_apply3(&M, pos[1]), tc = m4_apply3(&M, pos[2]);
    float S = 2.0f * tri_area(&M, pos);
    float s2 = length3(cross3(sub3(Pos, tc), sub3(Pos, ta)));
    float s3 = length3(cross3(sub3(Pos, tc), sub3(Pos, tb)));
    float m1 = s3 / S, m2 = s2 / S, m3 = 1.0f - m1 - m2;
    f2 UV = { vt[0].x * m1 + vt[1].x * m2 + vt[2].x * m3, vt[0].y * m1 + vt[1].y * m2 + vt[2].y * m3 };   /* :68 = rec.uv (:117,132) */
    if (m3 > m2) {                                                  /* :69-84: 0.01, 0.02 are doubles */
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
    f2 anyUV = { vt[0].x * m1 + vt[1].x * m2 + vt[2].x * m3, vt[0].y * m1 + vt[1].y * m2 + vt[2].y * m3 };   /* :86 */
    f3 anyPos = add3(scale3(sub3(Pos, ta), m1), scale3(sub3(Pos, tb), m2));                                   /* :87 */
    /* getdxdy(rec.pos, UV, anyPos, anyUV) */
    f2 dx = { 0.0f, 0.0f }, dy = { 0.0f, 0.0f };
    if (g_cam_cur) {
        f2 duv = { anyUV.x - UV.x, anyUV.y - UV.y };
        f2 xy1 = cam_getxy(g_cam_cur, Pos), xy2 = cam_getxy(g_cam_cur, anyPos);
        f2 dxy = { xy2.x - xy1.x, xy2.y - xy1.y };
        float rx = duv.x / dxy.x, ry = duv.y / dxy.x, rz = duv.x / dxy.y, rw = duv.y / dxy.y;   /* :69-72 */
        if (duv.x == 0.0f)                                          /* :74-75 */
            rz = rx = 0.0f;
        /* :76 `if (duv.y = 0.0f)` assigns: never taken */
        dx.x = fabsf(rx); dx.y = fabsf(ry); dy.x = fabsf(rz); dy.y = fabsf(rw);                 /* :79 */
    }
    return tex_grad(tex, UV.x, UV.y, dx, dy);                       /* Model.cu:309 */
}

/* ---------------------------------------------------------------- pathTracer */
typedef struct { uint32_t pixel, draw; } rng_t;
static void rng_uniform4(rng_t *g, float out[4])
{
    orc_pixel_uniform4(g->pixel, g->draw, out);
    g->draw++;
}

/* Scene/Scene.cu:69-124 sampleAllLight: one curand_uniform4 per light (:87),
 * light types in enum order; point, directional and triangle lights in scope. */
static f3 sample_all_light(const orc_scene *s, const rec_t *rec, rng_t *g, orc_counts *cnt)
{
    f3 res = mk3(0.0f, 0.0f, 0.0f);
    for (int j = 0; j < s->n_point_lights; j++) {           /* light::POINT_LIGHT */
        float tmp[4];
        rng_uniform4(g, tmp);
        res = add3(res, eval_direct_delta(s, 0, j, *rec, cnt));
    }
    for (int j = 0; j < s->n_dir_lights; j++) {             /* light::DIR_LIGHT */
        float tmp[4];
        rng_uniform4(g, tmp);
        res = add3(res, eval_direct_delta(s, 1, j, *rec, cnt));
    }
    for (int j = 0; j < s->n_tri_lights; j++) {             /* light::TRIANGLE_LIGHT */
        float tmp[4];
        rng_uniform4(g, tmp);                               /* :87-89 */
        f2 sample_light = { tmp[0], tmp[1] }, sample_surface = { tmp[2], tmp[3] };
        res = add3(res, eval_direct_area(s, j, *rec, sample_light, sample_surface, cnt));
    }
    const env_t *e = scene_env(s);
    for (int j = 0; e && j < s->n_env_lights; j++) {        /* light::ENVIRONMENT_LIGHT (one at most) */
        float tmp[4];
        rng_uniform4(g, tmp);
        f2 sample_light = { tmp[0], tmp[1] }, sample_surface = { tmp[2], tmp[3] };
        res = add3(res, eval_direct_env(s, e, *rec, sample_light, sample_surface, cnt));
    }
    return res;
}

/* core/PathTracer.cu:11-92 with MAX_DEPTH as a run-time argument. */
static f3 path_tracer(const orc_scene *s, ray_t r, rng_t *g, int max_depth, orc_counts *cnt,
                      float *first_t, int *first_prim)
{
    rec_t rec;
    int ishit, specular_bounce = 0;
    float pdf;
    f3 le, wi = mk3(0.0f, 0.0f, 0.0f);                              /* :18 uninitialised: H9 */
    f3 beta = mk3(1.0f, 1.0f, 1.0f);
    f3 res = mk3(0.0f, 0.0f, 0.0f);
    float tmp[4];
    for (int bounces = 0;; bounces++) {
        rng_uniform4(g, tmp);                                       /* :27 */
        f2 sample_scatter = { tmp[0], tmp[1] };
        rec_init(&rec);                                             /* :29 */
        const env_t *env = scene_env(s);
        if (bounces > max_depth && !(specular_bounce && (s->n_tri_lights > 0 || env))) {
            /* :31 traces this ray too, but nothing reads the result before the break at :64
             * unless the ray left a specular surface and there is a light to land on (:33-52). */
            break;
        }
        ishit = scene_hit(s, &r, &rec);                             /* :31 */
        if (bounces == 0) {
            cnt->primary++;
            *first_t = rec.t;
            *first_prim = rec.prim;
        } else {
            cnt->bounce++;
        }
        rec.wo = r;                                                 /* :32 */
        /* :33-52 emission of a triangle light the camera ray, or a ray off a specular surface,
         * lands on */
        if ((bounces == 0 || specular_bounce) && ishit && rec.isLight &&
            rec.light_type >= ORC_LIGHT_TRIANGLE && rec.lightidx >= 0 && rec.lightidx < s->n_tri_lights) {
            tl_t l = tl_load(s, rec.lightidx);
            res = add3(res, mul3(beta, tl_L(&l, neg3(r.d))));       /* :47 */
        }
        if ((bounces == 0 || specular_bounce) && !ishit && env)     /* :53-61 */
            res = add3(res, mul3(beta, env_L(env, r.d)));
        if (!ishit || bounces > max_depth)                          /* :64 */
            break;
        specular_bounce = rec.material_type & ORC_MAT_SPECULAR_BIT; /* :67 */
        if (!specular_bounce) {
            le = sample_all_light(s, &rec, g, cnt);                 /* :72 */
            res = add3(res, mul3(beta, le));                        /* :73 */
        }
        le = rec_sample_f(s, &rec, neg3(r.d), &wi, &pdf, sample_scatter); /* :76 */
        if (pdf < 0.0001f || length3(le) < 0.0001f)                 /* :77 */
            return res;
        beta = mul3(beta, div3s(scale3(le, fabsf(dot3(wi, rec.normal))), pdf)); /* :79 */
        r = rec_spawn_ray(&rec, wi);                                /* :80 */
        if (bounces > 3) {                                          /* :82-88 */
            float q = 0.5f;
            if (tmp[2] < q)
                break;
            beta = div3s(beta, 1.0f - q);
        }
    }
    return res;
}

/* ---------------------------------------------------------------- driver */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* kernel.cu:50-79 `debug` for one pixel, with the H4 conventions: jitter = (.x,.y) of one
 * uniform4 from the pixel's own stream, cnt = 1-based sample index. */
static void render_pixel(const orc_scene *scene, const cam_t *cam, int width, int height, int x,
                         int y, int spp, int max_depth, float *rgba, int32_t *prim_id,
                         float *t_out, orc_counts *c)
{
    rng_t g = { (uint32_t)(y * width + x), 0u };                    /* kernel.cu:46-47 */
    f3 px = mk3(0.0f, 0.0f, 0.0f);
    float ft = ORC_INF;
    int fprim = -1;
    for (int cnt = 1; cnt <= spp; cnt++) {
        float frac1 = 1.0f / cnt;                                   /* :65 */
        float frac2 = frac1 * (cnt - 1);                            /* :66 */
        float j[4];
        rng_uniform4(&g, j);                                        /* :67 (H4) */
        float xoff = j[0], yoff = j[1];
        ray_t r = cam_generate_ray(cam, x - width / 2 + xoff, y - height / 2 + yoff);    /* :73 */
        f3 col = path_tracer(scene, r, &g, max_depth, c, &ft, &fprim);                   /* :74 */
        px = add3(scale3(px, frac2), scale3(col, frac1));           /* :75 */
    }
    if (rgba) {
        rgba[0] = px.x;
        rgba[1] = px.y;
        rgba[2] = px.z;
        rgba[3] = 1.0f;                                             /* :77 */
    }
    if (prim_id) *prim_id = fprim;
    if (t_out) *t_out = ft;
}

int orc_render(const orc_scene *scene, const orc_camera_args *cam_args, int width, int height,
               int x0, int y0, int x1, int y1, int spp, int max_depth, int threads,
               float *rgba, int32_t *prim_id, float *t_out, orc_counts *counts)
{
    if (!scene || !cam_args || x0 < 0 || y0 < 0 || x1 > width || y1 > height || x0 >= x1 ||
        y0 >= y1 || spp < 1)
        return -1;
    cam_t cam = cam_make(cam_args);
    env_begin(scene);
    tex_begin(scene, &cam);
    int cw = x1 - x0, ch = y1 - y0;
    uint64_t n_primary = 0, n_shadow = 0, n_bounce = 0;
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    /* small windows over big meshes: one pixel per task, or most threads would idle */
    const int grain = cw * ch >= 4096 ? 16 : 1;
#pragma omp parallel for schedule(dynamic, grain) reduction(+ : n_primary, n_shadow, n_bounce)
    for (int i = 0; i < cw * ch; i++) {
        orc_counts c = { 0, 0, 0 };
        render_pixel(scene, &cam, width, height, x0 + i % cw, y0 + i / cw, spp, max_depth,
                     rgba ? rgba + 4 * i : NULL, prim_id ? prim_id + i : NULL,
                     t_out ? t_out + i : NULL, &c);
        n_primary += c.primary;
        n_shadow += c.shadow;
        n_bounce += c.bounce;
    }
    if (counts) {
        counts->primary = n_primary;
        counts->shadow = n_shadow;
        counts->bounce = n_bounce;
    }
    tex_end();
    env_end();
    return 0;
}

/* The same for an explicit list of pixels (xy[2 i], xy[2 i + 1]): what the full-size parity
 * tests use to check a handful of pixels of a 4K frame over a mesh of millions of triangles
 * with every host thread busy. */
int orc_render_pixels(const orc_scene *scene, const orc_camera_args *cam_args, int width,
                      int height, int n, const int32_t *xy, int spp, int max_depth, int threads,
                      float *rgba, int32_t *prim_id, float *t_out, orc_counts *counts)
{
    if (!scene || !cam_args || n < 0 || (n > 0 && !xy) || spp < 1)
        return -1;
    for (int i = 0; i < n; i++)
        if (xy[2 * i] < 0 || xy[2 * i] >= width || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= height)
            return -1;
    cam_t cam = cam_make(cam_args);
    env_begin(scene);
    tex_begin(scene, &cam);
    uint64_t n_primary = 0, n_shadow = 0, n_bounce = 0;
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_primary, n_shadow, n_bounce)
    for (int i = 0; i < n; i++) {
        orc_counts c = { 0, 0, 0 };
        render_pixel(scene, &cam, width, height, xy[2 * i], xy[2 * i + 1], spp, max_depth,
                     rgba ? rgba + 4 * i : NULL, prim_id ? prim_id + i : NULL,
                     t_out ? t_out + i : NULL, &c);
        n_primary += c.primary;
        n_shadow += c.shadow;
        n_bounce += c.bounce;
    }
    if (counts) {
        counts->primary = n_primary;
        counts->shadow = n_shadow;
        counts->bounce = n_bounce;
    }
    tex_end();
    env_end();
    return 0;
}

int orc_scene_hit(const orc_scene *scene, int n, const float *org, const float *dir, float *t,
                  int32_t *prim_id, int32_t *ishit)
{
    if (!scene || n < 0)
        return -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; i++) {
        ray_t r = mk_ray(mk3(org[3 * i], org[3 * i + 1], org[3 * i + 2]),
                         mk3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        rec_t rec;
        rec_init(&rec);
        int h = scene_hit(scene, &r, &rec);
        if (t) t[i] = rec.t;
        if (prim_id) prim_id[i] = rec.prim;
        if (ishit) ishit[i] = h;
    }
    return 0;
}

/* Camera::generateRay on explicit raster coordinates (Camera/Camera.cu:16-24). */
int orc_camera_ray(const orc_camera_args *cam_args, int n, const float *xy, float *org, float *dir)
{
    if (!cam_args || n < 0)
        return -1;
    cam_t cam = cam_make(cam_args);
    for (int i = 0; i < n; i++) {
        ray_t r = cam_generate_ray(&cam, xy[2 * i], xy[2 * i + 1]);
        org[3 * i] = r.o.x; org[3 * i + 1] = r.o.y; org[3 * i + 2] = r.o.z;
        dir[3 * i] = r.d.x; dir[3 * i + 1] = r.d.y; dir[3 * i + 2] = r.d.z;
    }
    return 0;
}

/* pathTracer on one explicit camera ray (core/PathTracer.cu:11-92); the RNG
 * stream is that of `pixel_index`, starting at draw number `draw0`. */
int orc_trace_ray(const orc_scene *scene, const float org[3], const float dir[3], int max_depth,
                  uint32_t pixel_index, uint32_t draw0, float rgb[3], float *t, int32_t *prim)
{
    if (!scene)
        return -1;
    ray_t r = mk_ray(mk3(org[0], org[1], org[2]), mk3(dir[0], dir[1], dir[2]));
    rng_t g = { pixel_index, draw0 };
    env_begin(scene);
    tex_begin(scene, NULL);
    orc_counts c = { 0, 0, 0 };
    float ft = ORC_INF;
    int fp = -1;
    f3 col = path_tracer(scene, r, &g, max_depth, &c, &ft, &fp);
    tex_end();
    env_end();
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    if (t) *t = ft;
    if (prim) *prim = fp;
    return 0;
}

/* ---------------------------------------------------------------- post-process */
/* PostProcess/PostProcess.cuh:11-18, 29-37 (HDR = ACES filmic), :39-56 */
static float saturate1(float v) { return fminf(1.0f, fmaxf(0.0f, v)); }
static float aces1(float x)
{
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return saturate1((x * (a * x + b)) / (x * (c * x + d) + e));
}
void orc_tonemap_aces(int n_px, const float *in, float *out)
{
    for (int i = 0; i < n_px; i++) {
        out[4 * i + 0] = aces1(in[4 * i + 0]);
        out[4 * i + 1] = aces1(in[4 * i + 1]);
        out[4 * i + 2] = aces1(in[4 * i + 2]);
        out[4 * i + 3] = 1.0f;
    }
}

/* PostProcess/PostProcess.cuh:59-104 filterKernel: 3x3 window gathered x-major
 * (i = x-1..x+1 outer, j = y-1..y+1 inner), the reference's own exchange sort by
 * luma (descending), output element count/2. */
void orc_median3x3(int width, int height, const float *in, float *out)
{
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            f4 temp[10];
            float ty[10];
            int count = 0;
            for (int i = x - 1; i <= x + 1; i++)
                for (int j = y - 1; j <= y + 1; j++)
                    if (i >= 0 && i < width && j >= 0 && j < height) {
                        const float *c = in + 4 * ((size_t)j * width + i);
                        temp[count] = mk4(c[0], c[1], c[2], c[3]);
                        ty[count++] = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
                    }
            for (int i = 0; i < count; i++)
                for (int j = 0; j < i; j++)
                    if (ty[i] > ty[j]) {
                        float sf = ty[i];
                        f4 s4 = temp[i];
                        ty[i] = ty[j];
                        temp[i] = temp[j];
                        ty[j] = sf;
                        temp[j] = s4;
                    }
            float *o = out + 4 * ((size_t)y * width + x);
            o[0] = temp[count / 2].x;
            o[1] = temp[count / 2].y;
            o[2] = temp[count / 2].z;
            o[3] = temp[count / 2].w;
        }
}
