/*
 * rt_host.cpp — host side of the render path (see include/rt_host.h).
 *
 * Scene construction, OBJ import, tone normalisation, sRGB encode and PNG
 * output stay on the CPU, as they do in the reference (src/main.rs:748-1083).
 * All f32 arithmetic follows the reference's operation order (rt_vec.h).
 */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../../include/rt_host.h"
#include "../rt_vec.h"
#include "../rt_luma.h"

using rt::V3;

struct rt_world {
    std::vector<rt_material> objects; /* Object { material } — primitives.rs:8-10 */
    std::vector<rt_triangle> triangles;
    std::vector<rt_sphere> spheres;
    std::vector<rt_light> lights;
};

static thread_local std::string g_host_error;
static int fail(int code, const std::string &msg) {
    g_host_error = msg;
    return code;
}

extern "C" {

const char *rt_host_last_error(void) { return g_host_error.c_str(); }

rt_world *rt_world_new(void) { return new (std::nothrow) rt_world(); }
void rt_world_free(rt_world *world) { delete world; }

int rt_world_push_object(rt_world *world, const rt_material *material) {
    if (!world || !material) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_object: null argument");
    world->objects.push_back(*material);
    return (int)world->objects.size() - 1;
}

int rt_world_push_triangle(rt_world *world, uint32_t object_index, const rt_vertex vertices[3]) {
    if (!world || !vertices) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_triangle: null argument");
    if (object_index >= world->objects.size()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_triangle: object index out of range");
    rt_triangle t;
    t.object_index = object_index;
    memcpy(t.vertices, vertices, sizeof t.vertices);
    world->triangles.push_back(t);
    return RT_OK;
}

int rt_world_push_sphere(rt_world *world, uint32_t object_index, const float center[3], float radius) {
    if (!world || !center) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_sphere: null argument");
    if (object_index >= world->objects.size()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_sphere: object index out of range");
    rt_sphere s;
    s.object_index = object_index;
    memcpy(s.center, center, sizeof s.center);
    s.radius = radius;
    world->spheres.push_back(s);
    return RT_OK;
}

int rt_world_push_light(rt_world *world, const rt_light *light) {
    if (!world || !light) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_light: null argument");
    if (light->kind > RT_LIGHT_POINT) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_light: unknown light kind");
    world->lights.push_back(*light);
    return RT_OK;
}

/* triangle(): src/main.rs:730-739 */
int rt_world_push_flat_triangle(rt_world *world, uint32_t object_index, const float positions[9], const float uvs[6]) {
    if (!world || !positions || !uvs) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_flat_triangle: null argument");
    V3 p0 = rt::v3p(positions), p1 = rt::v3p(positions + 3), p2 = rt::v3p(positions + 6);
    V3 a = p1 - p0;
    V3 b = p2 - p1;
    V3 n = rt::normalize(rt::cross(a, b));
    rt_vertex v[3];
    for (int i = 0; i < 3; ++i) {
        memcpy(v[i].position, positions + 3 * i, 3 * sizeof(float));
        v[i].normal[0] = n.x; v[i].normal[1] = n.y; v[i].normal[2] = n.z;
        v[i].uv[0] = uvs[2 * i]; v[i].uv[1] = uvs[2 * i + 1];
    }
    return rt_world_push_triangle(world, object_index, v);
}

/* square(): src/main.rs:741-746 */
int rt_world_push_square(rt_world *world, uint32_t object_index, const float positions[12], const float uvs[8]) {
    if (!world || !positions || !uvs) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_push_square: null argument");
    static const int corner[2][3] = {{0, 1, 2}, {0, 2, 3}};
    for (int t = 0; t < 2; ++t) {
        float p[9], uv[6];
        for (int i = 0; i < 3; ++i) {
            memcpy(p + 3 * i, positions + 3 * corner[t][i], 3 * sizeof(float));
            memcpy(uv + 2 * i, uvs + 2 * corner[t][i], 2 * sizeof(float));
        }
        int rc = rt_world_push_flat_triangle(world, object_index, p, uv);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

/* load_obj: src/main.rs:778-807 (tobj: first model, positions + triangle indices only) */
int rt_world_load_obj(rt_world *world, uint32_t object_index, const char *path, float divisor, const float offset[3]) {
    if (!world || !path || !offset) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_obj: null argument");
    FILE *f = fopen(path, "r");
    if (!f) return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_load_obj: cannot open ") + path + ": " + strerror(errno));
    std::vector<float> positions;
    std::vector<uint32_t> faces;
    char line[512];
    int models_seen = 0;
    bool bad = false;
    while (fgets(line, sizeof line, f)) {
        char *s = line;
        while (*s == ' ' || *s == '\t') ++s;
        if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t')) {
            /* tobj starts a new model at each o/g record once faces exist; only models[0] is used */
            if (!faces.empty()) { models_seen = 2; break; }
            models_seen = 1;
        } else if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char *e = s + 1;
            for (int i = 0; i < 3; ++i) {
                char *next = nullptr;
                float v = strtof(e, &next); /* correctly rounded, like Rust's str::parse::<f32> */
                if (next == e) { bad = true; break; }
                positions.push_back(v);
                e = next;
            }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            char *e = s + 1;
            uint32_t idx[3];
            int n = 0;
            while (n < 3) {
                char *next = nullptr;
                long v = strtol(e, &next, 10);
                if (next == e) break;
                /* skip /vt/vn suffixes if present */
                while (*next && *next != ' ' && *next != '\t' && *next != '\n' && *next != '\r') ++next;
                long count = (long)(positions.size() / 3);
                long zero_based = v > 0 ? v - 1 : count + v; /* OBJ allows negative (relative) indices */
                if (zero_based < 0 || zero_based >= count) { bad = true; break; }
                idx[n++] = (uint32_t)zero_based;
                e = next;
            }
            if (n != 3) bad = true;
            if (bad) break;
            faces.insert(faces.end(), idx, idx + 3);
        }
        if (bad) break;
    }
    fclose(f);
    (void)models_seen;
    if (bad) return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_load_obj: malformed record in ") + path);
    int pushed = 0;
    V3 off = rt::v3p(offset);
    for (size_t t = 0; t + 2 < faces.size(); t += 3) {
        float p[9];
        static const float uv[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < 3; ++i) {
            V3 v = rt::v3p(&positions[3 * faces[t + i]]);
            V3 q = v / divisor + off; /* p.position / 3.0 + Vector3::new(0.7, 1.0, -0.5) */
            p[3 * i] = q.x; p[3 * i + 1] = q.y; p[3 * i + 2] = q.z;
        }
        int rc = rt_world_push_flat_triangle(world, object_index, p, uv);
        if (rc != RT_OK) return rc;
        ++pushed;
    }
    return pushed;
}

/* ---- the literal scene of main(): src/main.rs:810-1075 ------------------- */

static rt_material color_material(float dr, float dg, float db, float shiness, float sr, float sg, float sb,
                                  float smoothness, float refraction_index, float opaque_decay, float transparency) {
    rt_material m;
    memset(&m, 0, sizeof m);
    m.diffuse_fn = RT_DIFFUSE_CONST;
    m.normal_fn = RT_NORMAL_CONST;
    m.normal[0] = 0.0f; m.normal[1] = 0.0f; m.normal[2] = 1.0f;
    m.diffuse_color[0] = dr; m.diffuse_color[1] = dg; m.diffuse_color[2] = db;
    m.shiness = shiness;
    m.specular_color[0] = sr; m.specular_color[1] = sg; m.specular_color[2] = sb;
    m.smoothness = smoothness;
    m.transparency = transparency;
    m.refraction_index = refraction_index;
    m.opaque_decay = opaque_decay;
    return m;
}

/* an axis-aligned box of 6 squares given as (corner order, uv order) rows; the
 * two glass slabs of main.rs:892-927 and 942-977 share this vertex pattern */
struct SquareRow {
    float p[12];
    float uv[8];
};

static int push_rows(rt_world *w, uint32_t obj, const SquareRow *rows, int n) {
    for (int i = 0; i < n; ++i) {
        int rc = rt_world_push_square(w, obj, rows[i].p, rows[i].uv);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

/* slab spanning x in [-hx, hx], y in [1.0, 1.5], z in [z0, z1] (z0 < z1), in
 * the face and vertex order of the reference */
static int push_glass_slab(rt_world *w, uint32_t obj, float hx, float z0, float z1) {
    const float ylo = 1.0f, yhi = 1.5f;
    const SquareRow rows[6] = {
        /* +z face */
        {{hx, yhi, z1, -hx, yhi, z1, -hx, ylo, z1, hx, ylo, z1}, {0, 0, 0, 1, 1, 0, 0, 1}},
        /* -z face */
        {{hx, ylo, z0, -hx, ylo, z0, -hx, yhi, z0, hx, yhi, z0}, {0, 1, 1, 0, 0, 1, 0, 0}},
        /* +y face */
        {{hx, yhi, z0, -hx, yhi, z0, -hx, yhi, z1, hx, yhi, z1}, {0, 1, 1, 0, 0, 1, 0, 0}},
        /* fourth and fifth faces differ between the two slabs; filled in by the caller */
        {{0}, {0}},
        {{0}, {0}},
        /* +x face */
        {{hx, ylo, z0, hx, yhi, z0, hx, yhi, z1, hx, ylo, z1}, {0, 1, 1, 0, 0, 1, 0, 0}},
    };
    SquareRow r[6];
    memcpy(r, rows, sizeof r);
    const SquareRow bottom = {{hx, ylo, z1, -hx, ylo, z1, -hx, ylo, z0, hx, ylo, z0}, {0, 1, 1, 0, 0, 1, 0, 0}};
    const SquareRow minus_x = {{-hx, yhi, z0, -hx, ylo, z0, -hx, ylo, z1, -hx, yhi, z1}, {0, 1, 1, 0, 0, 1, 0, 0}};
    if (hx == 0.5f) { /* first slab: ..., -y face (main.rs:910-915), -x face (916-921), +x */
        r[3] = bottom;
        r[4] = minus_x;
    } else {          /* second slab: ..., -x face (main.rs:960-965), -y face (966-971), +x */
        r[3] = minus_x;
        r[4] = bottom;
    }
    return push_rows(w, obj, r, 6);
}

int rt_world_build_reference_scene(rt_world *world, const char *obj_path) {
    if (!world || !obj_path) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_build_reference_scene: null argument");
    int rc;
    /* object 0: the imported dodecahedron (main.rs:812-825) */
    rt_material m0 = color_material(1.0f, 1.0f, 1.0f, 0.1f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f);
    int o0 = rt_world_push_object(world, &m0);
    const float obj_offset[3] = {0.7f, 1.0f, -0.5f};
    rc = rt_world_load_obj(world, (uint32_t)o0, obj_path, 3.0f, obj_offset);
    if (rc < 0) return rc;

    /* object 1: the floor (main.rs:826-844) */
    rt_material m1 = color_material(1.0f, 0.8f, 0.6f, 0.5f, 1.0f, 1.0f, 1.0f, 0.01f, 1.0f, 0.0f, 0.0f);
    int o1 = rt_world_push_object(world, &m1);
    const SquareRow floor_sq = {{-2, 0, -2, -2, 0, 2, 2, 0, 2, 2, 0, -2}, {0, 0, 0, 1, 1, 0, 0, 1}};
    if ((rc = push_rows(world, (uint32_t)o1, &floor_sq, 1)) != RT_OK) return rc;

    /* object 2: the striped, wavy wall — GenerativeMaterial (main.rs:845-877) */
    rt_material m2 = color_material(0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, 0.00001f, 1.0f, 0.0f, 0.0f);
    m2.diffuse_fn = RT_DIFFUSE_STRIPE_V;
    m2.normal_fn = RT_NORMAL_WAVE_U;
    m2.tex_color_a[0] = 1.0f; m2.tex_color_a[1] = 1.0f; m2.tex_color_a[2] = 1.0f;
    m2.tex_color_b[0] = 0.5f; m2.tex_color_b[1] = 0.5f; m2.tex_color_b[2] = 1.0f;
    m2.tex_frequency = 20.0f;
    m2.normal_frequency = 10.0f;
    int o2 = rt_world_push_object(world, &m2);
    const SquareRow wall_sq = {{-2, 2, -2, -2, 2, 2, -2, -2, 2, -2, -2, -2}, {0, 0, 0, 1, 1, 0, 1, 1}};
    if ((rc = push_rows(world, (uint32_t)o2, &wall_sq, 1)) != RT_OK) return rc;

    /* objects 3 and 4: two glass slabs (main.rs:879-977) */
    rt_material glass = color_material(1.0f, 0.8f, 0.6f, 1.0f, 1.0f, 1.0f, 1.0f, 0.00001f, 1.6f, 0.1f, 1.0f);
    int o3 = rt_world_push_object(world, &glass);
    if ((rc = push_glass_slab(world, (uint32_t)o3, 0.5f, 0.6f, 0.7f)) != RT_OK) return rc;
    int o4 = rt_world_push_object(world, &glass);
    if ((rc = push_glass_slab(world, (uint32_t)o4, 0.3f, 0.71f, 0.81f)) != RT_OK) return rc;

    /* object 5: red sphere (main.rs:979-996); yellow = (1,1,0) consts.rs:17 */
    rt_material m5 = color_material(1.0f, 0.2f, 0.2f, 0.2f, 1.0f, 1.0f, 0.0f, 0.2f, 1.0f, 0.0f, 0.0f);
    int o5 = rt_world_push_object(world, &m5);
    const float c5[3] = {-0.5f, 0.5f, 0.5f / rtdm::f_sqrt(3.0f)};
    if ((rc = rt_world_push_sphere(world, (uint32_t)o5, c5, 0.5f)) != RT_OK) return rc;

    /* object 6: clear sphere (main.rs:998-1014) */
    rt_material m6 = color_material(1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 0.001f, 1.12f, 0.3f, 0.96f);
    int o6 = rt_world_push_object(world, &m6);
    const float c6[3] = {0.5f, 0.5f, 0.5f / rtdm::f_sqrt(3.0f)};
    if ((rc = rt_world_push_sphere(world, (uint32_t)o6, c6, 0.5f)) != RT_OK) return rc;

    /* object 7: checkered sphere — GenerativeMaterial (main.rs:1016-1038); blue = (0,0,1) consts.rs:15 */
    rt_material m7 = color_material(0.0f, 0.0f, 0.0f, 0.3f, 0.0f, 0.0f, 1.0f, 0.7f, 1.0f, 0.0f, 0.0f);
    m7.diffuse_fn = RT_DIFFUSE_STRIPE_SUM;
    m7.normal_fn = RT_NORMAL_CONST;
    m7.tex_color_a[0] = 1.0f; m7.tex_color_a[1] = 0.1f; m7.tex_color_a[2] = 0.1f;
    m7.tex_color_b[0] = 0.1f; m7.tex_color_b[1] = 0.1f; m7.tex_color_b[2] = 1.0f;
    m7.tex_frequency = 10.0f;
    int o7 = rt_world_push_object(world, &m7);
    const float c7[3] = {0.0f, 0.5f, -1.0f / rtdm::f_sqrt(3.0f)};
    if ((rc = rt_world_push_sphere(world, (uint32_t)o7, c7, 0.5f)) != RT_OK) return rc;

    /* object 8: green sphere on top (main.rs:1040-1056) */
    rt_material m8 = color_material(0.5f, 1.0f, 0.2f, 0.5f, 1.0f, 1.0f, 1.0f, 0.01f, 1.0f, 0.0f, 0.0f);
    int o8 = rt_world_push_object(world, &m8);
    const float c8[3] = {0.0f, 0.5f + rtdm::f_sqrt(2.0f / 3.0f), 0.0f};
    if ((rc = rt_world_push_sphere(world, (uint32_t)o8, c8, 0.5f)) != RT_OK) return rc;

    /* lights (main.rs:1058-1075) */
    rt_light l;
    memset(&l, 0, sizeof l);
    l.kind = RT_LIGHT_DIRECTIONAL;
    l.has_origin = 0;
    V3 d0 = rt::normalize(rt::v3(-1.0f, -1.0f, 0.0f));
    l.direction[0] = d0.x; l.direction[1] = d0.y; l.direction[2] = d0.z;
    l.color[0] = 1.0f; l.color[1] = 0.98f; l.color[2] = 0.95f;
    if ((rc = rt_world_push_light(world, &l)) != RT_OK) return rc;

    memset(&l, 0, sizeof l);
    l.kind = RT_LIGHT_SPOT;
    l.has_origin = 1;
    l.origin[0] = 0.0f; l.origin[1] = 10.0f; l.origin[2] = 0.0f;
    V3 d1 = rt::normalize(rt::v3(0.0f, -1.0f, -0.0f));
    l.direction[0] = d1.x; l.direction[1] = d1.y; l.direction[2] = d1.z;
    l.angle = 60.0f * (float)(3.14159265358979323846 / 180.0); /* cgmath Deg -> Rad */
    l.softness = 1.0f;
    l.color[0] = 1.0f * 1.0f; l.color[1] = 0.5f * 1.0f; l.color[2] = 0.9f * 1.0f;
    if ((rc = rt_world_push_light(world, &l)) != RT_OK) return rc;

    memset(&l, 0, sizeof l);
    l.kind = RT_LIGHT_POINT;
    l.has_origin = 1;
    l.origin[0] = 0.0f; l.origin[1] = 0.1f; l.origin[2] = 0.0f;
    l.color[0] = 0.8f; l.color[1] = 0.8f; l.color[2] = 1.0f;
    if ((rc = rt_world_push_light(world, &l)) != RT_OK) return rc;
    return RT_OK;
}

/* main.rs:1077-1083 */
void rt_reference_camera(rt_camera *out) {
    if (!out) return;
    out->fovy = 60.0f * (float)(3.14159265358979323846 / 180.0);
    out->center[0] = 2.0f; out->center[1] = 2.5f; out->center[2] = 2.0f;
    V3 t = rt::normalize(rt::v3(-1.0f, -1.0f, -1.0f));
    out->toward[0] = t.x; out->toward[1] = t.y; out->toward[2] = t.z;
    V3 u = rt::normalize(rt::v3(0.0f, 1.0f, 0.0f));
    out->up[0] = u.x; out->up[1] = u.y; out->up[2] = u.z;
    out->near = -0.1f;
}

void rt_world_desc(const rt_world *world, rt_scene_desc *out) {
    if (!world || !out) return;
    out->triangles = world->triangles.data(); out->n_triangles = (uint32_t)world->triangles.size();
    out->spheres = world->spheres.data();     out->n_spheres = (uint32_t)world->spheres.size();
    out->materials = world->objects.data();   out->n_materials = (uint32_t)world->objects.size();
    out->lights = world->lights.data();       out->n_lights = (uint32_t)world->lights.size();
}

/* ---- the scene as a data format -----------------------------------
 * One flat little-endian file: a 128-byte header, then the four arrays of rt_scene_desc exactly as they sit in memory
 * (rt_material[], rt_triangle[], rt_sphere[], rt_light[]: include/rt_amd.h).  Array order is kept (ties and the light
 * sum depend on it).  The header records each record's size so that a reader built against another ABI revision
 * refuses the file instead of misreading it. */
struct SceneFileHeader {
    char magic[8];          /* "RTSCENE\0" */
    uint32_t version;       /* 1 */
    uint32_t endian;        /* 0x01020304 as written by the producer */
    uint32_t n_materials, n_triangles, n_spheres, n_lights;
    uint32_t sizeof_material, sizeof_triangle, sizeof_sphere, sizeof_light;
    uint32_t has_camera;
    rt_camera camera;       /* 11 f32 */
    uint32_t reserved[8];
};
static_assert(sizeof(SceneFileHeader) == 128, "scene file header");
static const char SCENE_MAGIC[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', 0};

int rt_world_save_scene(const rt_world *world, const rt_camera *camera, const char *path) {
    if (!world || !path) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_save_scene: null argument");
    SceneFileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, SCENE_MAGIC, 8);
    h.version = 1u;
    h.endian = 0x01020304u;
    h.n_materials = (uint32_t)world->objects.size();
    h.n_triangles = (uint32_t)world->triangles.size();
    h.n_spheres = (uint32_t)world->spheres.size();
    h.n_lights = (uint32_t)world->lights.size();
    h.sizeof_material = (uint32_t)sizeof(rt_material);
    h.sizeof_triangle = (uint32_t)sizeof(rt_triangle);
    h.sizeof_sphere = (uint32_t)sizeof(rt_sphere);
    h.sizeof_light = (uint32_t)sizeof(rt_light);
    if (camera) { h.has_camera = 1u; h.camera = *camera; }
    const std::string tmp = std::string(path) + ".tmp"; /* written beside the target, then renamed over it (as write_to_file) */
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_save_scene: cannot open ") + tmp + ": " + strerror(errno));
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    auto put = [&](const void *data, size_t size, size_t n) { if (ok && n) ok = fwrite(data, size, n, f) == n; };
    put(world->objects.data(), sizeof(rt_material), world->objects.size());
    put(world->triangles.data(), sizeof(rt_triangle), world->triangles.size());
    put(world->spheres.data(), sizeof(rt_sphere), world->spheres.size());
    put(world->lights.data(), sizeof(rt_light), world->lights.size());
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) {
        remove(tmp.c_str());
        return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_save_scene: write failed for ") + path);
    }
    return RT_OK;
}

int rt_world_load_scene(rt_world *world, const char *path, rt_camera *out_camera, int *out_has_camera) {
    if (!world || !path) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_scene: null argument");
    if (out_has_camera) *out_has_camera = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_load_scene: cannot open ") + path + ": " + strerror(errno));
    SceneFileHeader h;
    auto bail = [&](const char *why) { fclose(f); return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_world_load_scene: ") + why + " in " + path); };
    if (fread(&h, sizeof h, 1, f) != 1) return bail("short header");
    if (memcmp(h.magic, SCENE_MAGIC, 8) != 0) return bail("not a scene file (bad magic)");
    if (h.endian != 0x01020304u) return bail("written on a machine of the other byte order");
    if (h.version != 1u) { fclose(f); return fail(RT_ERR_UNSUPPORTED, std::string("rt_world_load_scene: unknown version in ") + path); }
    if (h.sizeof_material != sizeof(rt_material) || h.sizeof_triangle != sizeof(rt_triangle) || h.sizeof_sphere != sizeof(rt_sphere) ||
        h.sizeof_light != sizeof(rt_light)) {
        fclose(f);
        return fail(RT_ERR_UNSUPPORTED, std::string("rt_world_load_scene: record sizes of another ABI revision in ") + path);
    }
    const uint64_t body = (uint64_t)h.n_materials * sizeof(rt_material) + (uint64_t)h.n_triangles * sizeof(rt_triangle) +
                          (uint64_t)h.n_spheres * sizeof(rt_sphere) + (uint64_t)h.n_lights * sizeof(rt_light);
    if (fseek(f, 0, SEEK_END) != 0) return bail("cannot seek");
    const long end = ftell(f);
    if (end < 0 || (uint64_t)end != sizeof h + body) return bail("file size does not match the counts in its header");
    if (fseek(f, (long)sizeof h, SEEK_SET) != 0) return bail("cannot seek");
    rt_world fresh; /* the world is replaced only when the whole file is good */
    bool ok = true;
    try {
        fresh.objects.resize(h.n_materials);
        fresh.triangles.resize(h.n_triangles);
        fresh.spheres.resize(h.n_spheres);
        fresh.lights.resize(h.n_lights);
    } catch (...) {
        fclose(f);
        return fail(RT_ERR_OUT_OF_MEMORY, "rt_world_load_scene: out of memory");
    }
    auto get = [&](void *data, size_t size, size_t n) { if (ok && n) ok = fread(data, size, n, f) == n; };
    get(fresh.objects.data(), sizeof(rt_material), fresh.objects.size());
    get(fresh.triangles.data(), sizeof(rt_triangle), fresh.triangles.size());
    get(fresh.spheres.data(), sizeof(rt_sphere), fresh.spheres.size());
    get(fresh.lights.data(), sizeof(rt_light), fresh.lights.size());
    if (!ok) return bail("short read");
    fclose(f);
    for (const rt_material &m : fresh.objects)
        if (m.diffuse_fn > RT_DIFFUSE_STRIPE_SUM || m.normal_fn > RT_NORMAL_WAVE_U) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_scene: unknown material function");
    for (const rt_triangle &t : fresh.triangles)
        if (t.object_index >= h.n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_scene: triangle object index out of range");
    for (const rt_sphere &sp : fresh.spheres)
        if (sp.object_index >= h.n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_scene: sphere object index out of range");
    for (const rt_light &l : fresh.lights)
        if (l.kind > RT_LIGHT_POINT) return fail(RT_ERR_INVALID_ARGUMENT, "rt_world_load_scene: unknown light kind");
    world->objects.swap(fresh.objects);
    world->triangles.swap(fresh.triangles);
    world->spheres.swap(fresh.spheres);
    world->lights.swap(fresh.lights);
    if (h.has_camera && out_camera) *out_camera = h.camera;
    if (out_has_camera) *out_has_camera = h.has_camera ? 1 : 0;
    return RT_OK;
}

void rt_frame_full(uint32_t width, uint32_t height, int32_t max_depth, rt_frame *out) {
    if (!out) return;
    out->width = width; out->height = height; out->max_depth = max_depth;
    out->x0 = 0; out->y0 = 0; out->x1 = width; out->y1 = height; out->y_step = 1;
}

/* ---- post_process / encode / PNG ------------------------------------------ */

/* the luma weights post_process uses (palette's LinSrgb::into_luma: the Y row of its rgb -> xyz matrix): for callers that take the
 * percentile themselves (dist.post_process_sharded: the p99 of a frame whose rows live on several ranks) */
void rt_luma_row(float *row3) {
    if (row3) rt::luma_row(row3);
}

/* main.rs:748-762 */
float rt_post_process(float *rgb, size_t n_pixels) {
    if (!rgb || n_pixels == 0) return 0.0f;
    float row[3];
    rt::luma_row(row);
    std::vector<float> lum;
    lum.reserve(n_pixels);
    for (size_t i = 0; i < n_pixels; ++i) {
        const float l = (row[0] * rgb[3 * i]) + (row[1] * rgb[3 * i + 1]) + (row[2] * rgb[3 * i + 2]);
        if (rtdm::is_normal(l)) lum.push_back(l);
    }
    if (lum.empty()) return 0.0f;
    size_t idx = (size_t)((float)lum.size() * 0.99f);
    if (idx >= lum.size()) idx = lum.size() - 1;
    /* the k-th smallest of a sorted copy == nth_element; order-independent, so bit-exact */
    std::nth_element(lum.begin(), lum.begin() + (ptrdiff_t)idx, lum.end());
    const float p98 = lum[idx];
    if (p98 > RT_F_EPSILON) {
        for (size_t i = 0; i < n_pixels * 3; ++i) rgb[i] = rgb[i] / p98;
        return p98;
    }
    return 0.0f;
}

/* image.rs:55-66: Linear -> sRGB transfer, then f32 -> u8 (x*255, clamp, truncate) */
/* photon.rs:25-28, once per surviving sample, epoch by epoch */
void rt_accumulate(const float *samples, const uint8_t *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight) {
    for (uint32_t e = 0; e < n_epochs; ++e) {
        const float *s = samples + (size_t)e * n_pixels * 3;
        const uint8_t *v = valid + (size_t)e * n_pixels;
        for (size_t i = 0; i < n_pixels; ++i) {
            if (!v[i]) continue;
            sum[3 * i] = sum[3 * i] + s[3 * i];
            sum[3 * i + 1] = sum[3 * i + 1] + s[3 * i + 1];
            sum[3 * i + 2] = sum[3 * i + 2] + s[3 * i + 2];
            weight[i] += 1.0f;
        }
    }
}

/* photon.rs:15-23 */
void rt_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb) {
    for (size_t i = 0; i < n_pixels; ++i) {
        const bool empty = weight[i] < 1.1920928955078125e-7f; /* std::f32::EPSILON */
        rgb[3 * i] = empty ? 0.0f : sum[3 * i] / weight[i];
        rgb[3 * i + 1] = empty ? 0.0f : sum[3 * i + 1] / weight[i];
        rgb[3 * i + 2] = empty ? 0.0f : sum[3 * i + 2] / weight[i];
    }
}

void rt_encode_srgb8(const float *rgb, size_t n_values, uint8_t *out) {
    if (!rgb || !out) return;
    for (size_t i = 0; i < n_values; ++i) {
        const float x = rgb[i];
        const float e = (x <= 0.0031308f) ? 12.92f * x : 1.055f * rtdm::powf(x, 1.0f / 2.4f) - 0.055f;
        /* palette 0.4 f32 -> u8: scale, clamp, truncating `as u8` (not rounding; the reference's
         * own output image pins this, see tests/test_oracle_reference_png.py) */
        float r = e * 255.0f;
        if (!(r > 0.0f)) r = 0.0f; /* also NaN -> 0 (saturating `as u8`) */
        if (r > 255.0f) r = 255.0f;
        out[i] = (uint8_t)r;
    }
}

static void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t> &png, const char type[4], const uint8_t *data, size_t n) {
    put_be32(png, (uint32_t)n);
    size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, png.data() + start, (uInt)(png.size() - start));
    put_be32(png, crc);
}

/* main.rs:764-776: encode RGB8 PNG to a temporary file, then rename over the target */
int rt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8 || width == 0 || height == 0) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_png: bad argument");
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0); /* filter type None */
        const uint8_t *row = rgb8 + (size_t)y * width * 3;
        raw.insert(raw.end(), row, row + (size_t)width * 3);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_png: deflate failed");
    std::vector<uint8_t> png;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    png.insert(png.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.push_back(8); /* bit depth */
    ihdr.push_back(2); /* colour type RGB */
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(png, "IDAT", z.data(), zn);
    put_chunk(png, "IEND", nullptr, 0);
    std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_png: cannot create " + tmp + ": " + strerror(errno));
    size_t wr = fwrite(png.data(), 1, png.size(), f);
    if (fclose(f) != 0 || wr != png.size()) return fail(RT_ERR_INVALID_ARGUMENT, "rt_write_png: short write to " + tmp);
    if (rename(tmp.c_str(), path) != 0) return fail(RT_ERR_INVALID_ARGUMENT, std::string("rt_write_png: rename failed: ") + strerror(errno));
    return RT_OK;
}

} /* extern "C" */
