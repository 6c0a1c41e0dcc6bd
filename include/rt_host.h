/*
 * rt_host.h — host side of the render path: what stays on the CPU on either
 * side of the C-ABI render call.
 *
 * The reference is a single Rust binary; there is no Rust toolchain in the
 * build image, so the host side is C++ behind these C entry points, mirroring
 * the reference's builder API by name:
 *
 *   World::new / push_object / push_light        src/main.rs:160-178
 *   ObjectProxy::push_triangle(s) / push_sphere  src/main.rs:700-728
 *   triangle() / square() flat-normal helpers    src/main.rs:730-746
 *   load_obj                                     src/main.rs:778-807
 *   the literal scene and camera of main()       src/main.rs:810-1083
 *   (the same scene as a file: rt_world_save_scene / rt_world_load_scene)
 *   post_process                                 src/main.rs:748-762
 *   Image::<Srgb<u8>>::convert_from              src/image.rs:55-66
 *   write_to_file (tmp file + rename)            src/main.rs:764-776
 *
 * Library: librt_host.so (no HIP dependency).
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_world rt_world; /* host-side World under construction */

rt_world *rt_world_new(void);
void rt_world_free(rt_world *world);

/* World::push_object: returns the new ObjectIndex (>= 0) or a negative rt_status. */
int rt_world_push_object(rt_world *world, const rt_material *material);
/* ObjectProxy::push_triangle / push_sphere */
int rt_world_push_triangle(rt_world *world, uint32_t object_index, const rt_vertex vertices[3]);
int rt_world_push_sphere(rt_world *world, uint32_t object_index, const float center[3], float radius);
/* World::push_light */
int rt_world_push_light(rt_world *world, const rt_light *light);

/* triangle(): positions[9] = 3 x xyz, uvs[6] = 3 x uv; the vertex normal of all
 * three vertices is normalize((v1-v0) x (v2-v1)). */
int rt_world_push_flat_triangle(rt_world *world, uint32_t object_index, const float positions[9], const float uvs[6]);
/* square(): 4 corners -> triangles (0,1,2) and (0,2,3). */
int rt_world_push_square(rt_world *world, uint32_t object_index, const float positions[12], const float uvs[8]);

/* load_obj: first model only, `v` and triangular `f` records, no vn/vt; every
 * position becomes p / divisor + offset (the reference uses 3.0 and
 * (0.7, 1.0, -0.5)); uv = (0,0); flat normals via triangle().  Returns the
 * number of triangles pushed or a negative rt_status. */
int rt_world_load_obj(rt_world *world, uint32_t object_index, const char *path, float divisor, const float offset[3]);

/* The whole literal scene of main(): 9 objects, 64 triangles, 4 spheres,
 * 3 lights.  obj_path is the dodecahedron.obj to import. */
int rt_world_build_reference_scene(rt_world *world, const char *obj_path);
/* Camera literal of main.rs:1077-1083. */
void rt_reference_camera(rt_camera *out);

/* View of the world's arrays (valid until the next push / free). */
void rt_world_desc(const rt_world *world, rt_scene_desc *out);

/* The scene as a data format: one flat little-endian file — a 128-byte header ("RTSCENE", version,
 * counts, record sizes, optionally the camera) followed by the arrays of rt_scene_desc exactly as declared in rt_amd.h,
 * in the order materials, triangles, spheres, lights; array order is preserved (it is semantically significant).
 * rt_world_save_scene writes "<path>.tmp" and renames it over path.  rt_world_load_scene REPLACES the world's contents
 * (only if the whole file is valid: magic, version, byte order, record sizes, file length, index and enum ranges);
 * *out_camera is filled and *out_has_camera set when the file carries a camera (both may be NULL). */
int rt_world_save_scene(const rt_world *world, const rt_camera *camera_or_null, const char *path);
int rt_world_load_scene(rt_world *world, const char *path, rt_camera *out_camera, int *out_has_camera);

/* Full-frame rt_frame helper: tile = whole image. */
void rt_frame_full(uint32_t width, uint32_t height, int32_t max_depth, rt_frame *out);

/* post_process: divide the image by the 99th-percentile luma (in place).
 * Returns the divisor used, 0 when the image has no normal luma (the reference
 * panics there, main.rs:754; this returns instead) or the percentile is
 * <= f32::EPSILON (image left untouched, main.rs:755). */
float rt_post_process(float *rgb, size_t n_pixels);
/* its luma weights: luma = (row3[0] * r + row3[1] * g) + row3[2] * b (palette 0.4, LinSrgb::into_luma via main.rs:750) */
void rt_luma_row(float *row3);

/* Linear f32 -> sRGB-encoded u8 (n_values = 3 * pixels). */
void rt_encode_srgb8(const float *rgb, size_t n_values, uint8_t *out);

/* PhotonAccumulator (src/photon.rs:9-34; defined but never used by the reference's main(): SURVEY §8f-4) — a true
 * running average as the alternative to main()'s sum-and-renormalise.  `sum` (3 f32 per pixel) and `weight` (1 f32 per
 * pixel) start at zero.  rt_accumulate applies accumulate() — sum = sum + photon, weight_sum += 1.0 — for every sample
 * whose filter flag is set (`samples`, `valid`: the n_epochs x n_pixels outputs of rt_render_distributed), in epoch
 * order; rt_accumulator_resolve is into_rgb_internal: black while weight < f32::EPSILON, else sum / weight. */
void rt_accumulate(const float *samples, const uint8_t *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight);
void rt_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb);

/* RGB8 PNG, written to "<path>.tmp" then renamed over path. */
int rt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);

const char *rt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
