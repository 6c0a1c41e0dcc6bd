/*
 * rt_oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see
 * the header of rt_oracle.cpp).  Loaded through ctypes by tests/, smoke() and
 * bench.py's cpu_baseline leg; never by the product.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include "../include/rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MATH_SIN = 0, ORC_MATH_COS = 1, ORC_MATH_TAN = 2, ORC_MATH_ACOS = 3, ORC_MATH_ATAN2 = 4, ORC_MATH_POW = 5 };

/* main.rs:69-81 Ray + Exclusion, flattened */
typedef struct orc_ray {
    float origin[3];
    float direction[3];
    uint32_t face_direction; /* 0 Front, 1 Back, 2 Both (main.rs:52-57) */
    uint32_t has_exclude;
    uint32_t exclude_kind;   /* 0 Sphere, 1 Triangle (primitives.rs:31-34) */
    uint32_t exclude_index;
    uint32_t exclude_face;
} orc_ray;

/* main.rs:139-147 Hit, flattened (the incoming ray travels separately) */
typedef struct orc_hit {
    uint32_t kind;  /* 0 Sphere, 1 Triangle */
    uint32_t index;
    uint32_t object_index;
    float position[3];
    float normal[3];
    float uv[2];
    uint32_t face_direction;
    float distance;
} orc_hit;

int orc_uses_libm(void);
void orc_math(int op, const float *x, const float *y, float *out, size_t n);
void orc_clip(uint32_t width, uint32_t height, uint32_t x, uint32_t y, float *clip_xy);
void orc_shoot(const rt_camera *cam, const float *clip_xy, orc_ray *out);
int orc_cast(const rt_scene_desc *scene, const orc_ray *ray, orc_hit *out);
int orc_refract_dir(const float *n, const float *l, float k, float *out);
void orc_reflect(const orc_hit *hit, const orc_ray *incoming, orc_ray *out);
/* returns 0 Escaped, 1 Infinite, 2 Trapped (main.rs:149-158) */
int orc_get_refract(const rt_scene_desc *scene, const orc_hit *hit, const orc_ray *incoming, float max_distance,
                    float *travel, orc_ray *escape);
int orc_light_directional(const rt_light *light, const float *position, float *direction, float *color,
                          float *origin, int *has_origin);
void orc_material_approx(const rt_material *m, const float *uv, float *out14);
void orc_adjust_normal(const float *material_normal, const float *normal, float *out);
void orc_diffuse_specular(const rt_material *m, const float *uv, const float *normal, const float *view,
                          const float *light_dir, float *diffuse, float *specular);
void orc_get_shade(const rt_scene_desc *scene, const orc_hit *hit, const orc_ray *incoming, float *rgb3,
                   uint64_t *casts);
void orc_ray_trace(const rt_scene_desc *scene, const orc_ray *ray, int32_t depth, float contribution, float *rgb3,
                   uint64_t *casts);
void orc_render_whitted(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float *out_rgb,
                        uint64_t *out_casts, int n_threads);
void orc_render_whitted_counts(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float *out_rgb,
                               uint32_t *per_pixel_casts, int n_threads);
float orc_post_process(float *rgb, size_t n_pixels, int luma_mode);
void orc_luma_row(int luma_mode, float *row3);
void orc_encode_srgb8(const float *rgb, size_t n_values, uint8_t *out);
/* PhotonAccumulator (src/photon.rs:9-34): accumulate() for every sample whose flag is set, epoch by epoch; resolve */
void orc_accumulate(const float *samples, const uint8_t *valid, uint32_t n_epochs, size_t n_pixels, float *sum, float *weight);
void orc_accumulator_resolve(const float *sum, const float *weight, size_t n_pixels, float *rgb);

size_t orc_rng_state_words(void);
void orc_rng_init(const rt_frame *frame, uint32_t *states);
void orc_rng_draw_u32(uint32_t *state, uint32_t *out, size_t n);
void orc_rng_draw_normal(uint32_t *state, double mean, double std_dev, double *out, size_t n);
void orc_rng_draw_range_f32(uint32_t *state, float low, float high, float *out, size_t n);
void orc_render_distributed(const rt_scene_desc *scene, const rt_camera *camera, const rt_frame *frame, float focus,
                            float blur, uint32_t *rng_states, uint32_t n_epochs, float *samples, uint8_t *valid,
                            uint64_t *out_casts, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
