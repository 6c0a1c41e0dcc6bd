/*
 * rt_vec.h — f32 vector / colour arithmetic in the operation order of the
 * crates the reference uses (cgmath 0.16 Vector3/Point3, palette 0.4 LinSrgb).
 * Host + device.  Nothing here may be contracted to FMA or reassociated: build
 * with -ffp-contract=off (see csrc/Makefile).
 *
 *   dot        (a.x*b.x + a.y*b.y) + a.z*b.z          cgmath InnerSpace::dot = elementwise product, summed x,y,z
 *   cross      (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
 *   magnitude  sqrt(dot(a,a))
 *   normalize  a * (1 / magnitude(a))                   cgmath normalize_to(1)
 *   distance   magnitude(other - self)                  cgmath MetricSpace for Point3
 *   a / s      component-wise division (not multiply by reciprocal)
 */
#ifndef RT_VEC_H
#define RT_VEC_H

#include "rt_detmath.h"

namespace rt {

struct V3 {
    float x, y, z;
};

RT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 v3p(const float *p) { return v3(p[0], p[1], p[2]); }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); } /* palette Rgb * Rgb */
RT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD float magnitude2(V3 a) { return dot(a, a); }
RT_HD float magnitude(V3 a) { return rtdm::f_sqrt(magnitude2(a)); }
RT_HD V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }
RT_HD float distance(V3 self, V3 other) { return magnitude(other - self); }

#define RT_F_PI 3.14159265358979323846f      /* std::f32::consts::PI */
#define RT_F_EPSILON 1.1920928955078125e-7f  /* std::f32::EPSILON */

} /* namespace rt */

#endif /* RT_VEC_H */
