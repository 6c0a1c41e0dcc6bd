/*
 * rt_luma.h — the luma weights palette 0.4 uses for LinSrgb::into_luma (post_process, main.rs:750).
 * Host only; shared by librt_host.so (rt_post_process) and librt_amd.so (rt_post_process_device).
 */
#ifndef RT_LUMA_H
#define RT_LUMA_H

#include "rt_detmath.h"

namespace rt {

/* palette 0.4 converts LinSrgb -> Luma through Xyz with a matrix it derives at
 * run time, in f32, from the sRGB primaries (as Yxy) and the D65 white point;
 * the luma is the Y row.  Restated from the crate's published algorithm
 * (matrix.rs: rgb_to_xyz_matrix); the crate is not in the image, so the last
 * bits of these coefficients are unverified (DESIGN.md "Parity status"). */
inline void luma_row(float row[3]) {
    struct X3 { float x, y, z; };
    auto from_yxy = [](float x, float y, float luma) {
        X3 r{0.0f, luma, 0.0f};
        if (rtdm::is_normal(y)) {
            r.x = luma * x / y;
            r.z = luma * (1.0f - x - y) / y;
        }
        return r;
    };
    const X3 r = from_yxy(0.6400f, 0.3300f, 0.212656f);
    const X3 g = from_yxy(0.3000f, 0.6000f, 0.715158f);
    const X3 b = from_yxy(0.1500f, 0.0600f, 0.072186f);
    const float a[9] = {r.x, g.x, b.x, r.y, g.y, b.y, r.z, g.z, b.z};
    const float c0 = a[4] * a[8] - a[5] * a[7];
    const float c1 = a[3] * a[8] - a[5] * a[6];
    const float c2 = a[3] * a[7] - a[4] * a[6];
    const float det = a[0] * c0 - a[1] * c1 + a[2] * c2;
    const float c3 = a[1] * a[8] - a[2] * a[7];
    const float c4 = a[0] * a[8] - a[2] * a[6];
    const float c5 = a[0] * a[7] - a[1] * a[6];
    const float c6 = a[1] * a[5] - a[2] * a[4];
    const float c7 = a[0] * a[5] - a[2] * a[3];
    const float c8 = a[0] * a[4] - a[1] * a[3];
    const float inv[9] = {c0 / det, -c3 / det, c6 / det, -c1 / det, c4 / det, -c7 / det, c2 / det, -c5 / det, c8 / det};
    const float wx = 0.95047f, wy = 1.0f, wz = 1.08883f;
    const float sr = (inv[0] * wx) + (inv[1] * wy) + (inv[2] * wz);
    const float sg = (inv[3] * wx) + (inv[4] * wy) + (inv[5] * wz);
    const float sb = (inv[6] * wx) + (inv[7] * wy) + (inv[8] * wz);
    row[0] = a[3] * sr;
    row[1] = a[4] * sg;
    row[2] = a[5] * sb;
}


} /* namespace rt */

#endif
