/*
 * rt_api_layout.hip — from the ABI arrays to the device records (rt_device_scene.h): per-triangle precompute in the reference's
 * operation order, plane sharing, bounding circles, the node tree with its clusters and the pair-wise dealing.  Host only: no
 * HIP call in here, so rt_scene_describe_nodes lets a test look at the node array without a GPU (tests/test_host_logic.py).
 * Called by rt_scene_create (rt_api.hip).
 */
#include "rt_api_internal.h"

int layout_scene(const rt_scene_desc *desc, SceneLayout &layout) {
    if ((desc->n_triangles && !desc->triangles) || (desc->n_spheres && !desc->spheres) || (desc->n_materials && !desc->materials) ||
        (desc->n_lights && !desc->lights))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: null array with non-zero count");
    if ((uint64_t)desc->n_triangles + desc->n_spheres >= 0x1fffffffull)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: too many primitives");
    if (desc->n_triangles > RT_MAX_TRIANGLES)
        return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: more than 2^24 triangles (the path is brute force by definition: one cast tests them all)");
    for (uint32_t i = 0; i < desc->n_triangles; ++i)
        if (desc->triangles[i].object_index >= desc->n_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: triangle object_index out of range");
    for (uint32_t i = 0; i < desc->n_spheres; ++i)
        if (desc->spheres[i].object_index >= desc->n_materials)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: sphere object_index out of range");
    for (uint32_t i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].kind > RT_LIGHT_POINT) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: unknown light kind");
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        if (desc->materials[i].diffuse_fn > RT_DIFFUSE_STRIPE_SUM || desc->materials[i].normal_fn > RT_NORMAL_WAVE_U)
            return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_create: unknown material function");

    using rt::V3;
    std::vector<rt::DevTri> &tris = layout.tris;
    std::vector<rt::DevTriAttr> &attrs = layout.attrs;
    tris.assign(desc->n_triangles, rt::DevTri());
    attrs.assign(desc->n_triangles, rt::DevTriAttr());
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        const rt_triangle &s = desc->triangles[i];
        rt::DevTri &t = tris[i];
        rt::DevTriAttr &a = attrs[i];
        memset(&t, 0, sizeof t);
        memset(&a, 0, sizeof a);
        const V3 v0 = rt::v3p(s.vertices[0].position), v1 = rt::v3p(s.vertices[1].position), v2 = rt::v3p(s.vertices[2].position);
        /* Triangle::face_normal, primitives.rs:36-42 */
        const V3 n = rt::normalize(rt::cross(v1 - v0, v2 - v1));
        t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
        t.d = rt::dot(n, v0); /* main.rs:203 */
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.v1[0] = v1.x; t.v1[1] = v1.y; t.v1[2] = v1.z;
        t.v2[0] = v2.x; t.v2[1] = v2.y; t.v2[2] = v2.z;
        t.obj = s.object_index;
        const V3 e0 = v2 - v1, e1 = v0 - v2, e2 = v1 - v0; /* main.rs:219-221 */
        t.e0[0] = e0.x; t.e0[1] = e0.y; t.e0[2] = e0.z;
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
        t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
        t.area = rt::dot(rt::cross(v1 - v0, v2 - v0), n); /* main.rs:235 */
        for (int k = 0; k < 3; ++k) {
            a.n0[k] = s.vertices[0].normal[k];
            a.n1[k] = s.vertices[1].normal[k];
            a.n2[k] = s.vertices[2].normal[k];
        }
        a.uv0x = s.vertices[0].uv[0]; a.uv0y = s.vertices[0].uv[1];
        a.uv1x = s.vertices[1].uv[0]; a.uv1y = s.vertices[1].uv[1];
        a.uv2x = s.vertices[2].uv[0]; a.uv2y = s.vertices[2].uv[1];
    }
    /* bounding spheres for the conservative rejection in the intersection loop (rt_device_scene.h) */
    double &scene_extent = layout.scene_extent;
    scene_extent = 0.0;
    for (uint32_t i = 0; i < desc->n_triangles; ++i)
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const double a = fabs((double)desc->triangles[i].vertices[v].position[k]);
                if (a > scene_extent) scene_extent = a; /* NaN never compares greater */
            }
    const bool filter_off = getenv("RT_AMD_NO_SPHERE_FILTER") != nullptr; /* A/B switch; results are the same either way */
    const char *frac_env = getenv("RT_AMD_FILTER_MAX_FRAC");
    /* a triangle as large as the scene rejects next to nothing: not worth its ten instructions */
    const double max_frac = (frac_env && *frac_env) ? atof(frac_env) : 0.5;
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        rt::DevTri &t = tris[i];
        t.bq = std::numeric_limits<float>::infinity();
        t.bcx = t.bcy = t.bcz = 0.0f;
        double P[3][3];
        bool finite = true;
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                P[v][k] = (double)desc->triangles[i].vertices[v].position[k];
                finite = finite && std::isfinite(P[v][k]);
            }
        if (!finite || filter_off || !(scene_extent <= 1e10)) continue;
        auto sub = [](const double *a, const double *b, double *o) { for (int k = 0; k < 3; ++k) o[k] = a[k] - b[k]; };
        auto dotd = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
        double ab[3], ac[3], bc[3];
        sub(P[1], P[0], ab); sub(P[2], P[0], ac); sub(P[2], P[1], bc);
        const double la = dotd(bc, bc), lb = dotd(ac, ac), lc = dotd(ab, ab); /* squared sides opposite A, B, C */
        if (!(la > 0.0 && lb > 0.0 && lc > 0.0)) continue;
        /* smallest angle, from sin and cos at each vertex */
        double cr[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
        const double twice_area = sqrt(dotd(cr, cr));
        const double angA = atan2(twice_area, dotd(ab, ac));
        const double angB = atan2(twice_area, -dotd(ab, bc));
        const double angC = atan2(twice_area, dotd(ac, bc));
        const double ang_min = angA < angB ? (angA < angC ? angA : angC) : (angB < angC ? angB : angC);
        if (!(ang_min >= 0.0201)) continue; /* sin(angle/2) >= 0.01 */
        double c[3], r2;
        if (la >= lb + lc) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[1][k] + P[2][k]); r2 = 0.25 * la; }
        else if (lb >= la + lc) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[0][k] + P[2][k]); r2 = 0.25 * lb; }
        else if (lc >= la + lb) { for (int k = 0; k < 3; ++k) c[k] = 0.5 * (P[0][k] + P[1][k]); r2 = 0.25 * lc; }
        else { /* acute: circumcentre */
            const double wa = la * (lb + lc - la), wb = lb * (lc + la - lb), wc = lc * (la + lb - lc);
            const double w = wa + wb + wc;
            for (int k = 0; k < 3; ++k) c[k] = (wa * P[0][k] + wb * P[1][k] + wc * P[2][k]) / w;
            double d0[3];
            sub(P[0], c, d0);
            r2 = dotd(d0, d0);
        }
        /* the sphere must contain the three vertices whatever the rounding above did */
        for (int v = 0; v < 3; ++v) {
            double dv[3];
            sub(P[v], c, dv);
            const double q = dotd(dv, dv);
            if (q > r2) r2 = q;
        }
        const double radius = sqrt(r2);
        if (!(radius <= max_frac * scene_extent)) continue;
        if (!(radius >= 1e-3 * scene_extent) || !std::isfinite(radius)) continue; /* tiny against the scene: p - c would cancel */
        t.bcx = (float)c[0]; t.bcy = (float)c[1]; t.bcz = (float)c[2];
        /* 1.05 R^2, plus the float rounding of the centre (<= 1e-7 * extent per axis, far inside the margin), rounded up */
        t.bq = std::nextafter((float)(1.05 * r2 * 1.0001), std::numeric_limits<float>::infinity());
    }
    /* The triangles as NODES for the intersection loop (rt_device_scene.h "segments"): a pre-order array of leaves (runs of
     * consecutive triangles) and inner nodes over them, each with a skip pointer.  A run of one object's >= 8 triangles, all of
     * which qualify for their own bounding-sphere rejection, becomes a tree: leaves of RT_LEAF_TRIANGLES, grouped 16 by 16;
     * every node that is small against the scene gets a bounding sphere and either up to 8 representative face normals or a
     * normal cone, and can then be skipped by a wave none of whose rays can hit anything in it.  Everything else is a plain
     * leaf that is always visited. */
    std::vector<rt::DevSegment> &segments = layout.segments;
    segments.clear();
    {
        const bool clusters_off = filter_off || getenv("RT_AMD_NO_CLUSTERS") != nullptr; /* A/B switch; results are the same either way */
        const bool flat_only = getenv("RT_AMD_NO_HIERARCHY") != nullptr; /* A/B: one cluster per object run, explicit normals only (round 1) */
        uint32_t single_leaf_max = 64u; /* A/B: objects up to this many triangles stay one leaf */
        if (const char *v = getenv("RT_AMD_SINGLE_LEAF_MAX")) { if (*v) single_leaf_max = (uint32_t)atoi(v); }
        /* A plain run may only grow the leaf before it if that leaf is not inside a subtree that is already closed: an inner
         * node's skip_to jumps over everything emitted below it, so triangles appended to a leaf in there would be skipped with
         * it.  merge_barrier = the number of nodes no later run may be merged into (moved whenever a subtree or a tree ends). */
        size_t merge_barrier = 0;
        auto push_plain = [&](uint32_t first, uint32_t count) {
            if (segments.size() > merge_barrier && segments.back().n_normals == 0u && segments.back().count != 0u &&
                segments.back().first + segments.back().count == first) {
                segments.back().count += count; /* adjacent plain runs are one leaf */
                return;
            }
            rt::DevSegment g;
            memset(&g, 0, sizeof g);
            g.first = first;
            g.count = count;
            g.skip_to = (uint32_t)segments.size() + 1u;
            segments.push_back(g);
        };
        /* bounding sphere + steepness data of the triangles [lo, hi); false: the node cannot be skipped */
        auto node_stats = [&](uint32_t lo_t, uint32_t hi_t, rt::DevSegment *g) -> bool {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (uint32_t k = lo_t; k < hi_t; ++k)
                for (int v = 0; v < 3; ++v)
                    for (int a = 0; a < 3; ++a) {
                        const double x = (double)desc->triangles[k].vertices[v].position[a];
                        if (x < lo[a]) lo[a] = x;
                        if (x > hi[a]) hi[a] = x;
                    }
            for (int a = 0; a < 3; ++a) g->c[a] = (float)(0.5 * (lo[a] + hi[a]));
            double r2 = 0.0;
            for (uint32_t k = lo_t; k < hi_t; ++k) { /* the sphere must contain every triangle's own bounding sphere */
                const double dx = (double)tris[k].bcx - g->c[0], dy = (double)tris[k].bcy - g->c[1], dz = (double)tris[k].bcz - g->c[2];
                const double reach = sqrt(dx * dx + dy * dy + dz * dz) + sqrt((double)tris[k].bq);
                if (reach * reach > r2) r2 = reach * reach;
            }
            const double radius = sqrt(r2);
            if (!(std::isfinite(radius) && radius <= max_frac * scene_extent && radius >= 1e-3 * scene_extent)) return false;
            /* (1.05 R)^2 with R already holding the triangles' own 1.05 margins: generous, and rounded up */
            g->r2_hi = std::nextafter((float)(r2 * 1.0001), std::numeric_limits<float>::infinity());
            /* one representative per face plane direction: sign canonicalised, merged within 1e-4 per component */
            bool explicit_ok = true;
            g->n_normals = 0u;
            double mean[3] = {0.0, 0.0, 0.0};
            for (uint32_t k = lo_t; k < hi_t; ++k) {
                float n[3] = {tris[k].n[0], tris[k].n[1], tris[k].n[2]};
                if (!(std::isfinite(n[0]) && std::isfinite(n[1]) && std::isfinite(n[2]))) return false;
                const int lead = fabsf(n[0]) > 1e-3f ? 0 : (fabsf(n[1]) > 1e-3f ? 1 : 2);
                if (n[lead] < 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
                bool known = false;
                for (uint32_t q = 0; q < g->n_normals && !known; ++q)
                    known = fabsf(g->normals[q][0] - n[0]) <= 1e-4f && fabsf(g->normals[q][1] - n[1]) <= 1e-4f && fabsf(g->normals[q][2] - n[2]) <= 1e-4f;
                if (!known && explicit_ok) {
                    if (g->n_normals == RT_SEGMENT_NORMALS) explicit_ok = false;
                    else {
                        g->normals[g->n_normals][0] = n[0]; g->normals[g->n_normals][1] = n[1]; g->normals[g->n_normals][2] = n[2];
                        g->n_normals += 1u;
                    }
                }
            }
            if (explicit_ok && g->n_normals != 0u) return true;
            if (flat_only) return false;
            /* More than 8 plane directions: a CONE.  Axis a (unit), half-angle theta >= the angle between a and every face normal
             * or its negative.  For a unit direction d and a unit normal n within theta of +-a:
             *     |n.d| >= |a.d| cos(theta) - sin(theta),
             * so |a.d| >= (1.01e-3 + sin theta) / cos theta =: K implies |n.d| >= 1.01e-3 for every triangle below — the
             * condition the explicit normals test one by one (the 1 % covers the binary32 evaluation of both sides and of the
             * loop's own n.d).  Stored as K^2 for the test (a.d)^2 >= K^2 (d.d); cones of 60 degrees and more are useless. */
            for (uint32_t k = lo_t; k < hi_t; ++k) { /* the axis: mean of the normals, each flipped into the first one's half-space */
                const double s = ((double)tris[k].n[0] * tris[lo_t].n[0] + (double)tris[k].n[1] * tris[lo_t].n[1] + (double)tris[k].n[2] * tris[lo_t].n[2]) < 0.0 ? -1.0 : 1.0;
                for (int a = 0; a < 3; ++a) mean[a] += s * (double)tris[k].n[a];
            }
            const double ml = sqrt(mean[0] * mean[0] + mean[1] * mean[1] + mean[2] * mean[2]);
            if (!(ml > 1e-6)) return false;
            float ax[3];
            for (int a = 0; a < 3; ++a) ax[a] = (float)(mean[a] / ml);
            const double al = sqrt((double)ax[0] * ax[0] + (double)ax[1] * ax[1] + (double)ax[2] * ax[2]); /* of the ROUNDED axis: what the kernel uses */
            double cos_min = 1.0;
            for (uint32_t k = lo_t; k < hi_t; ++k) {
                const double nl = sqrt((double)tris[k].n[0] * tris[k].n[0] + (double)tris[k].n[1] * tris[k].n[1] + (double)tris[k].n[2] * tris[k].n[2]);
                const double c = fabs(((double)tris[k].n[0] * ax[0] + (double)tris[k].n[1] * ax[1] + (double)tris[k].n[2] * ax[2]) / (nl * al));
                if (!(c <= 1.0)) { if (c > 1.0 && c < 1.0 + 1e-9) continue; return false; }
                if (c < cos_min) cos_min = c;
            }
            const double theta = acos(cos_min) + 1e-5; /* slack for everything rounded on the way */
            if (!(theta < 1.0471975511965976)) return false; /* 60 degrees */
            const double K = (1.01e-3 + sin(theta)) / cos(theta) * 1.0001;
            if (!(K < 1.0)) return false;
            g->n_normals = RT_SEGMENT_CONE;
            /* the kernel compares (a.d)^2 with K^2 (d.d) where a is the rounded axis of length al: fold al^2 in, round up */
            g->normals[0][0] = ax[0]; g->normals[0][1] = ax[1]; g->normals[0][2] = ax[2];
            g->normals[0][3] = std::nextafter((float)(K * K * al * al * 1.0001), std::numeric_limits<float>::infinity());
            return true;
        };
        /* pre-order emission of the tree over the leaves [l0, l1) of the object run [run_lo, run_hi) */
        struct Emit {
            static void go(uint32_t l0, uint32_t l1, uint32_t run_lo, uint32_t run_hi, std::vector<rt::DevSegment> &out,
                           const std::function<bool(uint32_t, uint32_t, rt::DevSegment *)> &stats,
                           const std::function<void(uint32_t, uint32_t)> &plain, size_t *barrier) {
                const uint32_t t0 = run_lo + l0 * RT_LEAF_TRIANGLES;
                const uint32_t t1 = std::min<uint64_t>(run_hi, (uint64_t)run_lo + (uint64_t)l1 * RT_LEAF_TRIANGLES);
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                const bool ok = stats(t0, t1, &g);
                if (l1 - l0 == 1u) {
                    if (!ok) { plain(t0, t1 - t0); return; }
                    g.first = t0;
                    g.count = t1 - t0;
                    g.skip_to = (uint32_t)out.size() + 1u;
                    out.push_back(g);
                    return;
                }
                size_t at = (size_t)-1;
                if (ok) { /* an inner node: count 0, skip_to patched once its subtree is out */
                    g.first = t0;
                    g.count = 0u;
                    at = out.size();
                    out.push_back(g);
                }
                uint32_t child = 1u; /* leaves per child: the largest power of 16 below the span */
                while ((uint64_t)child * 16u < (uint64_t)(l1 - l0)) child *= 16u;
                for (uint32_t c0 = l0; c0 < l1; c0 += child) go(c0, std::min(l1, c0 + child), run_lo, run_hi, out, stats, plain, barrier);
                if (at != (size_t)-1) {
                    out[at].skip_to = (uint32_t)out.size();
                    *barrier = out.size(); /* the subtree is closed: nothing may be appended to a leaf inside it */
                }
            }
        };
        for (uint32_t i = 0; i < desc->n_triangles;) {
            uint32_t j = i;
            while (j < desc->n_triangles && desc->triangles[j].object_index == desc->triangles[i].object_index) ++j;
            bool ok = !clusters_off && j - i >= 8u;
            for (uint32_t k = i; ok && k < j; ++k) ok = std::isfinite(tris[k].bq); /* every triangle qualifies for its own rejection */
            if (!ok) {
                push_plain(i, j - i);
            } else if (flat_only || j - i <= single_leaf_max) {
                /* a small object is ONE leaf (the reference scene's dodecahedron: 36 triangles, 6 plane directions — one test per
                 * cast decides it; as a tree of three leaves it cost the bench frame 3 %) */
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                if (node_stats(i, j, &g)) { g.first = i; g.count = j - i; g.skip_to = (uint32_t)segments.size() + 1u; segments.push_back(g); }
                else if (flat_only) push_plain(i, j - i);
                else Emit::go(0u, (j - i + RT_LEAF_TRIANGLES - 1u) / RT_LEAF_TRIANGLES, i, j, segments, node_stats, push_plain, &merge_barrier);
            } else {
                const uint32_t n_leaves = (j - i + RT_LEAF_TRIANGLES - 1u) / RT_LEAF_TRIANGLES;
                Emit::go(0u, n_leaves, i, j, segments, node_stats, push_plain, &merge_barrier);
            }
            i = j;
        }
        /* Neighbouring clustered leaves whose common bounding sphere is hardly larger than the larger of their own become ONE leaf
         * (the reference scene's two glass slabs, main.rs:879-977: 12 + 12 triangles an arm's length apart, the same three plane
         * directions): a ray that needs one nearly always needs the other, and a leaf is a bounding-sphere test, a set of plane
         * directions and — pair-wise — a set-up of its own.  A leaf is any run of consecutive triangles that all qualify for their own
         * rejection, so nothing else changes.  Only leaves with the same ancestors are joined (no subtree ends between them). */
        if (!clusters_off && getenv("RT_AMD_NO_LEAF_MERGE") == nullptr) {
            for (size_t k = 0; k + 1u < segments.size();) {
                const rt::DevSegment a = segments[k], b = segments[k + 1u];
                bool ok = a.count != 0u && b.count != 0u && a.n_normals != 0u && b.n_normals != 0u && a.n_normals != RT_SEGMENT_CONE &&
                          b.n_normals != RT_SEGMENT_CONE && a.first + a.count == b.first && a.count + b.count <= 64u;
                for (size_t j = 0; ok && j < k; ++j) ok = !(segments[j].count == 0u && segments[j].skip_to == k + 1u);
                rt::DevSegment g;
                memset(&g, 0, sizeof g);
                ok = ok && node_stats(a.first, b.first + b.count, &g) && g.n_normals != RT_SEGMENT_CONE &&
                     g.r2_hi <= 1.15f * std::max(a.r2_hi, b.r2_hi);
                if (!ok) { ++k; continue; }
                g.first = a.first;
                g.count = a.count + b.count;
                g.skip_to = (uint32_t)k + 1u;
                segments[k] = g;
                segments.erase(segments.begin() + (ptrdiff_t)k + 1);
                for (rt::DevSegment &n : segments)
                    if (n.skip_to > k + 1u) n.skip_to -= 1u;
                /* and again from the same node: it may take the next one too */
            }
        }
        /* clustered leaves: how their triangles are dealt to the lanes of a pair-wise pass (rt_device_scene.h RT_SEG_PAIR_*) */
        const bool pairs_off = getenv("RT_AMD_NO_PAIRS") != nullptr; /* A/B switch; results are the same either way */
        for (rt::DevSegment &g : segments) {
            if (g.count == 0u || g.n_normals == 0u || pairs_off || g.count > 64u) continue;
            uint32_t best_k = 0u, best_ck = 0u, best_r = 0u;
            double best_fill = 0.0;
            for (uint32_t K = 1u; K <= 8u; ++K) {
                const uint32_t ck = (g.count + K - 1u) / K;
                if (ck < 4u && K > 1u) break;
                const uint32_t R = 64u / ck;
                const double fill = (double)R * g.count / K; /* pairs per full pass */
                if (fill > best_fill * 1.05) { best_fill = fill; best_k = K; best_ck = ck; best_r = R; }
            }
            if (best_k == 0u) continue;
            auto put = [](float *slot, uint32_t v) { memcpy(slot, &v, sizeof v); };
            put(&g.normals[1][3], best_ck | (best_k << 8) | (best_r << 16));
            put(&g.normals[2][3], 65535u / best_ck + 1u);
            put(&g.normals[3][3], 65535u / best_k + 1u);
        }
    }
    /* triangles on their predecessor's plane (rt_device_scene.h RT_TRI_FOLLOWS): same segment; n and d equal bit for bit, or
     * (WEAK) equal up to the signs of zero components */
    if (getenv("RT_AMD_NO_PLANE_SHARING") == nullptr && desc->n_materials <= RT_TRI_OBJ_MASK) { /* A/B switch; results are the same either way */
        const bool weak_ok = getenv("RT_AMD_NO_WEAK_PLANE_SHARING") == nullptr;
        /* any two consecutive triangles of one object: a call of the loop covers consecutive records and treats its first triangle
         * as a leader whatever its flag says, so a pair may straddle leaves */
        for (uint32_t i = 1u; i < desc->n_triangles; ++i) {
            if (desc->triangles[i].object_index != desc->triangles[i - 1u].object_index) continue;
            {
                const float a[4] = {tris[i - 1u].n[0], tris[i - 1u].n[1], tris[i - 1u].n[2], tris[i - 1u].d};
                const float b[4] = {tris[i].n[0], tris[i].n[1], tris[i].n[2], tris[i].d};
                bool exact = true, weak = true;
                for (int k = 0; k < 4; ++k) {
                    const bool same_bits = memcmp(&a[k], &b[k], sizeof(float)) == 0;
                    exact = exact && same_bits;
                    weak = weak && (same_bits || (a[k] == 0.0f && b[k] == 0.0f));
                }
                if (exact) tris[i].obj |= RT_TRI_FOLLOWS;
                else if (weak && weak_ok) tris[i].obj |= RT_TRI_FOLLOWS | RT_TRI_FOLLOWS_WEAK;
            }
        }
    }
    std::vector<rt::DevTriHead> &heads = layout.heads;
    heads.assign(desc->n_triangles, rt::DevTriHead());
    for (uint32_t i = 0; i < desc->n_triangles; ++i) {
        rt::DevTriHead &h = heads[i];
        const rt::DevTri &t = tris[i];
        h.n[0] = t.n[0]; h.n[1] = t.n[1]; h.n[2] = t.n[2]; h.d = t.d;
        h.bc[0] = t.bcx; h.bc[1] = t.bcy; h.bc[2] = t.bcz; h.bq = t.bq;
    }
    std::vector<rt::DevSphere> &spheres = layout.spheres;
    spheres.assign(desc->n_spheres, rt::DevSphere());
    for (uint32_t i = 0; i < desc->n_spheres; ++i) {
        const rt_sphere &s = desc->spheres[i];
        rt::DevSphere &d = spheres[i];
        memset(&d, 0, sizeof d);
        d.c[0] = s.center[0]; d.c[1] = s.center[1]; d.c[2] = s.center[2];
        d.radius = s.radius;
        d.r2 = s.radius * s.radius; /* radius.powi(2), main.rs:272 */
        d.obj = s.object_index;
        d.q_miss = std::numeric_limits<float>::infinity();
        if (std::isfinite(s.radius) && s.radius > 0.0f) {
            const double up = (double)s.radius * (1.0 + 0x1p-22);
            const float q = std::nextafter((float)(up * up), std::numeric_limits<float>::infinity());
            if (std::isfinite(q)) d.q_miss = q;
        }
    }
    return RT_OK;
}

extern "C" {
/* Diagnostics: the node array (rt_device_scene.h) rt_scene_create would build for `desc`, six words per node — first, count,
 * n_normals, skip_to, the pair-wise dealing word, 0 — without touching a device. */
int rt_scene_describe_nodes(const rt_scene_desc *desc, uint32_t *out_words, uint32_t cap_nodes, uint32_t *n_nodes) {
    if (!desc || !n_nodes || (cap_nodes && !out_words)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_scene_describe_nodes: null argument");
    SceneLayout layout;
    const int rc = layout_scene(desc, layout);
    if (rc != RT_OK) return rc;
    *n_nodes = (uint32_t)layout.segments.size();
    for (uint32_t k = 0; k < *n_nodes && k < cap_nodes; ++k) {
        const rt::DevSegment &g = layout.segments[k];
        uint32_t pair_word;
        memcpy(&pair_word, &g.normals[1][3], sizeof pair_word);
        uint32_t *o = out_words + (size_t)k * 6u;
        o[0] = g.first; o[1] = g.count; o[2] = g.n_normals; o[3] = g.skip_to; o[4] = pair_word; o[5] = 0u;
    }
    return RT_OK;
}

} /* extern "C" */
