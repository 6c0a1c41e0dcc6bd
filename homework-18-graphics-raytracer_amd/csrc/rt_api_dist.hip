/*
 * rt_api_dist.hip — the depth-of-field pass behind the C ABI (main.rs:1117-1167): the per-pixel generators (rt_rng_*) and
 * rt_render_distributed — how many epochs make a batch, the workspace(s) a call runs through, and which stream each of a batch's
 * kernels (look-ahead, chain, shade, unwind; rt_distributed.hip) is put on.
 */
#include "rt_api_internal.h"

#ifndef RT_DIST_SPLIT_DEFAULT
#define RT_DIST_SPLIT_DEFAULT 1
#endif
static std::atomic<int> g_dist_split{-1}; /* -1: RT_AMD_DIST_SPLIT or the default */
extern "C" int rt_set_distributed_split(int on) { g_dist_split.store(on < 0 ? -1 : (on > 1 ? 1 : on)); return 0; } /* 2 (round 2's queued chain) is 1 now */

/* ---- timing of the pass's kernels (rt_profile_enable / rt_profile_read_distributed): while profiling is on, every launch is
 * bracketed by an event pair on the stream it is put on — kernels of a pipelined call overlap, so the per-kernel sums add up to
 * more than the call takes; what they give is each kernel's own duration under that overlap (what rocprofv3 --kernel-trace shows) */
enum { DK_PREPARE = 0, DK_CHAIN, DK_SHADE, DK_UNWIND, DK_KINDS };
struct DistProfile {
    std::mutex mutex;
    std::vector<hipEvent_t> events; /* pairs */
    std::vector<int> kind;          /* per pair */
    size_t used = 0;                /* pairs */
};
static DistProfile g_dprof;
void dist_profile_reset() {
    std::lock_guard<std::mutex> lock(g_dprof.mutex);
    g_dprof.used = 0;
}
/* n pairs of consecutive kinds starting at `kind`; false (and no events) when profiling is off or events cannot be made */
static bool dist_profile_pairs(int kind, int n, hipEvent_t *out) {
    if (!profiling_on()) return false;
    std::lock_guard<std::mutex> lock(g_dprof.mutex);
    while (g_dprof.events.size() < 2u * (g_dprof.used + (size_t)n)) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return false; }
        g_dprof.events.push_back(e);
    }
    g_dprof.kind.resize(g_dprof.events.size() / 2u);
    for (int i = 0; i < n; ++i) {
        out[2 * i] = g_dprof.events[2u * (g_dprof.used + (size_t)i)];
        out[2 * i + 1] = g_dprof.events[2u * (g_dprof.used + (size_t)i) + 1u];
        g_dprof.kind[g_dprof.used + (size_t)i] = kind + i;
    }
    g_dprof.used += (size_t)n;
    return true;
}

extern "C" {

int rt_profile_read_distributed(double ms_sum[4], unsigned n_launches[4]) {
    if (!ms_sum || !n_launches) return fail(RT_ERR_INVALID_ARGUMENT, "rt_profile_read_distributed: null argument");
    RT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_dprof.mutex);
    for (int k = 0; k < DK_KINDS; ++k) { ms_sum[k] = 0.0; n_launches[k] = 0u; }
    for (size_t i = 0; i < g_dprof.used; ++i) {
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, g_dprof.events[2u * i], g_dprof.events[2u * i + 1u]));
        ms_sum[g_dprof.kind[i]] += ms;
        n_launches[g_dprof.kind[i]] += 1u;
    }
    g_dprof.used = 0;
    return RT_OK;
}

/* ---- distributed pass ------------------------------------------------------- */

struct rt_rng {
    int device;
    uint32_t *d_states; /* RT_RNG_DEVICE_WORDS per pixel */
    uint32_t *d_list;   /* scratch of the look-ahead pass: 1 + pixels words */
    uint32_t compute_units;
    /* The look-ahead for the NEXT batch runs on a stream of its own next to this batch's shade kernel (nothing after
     * the chain kernel touches the records).  ahead = every pixel has its next block, as of the work enqueued so far. */
    hipStream_t aux;
    hipEvent_t ev_chain, ev_prepared;
    bool ahead;
    /* The shade and unwind kernels of a batch run on a third stream, beside the NEXT batch's chain kernel (two workspaces, used
     * in turn): they fill what its tail leaves idle.  ev_tail[b]: the unwind that read workspace b has finished. */
    hipStream_t tail;
    hipEvent_t ev_tail[2];
    /* the chain kernel's pixels grouped by what their samples cost (rt_kernels.h DistParams::pixel_order): per pixel its cost in the
     * last batch unwound | two orders, one per workspace of a pipelined call | 512 words of scratch */
    uint32_t *d_pix;
    bool order_valid[2];
    hipStream_t main_stream; /* of the call in progress (for the after-chain hook) */
    uint32_t cols, rows, x0, y0, y_step;
};

int rt_rng_state_words(void) { return (int)RT_RNG_STATE_WORDS; }

int rt_rng_create(const rt_frame *frame, rt_rng **out_rng) {
    if (!out_rng) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_create: null argument");
    *out_rng = nullptr;
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_create: bad frame");
    if (!frame_fits(frame)) return fail(RT_ERR_UNSUPPORTED, "rt_rng_create: tile of 2^32 pixels or more");
    rt_rng *r = new (std::nothrow) rt_rng();
    if (!r) return fail(RT_ERR_OUT_OF_MEMORY, "rt_rng_create: host allocation failed");
    r->cols = frame->x1 - frame->x0;
    r->rows = rt_frame_rows(frame);
    r->x0 = frame->x0;
    r->y0 = frame->y0;
    r->y_step = frame->y_step;
    r->d_states = nullptr;
    r->d_list = nullptr;
    r->compute_units = 256;
    r->aux = nullptr;
    r->ev_chain = r->ev_prepared = nullptr;
    r->tail = nullptr;
    r->ev_tail[0] = r->ev_tail[1] = nullptr;
    r->d_pix = nullptr;
    r->order_valid[0] = r->order_valid[1] = false;
    r->ahead = false;
    r->main_stream = nullptr;
    const size_t bytes = (size_t)r->cols * r->rows * RT_RNG_DEVICE_WORDS * sizeof(uint32_t);
    hipError_t e = hipGetDevice(&r->device);
    if (e == hipSuccess) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r->device) == hipSuccess && cus > 0) r->compute_units = (uint32_t)cus;
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_states), bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_list), ((size_t)r->cols * r->rows + 1u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&r->d_pix), ((size_t)r->cols * r->rows * 3u + 512u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(r->d_pix, 0, ((size_t)r->cols * r->rows * 3u + 512u) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_chain, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_prepared, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->tail, hipStreamNonBlocking); /* (at the lowest stream priority: no different, 1 226 against 1 229 Msamples/s) */
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_tail[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_tail[1], hipEventDisableTiming);
    if (e == hipSuccess) {
        rt::KernelFrame kf;
        memset(&kf, 0, sizeof kf);
        kf.cols = r->cols; kf.rows = r->rows; kf.x0 = r->x0; kf.y0 = r->y0; kf.y_step = r->y_step;
        e = rt::launch_rng_seed(r->d_states, kf, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (r->d_states) (void)hipFree(r->d_states);
        if (r->d_list) (void)hipFree(r->d_list);
        if (r->d_pix) (void)hipFree(r->d_pix);
        if (r->ev_chain) (void)hipEventDestroy(r->ev_chain);
        if (r->ev_prepared) (void)hipEventDestroy(r->ev_prepared);
        if (r->aux) (void)hipStreamDestroy(r->aux);
        for (int b = 0; b < 2; ++b) if (r->ev_tail[b]) (void)hipEventDestroy(r->ev_tail[b]);
        if (r->tail) (void)hipStreamDestroy(r->tail);
        delete r;
        return fail_hip("rt_rng_create", e);
    }
    *out_rng = r;
    return RT_OK;
}

int rt_rng_destroy(rt_rng *rng) {
    if (!rng) return RT_OK;
    if (rng->aux) (void)hipStreamSynchronize(rng->aux); /* a look-ahead pass may still be writing the records */
    if (rng->tail) (void)hipStreamSynchronize(rng->tail);
    hipError_t e = rng->d_states ? hipFree(rng->d_states) : hipSuccess;
    if (rng->d_list) (void)hipFree(rng->d_list);
    if (rng->d_pix) (void)hipFree(rng->d_pix);
    if (rng->ev_chain) (void)hipEventDestroy(rng->ev_chain);
    if (rng->ev_prepared) (void)hipEventDestroy(rng->ev_prepared);
    if (rng->aux) (void)hipStreamDestroy(rng->aux);
    for (int b = 0; b < 2; ++b) if (rng->ev_tail[b]) (void)hipEventDestroy(rng->ev_tail[b]);
    if (rng->tail) (void)hipStreamDestroy(rng->tail);
    delete rng;
    if (e != hipSuccess) return fail_hip("rt_rng_destroy: hipFree", e);
    return RT_OK;
}

int rt_rng_download(const rt_rng *rng, uint32_t *h_states) {
    if (!rng || !h_states) return fail(RT_ERR_INVALID_ARGUMENT, "rt_rng_download: null argument");
    /* the device keeps two banks per pixel (the block in use and the next one, generated ahead); what leaves is the
     * reference's record: the bank in use + the position */
    const size_t bytes = (size_t)rng->cols * rng->rows * RT_RNG_STATE_WORDS * sizeof(uint32_t);
    if (bytes == 0) return RT_OK;
    RT_HIP(hipDeviceSynchronize());
    uint32_t *d_tmp = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_tmp), bytes));
    hipError_t e = rt::launch_rng_export(rng->d_states, rng->cols * rng->rows, d_tmp, nullptr);
    if (e == hipSuccess) e = hipMemcpy(h_states, d_tmp, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail_hip("rt_rng_download", e);
    return RT_OK;
}

/* after the chain kernel of a batch: look-ahead for the next batch on the aux stream */
static hipError_t lookahead_after_chain(void *ctx) {
    rt_rng *rng = static_cast<rt_rng *>(ctx);
    hipError_t e = hipEventRecord(rng->ev_chain, rng->main_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(rng->aux, rng->ev_chain, 0);
    hipEvent_t pe[2];
    const bool prof = dist_profile_pairs(DK_PREPARE, 1, pe);
    if (e == hipSuccess && prof) e = hipEventRecord(pe[0], rng->aux);
    if (e == hipSuccess) e = rt::launch_rng_prepare(rng->d_states, rng->cols * rng->rows, rng->d_list, rng->compute_units, rng->aux);
    if (e == hipSuccess && prof) e = hipEventRecord(pe[1], rng->aux);
    if (e == hipSuccess) e = hipEventRecord(rng->ev_prepared, rng->aux);
    if (e == hipSuccess) rng->ahead = true;
    return e;
}

int rt_render_distributed(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                          rt_rng *rng, uint32_t n_epochs, float *d_accum, float *d_samples, unsigned char *d_valid,
                          unsigned long long *d_ray_count, void *hip_stream) {
    if (!scene || !rng) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: null argument");
    if (!d_accum && !d_samples) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: need d_accum or d_samples");
    rt::KernelFrame kf;
    int rc = make_kernel_frame(camera, frame, &kf);
    if (rc != RT_OK) return rc;
    if (kf.cols != rng->cols || kf.rows != rng->rows || kf.x0 != rng->x0 || kf.y0 != rng->y0 || kf.y_step != rng->y_step)
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed: the RNG was created for a different tile");
    rt::DistParams dp;
    dp.rng_states = rng->d_states;
    dp.n_epochs = n_epochs;
    dp.focus = focus;
    dp.blur = blur;
    dp.accum = d_accum;
    dp.samples = d_samples;
    dp.valid = d_valid;
    dp.ray_count = d_ray_count;
    dp.work_queue = nullptr;
    dp.pixel_order = nullptr;
    dp.pixel_cost = nullptr;
    dp.own_first_chunk = 0u;
    dp.bfs_scratch = nullptr;
    dp.bfs_items_cap = dp.bfs_jobs_cap = 0u;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    uint32_t dist_waves = scene->resident_waves;
    int split = g_dist_split.load();
    if (split < 0) split = rt::option(rt::OPT_DIST_SPLIT, RT_DIST_SPLIT_DEFAULT) != 0 ? 1 : 0;
    if (scene->ks.bfs_walk != 0u) split = 0; /* a scene beyond the caches: the one-kernel organisation has the breadth-first walk (below) */
    const size_t n_pixels = (size_t)kf.cols * kf.rows;
    if (n_pixels == 0 || n_epochs == 0) return RT_OK;
    /* the switches (rt_kernels.h Option; rt_set_option or, once per process, the environment) */
    const bool lookahead = rt::option(rt::OPT_RNG_LOOKAHEAD, 1) != 0; /* 0 leaves every IsaacCore::generate to the render kernels */
    const bool overlap = rt::option(rt::OPT_RNG_OVERLAP, 1) != 0;     /* 0 runs the look-ahead in line, before each chain kernel */
    rt_scene *mut = const_cast<rt_scene *>(scene);
    if (split && kf.max_depth <= 254) {
        /* chain / shade / unwind kernels over batches of epochs (rt_distributed.hip "the split pass"); a batch is as
         * many epochs as fit the workspace cap (RT_AMD_DIST_WS_MB, default 32 GiB for the two workspaces of a call of several
         * batches; one epoch at least) and never more than 16 — a visit of a pixel should not need more random words than the block
         * in use plus the one prepared ahead */
        const uint32_t slots = (uint32_t)(kf.max_depth > 0 ? kf.max_depth : 0) + 1u;
        const size_t per_epoch = rt::distributed_split_bytes_per_sample(kf.max_depth) * n_pixels + 4096;
        /* Two workspaces, used in turn, when the call has more than one batch: batch k's shade and unwind kernels then run on a
         * stream of their own beside batch k+1's chain kernel (A/B: RT_AMD_DIST_PIPELINE=0: one workspace, everything in line) */
        const bool pipeline = rt::option(rt::OPT_DIST_PIPELINE, 1) != 0;
        /* the chain kernel's pixels grouped by cost when a lane gets two of them at most (rt_kernels.h DistParams::pixel_order);
         * A/B: RT_AMD_DIST_BY_COST=0 never, =1 always */
        const bool small_share = (n_pixels + 63u) / 64u <= 2u * (size_t)rt::dist_chain_waves(dist_waves);
        const bool by_cost = rt::option(rt::OPT_DIST_BY_COST, small_share ? 1 : 0) != 0;
        dp.own_first_chunk = rt::option(rt::OPT_DIST_OWN_FIRST, small_share ? 1 : 0) != 0 ? 1u : 0u; /* rt_kernels.h */
        const bool prep_first = rt::option(rt::OPT_DIST_PREP_FIRST, 1) != 0; /* 0: shade kernel and look-ahead start together */
        const size_t cap = (size_t)std::max<long long>(0, rt::option(rt::OPT_DIST_WS_MB, pipeline ? 32768 : 16384)) << 20;
        uint32_t batch = (uint32_t)std::min<size_t>(std::min<size_t>(n_epochs, 16), std::max<size_t>(1, cap / per_epoch));
        if (pipeline && batch < n_epochs) /* more than one batch: each workspace gets half the cap */
            batch = (uint32_t)std::min<size_t>(batch, std::max<size_t>(1, cap / 2u / per_epoch));
        if (batch < n_epochs) batch = (n_epochs + (n_epochs + batch - 1u) / batch - 1u) / ((n_epochs + batch - 1u) / batch); /* as many batches, of equal size */
        uint32_t n_buf = pipeline && batch < n_epochs ? 2u : 1u;
        size_t o_hdr = 0, o_req = 0, o_shade = 0, o_frame = 0;
        auto layout = [&](uint32_t epochs) { /* -> bytes of ONE workspace */
            auto carve = [](size_t &off, size_t bytes) { const size_t at = off; off = (off + bytes + 255u) & ~(size_t)255u; return at; };
            const size_t n_samples = n_pixels * epochs;
            size_t off = 0;
            o_hdr = carve(off, n_samples * sizeof(uint32_t));
            o_req = carve(off, n_samples * slots * 4u * sizeof(uint4));
            o_shade = carve(off, n_samples * slots * sizeof(float4));
            o_frame = carve(off, n_samples * (slots - 1u) * sizeof(float4));
            return off;
        };
        auto total_bytes = [&](uint32_t epochs, uint32_t bufs) { return layout(epochs) * bufs; };
        char *base = nullptr;
        size_t buf_stride = 0;
        {
            std::lock_guard<std::mutex> lock(mut->ws_mutex);
            Workspace &ws = mut->workspaces[stream];
            RT_HIP(ensure_counters(ws));
            size_t need = total_bytes(batch, n_buf);
            if (ws.d_split && ws.split_bytes < need) {
                /* A workspace that holds at least half the batch wanted is used as it is: giving back and obtaining
                 * gigabytes costs far more than the shorter batches do (measured: 0.65 s to replace a 14 GB workspace
                 * by a 16 GB one, against 0.15 s for the 64 epochs the call was made for; profiles/README.md) */
                uint32_t fit = batch;
                while (fit > 1u && total_bytes(fit, n_buf) > ws.split_bytes) fit -= 1u;
                if (total_bytes(fit, n_buf) <= ws.split_bytes && fit * 2u >= batch) batch = fit;
                need = total_bytes(batch, n_buf);
            }
            if (ws.split_bytes < need) {
                if (ws.d_split) {
                    RT_HIP(hipStreamSynchronize(stream));
                    RT_HIP(hipFree(ws.d_split));
                    ws.d_split = nullptr;
                    ws.split_bytes = 0;
                }
                /* no room for the batch the cap allows: halve it; no room for one epoch: the one-kernel organisation */
                int refuse = (int)rt::option(rt::OPT_DIAG_WS_REFUSE, 0); /* test hook: pretend the first n allocations fail (tests/test_gpu_distributed_parity.py) */
                while (refuse-- > 0 || hipMalloc(&ws.d_split, need) != hipSuccess) {
                    (void)hipGetLastError();
                    ws.d_split = nullptr;
                    if (batch == 1u && n_buf == 1u) break;
                    if (batch == 1u) n_buf = 1u;
                    else batch = (batch + 1u) / 2u;
                    need = total_bytes(batch, n_buf);
                }
                ws.split_bytes = ws.d_split ? need : 0;
            }
            dp.work_queue = ws.d_counters;
            base = static_cast<char *>(ws.d_split);
            buf_stride = layout(batch); /* sets the offsets for the batch size settled on */
        }
        if (base == nullptr) goto one_kernel;
        if (batch >= n_epochs) n_buf = 1u;
        dp.sp_slots = slots;
        uint32_t k = 0;
        bool tail_used[2] = {false, false};
        hipError_t e = hipSuccess;
        for (uint32_t e0 = 0; e0 < n_epochs && e == hipSuccess; e0 += batch, ++k) {
            /* the layout is [slot][sample of THIS batch]: a short last batch just uses a prefix of every array */
            const uint32_t b = n_buf == 2u ? (k & 1u) : 0u;
            char *const ws_base = base + (size_t)b * buf_stride;
            dp.sp_hdr = reinterpret_cast<uint32_t *>(ws_base + o_hdr);
            dp.sp_req = reinterpret_cast<uint4 *>(ws_base + o_req);
            dp.sp_shade = reinterpret_cast<float4 *>(ws_base + o_shade);
            dp.sp_frame = reinterpret_cast<float4 *>(ws_base + o_frame);
            dp.epoch0 = e0;
            dp.n_epochs = std::min(batch, n_epochs - e0);
            e = hipMemsetAsync(dp.work_queue, 0, sizeof(uint32_t), stream);
            if (e == hipSuccess && tail_used[b]) e = hipStreamWaitEvent(stream, rng->ev_tail[b], 0); /* the unwind two batches ago has read this workspace */
            hipEvent_t pe[8]; /* profiling: prepare | chain | shade, unwind */
            if (e == hipSuccess && lookahead && !rng->ahead) {
                const bool prof = dist_profile_pairs(DK_PREPARE, 1, pe);
                if (prof) e = hipEventRecord(pe[0], stream);
                if (e == hipSuccess) e = rt::launch_rng_prepare(rng->d_states, (uint32_t)n_pixels, rng->d_list, rng->compute_units, stream);
                if (e == hipSuccess && prof) e = hipEventRecord(pe[1], stream);
            }
            rng->ahead = false; /* the chain kernel uses blocks up */
            const bool prof_chain = dist_profile_pairs(DK_CHAIN, 1, pe + 2);
            const bool prof_tail = dist_profile_pairs(DK_SHADE, 2, pe + 4);
            const hipEvent_t *const tail_ev = prof_tail ? pe + 4 : nullptr;
            rng->main_stream = stream;
            /* the pixels in the order of what they cost in the batch that used this workspace last (two batches ago in a pipelined
             * call, the last one else): rt_kernels.h DistParams::pixel_order */
            uint32_t *const pix_cost = rng->d_pix, *const pix_order = rng->d_pix + (size_t)(1u + b) * n_pixels, *const pix_scratch = rng->d_pix + 3u * n_pixels;
            dp.pixel_cost = by_cost ? pix_cost : nullptr;
            dp.pixel_order = by_cost && rng->order_valid[b] ? pix_order : nullptr;
            if (e == hipSuccess && prof_chain) e = hipEventRecord(pe[2], stream);
            if (e == hipSuccess) e = rt::launch_dist_chain(scene->ks, kf, dp, dist_waves, stream);
            if (e == hipSuccess && prof_chain) e = hipEventRecord(pe[3], stream);
            /* from here on this batch does not touch the RNG records: the look-ahead for the next one, on its own stream */
            if (e == hipSuccess && lookahead && overlap) e = lookahead_after_chain(rng);
            if (n_buf == 2u) {
                if (e == hipSuccess) e = hipEventRecord(rng->ev_tail[b], stream); /* first: the chain kernel has written workspace b ... */
                if (e == hipSuccess) e = hipStreamWaitEvent(rng->tail, rng->ev_tail[b], 0);
                /* the next chain kernel waits for the look-ahead, and the look-ahead's workgroups need 64 KB of LDS each: the shade
                 * kernel starts after it instead of taking that LDS first (between two chain kernels of a 1/8 share of the 1080p
                 * frame 1.1 -> 0.3 ms: 0.38 -> 0.365 ms per epoch, a 1/4 share 0.553 -> 0.517, the whole frame 1.685 -> 1.669) */
                if (e == hipSuccess && prep_first && rng->ahead) e = hipStreamWaitEvent(rng->tail, rng->ev_prepared, 0);
                if (e == hipSuccess) e = rt::launch_dist_shade_unwind(scene->ks, kf, dp, rng->tail, tail_ev);
                if (e == hipSuccess && by_cost) {
                    e = rt::launch_dist_pixel_order(pix_cost, pix_order, (uint32_t)n_pixels, pix_scratch, rng->tail);
                    rng->order_valid[b] = e == hipSuccess;
                }
                if (e == hipSuccess) e = hipEventRecord(rng->ev_tail[b], rng->tail); /* ... then: and the unwind has read it */
                tail_used[b] = e == hipSuccess;
            } else if (e == hipSuccess) {
                e = rt::launch_dist_shade_unwind(scene->ks, kf, dp, stream, tail_ev);
                if (e == hipSuccess && by_cost) {
                    e = rt::launch_dist_pixel_order(pix_cost, pix_order, (uint32_t)n_pixels, pix_scratch, stream);
                    rng->order_valid[b] = e == hipSuccess;
                }
            }
            /* the next chain kernel — of this call or, on whatever stream is ordered after this one, of the next — needs the prepared blocks */
            if (e == hipSuccess && rng->ahead) e = hipStreamWaitEvent(stream, rng->ev_prepared, 0);
        }
        /* everything the call started is behind the caller's stream again */
        for (uint32_t b = 0; b < 2u; ++b)
            if (tail_used[b]) { const hipError_t e2 = hipStreamWaitEvent(stream, rng->ev_tail[b], 0); if (e == hipSuccess) e = e2; }
        if (e != hipSuccess) return fail_hip("rt_render_distributed: launch", e);
        return RT_OK;
    }
one_kernel:
    dp.n_epochs = n_epochs;
    dp.epoch0 = 0;
    dp.sp_hdr = nullptr; dp.sp_req = nullptr; dp.sp_shade = nullptr; dp.sp_frame = nullptr; dp.sp_slots = 0;
    {
        if (rt::option(rt::OPT_DIST_STATIC, 0) != 1) { /* 1 (A/B): one 64-pixel chunk per wave instead of persistent lanes */
            std::lock_guard<std::mutex> lock(mut->ws_mutex);
            Workspace &ws = mut->workspaces[stream];
            RT_HIP(ensure_counters(ws));
            dp.work_queue = ws.d_counters;
            if (scene->ks.bfs_walk != 0u) { /* the breadth-first walk's lists, one set per wave of the grid (shared with the Whitted path of this stream) */
                const uint32_t waves = rt::dist_bfs_waves(rng->compute_units);
                const size_t words = (size_t)waves * rt::pwf_bfs_scratch_words_per_wave();
                if (ws.bfs_words < words) {
                    if (ws.d_bfs) (void)hipFree(ws.d_bfs);
                    ws.d_bfs = nullptr;
                    ws.bfs_words = 0;
                    if (hipMalloc(reinterpret_cast<void **>(&ws.d_bfs), words * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); ws.d_bfs = nullptr; }
                    else ws.bfs_words = words;
                }
                if (ws.d_bfs != nullptr) { /* (no room: the wave-uniform walk renders the same samples) */
                    dp.bfs_scratch = ws.d_bfs;
                    dp.bfs_items_cap = RT_BFS_ITEMS_CAP;
                    dp.bfs_jobs_cap = RT_BFS_JOBS_CAP;
                    const long long cap = rt::option(rt::OPT_DIAG_BFS_CAP, 0);
                    if (cap > 0) {
                        dp.bfs_items_cap = (uint32_t)std::min<long long>(cap, RT_BFS_ITEMS_CAP);
                        dp.bfs_jobs_cap = (uint32_t)std::min<long long>(cap, RT_BFS_JOBS_CAP);
                    }
                    if (dist_waves > waves) dist_waves = waves;
                }
            }
        }
    }
    hipError_t e = hipSuccess;
    if (dp.work_queue) e = hipMemsetAsync(dp.work_queue, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess && lookahead && !rng->ahead) e = rt::launch_rng_prepare(rng->d_states, (uint32_t)n_pixels, rng->d_list, rng->compute_units, stream);
    rng->ahead = false;
    if (e == hipSuccess) e = rt::launch_distributed(scene->ks, kf, dp, dist_waves, stream);
    if (e != hipSuccess) return fail_hip("rt_render_distributed: launch", e);
    return RT_OK;
}

int rt_render_distributed_host(const rt_scene *scene, const rt_camera *camera, const rt_frame *frame, float focus, float blur,
                               rt_rng *rng, uint32_t n_epochs, float *h_accum, unsigned long long *h_ray_count) {
    if (!scene || !rng || !h_accum) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed_host: null argument");
    if (!frame_ok(frame)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_distributed_host: bad frame");
    const size_t bytes = (size_t)rt_frame_pixels(frame) * 3 * sizeof(float);
    float *d_accum = nullptr;
    unsigned long long *d_cnt = nullptr;
    RT_HIP(hipMalloc(reinterpret_cast<void **>(&d_accum), bytes));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpy(d_accum, h_accum, bytes, hipMemcpyHostToDevice); /* img continues from the caller's sums */
    int rc = RT_OK;
    if (e == hipSuccess) {
        rc = rt_render_distributed(scene, camera, frame, focus, blur, rng, n_epochs, d_accum, nullptr, nullptr, d_cnt, nullptr);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(h_accum, d_accum, bytes, hipMemcpyDeviceToHost);
            unsigned long long cnt = 0;
            if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
            if (e == hipSuccess && h_ray_count) *h_ray_count = cnt;
        }
    }
    (void)hipFree(d_accum);
    if (d_cnt) (void)hipFree(d_cnt);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail_hip("rt_render_distributed_host", e);
    return RT_OK;
}

} /* extern "C" */
