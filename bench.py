#!/usr/bin/env python3
"""Benchmark of the hot path: Whitted render of the reference scene at 1920x1080, depth 8.

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame: every rank renders its interleaved row band of the frame with the HIP kernel
(homework-18-graphics-raytracer_amd/csrc/rt_pwf.hip by default, through the C ABI) and, for N > 1, the bands
are gathered to rank 0 over RCCL.  Rays = World::cast evaluations (primary + shadow + reflection +
refraction casts), counted by the kernel itself.  Rank 0 prints ONE JSON line.

For N > 1 the driver launches this file under `python -m torch.distributed.run`; when started plainly
with --gpus N > 1 it starts that launcher itself as a child process.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# algorithmic work per cast = 64 triangle tests + 4 sphere tests.  Triangle test 77 flop
# (5 cull dot + 12 plane-t + 6 point + 51 three areas + 3 compares), sphere test 28 flop.
FLOP_PER_TRIANGLE_TEST = 77
FLOP_PER_SPHERE_TEST = 28
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (vector)" (= the f32-input MFMA dense peak)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--depth", type=int, default=8)
    p.add_argument("--variant", type=int, default=None, help="render path (include/rt_amd.h): 18 = the persistent wavefront kernel (default); 2 = the per-pixel kernel (scalar triangle fetches), 3 = with its triangle records staged in LDS; 19 = 18 with 3 as its fallback")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-pipelined", action="store_true", help="skip the extra measurement with two frames in flight (after the headline's timed region)")
    p.add_argument("--no-stochastic", action="store_true", help="skip the depth-of-field pass (configs[3]: 64 samples per pixel, sharded like the frame), measured after the headline's timed region")
    p.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = all cores)")
    p.add_argument("--frames-in-flight", type=int, default=0,
                   help="N > 1 only: frames a rank renders at once, one per stream (dist.FramePipeline in_flight); 0 = 4, each launch on a quarter of the "
                        "device (a share of the frame leaves the GPU mostly idle: profiles/r04_frames_in_flight.txt); 1 = one after the other")
    p.add_argument("--no-extras", action="store_true", help="skip the sharded_finish, share_timing and large_scene objects (after everything else)")
    return p.parse_args()


def relaunch_under_torchrun(args) -> int:
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def cpu_baseline(world_desc, camera, width, height, depth, threads, gpu_frame):
    """The oracle (CPU restatement of the reference algorithm) timed on this box's host cores.

    Checker/baseline use of oracle/ only: it is never the thing measured as `value`.
    """
    sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    import _oracle
    import homework_18_graphics_raytracer_amd as rt

    if threads > 0:
        cores = threads
    else:  # the cores this process may actually run on (a GPU box shares its host between jobs)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
    frame = rt.Frame.full(width, height, depth)
    t0 = time.perf_counter()
    img, casts = _oracle.render_whitted(world_desc, camera, frame, threads=cores)
    dt = time.perf_counter() - t0
    out = {
        "value": round(casts / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": f"one full {width}x{height} depth-{depth} frame of the same scene ({casts} casts, {dt:.2f} s), "
                  f"oracle/rt_oracle.cpp with {cores} threads over rows",
        "ms_per_frame": round(dt * 1e3, 2),
    }
    if gpu_frame is not None:
        same = bool(np.array_equal(img.view(np.uint32), gpu_frame.view(np.uint32)))
        out["gpu_frame_bit_identical_to_cpu"] = same
    return out


STOCHASTIC_EPOCHS = 64  # BASELINE.json configs[3]: 64 depth-of-field samples per pixel


SCATTER_EPOCHS = 5      # BASELINE.json configs[4] / SURVEY §8(d) "Config 5": 10 368 000 = 1920 x 1080 x 5 (pixel, epoch) samples


def traffic_record(name, match):
    """profiles/<name> when it describes THIS workload (`match`: key -> value) and was collected with THESE sources
    (`sources_sha256`, tools/make_traffic.py) — else (None, why): a kernel change without a fresh counter profile must read as null,
    not as a stale number."""
    from homework_18_graphics_raytracer_amd import _capi

    path = ROOT / "profiles" / name
    if not path.exists():
        return None, f"no profiles/{name}"
    try:
        rec = json.loads(path.read_text())
    except Exception as exc:  # noqa: BLE001
        return None, f"profiles/{name}: {exc}"
    for k, v in match.items():
        if rec.get(k, v if k == "variant" and v == 2 else None) != v:
            return None, f"profiles/{name} is a profile of another workload ({k} = {rec.get(k)!r}, this run {v!r})"
    if rec.get("sources_sha256") != _capi.sources_sha256():
        return None, f"profiles/{name} was collected with other kernel sources (sha256 {str(rec.get('sources_sha256'))[:12]}..., this tree {_capi.sources_sha256()[:12]}...): re-run tools/profile_round.sh + tools/make_traffic.py"
    return rec, None


def hbm_view(counter_bytes, alg_bytes, ms, rec):
    """north_star: "achieved HBM GB/s and L2-hit rate".  `achieved` is what HBM SAW — FETCH_SIZE (x2 on gfx950) + WRITE_SIZE of the
    committed rocprofv3 --pmc profile of this very workload, over the live kernel time; the algorithmic bytes sit
    beside it under their own name.  None when no committed profile matches the configuration."""
    out = {"achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None,
           "algorithmic": {"bytes_per_launch": alg_bytes, "GB_per_s": round(alg_bytes / (ms * 1e-3) / 1e9, 3),
                           "frac": round(alg_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6)},
           "l2_hit_rate": None, "source": None}
    if counter_bytes:
        gbs = counter_bytes / (ms * 1e-3) / 1e9
        out.update({"achieved": round(gbs, 2), "frac": round(gbs / PEAK_HBM_GBS, 5), "counter_bytes_per_launch": counter_bytes,
                    "counter_over_algorithmic": round(counter_bytes / alg_bytes, 2)})
    if rec:
        out["l2_hit_rate"] = rec.get("l2_hit_rate")
        out["source"] = rec.get("source")
    return out


def stochastic_pass(scene, camera, width, height, depth, rank, world_size, distributed, world_desc=None, cpu_threads=0,
                    epochs=STOCHASTIC_EPOCHS, config="configs[3]", traffic_file="traffic_stochastic.json"):
    """The other render loop of the reference (shoot_focus + distributed_ray_trace, main.rs:1117-1167) as BASELINE.json
    configs[3] states it: 64 samples per pixel of the 1920x1080 depth-8 frame, image rows sharded over the ranks, the
    accumulated bands gathered to rank 0 over RCCL.  ONE convention for every number of this pass: the whole job from
    freshly seeded streams (IsaacRng::new_from_u64 per pixel, main.rs:1117-1127, outside the timed region exactly as it is
    outside the reference's stopwatch), `epochs` epochs in one rt_render_distributed call per rank, the gather inside the
    timed region, barrier + synchronize on both sides, max over ranks.  Returns the JSON object (rank 0) or None."""
    import torch
    import torch.distributed as dist

    import homework_18_graphics_raytracer_amd as rt
    from homework_18_graphics_raytracer_amd import _capi
    from homework_18_graphics_raytracer_amd import dist as rtdist

    frame = rtdist.shard_frame(width, height, depth, rank, world_size)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    warm = rt.Rng(frame)  # untimed: loads the kernels and sizes the per-stream workspace for calls of this many epochs
    rt.render_distributed(scene, camera, frame, warm, epochs, accum=accum)
    torch.cuda.synchronize()
    warm.close()
    del warm
    accum.zero_()
    rng = rt.Rng(frame)
    staging = None
    if distributed and rank == 0:
        staging = torch.empty((world_size, rtdist.band_rows(height, 0, world_size), width, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib = _capi.amd_lib()
    _capi.check(lib.rt_profile_enable(1))  # an event pair around every kernel of the pass, on the stream it is launched on
    t0 = time.perf_counter()
    e0.record()
    rt.render_distributed(scene, camera, frame, rng, epochs, accum=accum, ray_count=cnt)
    e1.record()
    full = rtdist.gather_frame(accum, height, rank, world_size, staging=staging) if distributed else accum
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    render_ms = e0.elapsed_time(e1)  # this rank's epochs alone, HIP events on the launch stream
    k_ms, k_n = (C.c_double * 4)(), (C.c_uint * 4)()
    _capi.check(lib.rt_profile_read_distributed(k_ms, k_n))
    _capi.check(lib.rt_profile_enable(0))
    t = torch.tensor([elapsed, render_ms], dtype=torch.float64, device="cuda")
    total = cnt.clone()
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    elapsed, render_ms = float(t[0].item()), float(t[1].item())
    samples = width * height * epochs
    casts = int(total.item())
    casts_rank = int(cnt.item())
    flop_per_cast = 64 * FLOP_PER_TRIANGLE_TEST + 4 * FLOP_PER_SPHERE_TEST
    if world_desc is not None:
        flop_per_cast = world_desc.n_triangles * FLOP_PER_TRIANGLE_TEST + world_desc.n_spheres * FLOP_PER_SPHERE_TEST
    tflops = casts_rank * flop_per_cast / (render_ms * 1e-3) / 1e12
    # the pass's kernels from this run's own HIP events (rank 0's): each one's summed duration under the overlap of a pipelined call
    names = ("rng look-ahead (rng_scan + rng_prepare)", "dist_chain_kernel", "dist_shade_kernel", "dist_unwind_kernel")
    k_total = sum(k_ms) or 1.0
    kernels = {names[i]: {"ms_sum": round(k_ms[i], 3), "launches": int(k_n[i]), "share_of_kernel_time": round(k_ms[i] / k_total, 3)} for i in range(4)}
    kernel_note = ("the pass's kernels together; live HIP events of this run, summed per kernel (they overlap in a pipelined call, so the sums exceed render_ms): "
                   + ", ".join(f"{names[i]} {100.0 * k_ms[i] / k_total:.0f} %" for i in sorted(range(4), key=lambda i: -k_ms[i])))
    rank_samples = frame.rows * frame.cols * epochs
    alg_bytes = rank_samples * (12 + 2 * 2064)  # 12 B out + the generator's record read and written, per sample
    out = {"metric": "Msamples/s, depth-of-field pass (one sample = shoot_focus + cast + distributed_ray_trace of one pixel)",
           "value": round(samples / elapsed / 1e6, 2), "unit": "Msamples/s", "n_gpus": world_size,
           "ms_per_epoch": round(elapsed * 1e3 / epochs, 4), "ms_total": round(elapsed * 1e3, 3),
           "Mrays_per_s": round(casts / elapsed / 1e6, 2), "casts_per_sample": round(casts / samples, 3),
           "config": {"workload": f"{config}: {epochs} depth-of-field samples per pixel ({samples} (pixel, epoch) samples), {width}x{height}, depth {depth}, focus 3.0, blur 0.04, "
                                  f"streams seeded y*2^33+x, interleaved rows over {world_size} rank(s)" + (", accumulators gathered to rank 0 over RCCL inside the timed region" if distributed else ""),
                      "epochs": epochs, "samples": samples},
           "roofline": {"bound": "valu_fp32", "achieved": round(tflops, 4), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tflops / PEAK_FP32_VECTOR_TFLOPS, 5), "traffic": None,
                        "kernel": kernel_note, "kernels": kernels,
                        "render_ms": round(render_ms, 3), "flop_per_cast": flop_per_cast, "casts_per_launch": casts_rank,
                        "note": "algorithmic flop = casts x (T x 77 + S x 28): an upper bound on useful work, as for the Whitted pass"},
           "parity": "tests/test_gpu_distributed_parity.py, tests/test_gpu_reference_pins.py"}
    rec, why = (None, "counter profiles are single-GPU") if world_size != 1 else traffic_record(traffic_file, {"width": width, "height": height, "depth": depth, "epochs": epochs})
    if rec:
        out["roofline"]["traffic"] = rec.get("hbm_bytes_per_launch")
    out["roofline"]["hbm"] = hbm_view(rec.get("hbm_bytes_per_launch") if rec else None, alg_bytes, render_ms, rec)
    if why:
        out["roofline"]["hbm"]["source"] = why
    if world_desc is not None and world_size == 1:
        # the oracle's restatement of the same loop on the host cores: the FIRST epoch of the same frame; the GPU's first
        # epoch (fresh streams, same seeds) must equal it bit for bit
        sys.path.insert(0, str(ROOT / "tests"))
        import numpy as np
        import _oracle

        try:
            cores = cpu_threads if cpu_threads > 0 else len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        states = _oracle.rng_init(frame)
        t0 = time.perf_counter()
        want_s, want_v, cpu_casts = _oracle.render_distributed(world_desc, camera, frame, states, 1, threads=cores)
        dt = time.perf_counter() - t0
        rng1 = rt.Rng(frame)
        got_s = torch.empty((1, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
        got_v = torch.empty((1, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
        rt.render_distributed(scene, camera, frame, rng1, 1, samples=got_s, valid=got_v)
        torch.cuda.synchronize()
        gs, ws = got_s.cpu().numpy(), want_s
        same = bool(np.array_equal(got_v.cpu().numpy(), want_v) and (((gs.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(gs) & np.isnan(ws))).all())
                    and np.array_equal(rng1.download(), states))
        out["cpu_baseline"] = {"value": round(frame.rows * frame.cols / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                               "sample": f"the first epoch of the same {width}x{height} depth-{depth} frame ({cpu_casts} casts, {dt:.2f} s), oracle/rt_oracle.cpp",
                               "gpu_epoch_bit_identical_to_cpu": same}
    return out


def share_timing(scene, camera, width, height, depth, steps):
    """What ONE GPU needs for a 1/2, 1/4, 1/8 share of either pass — rank 0's interleaved rows of that many ranks, exactly the
    frames an N-GPU job gives a rank.  A prediction from one GPU (the critical path of the share's dearest pixels does not shrink
    with the share), NOT a scaling measurement: the driver's multi-GPU run is that."""
    import torch

    import homework_18_graphics_raytracer_amd as rt
    from homework_18_graphics_raytracer_amd import dist as rtdist

    out = {"note": "predicted from one GPU, not a scaling measurement: this GPU's time for rank 0's share of an N-rank job "
                   "(interleaved rows, no gather); speedup_if_all_ranks_alike = whole frame, one after the other / share; four_in_flight: as this script "
                   "renders a share for N > 1 (dist.FramePipeline(in_flight=4), each launch on a quarter of the device)", "shares": {}}

    def whitted_ms(frame):
        band = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
        for _ in range(3):
            rt.render_whitted(scene, camera, frame, out=band)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            rt.render_whitted(scene, camera, frame, out=band)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps

    def dof_ms_per_epoch(frame):
        accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
        warm = rt.Rng(frame)
        rt.render_distributed(scene, camera, frame, warm, STOCHASTIC_EPOCHS, accum=accum)
        torch.cuda.synchronize()
        warm.close()
        rng = rt.Rng(frame)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rt.render_distributed(scene, camera, frame, rng, STOCHASTIC_EPOCHS, accum=accum)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / STOCHASTIC_EPOCHS
        rng.close()
        return dt

    def whitted_in_flight_ms(frame, in_flight=4):
        """As this script renders a share for N > 1: `in_flight` frames at once, one per stream, each launch on its part of the device."""
        bands = [torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda") for _ in range(in_flight)]
        rt.set_option("RT_AMD_WF_SHARE", in_flight)
        try:
            streams, _ = rtdist.choose_streams(lambda i: rt.render_whitted(scene, camera, frame, out=bands[i]), in_flight)

            def run(n):
                for k in range(n):
                    with torch.cuda.stream(streams[k % in_flight]):
                        rt.render_whitted(scene, camera, frame, out=bands[k % in_flight])
                torch.cuda.synchronize()

            run(2 * in_flight)
            n = max(4 * steps, 16 * in_flight)
            t0 = time.perf_counter()
            run(n)
            return (time.perf_counter() - t0) * 1e3 / n
        finally:
            rt.set_option("RT_AMD_WF_SHARE", None)

    whole = rt.Frame.full(width, height, depth)
    w1, d1 = whitted_ms(whole), dof_ms_per_epoch(whole)
    out["whole_frame"] = {"whitted_ms_per_frame": round(w1, 4), "dof_ms_per_epoch": round(d1, 4)}
    for n in (2, 4, 8):
        f = rtdist.shard_frame(width, height, depth, 0, n)
        w, d, w4 = whitted_ms(f), dof_ms_per_epoch(f), whitted_in_flight_ms(f)
        out["shares"][f"1/{n}"] = {"whitted_ms_per_frame": round(w, 4), "whitted_ms_per_frame_four_in_flight": round(w4, 4), "dof_ms_per_epoch": round(d, 4),
                                    "speedup_if_all_ranks_alike": {"whitted": round(w1 / w, 2), "whitted_four_in_flight": round(w1 / w4, 2), "dof": round(d1 / d, 2)}}
    return out


def large_scene(camera, width, height, depth, levels=6):
    """the same literal scene around the dodecahedron tessellated 4^levels ways (147 484 triangles at 6: 28 MB of
    records, beyond every cache), the headline's frame.  rt_scene_create gives such a scene the breadth-first walk of the node tree
    (rt_cast_bfs.h cast_bfs; bit-identical to the oracle at every size: tests/test_gpu_scene_sizes.py, tools/scene_sweep.py).  Timed like
    the headline, three frames after a warm one; 'algorithmic' triangle tests = casts x triangles, what the reference's loop runs."""
    import subprocess
    import tempfile

    import torch

    import homework_18_graphics_raytracer_amd as rt

    with tempfile.TemporaryDirectory() as tmp:
        obj = Path(tmp) / "dodecahedron.obj"
        subprocess.run([sys.executable, str(ROOT / "tools" / "make_tessellated_obj.py"), rt.DEFAULT_OBJ, str(obj), "--levels", str(levels)], check=True, capture_output=True)
        world = rt.reference_world(str(obj))
    desc = world.desc()
    scene = rt.Scene(world)
    frame = rt.Frame.full(width, height, depth)
    out = torch.empty((height, width, 3), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    rt.render_whitted(scene, camera, frame, out=out, ray_count=cnt)
    torch.cuda.synchronize()
    casts = int(cnt.item())
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        rt.render_whitted(scene, camera, frame, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    # ... and four epochs of the depth-of-field pass in one call (the one-kernel organisation with the same walk), fresh streams, after a warm one
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    warm = rt.Rng(frame)
    rt.render_distributed(scene, camera, frame, warm, 1, accum=accum)
    torch.cuda.synchronize()
    warm.close()
    rng = rt.Rng(frame)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rt.render_distributed(scene, camera, frame, rng, 4, accum=accum)
    torch.cuda.synchronize()
    dof_ms = (time.perf_counter() - t0) * 1e3 / 4
    rng.close()
    return {"dof_ms_per_epoch": round(dof_ms, 2), "dof_Msamples_per_s": round(width * height / dof_ms / 1e3, 2),
            "workload": f"the reference scene around a dodecahedron of {36 * 4 ** levels} flat triangles ({desc.n_triangles} triangles, {desc.n_triangles * 128 / 1e6:.1f} MB of records), "
                        f"{width}x{height}, depth {depth}, Whitted pass, breadth-first walk of the node tree",
            "ms_per_frame": round(ms, 3), "Mrays_per_s": round(casts / ms / 1e3, 2), "casts_per_frame": casts,
            "algorithmic_T_triangle_tests_per_s": round(casts * desc.n_triangles / ms / 1e9, 2),
            "parity": "tests/test_gpu_scene_sizes.py (forced on at every size), tools/scene_sweep.py at this size on a small frame (profiles/r04_scene_sweep_flat_bfs.jsonl)"}


def sharded_finish(band, height, rank, world_size, distributed, steps):
    """What follows a frame when its bands stay on their ranks (main.rs:1113-1114 over N ranks): post_process with the p99 luma
    taken over all ranks (keys -> all-reduce -> 4 x (histogram -> all-reduce -> pick) -> scale, stream-ordered device work),
    sRGB / u8 encode of each band where it is, ONE gather of u8 rows to rank 0.  Timed per frame like the headline (barrier +
    synchronize both sides, max over ranks).  With one rank and no process group a single-rank RCCL group is made for the
    measurement, so that the 1-GPU line already pays for the collectives' launches."""
    import torch
    import torch.distributed as dist

    from homework_18_graphics_raytracer_amd import dist as rtdist

    made_group = False
    out = {"what": "frame -> dist.finish_frame_sharded (post_process over the ranks' bands, sRGB/u8 per band, one u8 gather) -> u8 frame on rank 0",
           "n_gpus": world_size}
    try:
        if not (dist.is_available() and dist.is_initialized()):
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
            made_group = True
        works = [band.clone() for _ in range(steps + 2)]  # finish normalises its band in place: every step gets the rendered one
        for k in range(2):
            rtdist.finish_frame_sharded(works[k], height, rank, world_size, sync=False)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u8 = None
        for k in range(steps):
            u8, divisor = rtdist.finish_frame_sharded(works[2 + k], height, rank, world_size, sync=False)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out.update({"ms_per_frame": round(float(t.item()) * 1e3 / steps, 4), "steps": steps, "divisor": float(divisor.item()),
                    "host_synchronisations_per_frame": 0, "collectives_per_frame": "5 all-reduces (1 + 4 x 256 counters) + 1 gather of u8 rows",
                    "u8_frame_bytes": int(u8.numel()) if u8 is not None else None,
                    "process_group": "single-rank RCCL group made for this measurement" if made_group else "the job's"})
    except Exception as exc:  # noqa: BLE001  (a box without a usable RCCL must not cost the headline)
        out["error"] = f"{type(exc).__name__}: {exc}"
    finally:
        if made_group:
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
    return out


def main() -> int:
    args = parse_args()
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_size == 1:
        return relaunch_under_torchrun(args)  # child process; nothing has touched the GPU yet

    # stdout carries ONE line, the JSON: whatever else writes to file descriptor 1 from here on (RCCL prints its version there when a
    # process group is made; make's output) goes to stderr instead, and the line is written to the real stdout at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import __graft_entry__

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # every rank, before anything imports the package or touches the GPU: the makes run under a file lock, so one rank builds (a
    # no-op when the in-tree libraries are current) and the others wait for it instead of loading a library that is being linked
    __graft_entry__.build()
    import torch
    import torch.distributed as dist

    # RT_BENCH_FORCE_DIST=1 runs the torch.distributed/RCCL plumbing even with one rank (rehearsal on a 1-GPU box)
    distributed = world_size > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    else:
        torch.cuda.set_device(0)

    import homework_18_graphics_raytracer_amd as rt
    from homework_18_graphics_raytracer_amd import _capi
    from homework_18_graphics_raytracer_amd import dist as rtdist

    if args.variant is not None:
        _capi.check(_capi.amd_lib().rt_set_variant(args.variant))
    variant = _capi.amd_lib().rt_get_variant()

    W, H, D = args.width, args.height, args.depth
    world = rt.reference_world()
    camera = rt.reference_camera()
    desc = world.desc()
    scene = rt.Scene(world)  # scene uploaded once; resident in HBM before the timed region
    frame = rtdist.shard_frame(W, H, D, rank, world_size)
    band = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    # N > 1: frame k's bands travel to rank 0 while frame k+1 is rendered (dist.FramePipeline); every frame is
    # assembled on rank 0 inside the timed region
    # ... and a rank has several frames of the sequence in flight, one per stream: a share of a frame ends on the critical path of
    # its deepest pixels with most of the GPU idle.  One rank (the headline): one frame after the other.
    in_flight = 1 if not distributed else (args.frames_in_flight if args.frames_in_flight > 0 else (1 if world_size == 1 else 4))
    device_share = in_flight if in_flight >= 3 else 1
    chosen, calibration = None, None
    if distributed and in_flight > 1:  # streams on which the frames in flight do run side by side (dist.choose_streams says why)
        scratch = [torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda") for _ in range(in_flight)]
        if device_share > 1:
            rt.set_option("RT_AMD_WF_SHARE", device_share)
        chosen, calibration = rtdist.choose_streams(lambda i: rt.render_whitted(scene, camera, frame, out=scratch[i]), in_flight)
        rt.set_option("RT_AMD_WF_SHARE", None)
        del scratch
    pipe = rtdist.FramePipeline(W, H, D, rank, world_size, in_flight=in_flight, streams=chosen) if distributed else None
    # from three in flight each launch takes its part of the device's workgroups, so that the frames run side by side instead of one
    # behind the other's tail (RT_AMD_WF_SHARE; profiles/r04_frames_in_flight.txt: a 1/8 share 0.168 -> 0.153 ms per frame with four,
    # and the larger shares, which four whole-device launches in flight would slow down, gain too)
    step_index = [0]

    def step(ev0=None, ev1=None):
        if not distributed:
            if ev0 is not None:
                ev0.record()
            rt.render_whitted(scene, camera, frame, out=band, ray_count=count)
            if ev1 is not None:
                ev1.record()
            return band
        k = step_index[0]
        step_index[0] += 1
        with pipe.stream(k):
            target = pipe.band(k)
            if ev0 is not None:
                ev0.record()
            rt.render_whitted(scene, camera, frame, out=target, ray_count=count)
            if ev1 is not None:
                ev1.record()
            return pipe.submit(k)

    if device_share > 1:
        rt.set_option("RT_AMD_WF_SHARE", device_share)
    full = None
    for _ in range(args.warmup):
        full = step()
    if distributed:
        full = pipe.finish()
    torch.cuda.synchronize()
    count.zero_()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    _capi.check(_capi.amd_lib().rt_profile_enable(1))  # HIP events around the render kernel alone, on its launch stream
    t0 = time.perf_counter()
    for e0, e1 in events:
        full = step(e0, e1)
    if distributed:
        full = pipe.finish()  # the last frame's gather and assembly belong to the timed region
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if device_share > 1:
        rt.set_option("RT_AMD_WF_SHARE", None)  # the other passes of this run use the whole device

    call_ms = sum(e0.elapsed_time(e1) for e0, e1 in events) / max(1, args.steps)  # whole call: probe + render kernels
    ksum, kn = C.c_double(0.0), C.c_uint(0)
    _capi.check(_capi.amd_lib().rt_profile_read(C.byref(ksum), C.byref(kn)))
    _capi.check(_capi.amd_lib().rt_profile_enable(0))
    kernel_ms = ksum.value / max(1, kn.value)  # the render kernel alone (rt::pwf_kernel, or rt::whitted_kernel)
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    total_casts = count.clone()
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_casts, op=dist.ReduceOp.SUM)
    elapsed_max, kernel_ms_max = float(t[0].item()), float(t[1].item())
    casts_all_steps = int(total_casts.item())
    casts_per_frame = casts_all_steps // max(1, args.steps)

    if rank == 0:
        ms_per_step = elapsed_max * 1e3 / args.steps
        mrays = casts_all_steps / elapsed_max / 1e6
        n_tri, n_sph = desc.n_triangles, desc.n_spheres
        flop_per_cast = n_tri * FLOP_PER_TRIANGLE_TEST + n_sph * FLOP_PER_SPHERE_TEST
        casts_this_rank = int(count.item()) // max(1, args.steps)
        achieved_tflops = casts_this_rank * flop_per_cast / (kernel_ms_max * 1e-3) / 1e12
        alg_hbm_bytes = frame.rows * frame.cols * 12 + 6752  # 12 B/pixel out + the scene once
        traffic, executed = None, None
        traffic_rec, traffic_why = (None, "counter profiles are single-GPU") if world_size != 1 else traffic_record("traffic.json", {"width": W, "height": H, "depth": D, "variant": variant})
        if traffic_rec:
            traffic = traffic_rec.get("hbm_bytes_per_launch")
            if traffic_rec.get("sq_insts_valu"):
                # executed VALU work from the SQ_INSTS_VALU counter of the committed profile (wave-instructions x 64
                # lanes), against the issue roof of a path that may not fuse multiply-add: half the FMA peak
                lane_ops = float(traffic_rec["sq_insts_valu"]) * 64.0
                tops = lane_ops / (kernel_ms_max * 1e-3) / 1e12
                executed = {"lane_ops_per_launch": lane_ops, "achieved": round(tops, 3), "peak_no_fma": PEAK_FP32_VECTOR_TFLOPS / 2,
                            "unit": "T lane-op/s", "frac": round(tops / (PEAK_FP32_VECTOR_TFLOPS / 2), 4), "source": traffic_rec.get("source")}
        line = {
            "metric": "Mrays/s (primary+secondary) at 1920x1080, depth 8",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: the reference's literal scene (main.rs:810-1083) rebuilt in-process",
            "config": {
                "workload": f"configs[2]/[1]: dodecahedron.obj scene (the reference's single scene: 64 triangles, 4 spheres, "
                            f"3 lights), {W}x{H}, depth {D}, Whitted pass, 1 spp",
                "width": W, "height": H, "max_depth": D,
                "tiling": f"interleaved rows over {world_size} rank(s)" + (", RCCL gather to rank 0 overlapped with the next frames' rendering" if distributed else ""),
                "frames_in_flight": in_flight, "device_share_per_launch": f"1/{device_share}", "stream_sets_tried_ms_per_frame": calibration,
                "kernel_variant": "persistent-wavefront" if variant & 16 else ("per-pixel, LDS-staged triangles" if variant & 1 else "per-pixel, scalar triangle fetches"),
            },
            "casts_per_frame": casts_per_frame,
            "casts_per_pixel": round(casts_per_frame / (W * H), 3),
            "roofline": {
                "bound": "valu_fp32",
                "achieved": round(achieved_tflops, 4),  # ALGORITHMIC flop/s: an upper bound on useful work (see "note")
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / PEAK_FP32_VECTOR_TFLOPS, 5),
                "traffic": traffic,
                "executed_valu": executed,
                "kernel": "rt::pwf_kernel (the persistent render kernel: one launch per frame; the no-op fallback launch behind it is in call_ms_avg)" if variant & 16
                          else "rt::whitted_kernel<8, %s>" % ("true" if variant & 1 else "false"),
                "kernel_ms_avg": round(kernel_ms_max, 4),  # (with frames_in_flight > 1 the launches overlap: each lasts longer than a step)
                "call_ms_avg": round(call_ms, 4),
                "flop_per_cast": flop_per_cast,
                "casts_per_launch": casts_this_rank,
                "note": "no MFMA and not HBM-bound: the scene is 6.75 KB, the binding roof is FP32 vector issue "
                        "(SURVEY §8d); hbm below is what the counters of the committed profile saw, the algorithmic bytes beside it. `achieved` counts "
                        "ALGORITHMIC flop — casts x (T x 77 + S x 28), as if every cast ran every primitive test — i.e. an "
                        "upper bound on useful work: the kernel's conservative rejections skip more than half of it, and "
                        "the path may not use FMA (parity). Executed VALU work is in executed_valu (from SQ_INSTS_VALU, "
                        "profiles/)",
                "hbm": hbm_view(traffic, alg_hbm_bytes, kernel_ms_max, traffic_rec),
            },
        }
        if traffic_why:
            line["roofline"]["hbm"]["source"] = traffic_why
        if not args.no_cpu_baseline and world_size == 1:
            gpu_frame = full.cpu().numpy() if full is not None else None
            line["cpu_baseline"] = cpu_baseline(desc, camera, W, H, D, args.cpu_threads, gpu_frame)
    # Extra, reported beside the headline and never as `value`: the same K frames with TWO in flight — frame k+1 rendered on a
    # second stream (its own workspace) while frame k drains.  A frame (and even more a 1/N share) ends with the critical path
    # of its deepest pixels while most of the GPU idles; independent frames fill that.  Same pixels.
    pipelined = None
    if not args.no_pipelined and in_flight == 1:  # (with several frames in flight the headline already is this)
        bands2 = [torch.empty_like(band), torch.empty_like(band)]
        streams, _ = rtdist.choose_streams(lambda i: rt.render_whitted(scene, camera, frame, out=bands2[i]), 2, attempts=3, frames_per_stream=4)  # (a pair that does overlap)
        count2 = torch.zeros(1, dtype=torch.int64, device="cuda")
        pipe2 = rtdist.FramePipeline(W, H, D, rank, world_size) if distributed else None

        def step2(k):
            with torch.cuda.stream(streams[k % 2]):
                target = pipe2.band(k) if distributed else bands2[k % 2]
                rt.render_whitted(scene, camera, frame, out=target, ray_count=count2)
                if distributed:
                    pipe2.submit(k)

        for k in range(4):  # each stream allocates its workspace, untimed
            step2(k)
        if distributed:
            pipe2.finish()
        torch.cuda.synchronize()
        count2.zero_()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(4, 4 + args.steps):
            step2(k)
        if distributed:
            with torch.cuda.stream(streams[(4 + args.steps - 1) % 2]):
                pipe2.finish()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        el2 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        c2 = count2.clone()
        if distributed:
            dist.all_reduce(el2, op=dist.ReduceOp.MAX)
            dist.all_reduce(c2, op=dist.ReduceOp.SUM)
        if rank == 0:
            same2 = all(bool(torch.equal(b.view(torch.int32), band.view(torch.int32))) for b in bands2) if not distributed else None
            pipelined = {"frames_in_flight": 2, "value": round(int(c2.item()) / float(el2.item()) / 1e6, 3), "unit": "Mrays/s",
                         "ms_per_frame": round(float(el2.item()) * 1e3 / args.steps, 4), "steps": args.steps,
                         "frames_bit_identical_to_the_headline_run": same2,
                         "note": "throughput with two independent frames in flight on two streams; latency per frame is the headline's"}

    # the other render loop (configs[3]), on every rank: after the headline's timed region, with its own barriers
    stochastic = None
    if not args.no_stochastic:
        stochastic = stochastic_pass(scene, camera, W, H, D, rank, world_size, distributed,
                                     world_desc=None if args.no_cpu_baseline else desc, cpu_threads=args.cpu_threads)
    # configs[4] ("photon.rs scatter pass: 10M photons" = SURVEY §8(d) Config 5): 10 368 000 (pixel, epoch) samples of
    # distributed_ray_trace — 5 epochs of the same frame, the same convention (fresh streams, one call, gather inside the timed region)
    scatter = None
    if not args.no_stochastic:
        scatter = stochastic_pass(scene, camera, W, H, D, rank, world_size, distributed, world_desc=None, epochs=SCATTER_EPOCHS,
                                  config="configs[4], the scatter pass", traffic_file="traffic_scatter.json")
    # after everything else: what the frame costs AFTER rendering when its bands stay sharded, and what a share of either pass costs
    finish, shares, large = None, None, None
    if not args.no_extras:
        finish = sharded_finish(band if not distributed else pipe.band(0), H, rank, world_size, distributed, min(args.steps, 20))
        if world_size == 1 and not distributed:
            shares = share_timing(scene, camera, W, H, D, min(args.steps, 20))
            large = large_scene(camera, W, H, D)
    status = 0
    if rank == 0:
        if finish is not None:
            line["sharded_finish"] = finish
        if shares is not None:
            line["share_timing"] = shares
        if large is not None:
            line["large_scene"] = large
        if pipelined is not None:
            line["pipelined"] = pipelined
        if stochastic is not None:
            line["stochastic_pass"] = stochastic
        if scatter is not None:
            scatter["parity"] = "tests/test_gpu_distributed_parity.py::test_scatter_job_of_configs4_equals_the_oracle (the whole job: samples, flags, generator records, casts)"
            line["scatter_pass"] = scatter
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        # a fast frame that differs from the reference algorithm's is not a result: fail loudly
        if line.get("cpu_baseline", {}).get("gpu_frame_bit_identical_to_cpu") is False:
            sys.stderr.write("bench.py: the GPU frame differs from the CPU oracle's\n")
            status = 1
        if (stochastic or {}).get("cpu_baseline", {}).get("gpu_epoch_bit_identical_to_cpu") is False:
            sys.stderr.write("bench.py: the GPU's depth-of-field epoch differs from the CPU oracle's\n")
            status = 1

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
