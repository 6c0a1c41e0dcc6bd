#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants in ONE process (cdna_hip_programming.md §5.4 rule 24).

    python tools/ab_bench.py --tags base,wg64 [--rounds 7] [--frames 5] [--width 1920 --height 1080 --depth 8]

Each tag is homework-18-graphics-raytracer_amd/variants/librt_amd_<tag>.so (make -C csrc variant TAG=.. EXTRA=..);
"main" is the in-tree librt_amd.so.  Every variant's frame is checked bit-for-bit against the first one.
"""
import argparse
import ctypes as C
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402  (first: one HIP runtime per process)

import homework_18_graphics_raytracer_amd as rt  # noqa: E402
from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tags", required=True)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--variant", type=int, default=18)
ap.add_argument("--world", type=int, default=1, help="render only rank 0's interleaved row band of this many ranks (a multi-GPU share)")
args = ap.parse_args()

ENV_KNOBS = ("RT_AMD_NO_SPHERE_FILTER", "RT_AMD_FILTER_MAX_FRAC", "RT_AMD_NO_CLUSTERS", "RT_AMD_PWF_RING", "RT_AMD_NO_PLANE_SHARING", "RT_AMD_NO_WEAK_PLANE_SHARING")
world = rt.reference_world()
cam = rt.reference_camera()
desc = world.desc()
frame = rt.Frame.full(args.width, args.height, args.depth) if args.world == 1 else rt.Frame.rows_of_rank(args.width, args.height, args.depth, 0, args.world)
libs = {}
for tag in args.tags.split(","):
    parts = tag.split(":")  # "name", "name:variant" or "name:variant:ENV=value[;ENV=value]"
    name, var = parts[0], (parts[1] if len(parts) > 1 else "")
    evict = None
    env = dict(kv.split("=", 1) for kv in parts[2].split(";")) if len(parts) == 3 else {}
    for k in ENV_KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)  # read by rt_scene_create (and, for some, at every render call: see run())
    path = _capi.PKG_DIR / ("librt_amd.so" if name == "main" else f"variants/librt_amd_{name}.so")
    lib = C.CDLL(str(path))
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
    lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
    lib_variant = int(var) if var else args.variant
    h = C.c_void_p()
    assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0, lib.rt_last_error()
    libs[tag] = (lib, h, lib_variant, evict, env)

out = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def run(tag, n):
    lib, h, lib_variant, evict, env = libs[tag]
    for k in ENV_KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    lib.rt_set_variant(lib_variant)
    for _ in range(n):
        rc = lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(stream))
        assert rc == 0, lib.rt_last_error()


ref = None
casts = None
for tag in libs:
    cnt.zero_()
    run(tag, 2)
    torch.cuda.synchronize()
    img = out.clone()
    c = int(cnt.item()) // 2
    if ref is None:
        ref, casts = img, c
    else:
        same = torch.equal(img.view(torch.int32), ref.view(torch.int32))
        print(f"# {tag}: bit-identical to {next(iter(libs))}: {same}; casts {c} vs {casts}")
times = {t: [] for t in libs}
for r in range(args.rounds):
    for tag in libs:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(tag, args.frames)
        e1.record()
        torch.cuda.synchronize()
        times[tag].append(e0.elapsed_time(e1) / args.frames)
for tag, ts in times.items():
    med, mn = statistics.median(ts), min(ts)
    print(f"{tag:16s} median {med:8.4f} ms  min {mn:8.4f} ms   {casts / med / 1e3:9.1f} Mrays/s (median)  {casts / mn / 1e3:9.1f} (best)")
