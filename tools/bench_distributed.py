#!/usr/bin/env python3
"""Timing of the distributed (stochastic / DoF) pass, configs[3]/[4] of BASELINE.json on ONE GPU's share.

    python tools/bench_distributed.py [--width 1920 --height 1080 --depth 8 --epochs 8 --calls 3 --rank 0 --world 1]

Prints one JSON line: ms per epoch, Msamples/s, Mrays/s (casts), and the RNG-state HBM traffic the pass implies.
Not the driver's bench (bench.py measures the Whitted headline); this is the measurement SURVEY §8d asks for rows 17-19.
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import dist as rtdist

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--calls", type=int, default=3)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--burn", type=int, default=0, help="untimed epochs first: the pixels' random streams drift out of step over the first ~100 epochs, which is the state a long run is in")
ap.add_argument("--warm", type=int, default=0, help="1: an untimed call of --epochs epochs on a generator of its own first, so that the timed calls find their workspace allocated (bench.py's convention)")
ap.add_argument("--fresh", type=int, default=0, help="1: no 1-epoch warm-up on the timed generator: the timed call starts from freshly seeded streams (bench.py's convention; use with --warm 1 --calls 1)")
ap.add_argument("--split", type=int, default=-1, help="1: chain/shade/unwind kernels, 0: the fused kernel, -1: library default")
ap.add_argument("--lib", default=None, help="variant tag: use variants/librt_amd_<tag>.so instead of the in-tree library")
a = ap.parse_args()

if a.lib:
    import ctypes as C
    from homework_18_graphics_raytracer_amd import _capi
    _capi._amd = None
    _orig = _capi._load
    _capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / f"librt_amd_{a.lib}.so")) if name == "librt_amd.so" else _orig(name)
from homework_18_graphics_raytracer_amd import _capi as _c
_c.amd_lib().rt_set_distributed_split(a.split)
world = rt.reference_world()
cam = rt.reference_camera()
scene = rt.Scene(world)
frame = rtdist.shard_frame(a.width, a.height, a.depth, a.rank, a.world)
if a.warm:
    warm = rt.Rng(frame)
    rt.render_distributed(scene, cam, frame, warm, a.epochs, accum=torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    warm.close()
    del warm
t0 = time.perf_counter()
rng = rt.Rng(frame)
torch.cuda.synchronize()
t_seed = time.perf_counter() - t0
accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
if not a.fresh:
    rt.render_distributed(scene, cam, frame, rng, 1, accum=accum)  # warm-up (also advances the stream; fine for timing)
if a.burn:
    rt.render_distributed(scene, cam, frame, rng, a.burn, accum=accum)
torch.cuda.synchronize()
cnt.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.calls):
    rt.render_distributed(scene, cam, frame, rng, a.epochs, accum=accum, ray_count=cnt)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
pixels = frame.rows * frame.cols
samples = pixels * a.epochs * a.calls
casts = int(cnt.item())
print(json.dumps({
    "pass": "distributed", "split": a.split, "burn": a.burn, "width": a.width, "height": a.height, "depth": a.depth, "tile_pixels": pixels,
    "epochs_per_call": a.epochs, "calls": a.calls, "ms_per_epoch": round(ms / (a.epochs * a.calls), 4),
    "Msamples_per_s": round(samples / ms / 1e3, 2), "Mrays_per_s": round(casts / ms / 1e3, 2),
    "casts_per_sample": round(casts / samples, 3), "rng_state_GB": round(pixels * 4128 / 1e9, 3),
    "rng_seed_ms": round(t_seed * 1e3, 2),
    "rng_algorithmic_bytes_per_sample_if_reloaded_every_epoch": 4128,
    "rng_note": "two 2064 B banks per pixel stay in HBM; a visit reads the words it draws (~40 B per sample) and the look-ahead pass moves 3 KB per 256-word block (~120 B per sample); measured HBM traffic of all kernels: profiles/r03_dist_pmc.txt",
}))
