#!/usr/bin/env python3
"""Experiment: does the stochastic pass have idle GPU time that independent work could fill?  The same pixels rendered as ONE
tile on one stream, or as K interleaved sub-tiles (each with its own generators and workspace) on K streams at once.

    python tools/exp_dist_streams.py [--world 1] [--epochs 64]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import homework_18_graphics_raytracer_amd as rt

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=1)
ap.add_argument("--epochs", type=int, default=64)
a = ap.parse_args()
W, H, D = 1920, 1080, 8
world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
for k in (1, 2, 4):
    n = a.world * k
    frames = [rt.Frame.rows_of_rank(W, H, D, j * a.world, n) for j in range(k)]  # the rows of rank 0 of `world`, dealt to k sub-tiles
    streams = [torch.cuda.Stream() for _ in range(k)]
    accs = [torch.zeros((f.rows, f.cols, 3), dtype=torch.float32, device="cuda") for f in frames]
    for f, s, acc in zip(frames, streams, accs):  # warm-up: workspaces
        r = rt.Rng(f)
        rt.render_distributed(scene, cam, f, r, a.epochs, accum=acc, stream=s)
        torch.cuda.synchronize()
        r.close()
    best = None
    for rep in range(3):
        rngs = [rt.Rng(f) for f in frames]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f, s, acc, r in zip(frames, streams, accs, rngs):
            rt.render_distributed(scene, cam, f, r, a.epochs, accum=acc, stream=s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / a.epochs
        best = dt if best is None else min(best, dt)
        for r in rngs:
            r.close()
    pixels = sum(f.rows * f.cols for f in frames)
    print(f"share 1/{a.world} ({pixels} pixels) as {k} sub-tile(s) on {k} stream(s): {best:.4f} ms per epoch, {pixels / best / 1e3:.1f} Msamples/s")
