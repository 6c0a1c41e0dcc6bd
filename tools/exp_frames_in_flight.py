#!/usr/bin/env python3
"""Experiment: frames in flight.  A frame's (or a 1/N share's) time ends with the critical path of its deepest pixels while most
of the GPU idles.  Independent frames rendered on several streams (each with its own workspace) let the
next frame's workgroups take the slots the draining frame frees.  Same pixels; throughput, not latency.

    python tools/exp_frames_in_flight.py [--world 1] [--frames 40]
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
ap.add_argument("--frames", type=int, default=40)
a = ap.parse_args()
W, H, D = 1920, 1080, 8
world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
frame = rt.Frame.full(W, H, D) if a.world == 1 else rt.Frame.rows_of_rank(W, H, D, 0, a.world)
ref = rt.render_whitted(scene, cam, frame).clone()
torch.cuda.synchronize()
for k in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(k)]
    outs = [torch.empty_like(ref) for _ in range(k)]
    for s, o in zip(streams, outs):  # warm-up: each stream allocates its workspace
        rt.render_whitted(scene, cam, frame, out=o, stream=s)
    torch.cuda.synchronize()
    best = None
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(a.frames):
            rt.render_whitted(scene, cam, frame, out=outs[i % k], stream=streams[i % k])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / a.frames
        best = dt if best is None else min(best, dt)
    same = all(torch.equal(o.view(torch.int32), ref.view(torch.int32)) for o in outs)
    print(f"share 1/{a.world}: {k} frame(s) in flight: {best:.4f} ms per frame  (bit-identical: {same})")
