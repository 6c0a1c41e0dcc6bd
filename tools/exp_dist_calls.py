#!/usr/bin/env python3
"""Experiment: what makes one 64-epoch rt_render_distributed call slow inside bench.py (16 ms/epoch vs 2.3)?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import homework_18_graphics_raytracer_amd as rt

world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
frame = rt.Frame.full(1920, 1080, 8)
accum = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")

def timed(label, rng, epochs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rt.render_distributed(scene, cam, frame, rng, epochs, accum=accum)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label}: {epochs} epochs, events {e0.elapsed_time(e1):.1f} ms, host call {1e3*(t1-t0):.1f} ms, wall {1e3*(t2-t0):.1f} ms", flush=True)

mode = sys.argv[1]
if mode == "a":   # warm 8, close, new rng, 64
    w = rt.Rng(frame); timed("warm8", w, 8); w.close(); del w
    r = rt.Rng(frame); timed("64 fresh rng", r, 64); timed("64 again", r, 64)
elif mode == "b":  # same rng: warm 8 then 64
    r = rt.Rng(frame); timed("warm8", r, 8); timed("64 same rng", r, 64); timed("64 again", r, 64)
elif mode == "c":  # whitted first
    out = rt.render_whitted(scene, cam, frame); torch.cuda.synchronize()
    r = rt.Rng(frame); timed("warm1", r, 1); timed("64", r, 64)
elif mode == "d":  # warm 8, keep warm alive, new rng 64
    w = rt.Rng(frame); timed("warm8", w, 8)
    r = rt.Rng(frame); timed("64 fresh rng, warm alive", r, 64)
