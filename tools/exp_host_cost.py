#!/usr/bin/env python3
"""Experiment: what a render call costs the HOST (Python wrapper, C ABI, the launches), against what it costs the GPU — for shares so
small that several frames in flight (dist.FramePipeline) approach the rate at which one thread can issue them."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402

scene = rt.Scene(rt.reference_world())
cam = rt.reference_camera()
for (w, h) in [(8, 8), (1920, 135)]:
    frame = rt.Frame.full(w, h, 8)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    for _ in range(20):
        rt.render_whitted(scene, cam, frame, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        rt.render_whitted(scene, cam, frame, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{w}x{h}: the host issues a call in {(t1 - t0) / 2000 * 1e6:.1f} us; with the GPU drained {(t2 - t0) / 2000 * 1e6:.1f} us per call")
