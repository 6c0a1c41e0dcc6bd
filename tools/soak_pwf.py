#!/usr/bin/env python3
"""Soak test of the wavefront kernel's queue protocol: many frames of several sizes, scenes and arena budgets on two streams,
every one compared bit for bit (radiance and cast count) with the first render of its configuration.  A race in the LDS
queues would show up as a mismatch (or as the spin limit's fallback, which the cast count of the per-pixel kernel gives away
only by its timing, so the kernel times are printed too).

    timeout -k 10 300 python tools/soak_pwf.py [--seconds 60]
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402
from homework_18_graphics_raytracer_amd import _capi  # noqa: E402
import _scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
a = ap.parse_args()

lib = _capi.amd_lib()
ref_world, ref_cam = rt.reference_world(), rt.reference_camera()
cases = []
for (w, h, d) in [(1920, 1080, 8), (640, 360, 8), (97, 61, 3), (1280, 960, 5), (2560, 1440, 4), (333, 777, 12), (3840, 2160, 2)]:
    cases.append((f"reference {w}x{h} d{d}", rt.Scene(ref_world), ref_cam, rt.Frame.full(w, h, d), 6))
for seed, nt, ns in [(5, 33, 4), (8, 131, 3), (9, 200, 9)]:
    world = _scenes.random_world(seed, nt, ns)
    cases.append((f"random {seed} {nt}+{ns} 480x270 d6", rt.Scene(world), _scenes.camera(seed), rt.Frame.full(480, 270, 6), 16))
cases.append(("reference 640x360 d8, budget 3 (overflows into the per-pixel kernel somewhere)", rt.Scene(ref_world), ref_cam, rt.Frame.full(640, 360, 8), 3))

streams = [torch.cuda.Stream(), torch.cuda.Stream()]
first = {}
frames = 0
t0 = time.time()
rounds = 0
while time.time() - t0 < a.seconds:
    for k, (name, scene, cam, frame, budget) in enumerate(cases):
        _capi.check(lib.rt_set_wavefront_budget(budget))
        outs = []
        for s in streams:
            cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
            with torch.cuda.stream(s):
                out = rt.render_whitted(scene, cam, frame, ray_count=cnt, stream=s)
            outs.append((out, cnt))
        torch.cuda.synchronize()
        for out, cnt in outs:
            img = out.view(torch.int32)
            if k not in first:
                first[k] = (img.clone(), int(cnt.item()))
            else:
                if not torch.equal(img, first[k][0]) or int(cnt.item()) != first[k][1]:
                    print(f"MISMATCH: {name}, round {rounds}: casts {int(cnt.item())} vs {first[k][1]}, {(img != first[k][0]).sum().item()} words differ")
                    sys.exit(1)
            frames += 1
    rounds += 1
_capi.check(lib.rt_set_wavefront_budget(6))
print(f"{frames} frames in {rounds} rounds over {len(cases)} configurations on two streams: all identical to their first render ({time.time() - t0:.0f} s)")
