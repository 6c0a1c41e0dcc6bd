#!/usr/bin/env python3
"""What finish_frame_sharded costs per frame (post_process over the ranks' bands + sRGB/u8 + u8 gather), this tree's stream-ordered
device passes against round 3's torch-op version (homework-18-graphics-raytracer_amd/_dist_r03_scratch.py, a scratch copy of the file
as it was at b82594e — `git show b82594e:homework-18-graphics-raytracer_amd/dist.py > homework-18-graphics-raytracer_amd/_dist_r03_scratch.py`
before the run; not committed), on one rank with a single-rank RCCL group.

    python tools/exp_sharded_finish.py [--steps 20]
"""
import argparse
import importlib
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import torch.distributed as dist

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import dist as rtdist

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H, D = 1920, 1080, 8
scene = rt.Scene(rt.reference_world())
band = rt.render_whitted(scene, rt.reference_camera(), rt.Frame.full(W, H, D))
torch.cuda.synchronize()
impls = {"round 4 (device passes, no host sync)": lambda b: rtdist.finish_frame_sharded(b, H, 0, 1, sync=False)}
old = ROOT / "homework-18-graphics-raytracer_amd" / "_dist_r03_scratch.py"
if old.exists():
    r03 = importlib.import_module("homework_18_graphics_raytracer_amd._dist_r03_scratch")
    impls["round 3 (torch ops, host round trips)"] = lambda b: r03.finish_frame_sharded(b, H, 0, 1)
ref = None
for name, fn in impls.items():
    works = [band.clone() for _ in range(a.steps + 2)]
    for k in range(2):
        fn(works[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        u8, d = fn(works[2 + k])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    d = float(d.item()) if torch.is_tensor(d) else float(d)
    same = True if ref is None else bool(torch.equal(u8, ref[0]) and d == ref[1])
    if ref is None:
        ref = (u8.clone(), d)
    print(f"{name}: {ms:.3f} ms per 1920x1080 frame, divisor {d!r}, same u8 frame and divisor as the first: {same}")
dist.destroy_process_group()
