#!/usr/bin/env python3
"""Stress run: the same 1920x1080 depth-8 frame many times through the persistent wavefront kernel (whose scheduling
differs from run to run); every frame must equal the per-pixel kernel's, bit for bit, with the same cast count."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = _capi.amd_lib()
world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
frame = rt.Frame.full(1920, 1080, 8)
out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
_capi.check(lib.rt_set_variant(2))
rt.render_whitted(scene, cam, frame, out=out, ray_count=cnt)
torch.cuda.synchronize()
ref = out.clone(); ref_casts = int(cnt.item())
_capi.check(lib.rt_set_variant(_capi.DEFAULT_VARIANT))
bad = 0
for k in range(n):
    cnt.zero_(); out.zero_()
    rt.render_whitted(scene, cam, frame, out=out, ray_count=cnt)
    torch.cuda.synchronize()
    if not torch.equal(out.view(torch.int32), ref.view(torch.int32)) or int(cnt.item()) != ref_casts:
        bad += 1
        print("MISMATCH at frame", k, int(cnt.item()), ref_casts, flush=True)
print(f"{n} frames: {bad} bad")
sys.exit(1 if bad else 0)
