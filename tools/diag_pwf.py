#!/usr/bin/env python3
"""Diagnostic: iteration / slot-use totals of the persistent-wavefront kernel (build: make -C csrc variant TAG=pwstats EXTRA=-DPW_STATS)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

W, H, depth = 1920, 1080, 8
lib = C.CDLL(str(_capi.PKG_DIR / "variants/librt_amd_pwstats.so"))
lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
world = rt.reference_world(); cam = rt.reference_camera(); desc = world.desc()
frame = rt.Frame.full(W, H, depth)
h = C.c_void_p(); assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
lib.rt_set_variant(18)
for _ in range(2):
    assert lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
g = (C.c_uint32 * 32)()
assert lib.rt_diag_read_pwf(g) == 0
wgs = g[17]
print(f"workgroups {wgs}; iterations total {g[8]} (mean {g[8] / wgs:.1f}, max {g[16]}); slots: node {g[9]} refr {g[10]} tiles {g[11]} shade {g[12]} (partial {g[13]})")
print(f"slot use {(g[9] + g[10] + g[11] + g[12]) / (8 * g[8]):.3f}; mean WG time {g[14] / wgs / 100:.1f} us, max {g[15] / 100:.1f} us; mean iteration {g[14] / g[8] / 100:.2f} us; max nodes in an arena {g[18]}")

print(f"waves busy with chunks: {g[19] / (8 * g[14]):.3f} of the workgroups' main-loop time")
import numpy as np
buf = np.zeros((1024, 8), dtype=np.uint32)
n = lib.rt_diag_read_pwf_groups(buf.ctypes.data_as(C.c_void_p), 1024)
r = buf[:n].astype(np.int64)
t0 = r[:, 4]; base = t0.min()
start = (t0 - base) / 100.0; dur = r[:, 1] / 100.0
order = np.argsort(dur)
print("per-WG: dur us pctl", np.percentile(dur, [0, 10, 50, 90, 99, 100]).round(0), "start us max", start.max().round(1))
print("iterations pctl", np.percentile(r[:, 0], [0, 10, 50, 90, 99, 100]), "tiles pctl", np.percentile(r[:, 3], [0, 10, 50, 90, 100]), "nodes pctl", np.percentile(r[:, 2], [0, 50, 90, 100]))
exh = r[:, 5] / 100.0
print("time tiles ran out (per WG) pctl", np.percentile(exh, [0, 10, 50, 90, 100]).round(0), "iterations after that pctl", np.percentile(r[:, 0] - r[:, 6], [0, 10, 50, 90, 100]))
pend = r[:, 7]
print("queued full chunks then: node", np.percentile(pend & 1023, [50, 90, 100]), "refr", np.percentile((pend >> 10) & 1023, [50, 90, 100]), "shade", np.percentile(pend >> 20, [50, 90, 100]))
for i in order[-8:]:
    print(f"  WG {i}: dur {dur[i]:.0f} us iters {r[i,0]} nodes {r[i,2]} tiles {r[i,3]} ran out at {exh[i]:.0f} us / iteration {r[i,6]}, queued n/f/s {pend[i] & 1023}/{(pend[i] >> 10) & 1023}/{pend[i] >> 20}")
