#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of the static-tiles kernel (build: make variant TAG=diag EXTRA=-DRT_DIAG_TIMELINE).
Prints the number of waves alive over time and per-iteration wave durations.  Never used for timing claims."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
lib = C.CDLL(str(_capi.PKG_DIR / "variants/librt_amd_diag.so"))
lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
lib.rt_diag_set_timeline.argtypes = [C.c_void_p]
world = rt.reference_world(); cam = rt.reference_camera(); desc = world.desc()
frame = rt.Frame.full(W, H, depth)
h = C.c_void_p(); assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0
n_waves = (W * H + 63) // 64
tl = torch.zeros((n_waves, 4), dtype=torch.int64, device="cuda")
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
lib.rt_set_variant(variant)
lib.rt_diag_set_timeline(C.c_void_p(tl.data_ptr()))
for _ in range(3):
    assert lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), None, None) == 0, lib.rt_last_error()
torch.cuda.synchronize()
t = tl.cpu().numpy()
t0, t1, it, cast_cyc = t[:, 0], t[:, 1], t[:, 2] & 0xFFFF, t[:, 3]
wave_cyc = t[:, 2] >> 16
base = t0.min()
s = (t0 - base) / 100.0  # us
e = (t1 - base) / 100.0
print(f"depth {depth} variant {variant}: kernel span {e.max():.1f} us; waves {len(s)}; mean wave {np.mean(e - s):.1f} us; max {np.max(e - s):.1f} us")
edges = np.linspace(0, e.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    alive = np.sum((s <= mid) & (e > mid))
    started = np.sum((s >= a) & (s < b))
    print(f"  t={mid:8.1f} us alive={alive:5d} started_in_bin={started:5d}")
dur = e - s
for k in (1, 2, 3, 4, 5, 8, 12, 20, 30, 40, 50):
    m = it == k
    if m.any():
        print(f"  iterations={k:3d}: waves={m.sum():6d} mean dur {dur[m].mean():8.1f} us  ({dur[m].mean() / k:6.1f} us/iter)")
for lo, hi in ((1, 9), (9, 21), (21, 100)):
    m = (it >= lo) & (it < hi)
    if m.any():
        print(f"  waves with {lo}..{hi - 1} trips: cast {cast_cyc[m].sum() / it[m].sum():.0f} cycles/trip, rest {(wave_cyc[m].sum() - cast_cyc[m].sum()) / it[m].sum():.0f} cycles/trip")
print(f"  share of wave cycles inside cast(): {cast_cyc.sum() / wave_cyc.sum():.3f}  (mean cast {cast_cyc.sum() / it.sum():.0f} cycles/iteration, wave {wave_cyc.sum() / it.sum():.0f} cycles/iteration)")
np.save("gpurun_out/timeline.npy", t)
