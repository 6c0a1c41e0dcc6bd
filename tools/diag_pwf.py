#!/usr/bin/env python3
"""Diagnostic: where the barrier-free persistent kernel's workgroups spend their time (build: make -C csrc variant TAG=pastats
EXTRA=-DPA_STATS)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

W, H, depth = 1920, 1080, 8
world_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tag = sys.argv[2] if len(sys.argv) > 2 else "pastats"  # another -DPA_STATS build
lib = C.CDLL(str(_capi.PKG_DIR / f"variants/librt_amd_{tag}.so"))
lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
world = rt.reference_world(); cam = rt.reference_camera(); desc = world.desc()
frame = rt.Frame.full(W, H, depth) if world_n == 1 else rt.Frame.rows_of_rank(W, H, depth, 0, world_n)
h = C.c_void_p(); assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0
out = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
lib.rt_set_variant(18)
lib.rt_diag_read_pwf_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for k in range(5):  # three untimed frames (the first touches the arenas' pages for the first time), then two counted ones
    if k == 3:
        assert lib.rt_diag_read_pwf_phases((C.c_ulonglong * 32)(), 1) == 0
    assert lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
g = (C.c_uint32 * 32)()
assert lib.rt_diag_read_pwf(g) == 0
wgs = g[14]
print(f"workgroups {wgs}; nodes/WG {g[15] / wgs:.0f}")
print(f"through the arena (HBM) per frame: {g[15]} NODE items (32 B), {g[25]} SHADE items (80 B), {g[26]} REFR items (48 B); nodes with ids from the top of the arena (roots + LDS-queued): {g[27]}")
print(f"  shadow casts {g[18]}: {g[19]} with a hit ({100.0 * g[19] / max(g[18], 1):.1f} %), {g[31]} occluded ({100.0 * g[31] / max(g[18], 1):.1f} %)")
print(f"  SHADE items that found no room in the LDS queue of light 0 / 1 / 2+: {g[28]} / {g[29]} / {g[30]}")
print(f"wave's own loop: mean {g[8] / (8 * wgs) / 100:.1f} us, max {g[9] / 100:.1f} us")
print(f"until the WG's last wave left the loop: mean {g[10] / wgs / 100:.1f} us, max {g[11] / 100:.1f} us")
print(f"fold: mean {g[12] / wgs / 100:.1f} us, max {g[13] / 100:.1f} us;  whole WG: min {g[16] / 100:.1f} us, max {g[17] / 100:.1f} us")
h = [g[20 + k] for k in range(5)]
print("chunks by item count (<=8, <=16, <=32, <64, 64):", h, "shares", [round(x / max(1, sum(h)), 3) for x in h])

# wave time by phase (shader-clock ticks summed over all waves, both frames above; the shares are what matters)
ph = (C.c_ulonglong * 32)()
assert lib.rt_diag_read_pwf_phases(ph, 1) == 0
ph = list(ph)
total = sum(ph[0:16]) + ph[24] + ph[25]
names = ["NODE", "REFR", "TILE", "SHADE"]
print(f"wave time by phase, share of {total:.3e} ticks (find / load / cast / after), chunks, mean lanes per chunk:")
for k, n in enumerate(names):
    chunks = max(1, ph[16 + k])
    print(f"  {n:5s} find {ph[k] / total:6.3f}  load {ph[4 + k] / total:6.3f}  cast {ph[8 + k] / total:6.3f}  after {ph[12 + k] / total:6.3f}"
          f"   chunks {ph[16 + k]:8d}  lanes/chunk {ph[20 + k] / chunks:5.1f}  ticks/chunk: load {ph[4 + k] / chunks:7.0f} cast {ph[8 + k] / chunks:7.0f} after {ph[12 + k] / chunks:7.0f}")
print(f"  idle/sleep {ph[24] / total:6.3f}   fold {ph[25] / total:6.3f}")
