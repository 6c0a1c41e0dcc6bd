#!/usr/bin/env python3
"""Diagnostic: where a wave of the chain kernel spends its time inside the pair-wise cast (build: make -C csrc variant
TAG=ptime EXTRA=-DRT_DIAG_PAIR_TIME).  Accumulated per wave in registers, added up once when the wave leaves.

    python tools/diag_pair_time.py [--epochs 8 --burn 32]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

LIB = next((x.split("=", 1)[1] for x in sys.argv if x.startswith("--lib=")), "ptime")
sys.argv = [x for x in sys.argv if not x.startswith("--lib=")]
_capi._amd = None
_orig = _capi._load
_capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / ("librt_amd_" + LIB + ".so"))) if name == "librt_amd.so" else _orig(name)
import homework_18_graphics_raytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--burn", type=int, default=32)
ap.add_argument("--world", type=int, default=1, help="rank 0's interleaved row band of this many ranks (a multi-GPU share)")
a = ap.parse_args()
lib = _capi.amd_lib()
lib.rt_diag_read_pair_time.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
world, cam = rt.reference_world(), rt.reference_camera()
frame = rt.Frame.full(a.width, a.height, a.depth) if a.world == 1 else rt.Frame.rows_of_rank(a.width, a.height, a.depth, 0, a.world)
scene = rt.Scene(world)
rng = rt.Rng(frame)
accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
if a.burn:
    rt.render_distributed(scene, cam, frame, rng, a.burn, accum=accum)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
assert lib.rt_diag_read_pair_time(buf, 1) == 0
if hasattr(lib, "rt_diag_read_pair_time_critical"):
    lib.rt_diag_read_pair_time_critical.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.rt_diag_read_pair_time_critical((C.c_ulonglong * 4)(), 1) == 0  # the burn-in is not part of the launch looked at
rt.render_distributed(scene, cam, frame, rng, a.epochs, accum=accum)
torch.cuda.synchronize()
assert lib.rt_diag_read_pair_time(buf, 1) == 0
steps, total = buf[0], buf[8]
names = {1: "classify nodes", 2: "wave-uniform runs", 3: "leaf set-up", 4: "pair passes", 5: "signed areas", 6: "finish"}
print(f"chain kernel, {a.epochs} epochs after {a.burn}: {steps} wave-steps, {total / max(steps, 1):.0f} ticks per step, the cast {100.0 * buf[7] / max(total, 1):.1f} % of the wave time")
if buf[14]:
    print(f"  lanes at work per step: {buf[14] / max(steps, 1):.1f} of 64; steps after the work queue ran dry: {100.0 * buf[15] / max(steps, 1):.1f} % of all")
if buf[9] or buf[10]:
    step = {9: "fetching work + shoot_focus", 10: "the hit + the level's factor", 12: "the level's draws + scatter_hit | get_refract's exit", 13: "the rest"}
    print("  outside the cast: " + ", ".join(f"{step[k]} {100.0 * buf[k] / max(total, 1):.1f} % ({buf[k] / max(steps, 1):.0f} ticks)" for k in (9, 10, 12, 13)))
if hasattr(lib, "rt_diag_read_pair_time_critical"):
    lib.rt_diag_read_pair_time_critical.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    cb = (C.c_ulonglong * 4)()
    assert lib.rt_diag_read_pair_time_critical(cb, 1) == 0
    if cb[2]:
        print(f"  critical path of the launch(es): {cb[2]} waves, {steps / cb[2]:.1f} steps per wave on average, {cb[0]} in the wave that made the most; the longest wave ran {cb[1]} ticks "
              f"= {cb[1] / max(cb[0], 1):.0f} ticks per step if it is that wave (a wave issues one VALU instruction per four cycles however few of its lanes work: "
              f"a launch cannot end before its dearest pixels' steps are made, one after the other)")
print("  of the cast: " + ", ".join(f"{names[k]} {100.0 * buf[k] / max(buf[7], 1):.1f} % ({buf[k] / max(steps, 1):.0f} ticks)" for k in range(1, 7)))

if hasattr(lib, "rt_diag_read_shade_time"):
    lib.rt_diag_read_shade_time.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    sb = (C.c_ulonglong * 8)()
    assert lib.rt_diag_read_shade_time(sb, 1) == 0
    st = max(sb[5], 1)
    # the per-request kernel: list building + sort | request load + material | lights -> directional, facing | the cast | diffuse / specular
    # the lights-as-phases kernel (default): place | prepare (request, adjust_normal, light_asks) | a chunk's set-up | the cast | lit bit + finish (colours, lights in order)
    nm = ["lists (place | build + sort)", "prepare (request load + material)", "chunk set-up (lights -> directional)", "the cast", "after the cast (lit, finish | diffuse / specular)", None, None]
    print(f"shade kernel (all calls since the start, burn included): {sb[6]} wave-casts, {sb[3] / max(sb[6], 1):.0f} ticks per wave-cast; wave time: "
          + ", ".join(f"{nm[k]} {100.0 * sb[k] / st:.1f} %" for k in range(5)))
