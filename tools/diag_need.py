#!/usr/bin/env python3
"""Diagnostic: of the triangle tests a wave runs, how many do its lanes NEED?  (build: make -C csrc variant TAG=need
EXTRA=-DRT_DIAG_NEED)

A wave runs a cluster's triangles for all its lanes as soon as ONE lane's ray may hit the cluster (rt_cast.h
cluster_skippable).  The counters compare, per kernel family, lane-tests run (active lanes x triangles visited) with
lane-tests needed (per lane: only the clusters its own ray cannot skip) — the room a cast that bins rays by the clusters they
can meet would have.  Whitted frame (rt::pwf_kernel) and the stochastic pass (chain kernel alone: the scene without its
lights has the same chains and no shadow casts; then with lights: chain + shade kernels).

    python tools/diag_need.py [--width 1920 --height 1080 --depth 8 --epochs 8]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

_capi._amd = None
_orig = _capi._load
_capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / "librt_amd_need.so")) if name == "librt_amd.so" else _orig(name)
import homework_18_graphics_raytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--epochs", type=int, default=8)
a = ap.parse_args()
lib = _capi.amd_lib()
for f in (lib.rt_diag_read_need_pwf, lib.rt_diag_read_need_dist):
    f.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]


def report(name, reader):
    buf = (C.c_ulonglong * 96)()
    assert reader(buf, 1) == 0
    casts, tris, run, need = buf[0], buf[1], buf[2], buf[3]
    if casts == 0:
        print(f"{name}: no casts")
        return
    print(f"{name}: {casts} wave-casts, {tris / casts:.1f} triangles visited per wave-cast, {run / max(tris, 1):.1f} active lanes; "
          f"lane-tests needed / run = {need / max(run, 1):.3f}  (needed {need / casts / 64:.1f} triangles per lane-cast of 64 lanes)")
    if buf[4] or buf[9]:
        hist = [buf[16 + n] for n in range(65)]
        tot = max(sum(hist), 1)
        cum, q = 0, {}
        for n, v in enumerate(hist):
            cum += v
            for pct in (10, 25, 50, 75, 90, 99):
                if pct not in q and cum * 100 >= pct * tot:
                    q[pct] = n
        print(f"    clustered leaves: {buf[4]} pair-wise ({buf[5] / max(buf[4], 1):.2f} passes each, {buf[6] / max(buf[5], 1):.2f} of the passes reach the signed areas "
              f"with {buf[7] / max(buf[6], 1):.1f} pairs, {buf[8]} pairs accepted), {buf[9]} wave-uniform; lanes needing a leaf: percentiles {q}; "
              f"casts redone for NaN {buf[10]}, with lanes missing {buf[11]}")


world, cam = rt.reference_world(), rt.reference_camera()
frame = rt.Frame.full(a.width, a.height, a.depth)
scene = rt.Scene(world)
report("(reset)", lib.rt_diag_read_need_pwf)
rt.render_whitted(scene, cam, frame)
torch.cuda.synchronize()
report("Whitted frame, rt::pwf_kernel", lib.rt_diag_read_need_pwf)

desc = world.desc()
n_lights = desc.n_lights
for lights, label in ((0, "stochastic pass, chain kernel alone (scene without lights)"), (n_lights, "stochastic pass, chain + shade kernels")):
    desc.n_lights = lights
    sc = rt.Scene(desc)
    rng = rt.Rng(frame)
    accum = torch.zeros((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    report("(reset)", lib.rt_diag_read_need_dist)
    rt.render_distributed(sc, cam, frame, rng, a.epochs, accum=accum)
    torch.cuda.synchronize()
    report(label, lib.rt_diag_read_need_dist)
