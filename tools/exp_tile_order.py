#!/usr/bin/env python3
"""Experiment: the tiles of the Whitted frame handed out DEAREST FIRST.  The persistent kernel records, per tile, how many of its
pixels recursed (rt_diag_set_tile_cost); sorted by that, descending, the order goes back in (rt_diag_set_tile_order): a workgroup's
own tiles are then the dear half, dealt round-robin, and the frame-wide counter ends with the cheapest tiles.

    python tools/exp_tile_order.py [--worlds 1 8] [--frames 200]
"""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402
from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, nargs="+", default=[1, 8])
ap.add_argument("--frames", type=int, default=200)
a = ap.parse_args()
lib = _capi.amd_lib()
lib.rt_diag_set_tile_order.argtypes = [C.c_void_p]
lib.rt_diag_set_tile_cost.argtypes = [C.c_void_p]
scene = rt.Scene(rt.reference_world())
cam = rt.reference_camera()
W, H, D = 1920, 1080, 8
for world in a.worlds:
    frame = rt.Frame.full(W, H, D) if world == 1 else rt.Frame.rows_of_rank(W, H, D, 0, world)
    out = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
    n_tiles = (frame.rows * frame.cols + 63) // 64

    def timed(n):
        for _ in range(5):
            rt.render_whitted(scene, cam, frame, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            rt.render_whitted(scene, cam, frame, out=out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n

    base = timed(a.frames)
    ref = out.clone()
    cost = torch.zeros(n_tiles, dtype=torch.int32, device="cuda")
    lib.rt_diag_set_tile_cost(C.c_void_p(cost.data_ptr()))
    rt.render_whitted(scene, cam, frame, out=out)
    torch.cuda.synchronize()
    lib.rt_diag_set_tile_cost(None)
    hist = torch.bincount(cost.clamp(0, 64), minlength=65).cpu().tolist()
    for name, order in (("dearest first", torch.argsort(cost, descending=True, stable=True)),
                        ("dearest first, equal costs scattered", torch.argsort(cost.to(torch.int64) * n_tiles + (torch.arange(n_tiles, device="cuda") * 20021) % n_tiles, descending=True)),
                        ("cheapest first", torch.argsort(cost, descending=False, stable=True))):
        order = order.to(torch.int32).contiguous()
        lib.rt_diag_set_tile_order(C.c_void_p(order.data_ptr()))
        ms = timed(a.frames)
        same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
        lib.rt_diag_set_tile_order(None)
        print(f"share 1/{world}: {name}: {ms:.4f} ms per frame against {base:.4f} in the golden-section order ({base / ms:.3f}x), frame identical: {same}", flush=True)
    print(f"share 1/{world}: tiles with 0 / 1-63 / 64 recursing pixels: {hist[0]} / {sum(hist[1:64])} / {hist[64]} of {n_tiles}", flush=True)
