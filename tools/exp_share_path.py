#!/usr/bin/env python3
"""Experiment: what decides the time of a 1/N share of the Whitted frame, and do two cheap levers move it?
 (i)  start the deepest tiles first, using the PREVIOUS frame's per-tile cast counts as the order (free in the progressive
      loop main.rs:1129-1173, where the same view is rendered again and again) — here the exact counts, from the oracle,
      handed to the kernel through the diagnostic hook rt_diag_set_tile_order: the best such a hint could do;
 (ii) workgroups of 4 waves instead of 8 on a small share (variant library built with -DPA_WAVES=4u).

    python tools/exp_share_path.py [--world 8] [--libs main,pw4]
"""
import argparse
import ctypes as C
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--libs", default="main")
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
W, H, D = 1920, 1080, 8
world = rt.reference_world(); cam = rt.reference_camera(); desc = world.desc()
frame = rt.Frame.full(W, H, D) if a.world == 1 else rt.Frame.rows_of_rank(W, H, D, 0, a.world)
rows, cols = frame.rows, frame.cols
ol = _oracle.lib()
ol.orc_render_whitted_counts.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_int]
img = np.zeros((rows, cols, 3), np.float32); cnt = np.zeros((rows, cols), np.uint32)
ol.orc_render_whitted_counts(C.byref(desc), C.byref(cam), C.byref(frame), img.ctypes.data, cnt.ctypes.data, 0)
# tile t = 64 consecutive slots of the band image: 8-row bands, column-major inside a band (ragged last band: fewer rows)
n_tiles = (rows * cols + 63) // 64
cost = np.zeros(n_tiles, np.int64)
slot = 0
for b0 in range(0, rows, 8):
    br = min(8, rows - b0)
    block = cnt[b0:b0 + br].T.reshape(-1)  # column-major inside the band
    idx = (slot + np.arange(block.size)) // 64
    np.add.at(cost, idx, block)
    slot += block.size
order = np.argsort(-cost, kind="stable").astype(np.uint32)
d_order = torch.from_numpy(order.astype(np.int32)).cuda()
# a proxy for the length of a tile's dependent chains rather than for its work: the casts that depth 8 adds to depth 2
shallow = rt.Frame(frame.width, frame.height, 2, frame.x0, frame.y0, frame.x1, frame.y1, frame.y_step)
cnt2 = np.zeros((rows, cols), np.uint32)
img2 = np.zeros((rows, cols, 3), np.float32)
ol.orc_render_whitted_counts(C.byref(desc), C.byref(cam), C.byref(shallow), img2.ctypes.data, cnt2.ctypes.data, 0)
deep = np.zeros(n_tiles, np.int64)
slot = 0
for b0 in range(0, rows, 8):
    br = min(8, rows - b0)
    block = (cnt[b0:b0 + br].astype(np.int64) - cnt2[b0:b0 + br].astype(np.int64)).T.reshape(-1)
    idx = (slot + np.arange(block.size)) // 64
    np.maximum.at(deep, idx, block)  # the deepest PIXEL of the tile decides
    slot += block.size
d_deep = torch.from_numpy(np.argsort(-deep, kind="stable").astype(np.int32)).cuda()
out = torch.empty((rows, cols, 3), dtype=torch.float32, device="cuda")
print(f"share 1/{a.world}: {rows}x{cols} pixels, {n_tiles} tiles, casts {int(cnt.sum())}, deepest tile {int(cost.max())} casts, mean {cost.mean():.0f}")
for tag in a.libs.split(","):
    path = _capi.PKG_DIR / ("librt_amd.so" if tag == "main" else f"variants/librt_amd_{tag}.so")
    lib = C.CDLL(str(path))
    lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
    lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_diag_set_tile_order.argtypes = [C.c_void_p]
    h = C.c_void_p(); assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0
    lib.rt_set_variant(18)
    res = {}
    for name, ptr in (("default order", None), ("most casts first", d_order.data_ptr()), ("longest chains first", d_deep.data_ptr())):
        lib.rt_diag_set_tile_order(C.c_void_p(ptr) if ptr else None)
        times = []
        for r in range(a.rounds + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), None, None) == 0
            e1.record(); torch.cuda.synchronize()
            if r >= 2:
                times.append(e0.elapsed_time(e1) / 5)
        res[name] = statistics.median(times)
        same = np.array_equal(out.cpu().numpy().view(np.uint32), img.view(np.uint32))
        print(f"  {tag:8s} {name:22s} median {res[name]:.4f} ms   bit-identical to the oracle: {same}")
    lib.rt_diag_set_tile_order(None)
