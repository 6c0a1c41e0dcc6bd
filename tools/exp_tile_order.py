#!/usr/bin/env python3
"""Experiment: how much would a better tile order buy the persistent kernel?  The orders tried: the ideal one (tiles by
descending cast count, taken from the CPU oracle's per-pixel counts), orders a cheap probe could produce (the centre
pixel's cast count, capped or not; the material its primary ray hits), ascending, and plain image order — handed to the
kernel through the diagnostic hook rt_diag_set_tile_order.  Results: profiles/README.md."""
import ctypes as C
import sys
import statistics
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle

W, H, D = 1920, 1080, 8
world = rt.reference_world(); cam = rt.reference_camera(); scene = rt.Scene(world)
frame = rt.Frame.full(W, H, D)
lib = _capi.amd_lib()
lib.rt_diag_set_tile_order.argtypes = [C.c_void_p]
ol = _oracle.lib()
ol.orc_render_whitted_counts.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_int]
img = np.zeros((H, W, 3), np.float32); cnt = np.zeros((H, W), np.uint32)
desc = world.desc()
ol.orc_render_whitted_counts(C.byref(desc), C.byref(cam), C.byref(frame), img.ctypes.data, cnt.ctypes.data, 0)
# tile t = 64 consecutive slots: 8-row bands, column-major inside a band
bands = H // 8
cost = cnt[: bands * 8].reshape(bands, 8, W // 8, 8).sum(axis=(1, 3)).reshape(-1)  # tile = band * (W/8) + col block
n_tiles = (W * H + 63) // 64
assert cost.size == n_tiles
# heuristic: the material the tile's centre pixel sees first (one primary cast per tile)
from homework_18_graphics_raytracer_amd._capi import Material
mats = [desc.materials[i] for i in range(desc.n_materials)]
weight = np.zeros(n_tiles, np.int64)
ray = _oracle.OrcRay(); hit = _oracle.OrcHit(); clip = (C.c_float * 2)()
tw = W // 8
for t in range(n_tiles):
    band, cb = divmod(t, tw)
    x, y = cb * 8 + 4, band * 8 + 4
    ol.orc_clip(W, H, x, y, clip)
    ol.orc_shoot(C.byref(cam), clip, C.byref(ray))
    if ol.orc_cast(C.byref(desc), C.byref(ray), C.byref(hit)):
        m = mats[hit.object_index]
        weight[t] = 1 + (8 if m.transparency > 0 else 0) + (4 if m.shiness > 0 else 0)
print("classes", np.unique(weight, return_counts=True))
stride_perm = (np.arange(n_tiles, dtype=np.int64) * int(n_tiles * 0.6180339887) ) % n_tiles  # not exactly coprime-corrected; fine for the experiment
centre = cnt[4:bands * 8:8, 4::8].reshape(-1).astype(np.int64)  # casts of each tile's centre pixel
def by_key(key):
    return np.array(sorted(range(n_tiles), key=lambda t: (-key[t], (t * 20023) % n_tiles)), dtype=np.uint32)
orders = {
    "centre casts, capped at 6": by_key(np.minimum(centre, 6)),
    "centre casts, capped at 12": by_key(np.minimum(centre, 12)),
    "centre casts, exact": by_key(centre),
    "centre-pixel material class": np.array(sorted(range(n_tiles), key=lambda t: (-weight[t], (t * 20023) % n_tiles)), dtype=np.uint32),
    "stride (default)": None,
    "cost descending": np.argsort(-cost.astype(np.int64), kind="stable").astype(np.uint32),
    "cost ascending": np.argsort(cost.astype(np.int64), kind="stable").astype(np.uint32),
    "image order": np.arange(n_tiles, dtype=np.uint32),
}
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
ref = None
for name, order in orders.items():
    d_order = torch.from_numpy(order.astype(np.int32)).cuda() if order is not None else None
    lib.rt_diag_set_tile_order(C.c_void_p(d_order.data_ptr()) if d_order is not None else None)
    ts = []
    for r in range(7):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rt.render_whitted(scene, cam, frame, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    same = True if ref is None else torch.equal(out.view(torch.int32), ref.view(torch.int32))
    if ref is None:
        ref = out.clone()
    print(f"{name:20s} median {statistics.median(ts):.4f} ms  min {min(ts):.4f} ms  identical {same}")
lib.rt_diag_set_tile_order(None)
