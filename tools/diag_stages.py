#!/usr/bin/env python3
"""Diagnostic: how far the triangle tests of the intersection loop get, per wave (build: make -C csrc variant TAG=stages
EXTRA=-DRT_DIAG_STAGES).  Prints, per kernel family, the number of wave-level executions of each stage per cast."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

W, H, depth = 1920, 1080, 8
lib = C.CDLL(str(_capi.PKG_DIR / "variants/librt_amd_stages.so"))
lib.rt_scene_create.argtypes = [C.POINTER(_capi.SceneDesc), C.POINTER(C.c_void_p)]
lib.rt_render_whitted.argtypes = [C.c_void_p, C.POINTER(_capi.Camera), C.POINTER(_capi.Frame), C.c_void_p, C.c_void_p, C.c_void_p]
world = rt.reference_world(); cam = rt.reference_camera(); desc = world.desc()
frame = rt.Frame.full(W, H, depth)
h = C.c_void_p(); assert lib.rt_scene_create(C.byref(desc), C.byref(h)) == 0
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
names = ["plane/cull", "divide", "point", "area0", "area1", "area2", "accept", "triangle"]
lib.rt_diag_read_stages_pwf.argtypes = lib.rt_diag_read_stages_kernels.argtypes
for variant, reader in ((2, lib.rt_diag_read_stages_kernels), (18, lib.rt_diag_read_stages_pwf)):
    lib.rt_set_variant(variant)
    buf = (C.c_ulonglong * 9)()
    reader(buf, 1)
    assert lib.rt_render_whitted(h, C.byref(cam), C.byref(frame), C.c_void_p(out.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
    reader(buf, 1)
    casts = buf[8]
    print(f"variant {variant}: {casts} calls of the triangle loop (one per visited segment per call)")
    for k, n in enumerate(names):
        print(f"   {n:9s} {buf[k] / casts:7.2f} per call")
