#!/usr/bin/env python3
"""Render N Whitted frames with a given variant (profiling driver: rocprofv3 --kernel-trace --stats -- python3 tools/run_frames.py ...)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import homework_18_graphics_raytracer_amd as rt  # noqa: E402
from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", type=int, default=18)
ap.add_argument("--world", type=int, default=1, help="render only rank 0's interleaved row band of this many ranks")
ap.add_argument("--lib", default=None, help="variant tag: use variants/librt_amd_<tag>.so instead of the in-tree library")
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--depth", type=int, default=8)
args = ap.parse_args()
if args.lib:
    import ctypes as C
    _capi._amd = None
    _orig = _capi._load
    _capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / f"librt_amd_{args.lib}.so")) if name == "librt_amd.so" else _orig(name)
_capi.check(_capi.amd_lib().rt_set_variant(args.variant))
world = rt.reference_world()
scene = rt.Scene(world)
cam = rt.reference_camera()
frame = rt.Frame.full(args.width, args.height, args.depth) if args.world == 1 else rt.Frame.rows_of_rank(args.width, args.height, args.depth, 0, args.world)
out = torch.empty((frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(args.frames):
    rt.render_whitted(scene, cam, frame, out=out, ray_count=cnt)
torch.cuda.synchronize()
print("casts/frame", int(cnt.item()) // args.frames)
