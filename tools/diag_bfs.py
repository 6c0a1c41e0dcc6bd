#!/usr/bin/env python3
"""Diagnostic: what the breadth-first walk (rt_cast_bfs.h cast_bfs) does per wave-cast on a tessellated scene (build: make -C csrc variant
TAG=bfsdiag EXTRA=-DRT_DIAG_BFS).

    python tools/diag_bfs.py [--levels 4 5 6] [--spherize]
"""
import argparse
import ctypes as C
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from homework_18_graphics_raytracer_amd import _capi  # noqa: E402

_capi._amd = None
_orig = _capi._load
_capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / "librt_amd_bfsdiag.so")) if name == "librt_amd.so" else _orig(name)
import homework_18_graphics_raytracer_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="+", default=[4, 5, 6])
ap.add_argument("--spherize", action="store_true")
a = ap.parse_args()
lib = _capi.amd_lib()
lib.rt_diag_read_bfs.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.rt_diag_read_bfs_ticks.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
rt.set_option("RT_AMD_BFS_WALK_TRIANGLES", 1)
cam = rt.reference_camera()
with tempfile.TemporaryDirectory() as tmp:
    for level in a.levels:
        obj = Path(tmp) / f"d{level}.obj"
        cmd = [sys.executable, str(ROOT / "tools" / "make_tessellated_obj.py"), rt.DEFAULT_OBJ, str(obj), "--levels", str(level)]
        subprocess.run(cmd + (["--spherize"] if a.spherize else []), check=True, capture_output=True)
        world = rt.reference_world(str(obj))
        scene = rt.Scene(world)
        W, H = (960, 540) if level <= 5 else (480, 270)
        frame = rt.Frame.full(W, H, 8)
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        buf = (C.c_ulonglong * 12)()
        tk = (C.c_ulonglong * 16)()
        lib.rt_diag_read_bfs(buf, 1)
        lib.rt_diag_read_bfs_ticks(tk, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rt.render_whitted(scene, cam, frame, out=out)
        e1.record()
        torch.cuda.synchronize()
        lib.rt_diag_read_bfs(buf, 1)
        lib.rt_diag_read_bfs_ticks(tk, 1)
        n = max(buf[0], 1)
        print(f"level {level} ({world.desc().n_triangles} triangles, {W}x{H}): {buf[0]} wave-casts; {buf[1]} sent to cast_asm for a full list, {buf[2]} rays with an accepted NaN distance (second pass over the jobs); "
              f"frame {e0.elapsed_time(e1):.1f} ms; wave cycles in the cast: levels {tk[0] / 1e6:.0f} M, jobs {tk[1] / 1e6:.0f} M, band jobs {tk[2] / 1e6:.0f} M, cast_finish {tk[3] / 1e6:.0f} M; "
              f"per wave-cast {buf[3] / n:.0f} (ray, node) pairs over {buf[6] / n:.1f} levels (most records in one level, any wave: {buf[5]}), {buf[4] / n:.0f} jobs with {buf[7] / n:.0f} (ray, triangle) pairs, "
              f"{buf[8] / n:.0f} band jobs with {buf[9] / n:.0f} pairs, {buf[10] / n:.1f} pairs as far as the signed areas; the longest walk of one wave-cast {buf[11] / 1e6:.2f} M cycles; waiting for memory at the top of a group / the rest: levels {tk[4] / 1e6:.0f} / {tk[5] / 1e6:.0f} M, jobs {tk[6] / 1e6:.0f} / {tk[7] / 1e6:.0f} M, band jobs {tk[8] / 1e6:.0f} / {tk[9] / 1e6:.0f} M; of a band group of 8 passes ({tk[12]} groups): issuing the next group's loads {tk[10] / max(tk[12], 1):.0f} cycles, its records and rays out of LDS {tk[11] / max(tk[12], 1):.0f}, all of it {(tk[8] + tk[9]) / max(tk[12], 1):.0f}")
