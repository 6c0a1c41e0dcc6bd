#!/usr/bin/env python3
"""Scene-size sweep: the reference scene around a tessellated dodecahedron, through the scene FILE.

    python tools/scene_sweep.py [--levels 0 1 2 3 4 5 6] [--variants 18 2 3] [--spherize] [--out profiles/r02_scene_sweep.jsonl]
    python tools/scene_sweep.py --levels 4 --variants 18 --frames 3 --no-parity      # one configuration, e.g. under rocprofv3

Per level k: tools/make_tessellated_obj.py writes the OBJ (36 * 4^k triangles), rt_world_build_reference_scene imports it
through load_obj (main.rs:778-807) into the literal scene (28 other triangles, 4 spheres, 3 lights), the world is SAVED
as a scene file and LOADED back (rt_world_save_scene / rt_world_load_scene), and the loaded scene is
  * checked against the CPU oracle on a small frame (radiance as u32, cast counts) with every variant asked for,
  * timed on a frame whose size keeps a run in seconds (1920x1080 up to a few thousand triangles, smaller beyond).
One JSON line per (level, variant): triangles, frame, ms per frame, Mrays/s, G triangle-tests/s (casts x T / time: what a
brute-force cast costs algorithmically, main.rs:183-262), and the scene's size in HBM.
"""
import argparse
import json
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--levels", type=int, nargs="+", default=[0, 1, 2, 3, 4, 5, 6])
ap.add_argument("--variants", type=int, nargs="+", default=[18, 2, 3])
ap.add_argument("--spherize", action="store_true")
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--frames", type=int, default=0, help="timed frames (0: chosen per size)")
ap.add_argument("--no-parity", action="store_true")
ap.add_argument("--dof-epochs", type=int, default=0, help="also time this many epochs of the depth-of-field pass on the timed frame (one call, fresh streams, after a warm call of one epoch)")
ap.add_argument("--size", type=int, nargs=2, default=None, metavar=("W", "H"), help="timed frame size (default: by scene size, 1920x1080 up to 3000 triangles, 960x540 up to 40000, 480x270 above)")
ap.add_argument("--out", default=None)
ap.add_argument("--tile-order", default="default", choices=["default", "image"], help="image: the wavefront kernel takes its 8x8 tiles in image order (rt_diag_set_tile_order) instead of scattered by the golden-section stride: all workgroups then work in one region of the image at a time")
ap.add_argument("--bfs-walk", type=int, default=None, help="RT_AMD_BFS_WALK_TRIANGLES for the scenes of this run: scenes of at least this many triangles are walked breadth-first by the wavefront kernel (rt_cast_bfs.h cast_bfs); 0 never, 1 always; default: the library's")
ap.add_argument("--lib", default=None, help="variant tag: use variants/librt_amd_<tag>.so instead of the in-tree library")
a = ap.parse_args()

import numpy as np
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi

if a.lib:
    import ctypes as C

    _capi._amd = None
    _orig = _capi._load
    _capi._load = lambda name: C.CDLL(str(_capi.PKG_DIR / "variants" / f"librt_amd_{a.lib}.so")) if name == "librt_amd.so" else _orig(name)
lib = _capi.amd_lib()
if a.bfs_walk is not None:
    rt.set_option("RT_AMD_BFS_WALK_TRIANGLES", a.bfs_walk)
cam = rt.reference_camera()
lines = []
with tempfile.TemporaryDirectory() as tmp:
    for level in a.levels:
        obj = Path(tmp) / f"dodecahedron_l{level}.obj"
        cmd = [sys.executable, str(ROOT / "tools" / "make_tessellated_obj.py"), rt.DEFAULT_OBJ, str(obj), "--levels", str(level)]
        subprocess.run(cmd + (["--spherize"] if a.spherize else []), check=True, capture_output=True)
        built = rt.reference_world(str(obj))
        path = Path(tmp) / f"scene_l{level}.rtscene"
        built.save_scene(path, cam)
        world, file_cam = rt.World.load_scene(path)
        desc = world.desc()
        T = desc.n_triangles
        scene = rt.Scene(world)
        parity = None
        if not a.no_parity:
            import _oracle

            small = rt.Frame.full(64, 48, 5) if T > 40000 else rt.Frame.full(128, 96, 5)
            want, wcasts = _oracle.render_whitted(desc, file_cam, small)
            parity = True
            for v in a.variants:
                _capi.check(lib.rt_set_variant(v))
                got, casts = rt.render_whitted_numpy(scene, file_cam, small)
                same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all() and casts == wcasts
                parity = parity and bool(same)
                if not same:
                    print(f"PARITY MISS level {level} variant {v}", file=sys.stderr)
        # frame size by scene size: a brute-force cast is linear in T
        if T <= 3000:
            W, H = 1920, 1080
        elif T <= 40000:
            W, H = 960, 540
        else:
            W, H = 480, 270
        if a.size:
            W, H = a.size
        frame = rt.Frame.full(W, H, a.depth)
        d_order = None
        if a.tile_order == "image":
            import ctypes as C
            lib.rt_diag_set_tile_order.argtypes = [C.c_void_p]
            d_order = torch.arange((W * H + 63) // 64, dtype=torch.int32, device="cuda")
            lib.rt_diag_set_tile_order(C.c_void_p(d_order.data_ptr()))
        out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        for v in a.variants:
            _capi.check(lib.rt_set_variant(v))
            rt.render_whitted(scene, file_cam, frame, out=out, ray_count=cnt)  # warm-up
            torch.cuda.synchronize()
            cnt.zero_()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rt.render_whitted(scene, file_cam, frame, out=out, ray_count=cnt)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            n = a.frames or max(1, min(20, int(1.0 / max(first, 1e-4))))
            cnt.zero_()
            e0.record()
            for _ in range(n):
                rt.render_whitted(scene, file_cam, frame, out=out, ray_count=cnt)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            casts = int(cnt.item()) // n
            rec = {"level": level, "spherize": a.spherize, "triangles": T, "spheres": desc.n_spheres, "variant": v,
                   "lds_staged": bool(v & 1) and not (v & 16) and T * 128 <= 96 * 1024,
                   "scene_bytes_device": T * (128 + 64), "scene_file_bytes": path.stat().st_size,
                   "width": W, "height": H, "depth": a.depth, "frames": n, "ms_per_frame": round(ms, 4), "casts_per_frame": casts,
                   "Mrays_per_s": round(casts / ms / 1e3, 2), "Gtri_tests_per_s": round(casts * T / ms / 1e6, 2),
                   "parity_vs_oracle_small_frame": parity, "tile_order": a.tile_order, "bfs_walk_from_triangles": a.bfs_walk}
            if a.dof_epochs and v == a.variants[0]:
                acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
                dcnt = torch.zeros(1, dtype=torch.int64, device="cuda")
                warm = rt.Rng(frame)
                rt.render_distributed(scene, file_cam, frame, warm, 1, accum=acc)
                torch.cuda.synchronize()
                warm.close()
                rng = rt.Rng(frame)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rt.render_distributed(scene, file_cam, frame, rng, a.dof_epochs, accum=acc, ray_count=dcnt)
                torch.cuda.synchronize()
                dms = (time.perf_counter() - t0) * 1e3 / a.dof_epochs
                rng.close()
                rec.update({"dof_epochs": a.dof_epochs, "dof_ms_per_epoch": round(dms, 3), "dof_Msamples_per_s": round(W * H / dms / 1e3, 3),
                            "dof_Mrays_per_s": round(int(dcnt.item()) / a.dof_epochs / dms / 1e3, 2)})
            lines.append(rec)
            print(json.dumps(rec), flush=True)
        _capi.check(lib.rt_set_variant(_capi.DEFAULT_VARIANT))
        if d_order is not None:
            lib.rt_diag_set_tile_order(None)
        del scene
if a.out:
    with open(a.out, "w") as f:
        for rec in lines:
            f.write(json.dumps(rec) + "\n")
