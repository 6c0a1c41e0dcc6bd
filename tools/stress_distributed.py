#!/usr/bin/env python3
"""Stress run (not part of the test suite) of the stochastic pass: random scenes, cameras, tile sizes, depths, epoch
counts split into random calls, random workspace caps (batches of epochs) and both organisations of the pass — samples,
filter flags, final RNG records and cast counts against the CPU oracle, bit for bit; a sample of the configurations
also GPU against GPU with the look-ahead switched off.

    python tools/stress_distributed.py [--configs 150] [--seed 0]
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle
import _scenes

ap = argparse.ArgumentParser()
ap.add_argument("--configs", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
lib = _capi.amd_lib()
rng = np.random.default_rng(a.seed)


def same(x, y):
    return bool((((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))).all()))


def run_gpu(scene, cam, frame, calls):
    r = rt.Rng(frame)
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    outs, flags = [], []
    for n in calls:
        s = torch.empty((n, frame.rows, frame.cols, 3), dtype=torch.float32, device="cuda")
        v = torch.empty((n, frame.rows, frame.cols), dtype=torch.uint8, device="cuda")
        rt.render_distributed(scene, cam, frame, r, n, samples=s, valid=v, ray_count=cnt)
        outs.append(s.cpu().numpy())
        flags.append(v.cpu().numpy())
    torch.cuda.synchronize()
    return np.concatenate(outs), np.concatenate(flags), int(cnt.item()), r.download()


bad = 0
for k in range(a.configs):
    seed = int(rng.integers(1, 1 << 30))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        world, cam = _scenes.random_world(seed, int(rng.integers(0, 150)), int(rng.integers(0, 6))), _scenes.camera(seed)
    elif kind == 1:
        world, cam = _scenes.clustered_world(seed, int(rng.integers(1, 8))), _scenes.camera(seed)
    else:
        world, cam = rt.reference_world(), _scenes.camera(seed)
    w, h, d = int(rng.integers(4, 90)), int(rng.integers(4, 70)), int(rng.choice([0, 1, 2, 3, 5, 8, 12, 20]))
    world_n = int(rng.choice([1, 1, 2, 3]))
    frame = rt.Frame.rows_of_rank(w, h, d, int(rng.integers(0, world_n)), world_n) if h >= world_n else rt.Frame.full(w, h, d)
    total = int(rng.choice([1, 2, 5, 9, 20, 45]))
    calls, left = [], total
    while left:
        n = int(rng.integers(1, left + 1))
        calls.append(n)
        left -= n
    split = int(rng.integers(0, 2))
    cap = str(int(rng.choice([1, 2, 8, 64, 16384])))
    lib.rt_set_distributed_split(split)
    rt.set_option("RT_AMD_DIST_WS_MB", cap)
    scene = rt.Scene(world)
    s, v, casts, st = run_gpu(scene, cam, frame, calls)
    want_st = _oracle.rng_init(frame)
    ws, wv, wcasts = _oracle.render_distributed(world.desc(), cam, frame, want_st, total)
    ok = same(s, ws) and np.array_equal(v, wv) and casts == wcasts and np.array_equal(st, want_st)
    if ok and k % 5 == 0:
        rt.set_option("RT_AMD_RNG_LOOKAHEAD", "0")
        s2, v2, casts2, st2 = run_gpu(scene, cam, frame, [total])
        rt.set_option("RT_AMD_RNG_LOOKAHEAD", None)
        ok = same(s2, s) and np.array_equal(v2, v) and casts2 == casts and np.array_equal(st2, st)
    if not ok:
        bad += 1
        print(f"MISMATCH config {k}: kind {kind} seed {seed} {w}x{h} d{d} rows {frame.rows} calls {calls} split {split} cap {cap} MB: casts {casts} vs {wcasts}", flush=True)
    if k % 10 == 0:
        print(f"... {k + 1} configurations, {bad} bad", flush=True)
lib.rt_set_distributed_split(-1)
rt.set_option("RT_AMD_DIST_WS_MB", None)
print(f"{a.configs} configurations: {bad} bad")
sys.exit(1 if bad else 0)
