#!/usr/bin/env python3
"""Stress run (not part of the test suite): many random scenes, cameras, sizes and depths; the persistent wavefront
kernel against the per-pixel kernel (both on the GPU: bit-identical radiance and equal cast counts), every
configuration several times because the wavefront kernel's scheduling is not deterministic, and a sample of them
against the CPU oracle.

    python tools/stress_parity.py [--configs 200] [--repeats 3] [--oracle-every 10] [--seed 0]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

import homework_18_graphics_raytracer_amd as rt
from homework_18_graphics_raytracer_amd import _capi
import _oracle
import _scenes

ap = argparse.ArgumentParser()
ap.add_argument("--configs", type=int, default=200)
ap.add_argument("--repeats", type=int, default=3)
ap.add_argument("--oracle-every", type=int, default=10)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
lib = _capi.amd_lib()
rng = np.random.default_rng(a.seed)


def same(x, y):
    return bool((((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))).all()))


bad = 0
for k in range(a.configs):
    seed = int(rng.integers(1, 1 << 30))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        world, cam = _scenes.random_world(seed, int(rng.integers(0, 150)), int(rng.integers(0, 6))), _scenes.camera(seed)
    elif kind == 1:
        world, cam = _scenes.clustered_world(seed, int(rng.integers(1, 8))), _scenes.camera(seed)
    elif kind == 2:
        eye = (float(rng.integers(-2, 3)) * 0.5, float(rng.integers(-1, 4)) * 0.5, float(rng.integers(3, 7)) * 0.5)
        world, cam = _scenes.clustered_world(seed, int(rng.integers(1, 8)), axis_aligned=True), _scenes.axis_camera(eye)
    else:
        world, cam = rt.reference_world(), _scenes.camera(seed)
    w, h, d = int(rng.integers(8, 400)), int(rng.integers(8, 300)), int(rng.choice([0, 1, 2, 3, 5, 8, 12]))
    frame = rt.Frame.full(w, h, d)
    scene = rt.Scene(world)
    _capi.check(lib.rt_set_variant(2))
    ref, ref_casts = rt.render_whitted_numpy(scene, cam, frame)
    _capi.check(lib.rt_set_variant(_capi.DEFAULT_VARIANT))
    _capi.check(lib.rt_set_wavefront_budget(int(rng.choice([6, 6, 6, 16, 64, 2]))))
    ok = True
    for r in range(a.repeats):
        img, casts = rt.render_whitted_numpy(scene, cam, frame)
        if not same(img, ref) or casts != ref_casts:
            ok = False
            print(f"MISMATCH config {k}: kind {kind} seed {seed} {w}x{h} d{d} repeat {r}: casts {casts} vs {ref_casts}", flush=True)
    if ok and a.oracle_every and k % a.oracle_every == 0:
        want, wcasts = _oracle.render_whitted(world.desc(), cam, frame)
        if not same(ref, want) or ref_casts != wcasts:
            ok = False
            print(f"ORACLE MISMATCH config {k}: kind {kind} seed {seed} {w}x{h} d{d}", flush=True)
    bad += 0 if ok else 1
    if k % 20 == 0:
        print(f"... {k + 1} configurations, {bad} bad", flush=True)
_capi.check(lib.rt_set_wavefront_budget(6))
print(f"{a.configs} configurations x {a.repeats} repeats: {bad} bad")
sys.exit(1 if bad else 0)
