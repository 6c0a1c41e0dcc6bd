#!/usr/bin/env python3
"""The look-ahead pass alone (rng_scan_kernel + rng_prepare_kernel over freshly seeded records), for diagnostic variants built
with -DRT_DIAG_PREPARE:   python3 tools/diag_prepare.py TAG [TAG...]     (variants/librt_amd_TAG.so)"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
N, REPS, CUS = 1 << 20, 5, 256
for tag in sys.argv[1:]:
    lib = C.CDLL(str(ROOT / "homework-18-graphics-raytracer_amd" / "variants" / f"librt_amd_{tag}.so"))
    out = (C.c_float * 2)()
    rc = lib.rt_diag_prepare_time(C.c_uint32(N), C.c_uint32(REPS), C.c_uint32(CUS), out)
    if rc != 0:
        print(f"{tag}: failed ({rc})")
        continue
    prep = out[0] - out[1]
    print(f"{tag:12s} {N} records: scan + prepare {out[0]:.3f} ms, scan alone {out[1]:.3f} ms -> prepare {prep:.3f} ms = {N / prep / 1e3:.0f} K records/ms, "
          f"{N * 3 * 1024 / prep / 1e6:.0f} GB/s of record traffic", flush=True)
