#!/usr/bin/env python3
"""profiles/traffic*.json from the --pmc summaries of a round (tools/profile_round.sh writes them):

    python3 tools/make_traffic.py r03        # reads profiles/r03_pwf_pmc.txt, r03_dist_pmc.txt, r03_scatter_pmc.txt

bench.py reads these records for `roofline.traffic` / `roofline.hbm` (the counters are collected in separate rocprofv3 passes,
never inside the timed run).  HBM bytes = FETCH_SIZE x 2 (gfx950 tallies a 128-byte read request as 64) + WRITE_SIZE, both in
KB in the summaries (MI355X_MICROARCH.md).  The depth-of-field summaries hold one line per counter and kernel with the number of
dispatches, their mean and their minimum.  Round 4 on: the passes are collected with `--warm 1 --fresh 1 --calls 1`, i.e.
the profiled process makes bench.py's two calls — the untimed one on a generator of its own, then the measured one from fresh
streams — so HALF of every sum belongs to the measured call (the two calls are the same job; `halve=True`).  Records made from
older summaries (a 1-epoch warm-up call whose chain / shade / unwind dispatches — the minimum of each — are taken out) keep that rule."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from homework_18_graphics_raytracer_amd import _capi  # noqa: E402  (the hash of the library's sources; no GPU needed)

P = ROOT / "profiles"
SOURCES = _capi.sources_sha256()  # run this on the tree the counters were collected with
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
LINE = re.compile(r"^(\w+)\s+n=\s*(\d+)\s+mean=(\S+)\s+min=(\S+)\s+max=(\S+)")


def sections(path):
    out, cur = {}, "_"
    for line in path.read_text().splitlines():
        if line.startswith("== "):
            cur = line[3:].strip()
            continue
        m = LINE.match(line)
        if m:
            out.setdefault(cur, {})[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return out


def whitted():
    c = sections(P / f"{tag}_pwf_pmc.txt")["_"]
    path = P / "traffic.json"
    rec = json.loads(path.read_text()) if path.exists() else {}
    mean = lambda k: c[k][1]
    rec.update({
        "width": 1920, "height": 1080, "depth": 8, "variant": 18, "sources_sha256": SOURCES,
        "source": f"profiles/{tag}_pwf_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / TCC_*, separate passes, rt::pwf_kernel, mean per launch; tools/make_traffic.py)",
        "fetch_size_kb": mean("FETCH_SIZE"), "write_size_kb": mean("WRITE_SIZE"),
        "hbm_bytes_per_launch": int(mean("FETCH_SIZE") * 2 * 1024 + mean("WRITE_SIZE") * 1024),
        "correction": "FETCH_SIZE x2 on gfx950 (reads tallied at 64 B per 128-B request); WRITE_SIZE taken as is",
        "sq_insts_valu": mean("SQ_INSTS_VALU"), "sq_insts_salu": mean("SQ_INSTS_SALU"),
        "l2_hit_rate": round(mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum")), 3),
        "l2_requests": mean("TCC_HIT_sum") + mean("TCC_MISS_sum"),
    })
    path.write_text(json.dumps(rec, indent=1) + "\n")
    print("traffic.json:", rec["hbm_bytes_per_launch"] / 1e9, "GB per frame, L2 hit", rec["l2_hit_rate"])


def dof(pmc_name, out_name, epochs):
    sec = sections(P / pmc_name)
    halve = "--warm 1 --fresh 1" in (P / pmc_name).read_text().splitlines()[0]
    per, fetch, write, valu, hit, miss = {}, 0.0, 0.0, 0.0, 0.0, 0.0
    for k, c in sec.items():
        warm = k in ("dist_chain", "dist_shade", "dist_unwind")  # one dispatch of each belongs to the untimed 1-epoch call: the minimum

        def total(name):
            if name not in c:
                return 0.0
            n, mean, mn = c[name]
            if halve:
                return n * mean / 2.0
            return n * mean - (mn if warm and n > 1 else 0.0)
        per[k] = {"fetch": round(total("FETCH_SIZE")), "write": round(total("WRITE_SIZE"))}
        fetch += total("FETCH_SIZE")
        write += total("WRITE_SIZE")
        valu += total("SQ_INSTS_VALU")
        hit += total("TCC_HIT_sum")
        miss += total("TCC_MISS_sum")
    samples = 1920 * 1080 * epochs
    rec = {
        "width": 1920, "height": 1080, "depth": 8, "epochs": epochs, "sources_sha256": SOURCES,
        "source": f"profiles/{pmc_name} (rocprofv3 --pmc, separate passes, tools/bench_distributed.py --epochs {epochs} --calls 1: sums over the dispatches of dist_chain / "
                  "dist_shade / dist_unwind / rng_prepare / rng_scan of the measured call — the untimed 1-epoch call's chain / shade / unwind dispatches taken out; tools/make_traffic.py)",
        "fetch_size_kb": fetch, "write_size_kb": write,
        "hbm_bytes_per_launch": int(fetch * 2 * 1024 + write * 1024),
        "correction": f"FETCH_SIZE x2 on gfx950; WRITE_SIZE as is; 'launch' = the whole {epochs}-epoch call",
        "l2_hit_rate": round(hit / (hit + miss), 3) if hit + miss > 0 else None,
        "per_kernel_kb": per,
    }
    if valu > 0:
        rec["sq_insts_valu"] = valu
        rec["sq_insts_valu_per_sample"] = round(valu / samples, 2)
    rec["hbm_bytes_per_sample"] = round(rec["hbm_bytes_per_launch"] / samples, 1)
    (P / out_name).write_text(json.dumps(rec, indent=1) + "\n")
    print(out_name + ":", rec["hbm_bytes_per_launch"] / 1e9, "GB,", rec["hbm_bytes_per_sample"], "B per sample,", rec.get("sq_insts_valu_per_sample"), "VALU wave-instructions per sample")


whitted()
dof(f"{tag}_dist_pmc.txt", "traffic_stochastic.json", 64)
dof(f"{tag}_scatter_pmc.txt", "traffic_scatter.json", 5)
