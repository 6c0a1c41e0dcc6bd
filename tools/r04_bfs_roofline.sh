#!/bin/bash
# The breadth-first walk against its roofs (run on the GPU box): tools/r04_bfs_roofline.sh <tag>
#   level 6 (147 484 triangles), flat and spherized, at the sweep's frame size (480 x 270) and at the headline's (1920 x 1080), the library's default walk (breadth-first from 8 192 triangles): the frame time, and
#   rocprofv3 --pmc passes (separate) of rt::pwf_kernel — FETCH_SIZE, WRITE_SIZE (KB; FETCH x2 on gfx950), SQ_INSTS_VALU, SQ_BUSY_CYCLES
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04}
OUT=$O/${TAG}_bfs_roofline.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
for sph in "" "--spherize"; do for level in 6; do for size in "480 270" "1920 1080"; do
  name="flat"; if [ -n "$sph" ]; then name="spherized"; fi
  line=$(python3 $R/tools/scene_sweep.py $sph --levels $level --variants 18 --no-parity --size $size 2>/dev/null | tail -1)
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_bp/$name$level/$t -- python3 $R/tools/scene_sweep.py $sph --levels $level --variants 18 --frames 2 --no-parity --size $size > /dev/null 2>&1
  done
  { echo "== $name level $level: $line"; python3 $R/tools/pmc_summary.py $O/${TAG}_bp/$name$level --kernel pwf_kernel; } >> $OUT
  rm -rf $O/${TAG}_bp
done; done; done
rm -rf $O/${TAG}_bp
python3 - "$OUT" <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
out = []
for block in txt.split("== ")[1:]:
    head, *rows = block.strip().split("\n")
    rec = json.loads(head[head.index("{"):])
    c = {}
    for r in rows:
        m = re.match(r"(\w+)\s+n=\s*\d+\s+mean=(\S+)", r)
        if m: c[m.group(1)] = float(m.group(2))
    ms = rec["ms_per_frame"]
    hbm = (c.get("FETCH_SIZE", 0) * 2 + c.get("WRITE_SIZE", 0)) * 1024
    valu = c.get("SQ_INSTS_VALU", 0)
    hits, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    parked = c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 0), 1)
    out.append(f"{head.split(':')[0]}: {rec['triangles']} triangles, {rec['width']}x{rec['height']}: {ms} ms per frame, {rec['Gtri_tests_per_s']/1e3:.2f} T algorithmic triangle-tests/s; "
               f"HBM {hbm/1e9:.2f} GB per frame = {hbm/ms/1e6:.0f} GB/s = {hbm/ms/1e6/8000:.3f} of 8 TB/s (L2 hit {hits/max(hits+miss,1):.2f}); "
               f"VALU {valu:.3g} wave-instructions = {valu*4/1024/2.4e9/(ms*1e-3):.2f} of the issue slots of 1 024 SIMDs at 2.4 GHz; waves parked at a wait {parked:.2f} of their time")
open(sys.argv[1], "a").write("\n# summary (roofline.bound: whichever fraction is the larger)\n" + "\n".join(out) + "\n")
print("\n".join(out))
PY
