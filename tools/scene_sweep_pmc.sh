#!/bin/bash
# Scene-size sweep with counters (run on the GPU box): tools/scene_sweep_pmc.sh r02 -> gpurun_out/<tag>_scene_sweep*.{jsonl,txt}
#   timings of every size (tools/scene_sweep.py, flat and spherized tessellations), then rocprofv3 --pmc passes (separate, as
#   MI355X_MICROARCH.md prescribes) of rt::pwf_kernel at three sizes: FETCH_SIZE, L2 hits / misses, scalar-cache hits / misses
set -e
TAG=${1:-round}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 tools/scene_sweep.py --out $O/${TAG}_scene_sweep_flat.jsonl > $O/${TAG}_scene_sweep_flat.log 2>&1
python3 tools/scene_sweep.py --spherize --out $O/${TAG}_scene_sweep_spherized.jsonl > $O/${TAG}_scene_sweep_spherized.log 2>&1
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_scene_sweep_pmc.txt
for level in 0 2 4 6; do
  for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_sweep_pmc/l$level/$tag -- python3 $R/tools/scene_sweep.py --spherize --levels $level --variants 18 --frames 2 --no-parity > /dev/null 2>&1
  done
  { echo "== spherized level $level (36 * 4^$level + 28 triangles), rt::pwf_kernel, mean per dispatch; FETCH_SIZE in KB (x2 on gfx950)"; cd $R; python3 tools/pmc_summary.py $O/${TAG}_sweep_pmc/l$level --kernel pwf_kernel; cd /tmp; } >> $O/${TAG}_scene_sweep_pmc.txt
done
rm -rf $O/${TAG}_sweep_pmc
