#!/bin/bash
# A/B of the shade kernel organisations on the GPU box: tools/r04_shade_ab.sh <out-tag>
#   RT_AMD_SHADE_KERNEL=0 the per-request light loop (round 3), =1 the lights as phases (round 4), with the requests per round swept;
#   --lib noclass: the lights as phases without the ordering of a light's list by cluster class
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_ab1}
cd $R
OUT=$O/$TAG.txt; : > $OUT
short() { sed 's/"pass": "distributed", "split": -1, "burn": [0-9]*, "width": 1920, "height": 1080, "depth": 8, "tile_pixels": 2073600, //' | cut -c1-150; }
run() { echo "$1: $(env $2 python3 tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $3 2>/dev/null | tail -1 | short)" >> $OUT; }
run "in line, per-request loop          " "RT_AMD_DIST_PIPELINE=0 RT_AMD_SHADE_KERNEL=0"
for cap in 384 512 768; do run "in line, lights as phases, cap $cap" "RT_AMD_DIST_PIPELINE=0 RT_AMD_SHADE_KERNEL=1 RT_AMD_SHADE_CAP=$cap"; done
for cap in 512 768; do run "in line, lights as phases, no classes, cap $cap" "RT_AMD_DIST_PIPELINE=0 RT_AMD_SHADE_KERNEL=1 RT_AMD_SHADE_CAP=$cap" "--lib noclass"; done
run2() { echo "$1: $(env $2 python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 2>/dev/null | tail -1 | short)" >> $OUT; }
run2 "64-epoch calls, per-request loop    " "RT_AMD_SHADE_KERNEL=0"
for cap in 512 768; do run2 "64-epoch calls, lights as phases, cap $cap" "RT_AMD_SHADE_KERNEL=1 RT_AMD_SHADE_CAP=$cap"; done
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  RT_AMD_SHADE_KERNEL=$k RT_AMD_DIST_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_k$k -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 > /dev/null 2>&1
  echo "kernel stats, in line, RT_AMD_SHADE_KERNEL=$k" >> $OUT
  grep -h "dist_shade\|dist_chain\|dist_unwind" $(find $O/${TAG}_k$k -name "*kernel_stats.csv" | head -1) >> $OUT
  rm -rf $O/${TAG}_k$k
  RT_AMD_SHADE_KERNEL=$k RT_AMD_DIST_PIPELINE=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_p$k -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 > /dev/null 2>&1
  echo "counters, in line, RT_AMD_SHADE_KERNEL=$k" >> $OUT
  python3 $R/tools/pmc_summary.py $O/${TAG}_p$k --kernel dist_shade >> $OUT
  rm -rf $O/${TAG}_p$k
done
cd $R
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_ptime.so ]; then python3 tools/diag_pair_time.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT; fi
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_need.so ]; then for k in 0 1; do echo "RT_AMD_SHADE_KERNEL=$k" >> $OUT; RT_AMD_SHADE_KERNEL=$k python3 tools/diag_need.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-260 >> $OUT; done; fi
cat $OUT
