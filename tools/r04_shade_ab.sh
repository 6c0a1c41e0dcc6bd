#!/bin/bash
# A/B of the shade stage on the GPU box: tools/r04_shade_ab.sh <out-tag>
#   RT_AMD_SHADE_KERNEL=0 the per-request light loop (round 3), =1 the shadow rays binned over the batch (round 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_ab4}
cd $R
OUT=$O/$TAG.txt; : > $OUT
short() { sed 's/"pass": "distributed", "split": -1, "burn": [0-9]*, "width": 1920, "height": 1080, "depth": 8, "tile_pixels": [0-9]*, //' | cut -c1-150; }
for k in 0 1; do
  echo "8-epoch calls in line, RT_AMD_SHADE_KERNEL=$k: $(RT_AMD_DIST_PIPELINE=0 RT_AMD_SHADE_KERNEL=$k python3 tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 2>/dev/null | tail -1 | short)" >> $OUT
  echo "8-epoch calls,         RT_AMD_SHADE_KERNEL=$k: $(RT_AMD_SHADE_KERNEL=$k python3 tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 2>/dev/null | tail -1 | short)" >> $OUT
  echo "64-epoch calls,        RT_AMD_SHADE_KERNEL=$k: $(RT_AMD_SHADE_KERNEL=$k python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 2>/dev/null | tail -1 | short)" >> $OUT
  echo "64 epochs, fresh,      RT_AMD_SHADE_KERNEL=$k: $(RT_AMD_SHADE_KERNEL=$k python3 tools/bench_distributed.py --epochs 64 --calls 1 --warm 1 --fresh 1 2>/dev/null | tail -1 | short)" >> $OUT
  echo "1/8 share, 64 epochs,  RT_AMD_SHADE_KERNEL=$k: $(RT_AMD_SHADE_KERNEL=$k python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 --world 8 2>/dev/null | tail -1 | short)" >> $OUT
done
cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  RT_AMD_SHADE_KERNEL=$k RT_AMD_DIST_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_k$k -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 > /dev/null 2>&1
  echo "kernel stats, 8-epoch calls in line, RT_AMD_SHADE_KERNEL=$k" >> $OUT
  grep -h "dist_\|rng_" $(find $O/${TAG}_k$k -name "*kernel_stats.csv" | head -1) | cut -c1-60,120- >> $OUT
  rm -rf $O/${TAG}_k$k
  RT_AMD_SHADE_KERNEL=$k RT_AMD_DIST_PIPELINE=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_p$k -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 > /dev/null 2>&1
  echo "counters, 8-epoch calls in line, RT_AMD_SHADE_KERNEL=$k" >> $OUT
  for kn in dist_shade dist_bin_prepare dist_bin_cast dist_bin_finish; do echo "-- $kn" >> $OUT; python3 $R/tools/pmc_summary.py $O/${TAG}_p$k --kernel $kn | grep "INSTS_VALU\|BUSY" >> $OUT; done
  rm -rf $O/${TAG}_p$k
done
cd $R
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_need.so ]; then for k in 0 1; do echo "RT_AMD_SHADE_KERNEL=$k" >> $OUT; RT_AMD_SHADE_KERNEL=$k python3 tools/diag_need.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-260 >> $OUT; done; fi
cat $OUT
