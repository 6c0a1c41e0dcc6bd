#!/bin/bash
# Where the shade kernel's VALU instructions go (run on the GPU box): tools/shade_attribution.sh <out-tag> lib [lib ...]
#   for each library variant (main = the in-tree one; stubN = make variant TAG=stubN EXTRA=-DRT_AB_SHADE_STUB=N: 1 without the cast,
#   2 without the light loop, 3 the lists only — their RESULTS are wrong on purpose): the kernels' average durations over 8-epoch
#   calls in line (one workspace, nothing beside the shade kernel) and their SQ_INSTS_VALU / SQ_BUSY_CYCLES per dispatch.
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export RT_AMD_DIST_PIPELINE=0
OUT=$O/${TAG}_shade_attribution.txt
: > $OUT
for lib in "$@"; do
  arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
  rm -rf $O/attr_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/attr_$lib/t -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg > $O/attr_$lib.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/attr_$lib/p/a -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg >> $O/attr_$lib.log 2>&1
  { echo "== $lib"; tail -1 $O/attr_$lib.log
    grep -h "dist_shade\|dist_chain\|dist_unwind" $(find $O/attr_$lib/t -name "*kernel_stats.csv" | head -1) | cut -c1-60,100-
    for k in dist_shade dist_chain; do echo "-- $k"; python3 $R/tools/pmc_summary.py $O/attr_$lib/p --kernel $k; done; } >> $OUT
  rm -rf $O/attr_$lib
done
cat $OUT
