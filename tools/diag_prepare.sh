#!/bin/bash
# GPU box: the look-ahead kernel alone (the first call after seeding prepares every pixel, in line) for the variant tags given
# — its max duration in the kernel trace is the stand-alone time for 2 073 600 records.  tools/diag_prepare.sh OUT TAG...
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
: > $O/$OUT
for lib in "$@"; do
  arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prep_$lib -- python3 $R/tools/bench_distributed.py --epochs 1 --calls 1 --burn 0 $arg > $O/prep_$lib.log 2>&1
  echo "== $lib" >> $O/$OUT
  grep "rng_prepare" $(find $O/prep_$lib -name "*kernel_stats.csv" | head -1) >> $O/$OUT
  rm -rf $O/prep_$lib $O/prep_$lib.log
done
