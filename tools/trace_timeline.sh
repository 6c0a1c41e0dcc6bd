#!/bin/bash
# GPU box: kernel timeline (start/end per launch) of the stochastic pass: tools/trace_timeline.sh OUT.csv [bench_distributed args]
set -e
OUT=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tl_tmp -- python3 $R/tools/bench_distributed.py "$@" > $O/tl_tmp.log 2>&1
cp $(find $O/tl_tmp -name "*kernel_trace.csv" | head -1) $O/$OUT
rm -rf $O/tl_tmp
