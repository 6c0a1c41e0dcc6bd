#!/bin/bash
# GPU box: instruction-cache counters of the render kernels (one rocprofv3 --pmc pass each): tools/pmc_icache.sh OUT.txt
set -e
OUT=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
G="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
rocprofv3 --pmc $G --output-format csv -d $O/ic_w/a -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stochastic --no-pipelined --no-extras > $O/ic_w.log 2>&1
rocprofv3 --pmc SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/ic_w/b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stochastic --no-pipelined --no-extras > $O/ic_w.log 2>&1
rocprofv3 --pmc $G --output-format csv -d $O/ic_d/a -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 1 --burn 8 > $O/ic_d.log 2>&1
rocprofv3 --pmc SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/ic_d/b -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 1 --burn 8 > $O/ic_d.log 2>&1
cd $R
{ echo "# rocprofv3 --pmc, instruction cache (SQC_ICACHE_*: per dispatch, summed over the SQCs), mean per dispatch"
  echo "== pwf_kernel (1080p d8 frame)"; python3 tools/pmc_summary.py $O/ic_w --kernel pwf_kernel
  for k in dist_chain dist_shade; do echo "== $k (8-epoch batch)"; python3 tools/pmc_summary.py $O/ic_d --kernel $k; done; } > $O/$OUT
rm -rf $O/ic_w $O/ic_d $O/ic_w.log $O/ic_d.log
