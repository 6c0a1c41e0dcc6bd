#!/bin/bash
# PMC passes over the stochastic pass (run on the GPU box): tools/pmc_dist.sh <outdir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_distributed.py --split 1 --burn 32 --calls 1 > $OUT.$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
for k in dist_chain dist_shade rng_prepare; do echo "== $k"; python3 tools/pmc_summary.py gpurun_out/$1 --kernel $k; done
