#!/bin/bash
# Round-1 evidence for the stochastic pass (run on the GPU box): tools/profile_dist.sh
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dist_final
mkdir -p $O
cd $R
{
python3 tools/bench_distributed.py --split 1 --burn 96
python3 tools/bench_distributed.py --split 1 --burn 0
python3 tools/bench_distributed.py --split 0 --burn 96
RT_AMD_RNG_LOOKAHEAD=0 python3 tools/bench_distributed.py --split 1 --burn 96
RT_AMD_RNG_LOOKAHEAD=0 python3 tools/bench_distributed.py --split 0 --burn 96
python3 tools/bench_distributed.py --split 1 --burn 96 --width 1280 --height 960 --depth 5
python3 tools/bench_distributed.py --split 0 --burn 96 --width 1280 --height 960 --depth 5
for w in 2 4 8; do python3 tools/bench_distributed.py --split 1 --burn 96 --world $w; done
} > $O/bench.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_distributed.py --split 1 --burn 96 > $O/stats.log 2>&1
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc/$tag -- python3 $R/tools/bench_distributed.py --split 1 --burn 32 --calls 1 > $O/pmc.$tag.log 2>&1
done
cd $R
for k in dist_chain dist_shade dist_unwind rng_prepare rng_scan; do echo "== $k"; python3 tools/pmc_summary.py gpurun_out/dist_final/pmc --kernel $k; done > $O/pmc.txt
