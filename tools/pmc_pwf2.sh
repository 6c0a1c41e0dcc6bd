#!/bin/bash
# usage: tools/pmc_pwf2.sh tag  -> gpurun_out/pmc2_<tag>.txt  (pwf kernel: instruction cache, scalar cache, waits)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/$tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stochastic --no-extras > $O/$tag.log 2>&1
done
cd $R && python3 tools/pmc_summary.py $O --kernel pwf_kernel > gpurun_out/pmc2_$1.txt
