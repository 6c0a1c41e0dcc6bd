#!/bin/bash
# usage: pmc.sh tag  -> gpurun_out/pmc_<tag>.txt  (pwf kernel counters, separate passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/$tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stochastic --no-extras > $O/$tag.log 2>&1
done
cd $R && python3 tools/pmc_summary.py $O --kernel pwf_kernel > gpurun_out/pmc_$1.txt
