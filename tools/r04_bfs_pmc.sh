#!/bin/bash
# where the breadth-first walk's time goes (run on the GPU box): tools/r04_bfs_pmc.sh <tag> [scene_sweep arguments]
# rocprofv3 --pmc passes (separate) of rt::pwf_kernel over one scene_sweep run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04x}; shift
ARGS=${@:---levels 6 --spherize --size 1920 1080}
OUT=$O/${TAG}_bfs_pmc.txt; : > $OUT
cd /tmp && export TMPDIR=/tmp
echo "== python3 tools/scene_sweep.py $ARGS --variants 18 --frames 2 --no-parity" >> $OUT
python3 $R/tools/scene_sweep.py $ARGS --variants 18 --no-parity 2>/dev/null | tail -1 | cut -c1-400 >> $OUT
n=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_pmc/p$n -- python3 $R/tools/scene_sweep.py $ARGS --variants 18 --frames 2 --no-parity > /dev/null 2>$O/${TAG}_pmc_err.txt || { echo "pass $n ($grp) failed: $(tail -2 $O/${TAG}_pmc_err.txt | cut -c1-300)" >> $OUT; continue; }
  python3 $R/tools/pmc_summary.py $O/${TAG}_pmc/p$n --kernel pwf_kernel >> $OUT
done
rm -rf $O/${TAG}_pmc
cat $OUT
