#!/bin/bash
# round-3 probe (GPU box): need counters of the diagnostic build, then per-kernel times (and, with PMC=1, instruction counters)
# of the stochastic pass for the in-tree library and for the variant tags given:  [PMC=1] tools/r03_probe.sh TAG [tags...]
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_need.so ]; then python3 tools/diag_need.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_need.txt; fi
cd /tmp && export TMPDIR=/tmp
for lib in main "$@"; do
  arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
  python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg 2>/dev/null | tail -1 > $O/${TAG}_bench_$lib.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_$lib -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg > $O/${TAG}_stats_$lib.log 2>&1
  cp $(find $O/${TAG}_stats_$lib -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats_$lib.csv
  rm -rf $O/${TAG}_stats_$lib
  if [ -n "$PMC" ]; then
    for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
      tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
      rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_pmc_$lib/$tag -- python3 $R/tools/bench_distributed.py --epochs 8 --calls 1 --burn 32 $arg > $O/${TAG}_pmc_$lib.$tag.log 2>&1
    done
    { for k in dist_chain dist_shade; do echo "== $k"; python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_$lib --kernel $k; done; } > $O/${TAG}_pmc_$lib.txt
    rm -rf $O/${TAG}_pmc_$lib $O/${TAG}_pmc_$lib.*.log
  fi
done
