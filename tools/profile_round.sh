#!/bin/bash
# Evidence for a round (run on the GPU box): tools/profile_round.sh r02  ->  gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
#   bench line, rocprofv3 kernel stats of the same command, PMC passes (separate, as MI355X_MICROARCH.md prescribes) for the
#   Whitted kernel and for the stochastic pass's kernels, the workgroup phase timeline (diagnostic build, if present)
set -e
TAG=${1:-round}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined --no-extras > $O/${TAG}_stats.log 2>&1
cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_pmc/$tag -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-stochastic --no-pipelined --no-extras > $O/${TAG}_pmc.$tag.log 2>&1
  rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_pmc_dist/$tag -- python3 $R/tools/bench_distributed.py --epochs 64 --calls 1 --warm 1 --fresh 1 > $O/${TAG}_pmc_dist.$tag.log 2>&1
done
cd $R
{ echo "# rocprofv3 --pmc (separate passes), bench.py --steps 5 --warmup 1, 1920x1080 depth 8: mean over the dispatches of rt::pwf_kernel; FETCH_SIZE / WRITE_SIZE in KB"; python3 tools/pmc_summary.py $O/${TAG}_pmc --kernel pwf_kernel; } > $O/${TAG}_pwf_pmc.txt
{ echo "# rocprofv3 --pmc (separate passes), tools/bench_distributed.py --epochs 64 --calls 1 --warm 1 --fresh 1 (configs[3] twice: bench.py's untimed call on a generator of its own, then the measured one from fresh seeds, 1920x1080 depth 8), per kernel: mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB"
  for k in dist_chain dist_shade dist_unwind rng_prepare rng_scan; do echo "== $k"; python3 tools/pmc_summary.py $O/${TAG}_pmc_dist --kernel $k; done; } > $O/${TAG}_dist_pmc.txt
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_pmc_scatter/$grp -- python3 $R/tools/bench_distributed.py --epochs 5 --calls 1 --warm 1 --fresh 1 > $O/${TAG}_pmc_scatter.$grp.log 2>&1
done
cd $R
{ echo "# rocprofv3 --pmc (separate passes), tools/bench_distributed.py --epochs 5 --calls 1 --warm 1 --fresh 1 (configs[4] twice, as above, 1920x1080 depth 8), per kernel: mean per dispatch; KB"
  for k in dist_chain dist_shade dist_unwind rng_prepare rng_scan; do echo "== $k"; python3 tools/pmc_summary.py $O/${TAG}_pmc_scatter --kernel $k; done; } > $O/${TAG}_scatter_pmc.txt
rm -rf $O/${TAG}_pmc_scatter $O/${TAG}_pmc_scatter.*.log
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_need.so ]; then python3 tools/diag_need.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_lane_tests_needed.txt; fi
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_ptime.so ]; then
  python3 tools/diag_pair_time.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_chain_step_time.txt
  # a 1/8 share: one batch of 16 epochs; its chain kernel cannot end before its dearest pixels' steps are made
  { echo "# tools/diag_pair_time.py --world 8 --epochs 16 (rank 0's share of an 8-rank job: 259 200 pixels, one pixel per lane)"; python3 tools/diag_pair_time.py --world 8 --epochs 16 2>&1 | grep -v amdgpu.ids
    echo "# the same share, timed: $(python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 --world 8 2>/dev/null | tail -1 | cut -c1-330)"; } > $O/${TAG}_share_critical_path.txt
fi
python3 tools/ab_bench.py --tags r02,main --rounds 7 --frames 10 2>&1 | grep -v amdgpu.ids > $O/${TAG}_whitted_vs_r02.txt
python3 tools/ab_bench.py --tags r02,main --rounds 7 --frames 10 --world 8 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_whitted_vs_r02.txt
# (round 2's library lacks entry points this tool binds: its number is in profiles/r02_*)
{ for lib in main nopairs; do arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi; echo "$lib $(python3 tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg 2>/dev/null | tail -1)"; done
  echo "main, 64-epoch calls (pipelined over two workspaces) $(python3 tools/bench_distributed.py --epochs 64 --calls 2 --burn 0 --warm 1 2>/dev/null | tail -1)"
  echo "main, 64-epoch calls, one workspace (RT_AMD_DIST_PIPELINE=0) $(RT_AMD_DIST_PIPELINE=0 python3 tools/bench_distributed.py --epochs 64 --calls 2 --burn 0 --warm 1 2>/dev/null | tail -1)"; } > $O/${TAG}_stochastic_vs_r02.txt
if [ -f homework-18-graphics-raytracer_amd/variants/librt_amd_pastats.so ]; then python3 tools/diag_pwf.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_pwf_phases.txt; fi
rm -rf $O/${TAG}_stats $O/${TAG}_pmc $O/${TAG}_pmc_dist $O/${TAG}_pmc.*.log $O/${TAG}_pmc_dist.*.log
