#!/bin/bash
# pwf_kernel A/B of round 4: tools/r04_pwf_ab.sh <tag>
#   outline = -DPA_OUTLINE (the queue helpers as functions), lq4 = -DPA_LDS_PAGES=4 (four LDS pages per light queue instead of two:
#   what a shared pool of pages could save in HBM traffic at best; 57 KB of LDS, two workgroups per CU instead of three)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_ab6}
cd $R
OUT=$O/$TAG.txt
{ echo "# tools/ab_bench.py, interleaved, medians; full 1080p d8 frame, then a 1/8 share"
  python3 tools/ab_bench.py --tags main,outline,lq4 --rounds 7 --frames 10 2>&1 | grep -v amdgpu.ids
  python3 tools/ab_bench.py --tags main,outline,lq4 --rounds 7 --frames 10 --world 8 2>&1 | grep -v amdgpu.ids; } > $OUT
cd /tmp && export TMPDIR=/tmp
for lib in main outline lq4; do
  arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
    t=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $grp --output-format csv -d $O/${TAG}_p/$lib/full/$t -- python3 $R/tools/run_frames.py --frames 6 $arg > /dev/null 2>&1
  done
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_p/$lib/share/i -- python3 $R/tools/run_frames.py --frames 6 --world 8 $arg > /dev/null 2>&1
  { echo "== $lib, full frame (mean per launch; FETCH_SIZE / WRITE_SIZE in KB, FETCH x2 on gfx950)"; python3 $R/tools/pmc_summary.py $O/${TAG}_p/$lib/full --kernel pwf_kernel
    echo "== $lib, 1/8 share"; python3 $R/tools/pmc_summary.py $O/${TAG}_p/$lib/share --kernel pwf_kernel; } >> $OUT
done
rm -rf $O/${TAG}_p
cat $OUT
