#!/bin/bash
# the sphere loop's wave-level miss test ahead of the square root (rt_cast.h cast_finish), against -DRT_NO_SPHERE_PRETEST: tools/r04_pretest_ab.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_ab7}
cd $R
{ echo "# tools/ab_bench.py, interleaved, medians; full 1080p d8 frame, then a 1/8 share"
  python3 tools/ab_bench.py --tags nopretest,main --rounds 9 --frames 10 2>&1 | grep -v amdgpu.ids
  python3 tools/ab_bench.py --tags nopretest,main --rounds 9 --frames 10 --world 8 2>&1 | grep -v amdgpu.ids
  echo "# tools/bench_distributed.py, 64-epoch calls, alternating"
  for r in 1 2; do for lib in nopretest main; do arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
    echo "$lib: $(python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 $arg 2>/dev/null | tail -1 | cut -c120-330)"; done; done; } > $O/$TAG.txt
cat $O/$TAG.txt
