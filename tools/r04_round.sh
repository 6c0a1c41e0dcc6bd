#!/bin/bash
# round-4 checks on the GPU box: tools/r04_round.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04}
cd $R
python3 tools/exp_sharded_finish.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_sharded_finish.txt
{ for lib in main noasks; do arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
    echo "$lib, 8-epoch calls in line: $(RT_AMD_DIST_PIPELINE=0 python3 tools/bench_distributed.py --epochs 8 --calls 3 --burn 32 $arg 2>/dev/null | tail -1 | cut -c1-330)"
    echo "$lib, 64-epoch calls: $(python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 $arg 2>/dev/null | tail -1 | cut -c1-330)"; done; } > $O/${TAG}_shade_light_asks.txt
python3 tools/scene_sweep.py --levels 5 6 --variants 18 --no-parity > $O/${TAG}_sweep_default.jsonl 2>/dev/null
python3 tools/scene_sweep.py --levels 5 6 --variants 18 --no-parity --tile-order image > $O/${TAG}_sweep_image_order.jsonl 2>/dev/null
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -3 $O/${TAG}_sharded_finish.txt; cat $O/${TAG}_shade_light_asks.txt | cut -c1-200; cat $O/${TAG}_sweep_default.jsonl $O/${TAG}_sweep_image_order.jsonl | cut -c1-400; tail -c 3000 $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
