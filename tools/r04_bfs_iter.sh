#!/bin/bash
# one iteration on the breadth-first walk (run on the GPU box): parity of the scene-size suite, the per-wave-cast diagnostics, timings at two frame sizes
# tools/r04_bfs_iter.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04x}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_scene_sizes.py tests/test_gpu_whitted_parity.py -x -q > $O/${TAG}_tests.log 2>&1; tail -2 $O/${TAG}_tests.log | cut -c1-300
timeout -k 10 200 python3 tools/diag_bfs.py --levels 5 6 > $O/${TAG}_bfs_diag.txt 2>&1 && timeout -k 10 200 python3 tools/diag_bfs.py --levels 5 6 --spherize >> $O/${TAG}_bfs_diag.txt 2>&1
grep -v amdgpu.ids $O/${TAG}_bfs_diag.txt | cut -c1-900
for sz in "480 270" "1920 1080"; do for sph in "" "--spherize"; do
  timeout -k 10 250 python3 tools/scene_sweep.py --levels 5 6 --variants 18 --no-parity --size $sz $sph 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$sph', d['triangles'], d['width'], d['ms_per_frame'], 'ms', d['Mrays_per_s'], 'Mrays/s', d['Gtri_tests_per_s'], 'Gtests/s')
" || exit 1
done; done
