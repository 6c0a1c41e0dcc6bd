#!/bin/bash
# the scene-size sweep with the wave-uniform walk and with the breadth-first walk (tools/scene_sweep.py --bfs-walk): tools/r04_sweep.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04}
cd $R
for sph in "" "--spherize"; do
  name=flat; if [ -n "$sph" ]; then name=spherized; fi
  python3 tools/scene_sweep.py --levels 3 4 5 6 --variants 18 --bfs-walk 1 $sph > $O/${TAG}_scene_sweep_${name}_bfs.jsonl 2>$O/${TAG}_sweep.err
  python3 tools/scene_sweep.py --levels 3 4 5 6 --variants 18 --bfs-walk 0 $sph > $O/${TAG}_scene_sweep_${name}_wave_uniform.jsonl 2>>$O/${TAG}_sweep.err
  # the same scenes at the frame size of the headline, where the GPU is full (the wave-uniform walk would take seconds per frame there)
  python3 tools/scene_sweep.py --levels 4 5 6 --variants 18 --bfs-walk 1 --size 1920 1080 --no-parity $sph > $O/${TAG}_scene_sweep_${name}_bfs_1080p.jsonl 2>>$O/${TAG}_sweep.err
done
for f in $O/${TAG}_scene_sweep_*.jsonl; do echo $f; python3 -c "
import json,sys
for l in open('$f'):
    d=json.loads(l); print(d['triangles'], d['width'], d['ms_per_frame'], 'ms', d['Mrays_per_s'], 'Mrays/s', d['Gtri_tests_per_s'], 'Gtests/s parity', d['parity_vs_oracle_small_frame'])
"; done
tail -3 $O/${TAG}_sweep.err
