#!/bin/bash
# the breadth-first walk's group sizes (passes whose loads are in flight together): tools/r04_bfs_groups.sh
# variants: make -C csrc variant TAG=lg2 EXTRA=-DRT_BFS_LEVEL_GROUP=2u (lg8, bg4, bg16: RT_BFS_BAND_GROUP, pg2, pg8: RT_BFS_PAIR_GROUP); main = 4 / 4 / 8
R=$GRAFT_REPO_ROOT; cd $R
for lib in main lg2 lg8 bg4 bg16 pg2 pg8; do
  arg=""; if [ "$lib" != main ]; then arg="--lib $lib"; fi
  for sph in "" "--spherize"; do for sz in "480 270" "1920 1080"; do
    python3 tools/scene_sweep.py --levels 6 --variants 18 --no-parity --size $sz $sph $arg 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$lib', 'spherized' if d['spherize'] else 'flat', d['triangles'], d['width'], d['ms_per_frame'], 'ms')
" || exit 1
  done; done
done
