#!/bin/bash
# two against three workspaces for the batches of a long depth-of-field call (RT_AMD_DIST_PIPELINE=2 / 3): tools/r04_ws3_ab.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r04_ab8}
cd $R
{ echo "# tools/bench_distributed.py, alternating runs; RT_AMD_DIST_PIPELINE = workspaces used in turn (2: 2 x 8 epochs in 32 GiB, 3: 3 x 8 epochs in 48 GiB)"
  for r in 1 2; do for n in 2 3; do
    echo "$n workspaces, 64-epoch calls x 2 after a warm call: $(RT_AMD_DIST_PIPELINE=$n python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 2>/dev/null | tail -1 | cut -c120-330)"
    echo "$n workspaces, 64 epochs from fresh streams (bench.py's job): $(RT_AMD_DIST_PIPELINE=$n python3 tools/bench_distributed.py --epochs 64 --calls 1 --warm 1 --fresh 1 2>/dev/null | tail -1 | cut -c120-330)"
  done; done
  for n in 2 3; do echo "$n workspaces, a 1/8 share, 64-epoch calls x 2: $(RT_AMD_DIST_PIPELINE=$n python3 tools/bench_distributed.py --epochs 64 --calls 2 --warm 1 --world 8 2>/dev/null | tail -1 | cut -c120-330)"; done; } > $O/$TAG.txt
cat $O/$TAG.txt
