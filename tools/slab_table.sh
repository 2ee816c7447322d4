#!/bin/bash
# GPU box: per-rank slab times on ONE GPU for the 1 / 2 / 4 / 8-GPU splits of 2048^2 (tools/p2p_slab_probe.py per slab height) -> gpurun_out/slab_table.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
: > $R/gpurun_out/slab_table.txt
for h in 2048 1024 512 256; do
  PH=$h python3 $R/tools/p2p_slab_probe.py 2>/dev/null | tail -1 | sed "s/^/H=$h /" >> $R/gpurun_out/slab_table.txt
done
cat $R/gpurun_out/slab_table.txt
