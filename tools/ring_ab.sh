#!/bin/bash
# Runs on the GPU box: the headline bench with the ring of p planes in its variants (THALLO_DELTA_PLANES: 1 = round 4's every-other-iteration update,
# 33 = ring with the update on the loop's stream (default), 33:W = the update next to the loop on at most W workgroups).  One JSON line per variant into gpurun_out/ring_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/ring_ab.txt
: > $out
for v in ${RING_VARIANTS:-1 33 33:64 33:128 33:256 33:512 33:1024 17 9}; do
  THALLO_DELTA_PLANES=$v python3 $R/bench.py --no-small --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4))" >> $out
done
cat $out
