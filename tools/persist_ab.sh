#!/bin/bash
# [RESEARCH build: make -C thallo_amd/csrc VARIANT=research; export THALLO_LIB=$R/tools/ab/libThallo_research.so -- the persistent loop is not in the product library since round 6]
# Runs on the GPU box: the headline bench with the persistent marching loop against one launch per iteration (THALLO_AB=persist=0), both visibility forms
# (THALLO_PERSIST_ACQ=1: one agent-scope acquire per wave and iteration instead of L1-bypassing loads).  One line per variant into gpurun_out/persist_ab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/persist_ab.txt
: > $out
run() { env "$@" python3 $R/bench.py --no-small --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,2))" >> $out; }
for rep in 1 2; do
run THALLO_AB=persist=0
run THALLO_AB=persist=1
run THALLO_AB=persist=1 THALLO_PERSIST_RES=0
run THALLO_AB=persist=1 THALLO_PERSIST_OCC=2
run THALLO_AB=persist=1 THALLO_PERSIST_OCC=2 THALLO_PERSIST_RES=0
done
cat $out
