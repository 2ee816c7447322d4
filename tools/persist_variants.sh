#!/bin/bash
# Runs on the GPU box: TIMING of tools builds of the persistent marching loop (tools/ab/libThallo_<variant>.so) next to the product and to a launch per iteration
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/persist_variants.txt
: > $out
run() { env "$@" python3 $R/bench.py --no-small --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*'.replace('$R/tools/ab/',''), round(d['value'],1), round(d['ms_per_step'],4))" >> $out; }
run THALLO_AB=persist=0
run THALLO_AB=persist=1
for v in "$@"; do run THALLO_AB=persist=1 THALLO_LIB=$R/tools/ab/libThallo_$v.so; done
run THALLO_AB=persist=0
cat $out
