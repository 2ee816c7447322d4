#!/bin/bash
# Runs on the GPU box: TIMING ONLY -- the persistent marching loop built with other cache policies for its r / p loads and stores (tools/ab/libThallo_ld*st*.so:
# make VARIANT=ld0st0 EXTRA="-DPST_LD_AUX=0 -DPST_ST_AUX=0" ...; anything but sc1 / sc1 is not coherent between workgroups, the results are not checked)
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/persist_policy.txt
: > $out
run() { env "$@" python3 $R/bench.py --no-small --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,2))" >> $out; }
run THALLO_PERSIST=1
for v in ld0st0 ld16st0 ld0st16 ld0st2; do run THALLO_PERSIST=1 THALLO_LIB=$R/tools/ab/libThallo_$v.so; done
run THALLO_PERSIST=0
cat $out
