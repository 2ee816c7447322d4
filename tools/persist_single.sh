#!/bin/bash
# Runs on the GPU box: the persistent marching kernel as ONE iteration per launch (THALLO_DELTA_PLANES=2) under rocprofv3, per library variant: its launch duration
# next to the launch-per-iteration kernel's.  TIMING ONLY for the non-product cache policies.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/persist_single.txt
for v in product "$@"; do
  lib=$R/thallo_amd/libThallo.so; [ $v != product ] && lib=$R/tools/ab/libThallo_$v.so
  rm -rf $R/gpurun_out/prof_single
  THALLO_LIB=$lib THALLO_PERSIST=1 THALLO_DELTA_PLANES=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_single -- python3 $R/bench.py --no-cpu-baseline --no-small --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_single -name "*kernel_stats.csv" | tail -1)
  python3 - "$f" $v >> $R/gpurun_out/persist_single.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_march_persist" in r["Name"] or "k_iter_march" in r["Name"]: print(sys.argv[2], r["Name"][28:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
done
rm -rf $R/gpurun_out/prof_single
cat $R/gpurun_out/persist_single.txt
