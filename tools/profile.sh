#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the headline bench, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md: TCC slots do not fit both).  Summaries land in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --no-cpu-baseline --no-small > $R/gpurun_out/prof_kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --no-cpu-baseline --no-small --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --no-cpu-baseline --no-small --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
grep "^{" gpurun_out/prof_kt.log | tail -1 > gpurun_out/bench_under_rocprof.json
python3 bench.py > gpurun_out/bench_plain.log 2>&1
grep "^{" gpurun_out/bench_plain.log | tail -1 > gpurun_out/bench_plain.json
ls -R gpurun_out | head -40
