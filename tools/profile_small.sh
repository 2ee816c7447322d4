#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the resident PCG loop at BASELINE config 1's size (512^2) and at one rank's slab of the
# 8-GPU run (2048 x 256), each next to one launch per iteration of the marching kernel with the same rows per segment (tools/resident_probe.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PW=512 PH=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small512 -- python3 $R/tools/resident_probe.py > $R/gpurun_out/prof_small512.log 2>&1
PW=2048 PH=256 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_small_slab -- python3 $R/tools/resident_probe.py > $R/gpurun_out/prof_small_slab.log 2>&1
cd $R
tail -1 gpurun_out/prof_small512.log; tail -1 gpurun_out/prof_small_slab.log
