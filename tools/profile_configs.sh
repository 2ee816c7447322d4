#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the secondary configurations (tools/bench_configs.py:
# image_warping 512^2, ARAP 102,400 vertices, shape_from_shading 2048^2, bundle adjustment ladybug-1723 shape), then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes.  Raw outputs in gpurun_out/cfg_*; tools/summarize_configs.py turns them into profiles/<round>/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg_kt -- python3 $R/tools/bench_configs.py > $R/gpurun_out/cfg_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/cfg_fetch -- python3 $R/tools/bench_configs.py > $R/gpurun_out/cfg_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/cfg_write -- python3 $R/tools/bench_configs.py > $R/gpurun_out/cfg_write.log 2>&1
# shape_from_shading at the size of the reference's data set (the resident loops; kernel trace only: every name of this run has one size)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cfg640_kt -- python3 $R/tools/bench_configs.py sfs640 > $R/gpurun_out/cfg640_kt.log 2>&1
cd $R
python3 tools/bench_configs.py sfs640 2>/dev/null | grep -v "^Initial" > gpurun_out/sfs_640x480.json
python3 tools/bench_configs.py 2>/dev/null | grep -v "^Initial" > gpurun_out/secondary_configs.json
ls gpurun_out/cfg_kt/* | head
