#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace of the bundle adjustment LM loop (tools/ba_time.py, BA_TIME_ONLY=lm: ladybug-1723 shape, 5 x 150), once in the
# single-reduction form (default) and once in the reference-shaped form (THALLO_AB=lm_fold_p=0).  tools/summarize_ba_lm.py turns the two traces into profiles/<round>/ba_lm_loops.json.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export BA_TIME_ONLY=lm
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ba_lm_new -- python3 $R/tools/ba_time.py > $R/gpurun_out/ba_lm_new.log 2>&1
export THALLO_AB=lm_fold_p=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ba_lm_old -- python3 $R/tools/ba_time.py > $R/gpurun_out/ba_lm_old.log 2>&1
