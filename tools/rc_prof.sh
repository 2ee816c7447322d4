#!/bin/bash
# GPU box: per-kernel-variant durations of the marching iteration without the A p plane (rocprofv3 --kernel-trace --stats of tools/rc_probe.py), one run per rows-per-segment value
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rows in ${RC_ROWS_LIST:-0 18}; do
  RC_ROWS=$rows timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rcprof_$rows -- python3 $R/tools/rc_probe.py 2048 6 > $R/gpurun_out/rcprof_$rows.log 2>&1
  f=$(ls -t $R/gpurun_out/rcprof_$rows/*/*_kernel_stats.csv | head -1)
  echo "rows=$rows"; grep -E "k_iter_march" $f | sed -E 's/\(thallo::MarchGeo[^"]*"/"/' | cut -d, -f1-4
done
