#!/bin/bash
# GPU box: counters of the GENERATED image_warping kernels (tools/generated_kernel_times.py) -- what the merged gather kernel waits for
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/gen_pmc_$tag -- python3 $R/tools/generated_kernel_times.py > $R/gpurun_out/gen_pmc_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/gen_pmc_*/")):
    fs = glob.glob(d + "*/*_counter_collection.csv")
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        for key in ("jtjgrp_0", "jtfgrp_0", "cost_0"):
            if n.startswith(key): agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items(): out[k][c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
PY
