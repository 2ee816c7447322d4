#!/bin/bash
# GPU box: where the bundle-adjustment camera kernel's traffic goes.  Separate --pmc passes (TCC slots), kernel trace only:
#   FETCH_SIZE / WRITE_SIZE (HBM-side bytes; FETCH x 2 on gfx950), TCP_TCC_READ_REQ_sum (L1 -> L2 read requests), TCC_MISS_sum / TCC_HIT_sum, TCC_EA0_RDREQ_sum / WRREQ
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in ${BA_PMC_COUNTERS:-FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_WRITE_REQ_sum}; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/ba_pmc_$ctr -- python3 $R/tools/lm_probe.py ba > $R/gpurun_out/ba_pmc_$ctr.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/ba_pmc_*/")):
    ctr = d.rstrip("/").split("ba_pmc_")[1]
    fs = glob.glob(d + "*/*_counter_collection.csv")
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != ctr: continue
        n = row["Kernel_Name"]
        for key in ("k_cam2", "k_pt2", "k_pcg_step2", "k_step2", "k_pupdate", "k_precompute", "k_pack_point"):
            if key in n: agg[key].append(float(row["Counter_Value"]))
    for k, v in agg.items(): out[k][ctr] = sum(v) / len(v); out[k]["launches"] = len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/ba_pmc.json", "w"), indent=1)
PY
