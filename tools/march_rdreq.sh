#!/bin/bash
# GPU box: the marching PCG iteration's HBM-side traffic from the REQUEST counters (TCC_EA0_RDREQ / RDREQ_32B / WRREQ / WRREQ_64B, one --pmc pass each) next to
# FETCH_SIZE / WRITE_SIZE of tools/profile.sh -- a cross-check of the x2 FETCH_SIZE correction bench.py's `traffic` figure rests on.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/mr_$ctr -- python3 $R/bench.py --no-cpu-baseline --no-small --steps 2 --warmup 1 > $R/gpurun_out/mr_$ctr.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/mr_*/")):
    ctr = d.rstrip("/").split("mr_")[1]
    fs = glob.glob(d + "*/*_counter_collection.csv")
    if not fs: continue
    f = max(fs, key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != ctr: continue
        n = row["Kernel_Name"]
        for key in ("k_iter_march_rc", "k_linear_update_n", "k_step1"):
            if key in n: agg[key].append(float(row["Counter_Value"]))
    for k, v in agg.items(): out[k][ctr] = sum(v) / len(v); out[k]["launches"] = len(v)
for k, v in out.items():
    if "TCC_EA0_RDREQ_sum" in v:
        r32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0); w64 = v.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        v["read_bytes_from_requests"] = (v["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32
        v["write_bytes_from_requests"] = w64 * 64 + (v.get("TCC_EA0_WRREQ_sum", 0.0) - w64) * 32
        v["FETCH_SIZE_bytes_x1"] = v.get("FETCH_SIZE", 0.0) * 1024; v["WRITE_SIZE_bytes"] = v.get("WRITE_SIZE", 0.0) * 1024
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/march_rdreq.json", "w"), indent=1)
PY
