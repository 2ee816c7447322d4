#!/bin/bash
# rocprofv3 counter passes over tools/march_pmc.py (run on the GPU box through gpurun); summaries into gpurun_out/pmc_*.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_REQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$i -- python3 $R/tools/march_pmc.py > $R/gpurun_out/pmc_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
out = open("gpurun_out/pmc_summary.txt", "w")
for d in sorted(glob.glob("gpurun_out/pmc_[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, c in acc.items():
            if "march" in k or "stream" in k or "k_iter" in k:
                out.write(k + "  " + "  ".join(f"{n}={sum(v)/len(v):.4g}(n={len(v)})" for n, v in sorted(c.items())) + "\n")
out.close()
print(open("gpurun_out/pmc_summary.txt").read())
PY
