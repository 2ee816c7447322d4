#!/bin/bash
# rocprofv3 counter passes over the shape_from_shading 2048^2 GN configuration (tools/sfs_pmc.py) -- what bounds the applyJTJ kernels (the LDS-tiled k_fused of
# THALLO_AB=sfs_march=0, the marching k_march variants by default).
# Run on the GPU box through gpurun; summary in gpurun_out/sfs_pmc_summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sfspmc_$i -- python3 $R/tools/sfs_pmc.py > $R/gpurun_out/sfspmc_$i.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/sfs_pmc_summary.txt", "w")
for d in sorted(glob.glob("gpurun_out/sfspmc_[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:110]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, c in acc.items():
            if "k_fused" in k or "k_pcg_update" in k or "k_march" in k or "k_precompute_march" in k:
                out.write(k + "\n   " + "  ".join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items())) + f"  (n={len(next(iter(c.values())))})\n")
out.close()
print(open("gpurun_out/sfs_pmc_summary.txt").read())
PY
