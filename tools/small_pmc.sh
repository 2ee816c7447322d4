#!/bin/bash
# rocprofv3 counter passes over the one-kernel PCG iteration at a SMALL size (default 512^2; sweep build): instruction mix, instruction cache,
# wave stall reasons -- for the tile kernel, the marching kernel and the streaming reference.  Run on the GPU box through gpurun.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export MB_W=${MB_W:-512} MB_H=${MB_H:-512} MB_MODE=pmcsmall MB_REPS=${MB_REPS:-20} MB_FIN=${MB_FIN:-0}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/spmc_$i -- python3 $R/tools/march_probe.py > $R/gpurun_out/spmc_$i.log 2>&1
  tail -2 $R/gpurun_out/spmc_$i.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections
out = open("gpurun_out/spmc_summary.txt", "w")
for d in sorted(glob.glob("gpurun_out/spmc_[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, c in acc.items():
            if "march" in k or "stream" in k or "k_iter" in k:
                out.write(k + "\n    " + "  ".join(f"{n}={sum(v)/len(v):.5g}" for n, v in sorted(c.items())) + f"  (n={len(next(iter(c.values())))})\n")
out.close()
print(open("gpurun_out/spmc_summary.txt").read())
PY
