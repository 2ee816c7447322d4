"""rocprofv3 --stats tables of tools/profile_ba_lm.sh (gpurun_out/ba_lm_new, ba_lm_old) -> profiles/<round>/ba_lm_loops.json: per kernel of the bundle adjustment LM loop the
launch count and average duration, in the single-reduction form and in the reference-shaped form, and the per-iteration times tools/ba_time.py printed under the profiler."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
out = {}
for tag in ("new", "old"):
    f = max(glob.glob(os.path.join(ROOT, "gpurun_out", "ba_lm_" + tag, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    rows = []
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        name = re.sub(r"\(.*", "", name)[:70]
        rows.append({"kernel": name, "launches": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 2)})
    rows.sort(key=lambda x: -x["total_ms"])
    log = open(os.path.join(ROOT, "gpurun_out", "ba_lm_%s.log" % tag)).read()
    m = re.findall(r'"us_per_pcg_iter": ([0-9.]+)', log)
    out["single_reduction_form" if tag == "new" else "reference_shaped_form (THALLO_AB=lm_fold_p=0)"] = {"us_per_pcg_iter_under_the_profiler": [float(x) for x in m], "kernels": rows[:10]}
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
json.dump({"source": "tools/profile_ba_lm.sh: rocprofv3 --kernel-trace --stats of tools/ba_time.py (BA_TIME_ONLY=lm), ladybug-1723 shape, LM 5 x 150 twice", "loops": out},
          open(os.path.join(ROOT, "profiles", rnd, "ba_lm_loops.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
