"""Turn the rocprofv3 outputs that tools/profile.sh left in gpurun_out/ into the small committed summaries under
profiles/<round>/ and profiles/traffic_latest.json (read by bench.py for roofline.traffic)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun_out/ accumulates earlier runs
ks = newest(os.path.join(g, "prof_kt", "*", "*_kernel_stats.csv"))
shutil.copy(ks, os.path.join(out, "bench_kernel_stats.csv"))
for f in ("bench_under_rocprof.json", "bench_plain.json"):
    if os.path.exists(os.path.join(g, f)):
        shutil.copy(os.path.join(g, f), os.path.join(out, f))


def short(name):
    for key, lab in (("k_iter_finish", "PCGScalars"), ("k_iter_march_rc<", "PCGIteration"), ("k_iter_march<", "PCGIteration"), ("k_iter<", "PCGIteration"), ("k_step1<true", "PCGStep1_fused"), ("k_step1<false", "applyJTJ_plain"), ("k_step2_iw", "PCGStep2"), ("k_step2<", "PCGStep2_generic"), ("k_init", "PCGInit1"),
                     ("k_linear_update_n", "PCGDeltaUpdate"), ("k_linear_update", "PCGLinearUpdate"), ("k_cost", "computeCost")):
        if key in name:
            return lab
    return None


pmc = {}
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(g, f"prof_{kind}", "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        lab = short(row["Kernel_Name"])
        if lab and row["Counter_Name"] == ctr:
            agg[lab].append(float(row["Counter_Value"]))
    for lab, v in agg.items():
        pmc.setdefault(lab, {})[ctr + "_KB_mean"] = sum(v) / len(v)
        pmc[lab]["launches_" + kind] = len(v)
# gfx950: FETCH_SIZE reports half of the bytes of coalesced streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact
for lab, d in pmc.items():
    if "FETCH_SIZE_KB_mean" in d and "WRITE_SIZE_KB_mean" in d:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * d["FETCH_SIZE_KB_mean"] + d["WRITE_SIZE_KB_mean"]) * 1024.0
json.dump(pmc, open(os.path.join(out, "pmc_fetch_write.json"), "w"), indent=1)
tl = {lab2: pmc[lab]["hbm_bytes_per_launch_corrected"] for lab, lab2 in (("PCGStep1_fused", "PCGStep1_bytes_per_launch"), ("PCGIteration", "PCGIteration_bytes_per_launch"))
      if lab in pmc and "hbm_bytes_per_launch_corrected" in pmc[lab]}
if tl:
    json.dump({**tl, "round": rnd,
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH doubled (gfx950 correction)"},
              open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(pmc, indent=1))
print(open(os.path.join(out, "bench_kernel_stats.csv")).read()[:1500])
