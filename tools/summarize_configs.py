"""rocprofv3 outputs of tools/profile_configs.sh (gpurun_out/cfg_*) -> profiles/<round>/secondary_kernel_stats.csv (the --stats table) and
secondary_kernels.json: per hot kernel the average duration, the corrected HBM-side traffic (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md
HBM section) per launch, the rate that gives and its fraction of the 8 TB/s peak, next to the algorithmic bytes DESIGN.md section 4 defines."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
shutil.copy(newest(os.path.join(g, "cfg_kt", "*", "*_kernel_stats.csv")), os.path.join(out, "secondary_kernel_stats.csv"))
if glob.glob(os.path.join(g, "cfg640_kt", "*", "*_kernel_stats.csv")):
    shutil.copy(newest(os.path.join(g, "cfg640_kt", "*", "*_kernel_stats.csv")), os.path.join(out, "sfs_640x480_kernel_stats.csv"))
if os.path.exists(os.path.join(g, "sfs_640x480.json")):
    shutil.copy(os.path.join(g, "sfs_640x480.json"), os.path.join(out, "sfs_640x480.json"))
if os.path.exists(os.path.join(g, "secondary_configs.json")):
    shutil.copy(os.path.join(g, "secondary_configs.json"), os.path.join(out, "secondary_configs.json"))

# kernel-name fragment -> (label, algorithmic bytes per launch at the profiled size or None)
NPX = 2048 * 2048
O_BA = 678718
KEYS = [# round 6: shape_from_shading on pixel pairs and packed planes (energy_sfs_pair.hip; template arguments SUMS, CTC, INIT, DIAG, OCC, PUPD, UPD, LMQ, FIN, MODEL, DEPTH)
        ("k_pmarch<true, false, false, false, 2, false, true, false, false, false", "shape_from_shading one-kernel GN iteration on pixel pairs: PCGUpdate + applyJTJ + three sums, p_k into the ring (2048^2; 40 B/pixel: Gx Gy Gz 12, flags+masks 4, r Ap p read 12 + written 12)", 40 * NPX),
        ("k_pmarch<true, true, false, false, 2, false, true, true", "shape_from_shading one-kernel LM iteration on pixel pairs (2048^2; 60 B/pixel: + delta read and written 8, CtC b M^-1 12)", 60 * NPX),
        ("k_pmarch<false, true, true, false", "shape_from_shading PCGInit1 J^T F on pixel pairs (2048^2; 44 B/pixel: Gx Gy Gz BI 16, flags+masks 4, X D 8 in; r z p delta 16 out)", 44 * NPX),
        ("k_pmarch<false, true, true, true, 2, false, false, false, true", "shape_from_shading PCGInit1 + PCGFinalizeDiagonal in one launch (LM; 2048^2; 56 B/pixel: 28 in; r z p delta CtC M^-1 b 28 out; SSq written at the first step only)", 56 * NPX),
        ("k_pmarch<false, false, false, false, 2, false, false, false, false, true", "shape_from_shading LM model cost in one launch: owed delta update + J^T J delta + two dots + savePreviousUnknowns + PCGLinearUpdate (2048^2; 44 B/pixel)", 44 * NPX),
        ("k_pmarch<false, false, false, false, 2, false, false, false, false, false", "shape_from_shading applyJTJ on pixel pairs (2048^2; 24 B/pixel)", 24 * NPX),
        ("k_pprecompute<false>", "shape_from_shading precompute, closed-form partials, packed planes (2048^2; 34 B/pixel: X D I 12 + masks 2 in; Gx Gy Gz BI 16 + flags/masks 4 out)", 34 * NPX),
        ("k_pprecompute<true>", "shape_from_shading precompute + computeCost in one launch (2048^2; 34 B/pixel)", 34 * NPX),
        ("k_march<true, false, false, false, 2, false, true>", "shape_from_shading one-kernel GN iteration: PCGUpdate + applyJTJ + three sums (marching kernel, 2048^2; 57 B/pixel)", 57 * NPX),
        ("k_march<false, true, false, false, 2, true, false>", "shape_from_shading LM: PCGStep3 + (J^T J + CtC) p (marching kernel, 2048^2; 45 B/pixel)", 45 * NPX),
        ("k_march<true, false, false, false, 2, false, false>", "shape_from_shading applyJTJ + three sums (marching kernel, 2048^2; 37 B/pixel)", 37 * NPX),
        ("k_march<false, true, false, false, 2, false, false>", "shape_from_shading (J^T J + CtC) p (marching kernel, LM, 2048^2; 37 B/pixel)", 37 * NPX),
        ("k_march<false, false, false, false", "shape_from_shading applyJTJ (marching kernel, 2048^2; 33 B/pixel)", 33 * NPX),
        ("k_march<false, true, true, false", "shape_from_shading PCGInit1 J^T F (marching kernel, 2048^2; 49 B/pixel: X D G Wt fl in, r z p delta out)", 49 * NPX),
        ("k_march<false, true, true, true", "shape_from_shading PCGInit1 J^T F + LM diagonal (marching kernel, 2048^2; 53 B/pixel)", 53 * NPX),
        ("k_precompute_march", "shape_from_shading precompute (marching kernel, 2048^2; 39 B/pixel: X D I masks in, G Wt fl out)", 39 * NPX),
        ("k_fused<1>", "shape_from_shading applyJTJ (fused, 2048^2)", 33 * NPX),
        ("k_fused<0>", "shape_from_shading PCGInit1 J^T F (fused, 2048^2)", None),
        ("k_cam2", "bundle_adjustment J^T(Jp) camera kernel, blocks rebuilt in closed form (ladybug-1723 shape; 36 B/observation: index 4, point 12, p of the point 12, J p out 8; rounds 2-3 loaded the 96-byte block: 116)", 36 * O_BA),
        ("k_pt2", "bundle_adjustment J^T(Jp) point kernel (ladybug-1723 shape; 36 B/observation: packed point block 24, index 4, J p 8)", 36 * O_BA),
        ("k_arap_resident", "ARAP resident PCG loop, 100 iterations per launch (102,400 vertices; per iteration 48 B/vertex of A p granules out and the ghosts' in, through the fabric: latency-, not byte-bound)", None),
        ("k_arap_apply_rc", "ARAP applyJTJ, per-edge blocks recomputed, + sums + in-kernel finish (102,400 vertices, 614,400 directed edges; 13.5 MB algorithmic, SURVEY 8d)", 13.5e6),
        ("k_arap_apply_ell", "ARAP applyJTJ, stored per-edge blocks (round 2's kernel)", 13.5e6),
        ("k_pcg_resident<3", "image_warping resident PCG loop, 100 iterations per launch (512^2; 99 B/pixel per iteration in the launch-per-iteration formulation)", 100 * 99 * 512 * 512),
        ("k_iter<3, 512", "image_warping one-kernel PCG iteration, LDS-tiled form (512^2)", 99 * 512 * 512),
        ("k_pcg_update", "PCGUpdate (flat)", None)]
dur = collections.defaultdict(list)
kt = newest(os.path.join(g, "cfg_kt", "*", "*_kernel_trace.csv"))
for row in csv.DictReader(open(kt)):
    for key, lab, _ in KEYS:
        if key in row["Kernel_Name"]:
            dur[lab].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
pmc = collections.defaultdict(dict)
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(g, f"cfg_{kind}", "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != ctr:
            continue
        for key, lab, _ in KEYS:
            if key in row["Kernel_Name"]:
                agg[lab].append(float(row["Counter_Value"]))
    for lab, v in agg.items():
        pmc[lab][ctr + "_KB_mean"] = sum(v) / len(v)
res = {}
for key, lab, alg in KEYS:
    if lab not in dur:
        continue
    d = sorted(dur[lab]); us = sum(d) / len(d)
    e = {"launches": len(d), "avg_us": round(us, 2), "median_us": round(d[len(d) // 2], 2)}
    if "FETCH_SIZE_KB_mean" in pmc[lab] and "WRITE_SIZE_KB_mean" in pmc[lab]:
        b = (2.0 * pmc[lab]["FETCH_SIZE_KB_mean"] + pmc[lab]["WRITE_SIZE_KB_mean"]) * 1024.0
        e.update(traffic_bytes_per_launch=b, traffic_GBps=round(b / us * 1e-3, 1), frac_of_8TBps_on_traffic=round(b / us * 1e-3 / 8000.0, 3))
    if alg:
        e.update(algorithmic_bytes_per_launch=alg, algorithmic_GBps=round(alg / us * 1e-3, 1), frac_of_8TBps_algorithmic=round(alg / us * 1e-3 / 8000.0, 3))
    res[lab] = e
json.dump(res, open(os.path.join(out, "secondary_kernels.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
