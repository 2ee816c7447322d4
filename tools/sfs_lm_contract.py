"""sfs_lm_contract.py -- GPU probe, the device-side converse of tests/golden/lm_rounding_experiment.json: shape_from_shading 2048^2, LM 12 x 10,
on the product library and on a build whose energy_sfs.hip is compiled with -ffp-contract=off (make -C thallo_amd/csrc VARIANT=sfsnc SFSFLAGS=-ffp-contract=off),
each against oracle build A (-ffp-contract=off) of that file.  Adds the row "device" to the JSON.  python tools/sfs_lm_contract.py [run <lib>]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "lm_rounding_experiment.json")
if len(sys.argv) > 1 and sys.argv[1] == "run":
    os.environ["THALLO_LIB"] = sys.argv[2]
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import thallo_amd
    from thallo_amd import api, synthetic as syn
    from helpers import to_device
    W = H = 2048
    dev = to_device(syn.shape_from_shading(W, H))
    s = api.ThalloSolver((W, H), thallo_amd.energy_file("shape_from_shading"), solverkind="levenberg_marquardt")
    s.enable_lm()
    final, costs = s.solve(dev, profiled=True, nIterations=12, lIterations=10)
    print("COSTS " + json.dumps([float(c) for c in costs]))
    sys.exit(0)
fx = json.load(open(FX))
A = fx["runs"]["A"]["costs"]
row = {"instance": fx["instance"].replace(", one thread", "") + ", MI355X", "against": "oracle build A (-ffp-contract=off)"}
for tag, lib in (("product (-ffp-contract=fast, hipcc's default, for energy_sfs.hip)", os.path.join(ROOT, "thallo_amd", "libThallo.so")),
                 ("energy_sfs.hip with -ffp-contract=off", os.path.join(ROOT, "tools", "ab", "libThallo_sfsnc.so"))):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "run", lib], capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("COSTS ")]
    assert line, out.stderr[-2000:]
    c = json.loads(line[0][6:])
    row[tag] = {"costs": c, "rel_diff_vs_A": [float(f"{abs(x - y) / abs(y):.3g}") for x, y in zip(c, A)]}
fx["device"] = row
json.dump(fx, open(FX, "w"), indent=1)
print(json.dumps(row))
