// shape_from_shading -- the reference's examples/shape_from_shading application over libThallo.so: loads
// <prefix>_targetIntensity / _targetDepth / _initialUnknown / _maskEdgeMap .imagedump and <prefix>.SFSSolverParameters
// (SFSSolverInput.h:47-66), binds them as SFSSolverInput.h:22-46 does, runs GN 60 x 10 (main.cpp:44-46) and leaves the reference's
// artefacts (finalCosts.json, perf.json, results/results_float.csv) plus sfsOutput.imagedump.
//
//   shape_from_shading [prefix] [-o energy.t] [-N nonLinearIter] [-L linearIter] [--lm] [--profile]
#include "data_formats.hpp"
#include "thallo_harness.hpp"

using namespace harness;

int main(int argc, char** argv)
{
    std::string prefix = "../data/shape_from_shading/default", energy = "shape_from_shading.t";
    int nonLinearIter = 60, linearIter = 10; bool use_lm = false, profile = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(1); } return std::string(argv[++i]); };
        if (a == "-o") energy = next(); else if (a == "-N") nonLinearIter = std::atoi(next().c_str()); else if (a == "-L") linearIter = std::atoi(next().c_str());
        else if (a == "--lm") use_lm = true; else if (a == "--profile") profile = true; else prefix = a;
    }
    ImageDump inten, depth, init, edges; SfsParameters prm;
    try {
        inten = read_imagedump(prefix + "_targetIntensity.imagedump"); depth = read_imagedump(prefix + "_targetDepth.imagedump");
        init = read_imagedump(prefix + "_initialUnknown.imagedump");   edges = read_imagedump(prefix + "_maskEdgeMap.imagedump");
        prm = read_sfs_parameters(prefix + ".SFSSolverParameters");
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    const unsigned W = (unsigned)init.width, H = (unsigned)init.height;
    const size_t N = (size_t)W * H;
    if (depth.f.size() != N || inten.f.size() != N || edges.u.size() < 2 * N) { std::fprintf(stderr, "inconsistent input sizes\n"); return 1; }
    size_t active = 0; for (float d : depth.f) active += d > 0.0f;
    std::printf("Num Active Unknowns: %zu\n", active);

    DeviceArray dX, dD, dI, dR, dC;
    dX.upload(init.f); dD.upload(depth.f); dI.upload(inten.f);
    dR.upload(std::vector<uint8_t>(edges.u.begin(), edges.u.begin() + N));            // row map, then column map (SFSSolverInput.h:42-44)
    dC.upload(std::vector<uint8_t>(edges.u.begin() + N, edges.u.begin() + 2 * N));
    // Inputs of shape_from_shading.t in index order: w_p w_s w_g f_x f_y u_x u_y L_1..L_9 (host floats), X D_i Im edgeMaskR edgeMaskC (device)
    std::vector<void*> params = { &prm.weightFitting, &prm.weightRegularizer, &prm.weightShading, &prm.fx, &prm.fy, &prm.ux, &prm.uy };
    for (int i = 0; i < 9; ++i) params.push_back(&prm.lightingCoefficients[i]);
    for (void* p : { dX.data(), dD.data(), dI.data(), dR.data(), dC.data() }) params.push_back(p);

    SolverParameters sp; sp.ints["nIterations"] = (unsigned)nonLinearIter; sp.ints["lIterations"] = (unsigned)linearIter;
    NamedRun run; run.name = use_lm ? "ThalloLM" : "ThalloGN";
    {
        ThalloSolver solver({ W, H }, energy, use_lm ? "levenberg_marquardt" : "gauss_newton");
        std::cout << "//////////// (" << run.name << ") ///////////////" << std::endl;
        run.final_cost = solver.solve(sp, params, profile, run.iters);
        run.perf = solver.summary();
    }
    save_artefacts("Shape From Shading", 1, { run }, profile);
    write_imagedump("sfsOutput.imagedump", (int)W, (int)H, dX.download<float>());
    return 0;
}
