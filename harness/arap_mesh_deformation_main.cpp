// arap_mesh_deformation -- the reference's examples/arap_mesh_deformation application over libThallo.so: a triangle mesh
// (.ply / .off) + its landmark file (.mrk), one-ring edge lists (CombinedSolver.h:120-150), weights sqrt(4) / sqrt(1) (main.cpp:115-116),
// numIter = 10 outer solves with the landmark targets interpolated from the rest position to the target (CombinedSolver.h:98-118),
// GN 20 x 100 each (main.cpp:91-93).  The reference subdivides the mesh once (OpenMesh sqrt(3)) before solving, which is why the shipped
// landmarks index beyond the raw vertex count; --split-faces inserts the face centroids (the same vertex numbering, without the edge
// flips and smoothing).  Artefacts: finalCosts.json, perf.json, results/results_float.csv, out.ply.
//
//   arap_mesh_deformation [mesh.ply|.off] [-o energy.t] [-n numIter] [-N nonLinearIter] [-L linearIter] [--split-faces] [--profile]
#include "data_formats.hpp"
#include "thallo_harness.hpp"

using namespace harness;

int main(int argc, char** argv)
{
    std::string file = "../data/small_armadillo.ply", energy = "arap_mesh_deformation.t";
    int numIter = 10, nonLinearIter = 20, linearIter = 100; bool split = false, profile = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(1); } return std::string(argv[++i]); };
        if (a == "-o") energy = next(); else if (a == "-n") numIter = std::atoi(next().c_str()); else if (a == "-N") nonLinearIter = std::atoi(next().c_str());
        else if (a == "-L") linearIter = std::atoi(next().c_str()); else if (a == "--split-faces") split = true; else if (a == "--profile") profile = true; else file = a;
    }
    Mesh mesh; Landmarks lm;
    try { mesh = read_mesh(file); lm = read_mrk(file.substr(0, file.size() - 3) + "mrk"); }
    catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    if (split) {
        const size_t nv = mesh.v.size(); std::vector<std::vector<int>> nf;
        for (size_t f = 0; f < mesh.f.size(); ++f) {
            const auto& fc = mesh.f[f]; std::array<float, 3> c = { 0, 0, 0 };
            for (int i : fc) for (int k = 0; k < 3; ++k) c[k] += mesh.v[i][k] / (float)fc.size();
            mesh.v.push_back(c);
            for (size_t k = 0; k < fc.size(); ++k) nf.push_back({ fc[k], fc[(k + 1) % fc.size()], (int)(nv + f) });
        }
        mesh.f = nf;
    }
    const unsigned N = (unsigned)mesh.v.size();
    std::printf("Faces: %d\nVertices: %d\n", (int)mesh.f.size(), (int)N);
    for (int i : lm.index) if (i < 0 || i >= (int)N) { std::fprintf(stderr, "landmark vertex %d outside the mesh (%u vertices): try --split-faces\n", i, N); return 1; }
    std::vector<int> v0, v1; directed_edges(mesh, v0, v1);
    const unsigned E2 = (unsigned)v0.size();

    std::vector<float> pos(3 * (size_t)N);
    for (unsigned i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) pos[3 * i + k] = mesh.v[i][k];
    DeviceArray dPos, dAng(3 * (size_t)N * sizeof(float)), dOrig, dCons, dV0, dV1;
    dOrig.upload(pos); dV0.upload(v0); dV1.upload(v1);
    auto set_constraints = [&](float alpha) {      // CombinedSolver.h:98-118: unconstrained = -infinity
        std::vector<float> c(3 * (size_t)N, -std::numeric_limits<float>::infinity());
        for (size_t i = 0; i < lm.index.size(); ++i) for (int k = 0; k < 3; ++k)
            c[3 * (size_t)lm.index[i] + k] = (1 - alpha) * mesh.v[lm.index[i]][k] + alpha * lm.target[i][k];
        dCons.upload(c);
    };
    float w_fit = std::sqrt(4.0f), w_reg = std::sqrt(1.0f);
    SolverParameters sp; sp.ints["nIterations"] = (unsigned)nonLinearIter; sp.ints["lIterations"] = (unsigned)linearIter;
    NamedRun run; run.name = "ThalloGN";
    {
        ThalloSolver solver({ N, E2 }, energy, "gauss_newton");
        dPos.upload(pos); dAng.zero(); set_constraints(1.0f);                                  // resetGPUMemory()
        // Inputs of arap_mesh_deformation.t in index order: w_fitSqrt w_regSqrt (host), Position Angle Original Constraints (device), the two edge lists
        std::vector<void*> params = { &w_fit, &w_reg, dPos.data(), dAng.data(), dOrig.data(), dCons.data(), dV0.data(), dV1.data() };
        for (int i = 0; i < numIter; ++i) {
            if (numIter > 1) std::cout << "//////////// ITERATION" << i << "  (" << run.name << ") ///////////////" << std::endl;
            else std::cout << "//////////// (" << run.name << ") ///////////////" << std::endl;
            set_constraints((float)(i + 1) / (float)numIter);
            run.final_cost = solver.solve(sp, params, profile, run.iters);
        }
        run.perf = solver.summary();
    }
    save_artefacts("Mesh Deformation ARAP", 1, { run }, profile);
    const auto res = dPos.download<float>();
    for (unsigned i = 0; i < N; ++i) for (int k = 0; k < 3; ++k) mesh.v[i][k] = res[3 * i + k];
    write_ply_ascii("out.ply", mesh);
    return 0;
}
