// image_warping -- the reference's examples/image_warping application over libThallo.so:
// same inputs (image.png + image_mask.png + image.constraints, or a synthetic instance), same problem set-up
// (examples/image_warping/src/CombinedSolver.h:122-204: UrShape = pixel grid, Offset0 = UrShape, Angle0 = 0, w_fit = sqrt(100),
// w_reg = sqrt(0.01), border pinned), the same outer continuation (numIter = 19 solves with the marker targets interpolated
// from the pixel to the marker, main.cpp:131-149), and the reference's artefacts: finalCosts.json, perf.json,
// results/results_float.csv.  The warp field itself is saved as warp_offset.f32 (W*H float2) instead of a resampled image.
//
//   image_warping [image.png] [-d downsample] [-o energy.t] [-n numIter] [-N nonLinearIter] [-L linearIter] [--lm] [--profile]
//                 [--synthetic W H] [--io-only]
#include <cmath>
#include <memory>

#include "image_io.hpp"
#include "thallo_harness.hpp"

using namespace harness;

struct float2_ { float x, y; };

struct Instance {
    unsigned W = 0, H = 0;
    std::vector<float> mask;                        // 0 = active pixel (image_warping.t:14-15)
    std::vector<std::vector<int>> constraints;      // x y tx ty
};

static Instance load_instance(const std::string& png, int downsample)
{
    Instance in;
    const std::string base = png.substr(0, png.size() - 4);
    const Image8 mask = read_png(base + "_mask.png");
    in.constraints = read_constraints(base + ".constraints");
    in.W = mask.width / downsample; in.H = mask.height / downsample;
    in.mask.resize((size_t)in.W * in.H);
    for (unsigned y = 0; y < in.H; ++y) for (unsigned x = 0; x < in.W; ++x) in.mask[(size_t)y * in.W + x] = (float)mask.at(x * downsample, y * downsample, 0);
    for (auto& c : in.constraints) for (int& v : c) v /= downsample;
    return in;
}

static Instance synthetic_instance(unsigned W, unsigned H)
{   // masked disc in the middle, a ring of markers pushed outwards
    Instance in; in.W = W; in.H = H;
    in.mask.assign((size_t)W * H, 0.0f);
    const float cx = 0.5f * W, cy = 0.5f * H, rad = 0.1f * (W < H ? W : H);
    for (unsigned y = 0; y < H; ++y) for (unsigned x = 0; x < W; ++x) if ((x - cx) * (x - cx) + (y - cy) * (y - cy) < rad * rad) in.mask[(size_t)y * W + x] = 255.0f;
    for (int k = 0; k < 8; ++k) {
        const float a = 6.2831853f * k / 8;
        const int x = (int)(cx + 0.3f * W * std::cos(a)), y = (int)(cy + 0.3f * H * std::sin(a));
        const int tx = (int)(cx + 0.36f * W * std::cos(a + 0.2f)), ty = (int)(cy + 0.36f * H * std::sin(a + 0.2f));
        in.constraints.push_back({ x, y, tx, ty });
    }
    return in;
}

int main(int argc, char** argv)
{
    std::string file, energy = "image_warping.t";
    int downsample = 1, numIter = 19, nonLinearIter = 8, linearIter = 100;
    bool use_lm = false, profile = false, io_only = false;
    unsigned synW = 0, synH = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "missing value after %s\n", a.c_str()); std::exit(1); } return std::string(argv[++i]); };
        if (a == "-d") downsample = std::max(1, std::atoi(next().c_str()));
        else if (a == "-o") energy = next();
        else if (a == "-n") numIter = std::atoi(next().c_str());
        else if (a == "-N") nonLinearIter = std::atoi(next().c_str());
        else if (a == "-L") linearIter = std::atoi(next().c_str());
        else if (a == "--lm") use_lm = true;
        else if (a == "--profile") profile = true;
        else if (a == "--io-only") io_only = true;        // read the inputs, print their summary, touch no GPU
        else if (a == "--synthetic") { synW = (unsigned)std::atoi(next().c_str()); synH = (unsigned)std::atoi(next().c_str()); }
        else file = a;
    }
    Instance in;
    try {
        in = synW ? synthetic_instance(synW, synH) : load_instance(file.empty() ? "../data/cat512.png" : file, downsample);
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    const unsigned W = in.W, H = in.H;
    const size_t N = (size_t)W * H;
    std::printf("width %u, height %u\n", W, H);
    size_t active = 0;
    for (float m : in.mask) active += (m == 0.0f);
    std::printf("numActivePixels: %zu\n", active);
    if (io_only) {
        long sum = 0;
        for (auto& m : in.constraints) sum += m[0] + 3L * m[1] + 5L * m[2] + 7L * m[3];
        std::printf("markers: %zu checksum %ld\n", in.constraints.size(), sum);
        return 0;
    }
    for (unsigned y = 0; y < H; ++y) for (unsigned x = 0; x < W; ++x)            // pinned border (main.cpp:119-129)
        if (y == 0 || x == 0 || y == H - 1 || x == W - 1) in.constraints.push_back({ (int)x, (int)y, (int)x, (int)y });

    // device state (CombinedSolver.h:107-111,158-176)
    std::vector<float2_> urshape(N);
    for (unsigned y = 0; y < H; ++y) for (unsigned x = 0; x < W; ++x) urshape[(size_t)y * W + x] = { (float)x, (float)y };
    DeviceArray d_urshape, d_offset, d_angle(N * sizeof(float)), d_cons, d_mask;
    d_urshape.upload(urshape); d_mask.upload(in.mask);
    auto set_constraints = [&](float alpha) {       // CombinedSolver.h:178-204
        std::vector<float2_> c(N, float2_{ -1.0f, -1.0f });
        for (auto& m : in.constraints) {
            const int x = m[0], y = m[1];
            if (x < 0 || y < 0 || x >= (int)W || y >= (int)H) continue;
            if (in.mask[(size_t)y * W + x] == 0.0f) c[(size_t)y * W + x] = { (1.0f - alpha) * (float)x + alpha * (float)m[2], (1.0f - alpha) * (float)y + alpha * (float)m[3] };
        }
        d_cons.upload(c);
    };
    float w_fit = std::sqrt(100.0f), w_reg = std::sqrt(0.01f);
    // problem parameters in the order of the .t's Inputs (image_warping.t:3-11): Offset, Angle, UrShape, Constraints, Mask, w_fitSqrt, w_regSqrt
    SolverParameters sp;
    sp.ints["nIterations"] = (unsigned)nonLinearIter; sp.ints["lIterations"] = (unsigned)linearIter;

    const std::string kind = use_lm ? "levenberg_marquardt" : "gauss_newton";
    NamedRun run; run.name = use_lm ? "ThalloLM" : "ThalloGN";
    {
        ThalloSolver solver({ W, H }, energy, kind);
        d_offset.upload(urshape); d_angle.zero(); set_constraints(1.0f);           // resetGPU()
        std::vector<void*> params = { d_offset.data(), d_angle.data(), d_urshape.data(), d_cons.data(), d_mask.data(), &w_fit, &w_reg };
        for (int i = 0; i < numIter; ++i) {
            if (numIter > 1) std::cout << "//////////// ITERATION" << i << "  (" << run.name << ") ///////////////" << std::endl;
            else std::cout << "//////////// (" << run.name << ") ///////////////" << std::endl;
            set_constraints((float)(i + 1) / (float)numIter);
            run.final_cost = solver.solve(sp, params, profile, run.iters);
        }
        run.perf = solver.summary();
    }
    save_artefacts("Image Warping", 1, { run }, profile);
    const auto off = d_offset.download<float2_>();
    { std::ofstream f("warp_offset.f32", std::ios::binary); f.write(reinterpret_cast<const char*>(off.data()), (std::streamsize)(off.size() * sizeof(float2_))); }
    // a quick look at the result: displacement magnitude as an 8-bit image
    Image8 vis; vis.width = W; vis.height = H; vis.channels = 1; vis.px.resize(N);
    float mx = 1e-6f;
    for (size_t i = 0; i < N; ++i) mx = std::max(mx, std::hypot(off[i].x - urshape[i].x, off[i].y - urshape[i].y));
    for (size_t i = 0; i < N; ++i) vis.px[i] = (uint8_t)(255.0f * std::hypot(off[i].x - urshape[i].x, off[i].y - urshape[i].y) / mx);
    write_png("out_displacement.png", vis);
    return 0;
}
