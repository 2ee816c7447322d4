// plugins.cpp -- the bundled energy plugins (host side).  Each one binds the caller's void**
// (API/src/util.t:609-643) and forwards to the C-ABI kernel shim (include/thallo_hip.h).
#include "plugin.hpp"
#include <cstdio>
#include <cstring>

namespace thallo {

int DeviceBuffer::alloc(size_t n)
{
    release();
    if (n == 0) return 0;
    hipError_t e = hipMalloc(&ptr, n);
    if (e != hipSuccess) { ptr = nullptr; set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); return -(int)e; }
    bytes = n;
    e = hipMemset(ptr, 0, n);
    return e == hipSuccess ? 0 : -(int)e;
}
void DeviceBuffer::release() { if (ptr) { hipFree(ptr); ptr = nullptr; bytes = 0; } }

// ------------------------------------------------------------------ tests/minimal/laplacian.t
class LaplacianImagePlugin : public EnergyPlugin {
    int W, H; float w_fit; int xguard;
    std::vector<UnknownImage> imgs;
    float* X = nullptr; const float* A = nullptr;
public:
    LaplacianImagePlugin(const unsigned* dims, float w, int xg) : W((int)dims[0]), H((int)dims[1]), w_fit(w), xguard(xg)
    { imgs.push_back({ 0, (long)W * H }); }
    const char* name() const override { return "laplacian_image"; }
    long n_unknowns() const override { return (long)W * H; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return false; }
    int bind(void** p) override { X = (float*)p[0]; A = (const float*)p[1]; return (X && A) ? 0 : -1; }
    float* unknown_ptr(int) override { return X; }
    int cost(LaunchCtx& c, float* out) override
    { TimedLaunch t(c, "computeCost"); return thallo_hip_lapimg_cost(W, H, X, A, w_fit, xguard, out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    { TimedLaunch t(c, "PCGInit1"); return thallo_hip_lapimg_pcg_init(W, H, X, A, w_fit, xguard, v.r, v.z, v.p[cur], v.delta, aN, c.stream); }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    { TimedLaunch t(c, "PCGStep1"); return thallo_hip_lapimg_pcg_step1(W, H, w_fit, xguard, v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.Ap, first ? 1 : 0, aN, aD, bN, out, c.stream); }
};

// ------------------------------------------------------------------ examples/image_warping/image_warping.t
class ImageWarpingPlugin : public EnergyPlugin {
    int W, H;
    std::vector<UnknownImage> imgs;
    float *offset = nullptr, *angle = nullptr;
    const float *urshape = nullptr, *constraints = nullptr, *mask = nullptr;
    float w_fit = 0, w_reg = 0;
    DeviceBuffer cs, flags;     // per-GN-iteration planes: (cos,sin) float2 and validity bits
public:
    ImageWarpingPlugin(const unsigned* dims) : W((int)dims[0]), H((int)dims[1])
    {
        const long N = (long)W * H;
        imgs.push_back({ 0, 2 * N }); imgs.push_back({ 1, N });
    }
    const char* name() const override { return "image_warping"; }
    long n_unknowns() const override { return 3L * W * H; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return true; }           // image_warping.t:11
    int bind(void** p) override
    {
        offset = (float*)p[0]; angle = (float*)p[1]; urshape = (const float*)p[2];
        constraints = (const float*)p[3]; mask = (const float*)p[4];
        if (!offset || !angle || !urshape || !constraints || !mask || !p[5] || !p[6]) { set_error("image_warping: null problem parameter"); return -1; }
        w_fit = *(const float*)p[5]; w_reg = *(const float*)p[6];       // host scalars, re-read every Init/Step
        const long N = (long)W * H;
        if (!cs.ptr) { if (cs.alloc(N * 8)) return -1; if (flags.alloc((N + 255) / 256 * 256)) return -1; }
        return 0;
    }
    float* unknown_ptr(int k) override { return k == 0 ? offset : angle; }
    int cost(LaunchCtx& c, float* out) override
    { TimedLaunch t(c, "computeCost"); return thallo_hip_iw_cost(W, H, 0, H, offset, angle, urshape, constraints, mask, w_fit, w_reg, out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_iw_pcg_init(W, H, 0, H, offset, angle, urshape, constraints, mask, w_fit, w_reg,
                                      v.r, v.pre, v.z, v.p[cur], v.delta, (float*)cs.ptr, (unsigned char*)flags.ptr, aN, c.stream);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_iw_pcg_step1(W, H, 0, H, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                       v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.Ap, first ? 1 : 0, aN, aD, bN, out, c.stream);
    }
};

EnergyPlugin* make_plugin(const ProblemSpec& spec, const unsigned* dims)
{
    auto cst = [&](const char* k, double dflt) { auto it = spec.constants.find(k); return it == spec.constants.end() ? dflt : it->second; };
    if (spec.energy == "laplacian_image") return new LaplacianImagePlugin(dims, (float)cst("w_fit", 0.2), (int)cst("xguard", 0));
    if (spec.energy == "image_warping")   return new ImageWarpingPlugin(dims);
    set_error("no gfx950 plugin for energy '%s' (%s)", spec.energy.c_str(), spec.file.c_str());
    return nullptr;
}

}  // namespace thallo
