// dsl_plugin.cpp -- an EnergyPlugin generated from a .t file at Plan time (dsl.hpp): the translation unit of dsl_codegen.cpp is compiled
// with hipRTC for gfx950 and driven through the same interface as the hand-written plugins, in the reference's unfused residual-wise
// schedule (gauss_newton.t:998-1015: PCGInit1 + _Finish, PCGStep1 = clear Ap_X, one kernel per residual group, PCGStep1_Finish;
// PCGStep2 / PCGStep3 are the energy-independent kernels of pcg_kernels.hip).
// Schedule lines honoured per residual (thallo.t:5757-5772; the reference's five J^T J p schedules, gauss_newton.t:1019-1047, 1332-1525, 560-622):
//   (none)                              JtJp inline: the generated applyJTJ kernel recomputes the partials every PCG iteration
//   r.X.Jp:set_materialize(true)        Jt[Jp]: generated applyJ into a materialized Jp vector, generated applyJt from it
//   r.X.J:set_materialize(true)         [Jt][[J]p]: the generated dumpJ kernel writes the residual's rows once per GN iteration (ELL: K entries per
//                                       row); every PCG iteration applies them with the energy-independent thallo_hip_ell_apply -- no derivative is
//                                       re-evaluated inside the PCG loop; with Jp materialized too, as the pair J p / J^T (J p)
//   J and JtJ materialized on EVERY residual and n <= THALLO_DENSE_JTJ_MAX (default 2048): dense [JtJ]p -- J^T J accumulated once per GN iteration,
//                                       one GEMV per PCG iteration (gauss_newton.t:560-622, 1216-1241); above that size JtJ requests run as [Jt][[J]p]
//                                       (the reference's [[Jt][J]]p forms the sparse product with csrgemm: the same operator, not built here)
#include "dsl.hpp"
#include "plugin.hpp"
#include <hip/hiprtc.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace thallo {

namespace {

class GeneratedPlugin : public EnergyPlugin {
    dsl::Problem P;
    dsl::Generated G;
    std::string label;
    std::vector<long> dimv;                        // dimension sizes
    std::vector<UnknownImage> imgs;
    std::vector<int> unknown_input;                // imgs[k] -> index into P.inputs
    std::vector<long> uoff;                        // per input: flat offset of the unknown image in the solver vectors (-1 otherwise)
    long n_unk = 0;
    hipModule_t mod = nullptr;
    std::vector<hipFunction_t> fn;                 // G.kernels order
    std::vector<unsigned char> ctx;                // the generated code's `Ctx`, by value into every launch
    size_t off_dim = 0, off_prm = 0, off_uoff = 0;
    std::vector<void*> bound;                      // problem parameters as last bound (by input index)
    DeviceBuffer jp;                               // Jt[Jp] schedule: the materialized Jp vector
    std::vector<long> nel, jp_off;                 // per residual: elements, offset of its rows in jp
    std::vector<DeviceBuffer*> jval, jcol;         // per residual with a materialized J: ELL values / unknown indices ([rows][K])
    DeviceBuffer dense;                            // dense [JtJ]p: the n x n matrix
    bool dense_ = false;
    bool ok_ = false;

    long elements(const dsl::Residual& r) const { long n = 1; for (int d : r.domain) n *= dimv[d]; return n; }
    int grid_for(long n, int cap) const
    {
        long g = (n + 255) / 256; if (g < 1) g = 1; if (g > cap) g = cap;
        return (int)g;
    }
    int launch(int kernel, int grid, void** args, hipStream_t s)
    {
        hipError_t e = hipModuleLaunchKernel(fn[kernel], grid, 1, 1, 256, 1, 1, 0, s, args, nullptr);
        return e == hipSuccess ? 0 : -(int)e;
    }
    int kernel_of(int residual, int kind) const { return residual * 6 + kind; }
    long rows_of(size_t ri) const { return nel[ri] * (long)P.residuals[ri].exprs.size(); }

public:
    GeneratedPlugin(const dsl::Problem& p, const unsigned* dims) : P(p)
    {
        label = "generated:" + P.file.substr(P.file.find_last_of('/') == std::string::npos ? 0 : P.file.find_last_of('/') + 1);
        for (size_t d = 0; d < P.dims.size(); ++d) dimv.push_back((long)dims[d]);
        uoff.assign(P.inputs.size(), -1);
        for (size_t i = 0; i < P.inputs.size(); ++i) {
            const dsl::Input& in = P.inputs[i];
            if (in.kind != dsl::InputKind::Unknown) continue;
            long px = 1; for (int d : in.dims) px *= dimv[d];
            uoff[i] = n_unk; imgs.push_back({ in.slot, px * in.channels }); unknown_input.push_back((int)i); n_unk += px * in.channels;
        }
        std::string err;
        if (!dsl::generate_source(P, G, err)) { set_error("%s: %s", P.file.c_str(), err.c_str()); return; }
        if (compile()) return;
        // Ctx layout of the generated code: const void* in[NIN]; int dim[max(NDIM,1)]; float prm[NIN]; long uoff[NIN]
        const size_t nin = P.inputs.size(), nd = P.dims.empty() ? 1 : P.dims.size();
        off_dim = 8 * nin; off_prm = off_dim + 4 * nd; off_uoff = (off_prm + 4 * nin + 7) / 8 * 8;
        ctx.assign(off_uoff + 8 * nin, 0);
        for (size_t d = 0; d < P.dims.size(); ++d) { const int v = (int)dimv[d]; memcpy(&ctx[off_dim + 4 * d], &v, 4); }
        for (size_t i = 0; i < nin; ++i) memcpy(&ctx[off_uoff + 8 * i], &uoff[i], 8);
        long rows = 0;
        for (auto& r : P.residuals) { nel.push_back(elements(r)); jp_off.push_back(rows); if (r.mat_Jp) rows += nel.back() * (long)r.exprs.size(); }
        if (rows && jp.alloc(sizeof(float) * (size_t)rows + 256)) { set_error("%s: out of device memory for the materialized Jp", label.c_str()); return; }
        bool all_jtj = true;
        for (auto& r : P.residuals) all_jtj = all_jtj && r.mat_J && r.mat_JtJ;
        long dense_max = 2048; if (const char* e = getenv("THALLO_DENSE_JTJ_MAX")) dense_max = atol(e);
        dense_ = all_jtj && n_unk <= dense_max;
        if (dense_ && dense.alloc(sizeof(float) * (size_t)n_unk * (size_t)n_unk)) { set_error("%s: out of device memory for the dense JtJ", label.c_str()); return; }
        jval.assign(P.residuals.size(), nullptr); jcol.assign(P.residuals.size(), nullptr);
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (!P.residuals[ri].mat_J && !P.residuals[ri].mat_JtJ) continue;
            const size_t ent = (size_t)rows_of(ri) * (size_t)G.slots_per_row[ri];
            jval[ri] = new DeviceBuffer(); jcol[ri] = new DeviceBuffer();
            if (jval[ri]->alloc(sizeof(float) * ent + 256) || jcol[ri]->alloc(sizeof(int) * ent + 256)) { set_error("%s: out of device memory for the materialized J of %s", label.c_str(), P.residuals[ri].name.c_str()); return; }
        }
        ok_ = true;
    }
    ~GeneratedPlugin() override { if (mod) (void)hipModuleUnload(mod); for (auto b : jval) delete b; for (auto b : jcol) delete b; }
    const char* schedule() const { return dense_ ? "dense [JtJ]p" : "per residual"; }
    bool ok() const { return ok_; }

    int compile()
    {
        hiprtcProgram prog = nullptr;
        const std::string src = "#include <hip/hip_runtime.h>\n" + G.source;
        if (hiprtcCreateProgram(&prog, src.c_str(), "thallo_generated.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { set_error("hiprtcCreateProgram failed"); return -1; }
        const char* opts[] = { "--offload-arch=gfx950", "-O3", "-munsafe-fp-atomics" };
        const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
        if (rc != HIPRTC_SUCCESS) {
            size_t n = 0; hiprtcGetProgramLogSize(prog, &n); std::string log(n, '\0'); if (n) hiprtcGetProgramLog(prog, &log[0]);
            set_error("%s: hipRTC compilation of the generated kernels failed:\n%.1500s", P.file.c_str(), log.c_str());
            hiprtcDestroyProgram(&prog); return -1;
        }
        size_t n = 0; hiprtcGetCodeSize(prog, &n); std::vector<char> code(n); hiprtcGetCode(prog, code.data());
        hiprtcDestroyProgram(&prog);
        if (const char* dump = getenv("THALLO_FRONTEND_DUMP")) { FILE* f = fopen(dump, "w"); if (f) { fputs(src.c_str(), f); fclose(f); } }
        if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { set_error("%s: hipModuleLoadData failed (no gfx950 device?)", label.c_str()); (void)hipGetLastError(); return -1; }
        for (auto& k : G.kernels) {
            hipFunction_t f = nullptr;
            if (hipModuleGetFunction(&f, mod, k.name.c_str()) != hipSuccess) { set_error("%s: generated kernel %s missing", label.c_str(), k.name.c_str()); return -1; }
            fn.push_back(f);
        }
        return 0;
    }

    const char* name() const override { return label.c_str(); }
    long n_unknowns() const override { return n_unk; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return P.use_preconditioner; }
    int bind(void** p) override
    {
        bound.assign(P.inputs.size(), nullptr);
        for (size_t i = 0; i < P.inputs.size(); ++i) {
            const dsl::Input& in = P.inputs[i];
            void* v = p[in.slot];
            if (!v) { set_error("%s: problem parameter %d (%s) is NULL", label.c_str(), in.slot, in.name.c_str()); return -1; }
            bound[i] = v;
            if (in.kind == dsl::InputKind::Param) { const float f = *(const float*)v; memcpy(&ctx[off_prm + 4 * i], &f, 4); v = nullptr; }     // host scalar, re-read every Init / Step
            memcpy(&ctx[8 * i], &v, 8);
        }
        return 0;
    }
    float* unknown_ptr(int k) override { return (float*)bound[unknown_input[k]]; }

    int cost(LaunchCtx& c, float* out) override
    {
        TimedLaunch t(c, "computeCost");
        const int cap = THALLO_HIP_MAX_PARTIALS / (int)P.residuals.size();
        int total = 0;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            const int g = grid_for(nel[ri], cap);
            float* o = out + total; void* args[] = { ctx.data(), &o };
            const int rc = launch(kernel_of((int)ri, 0), g, args, c.stream); if (rc < 0) return rc;
            total += g;
        }
        return total;
    }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        TimedLaunch t(c, "PCGInit1");
        hipStream_t s = c.stream;
        const size_t bytes = (size_t)v.n_alloc * sizeof(float);
        if (hipMemsetAsync(v.r, 0, bytes, s) != hipSuccess || hipMemsetAsync(v.pre, 0, bytes, s) != hipSuccess || hipMemsetAsync(v.p[cur], 0, bytes, s) != hipSuccess ||
            hipMemsetAsync(v.delta, 0, bytes, s) != hipSuccess) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {            // evalJTF: r = -J^T F, pre = diag(J^T J)   (thallo.t:3898-3902)
            float *r = v.r, *pre = v.pre; void* args[] = { ctx.data(), &r, &pre };
            const int rc = launch(kernel_of((int)ri, 1), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
        }
        if (v.diag && hipMemcpyAsync(v.diag, v.pre, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;      // LM: the raw diagonal
        if (dense_ && hipMemsetAsync(dense.ptr, 0, sizeof(float) * (size_t)n_unk * (size_t)n_unk, s) != hipSuccess) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {            // precomputeJ (gauss_newton.t:1019-1025): once per GN iteration
            if (!jval[ri]) continue;
            float* jv = (float*)jval[ri]->ptr; int* jc = (int*)jcol[ri]->ptr; void* args[] = { ctx.data(), &jv, &jc };
            int rc = launch(kernel_of((int)ri, 5), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
            if (dense_ && (rc = thallo_hip_dense_jtj_accumulate(rows_of(ri), G.slots_per_row[ri], jv, jc, n_unk, (float*)dense.ptr, s)) < 0) return rc;
        }
        return thallo_hip_pcg_init_finish(v.r, v.pre, v.pre, v.z, v.n, P.use_preconditioner ? 1 : 0, aN, s);           // PCGInit1_Finish (gauss_newton.t:712-731)
    }
    int apply(LaunchCtx& c, const float* p, float* Ap, float* out, long n_alloc)
    {
        hipStream_t s = c.stream;
        if (dense_) {                                  // dense [JtJ]p
            const int rc = thallo_hip_dense_gemv(n_unk, (const float*)dense.ptr, p, Ap, s); if (rc < 0) return rc;
            return thallo_hip_dot(p, Ap, n_unk, out, s);
        }
        if (hipMemsetAsync(Ap, 0, (size_t)n_alloc * sizeof(float), s) != hipSuccess) return -1;                         // Ap_X:clear() (gauss_newton.t:1633-1635)
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            const int g = grid_for(nel[ri], 4096);
            if (jval[ri]) {                        // materialized J: no derivative evaluation inside the PCG loop
                const float* jv = (const float*)jval[ri]->ptr; const int* jc = (const int*)jcol[ri]->ptr; const int K = G.slots_per_row[ri];
                int rc;
                if (P.residuals[ri].mat_Jp) {
                    float* j = (float*)jp.ptr + jp_off[ri];
                    rc = thallo_hip_ell_apply(1, rows_of(ri), K, jv, jc, p, j, nullptr, s); if (rc < 0) return rc;
                    rc = thallo_hip_ell_apply(2, rows_of(ri), K, jv, jc, nullptr, j, Ap, s);
                } else rc = thallo_hip_ell_apply(0, rows_of(ri), K, jv, jc, p, nullptr, Ap, s);
                if (rc < 0) return rc;
            } else if (P.residuals[ri].mat_Jp) {          // Jt[Jp]: Jp = J p materialized, then Ap += J^T Jp
                float* j = (float*)jp.ptr + jp_off[ri];
                void* a1[] = { ctx.data(), &p, &j }; int rc = launch(kernel_of((int)ri, 3), g, a1, s); if (rc < 0) return rc;
                void* a2[] = { ctx.data(), &j, &Ap }; rc = launch(kernel_of((int)ri, 4), g, a2, s); if (rc < 0) return rc;
            } else {
                void* args[] = { ctx.data(), &p, &Ap };
                const int rc = launch(kernel_of((int)ri, 2), g, args, s); if (rc < 0) return rc;
            }
        }
        return thallo_hip_dot(p, Ap, n_unk, out, s);                  // PCGStep1_Finish: alphaD = p . Ap_X
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    { TimedLaunch t(c, "PCGStep1"); return apply(c, p, Ap, out, thallo_hip_vector_elems(n_unk)); }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1");
        return apply(c, v.p[cur ^ 1], v.Ap, out, v.n_alloc);
    }
};

}  // namespace

EnergyPlugin* make_generated_plugin(const char* filename, const unsigned* dims)
{
    dsl::Problem p; std::string err;
    if (!dsl::run_problem_file(filename, p, err)) { set_error("%s", err.c_str()); return nullptr; }
    GeneratedPlugin* g = new GeneratedPlugin(p, dims);
    if (!g->ok()) { delete g; return nullptr; }
    return g;
}

}  // namespace thallo

// The front-end without a device (tests, tooling): what = 0 the declarations as text (dsl::describe), 1 the generated HIP translation unit.
// Returns the length of the text (which is truncated to cap - 1), or -1 with ThalloX_LastError() set.
extern "C" int ThalloX_FrontendText(const char* filename, int what, char* out, int cap)
{
    if (!filename || !out || cap < 1) return -1;
    thallo::dsl::Problem p; std::string err, text;
    if (!thallo::dsl::run_problem_file(filename, p, err)) { thallo::set_error("%s", err.c_str()); return -1; }
    if (what == 0) text = thallo::dsl::describe(p);
    else {
        thallo::dsl::Generated g;
        if (!thallo::dsl::generate_source(p, g, err)) { thallo::set_error("%s: %s", filename, err.c_str()); return -1; }
        text = "#include <hip/hip_runtime.h>\n" + g.source;
    }
    snprintf(out, (size_t)cap, "%s", text.c_str());
    return (int)text.size();
}
