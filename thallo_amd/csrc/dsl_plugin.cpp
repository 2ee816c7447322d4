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
//                                       one GEMV per PCG iteration (gauss_newton.t:560-622, 1216-1241).  With <handle>:set_direct_solve(true) in the file AND
//                                       THALLO_ENABLE_DIRECT_SOLVE=1 (the reference compiles this out: enable_direct_solve = false, gauss_newton.t:22) the GN step
//                                       solves the dense system by Cholesky instead of running PCG (:1280-1328, 1612-1613)
//   ... and n above that size:          sparse [[Jt][J]]p (:1394-1441 csrgemm, :1462-1481 one csrmv per PCG iteration): the CSR pattern of J^T J is built from
//                                       the rows' unknown indices once per Init on the host, its values are re-accumulated once per GN iteration on the device
// Where a residual's kernels run (thallo.t:5173-5190,5273-5306: the reference maps a group at its output when it can; r.X:compute_at_output(true) or useAutoscheduler = 1 here):
//   on its unknowns' own grid                 the stencil gather, one merged kernel per iteration domain (dsl_codegen.cpp "gather group")
//   through Sparse maps / other index forms   per-owner instance lists, built per Init from the residual's own index evaluation (build_incidence below), a thread or a wave per owner
//   otherwise (or asked not to)               residual-wise, atomics
#include "dsl.hpp"
#include "plugin.hpp"
#include <hip/hiprtc.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace thallo {

namespace {

// hipRTC: one translation unit -> a loaded module, for the architecture of the current device
int rtc_build(const std::string& src, bool wide, const char* what, hipModule_t* mod)
{
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "thallo_generated.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { set_error("hiprtcCreateProgram failed"); return -1; }
    // the architecture of the device the plan will run on (a library built for another ARCH must not generate gfx950 code and then blame the device)
    std::string arch = "--offload-arch=gfx950";
    {   int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.gcnArchName[0]) {
            std::string a = pr.gcnArchName; const size_t colon = a.find(':'); if (colon != std::string::npos) a.resize(colon);      // "gfx950:sramecc+:xnack-" -> "gfx950"
            arch = "--offload-arch=" + a;
        } else (void)hipGetLastError();
    }
    // (a unit with a wide residual -- hundreds of 32-wide dual operations in one function -- takes minutes at -O3 with the loops unrolled: 128 s for the
    //  reference's 17 x 17 deconvolution against 25 s without unrolling)
    // (-I: under rocprofv3 the runtime compiler does not find its own <hip/hip_runtime.h>)
    std::string inc = "-I";        // ROCM_PATH / HIP_PATH as hipcc reads them, else the image's default
    { const char* rp = getenv("ROCM_PATH"); const char* hp = getenv("HIP_PATH"); inc += (rp && rp[0]) ? rp : (hp && hp[0]) ? hp : "/opt/rocm"; inc += "/include"; }
    const char* opts[] = { arch.c_str(), "-O3", "-munsafe-fp-atomics", inc.c_str(), "-fno-unroll-loops" };
    const hiprtcResult rc = hiprtcCompileProgram(prog, wide ? 5 : 4, opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0; hiprtcGetProgramLogSize(prog, &n); std::string log(n, '\0'); if (n) hiprtcGetProgramLog(prog, &log[0]);
        set_error("%s: hipRTC compilation of the generated kernels failed:\n%.1500s", what, log.c_str());
        hiprtcDestroyProgram(&prog); return -1;
    }
    size_t n = 0; hiprtcGetCodeSize(prog, &n); std::vector<char> code(n); hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (hipModuleLoadData(mod, code.data()) != hipSuccess) { set_error("%s: hipModuleLoadData failed for code generated with %s", what, arch.c_str()); (void)hipGetLastError(); return -1; }
    return 0;
}

class GeneratedPlugin : public EnergyPlugin, public EnergyPlugin64 {
    dsl::Problem P;
    bool f64_ = false;                             // doublePrecision = 1: the unit is compiled with thallo_float = double, only the EnergyPlugin64 interface is used
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
    bool direct_ = false;                          // dense + set_direct_solve(true) + THALLO_ENABLE_DIRECT_SOLVE=1
    DeviceBuffer info_;                            // Cholesky status word
    bool sparse_jtj_ = false, sp_ready_ = false;   // sparse [[Jt][J]]p; pattern built (once per Init)
    DeviceBuffer sp_rowptr, sp_col, sp_val;
    std::vector<DeviceBuffer*> jdest;              // per residual: [rows][K][K] positions of the products in sp_val
    long sp_nnz = 0;
    bool ok_ = false;
    // materialized computed arrays (round 6; thallo.t:1868-1937,4046-4094): their planes (values + gradient images), their precompute kernels, and whether the planes
    // belong to the unknowns as they are now -- precompute() runs in front of whatever evaluates residuals (cost, PCGInit1) once per change of the unknowns: per GN
    // iteration and after an LM revert (gauss_newton.t:979-986,1748)
    std::vector<hipFunction_t> ca_fn;
    std::vector<DeviceBuffer*> ca_buf;
    std::vector<long> ca_nel;
    size_t off_cap = 0;
    bool ca_valid_ = false;
    std::string sched_;
    int precompute(LaunchCtx& c)
    {
        if (ca_valid_ || G.computed.empty()) return 0;
        TimedLaunch t(c, "precompute");
        for (size_t k = 0; k < G.computed.size(); ++k) {
            void* args[] = { ctx.data() };
            const int rc = launch_fn(ca_fn[k], grid_for(ca_nel[k], 4096), args, c.stream); if (rc < 0) return rc;
        }
        ca_valid_ = true;
        return 0;
    }

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
    int kernel_of(int residual, int kind) const { return residual * dsl::GEN_KINDS + kind; }
    std::vector<char> gather_;                     // per residual: the unknown-wise (gather) kernels run (compute_at_output, or the autoscheduler's choice where eligible)
    // merged gather kernels, one per iteration domain (G.groups): group_of_[ri] = the group residual ri runs in (-1: its own kernel); group_nel_: elements of the domain
    std::vector<hipFunction_t> grp_jtj, grp_jtf, grp_cost;      // (grp_cost[gi] == nullptr: the members' own cost kernels)
    std::vector<int> group_of_;
    std::vector<long> group_nel_;
    bool whole_ = false;                           // ONE group holds every residual and writes every unknown: its applyJTJ stores (no Ap clear, no read-modify-write) and alphaD rides along
    int launch_fn(hipFunction_t f, int grid, void** args, hipStream_t s)
    {
        hipError_t e = hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, s, args, nullptr);
        return e == hipSuccess ? 0 : -(int)e;
    }
    long rows_of(size_t ri) const { return nel[ri] * (long)P.residuals[ri].exprs.size(); }
    // unknown-wise lowering through index maps (dsl.hpp IncResidual): per (residual, owner group) the kernel pair and the owners' instance lists (CSR), rebuilt per Init
    struct IncRun { int ri = -1, g = -1; hipFunction_t jtj = nullptr, jtf = nullptr; DeviceBuffer ptr, els; long npix = 0; int wave = 0, lanes = 1; };
    std::vector<IncRun*> inc_runs_;
    std::vector<hipFunction_t> inc_uidx_;          // G.inc order
    std::vector<char> use_inc_;                    // per residual: gathered through per-owner instance lists for THIS Init's Sparse maps and dims
    std::vector<char> want_inc_;                   // ... what the constructor decided from the schedule (build_incidence may say no for one Init: few owners with enormous lists)
    bool inc_ready_ = false;
    long group_pixels(int g) const { long n = 1; for (int d : G.inc_groups[(size_t)g].dims) n *= dimv[(size_t)d]; return n; }
    // the owners' lists: the residual's own index evaluation (uidx_<ri>: K flat unknown indices per instance, -1 = absent), inverted on the host
    int build_incidence(hipStream_t s)
    {
        for (size_t a = 0; a < G.inc.size(); ++a) {
            const dsl::IncResidual& ir = G.inc[a];
            if (!use_inc_[(size_t)ir.ri]) continue;
            const long n = nel[(size_t)ir.ri]; const int K = ir.K;
            DeviceBuffer col;
            if (col.alloc(sizeof(int) * (size_t)n * (size_t)K + 256)) { set_error("%s: out of device memory for the index evaluation of %s", label.c_str(), P.residuals[(size_t)ir.ri].name.c_str()); return -1; }
            int* cp = (int*)col.ptr; void* args[] = { ctx.data(), &cp };
            if (launch_fn(inc_uidx_[a], grid_for(n, 4096), args, s) < 0) return -1;
            DeviceBuffer tot;
            if (tot.alloc(64)) return -1;
            bool few_owners = false;
            for (IncRun* run : inc_runs_) {
                if (run->ri != ir.ri) continue;
                const long npix = group_pixels(run->g);
                std::vector<int> slot_ch((size_t)K, 0); std::vector<long> slot_base((size_t)K, -1);       // slots of this group: base offset and channel count of their image
                for (int q = 0; q < K; ++q) {
                    const int in = ir.slot_input[(size_t)q];
                    bool mine = false; for (int x : G.inc_groups[(size_t)run->g].inputs) mine = mine || x == in;
                    if (P.inputs[(size_t)in].dims != G.inc_groups[(size_t)run->g].dims) mine = false;
                    if (mine) { slot_base[(size_t)q] = uoff[(size_t)in]; slot_ch[(size_t)q] = P.inputs[(size_t)in].channels; }
                }
                // the inversion runs on the device (round 5; the host loop over n x K indices it replaces took 25-50 ms of a first Init at the benchmark sizes): count per
                // owner -> prefix sums -> fill through atomic cursors -> every list sorted (ascending instance order, as the host built them: same sums from run to run)
                long total = 0;
                if (run->ptr.alloc((size_t)(npix + 1) * sizeof(int) + 64) ||
                    thallo_hip_incidence_count(cp, n, K, slot_base.data(), slot_ch.data(), npix, (int*)run->ptr.ptr, (long*)tot.ptr, s) < 0 ||
                    hipMemcpyAsync(&total, tot.ptr, sizeof(long), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
                    set_error("%s: the instance lists of %s could not be counted (out of device memory?)", label.c_str(), P.residuals[(size_t)ir.ri].name.c_str()); return -1;
                }
                if (total > 0x7fffffffL) { set_error("%s: more than 2^31 (instance, unknown) pairs in residual %s", label.c_str(), P.residuals[(size_t)ir.ri].name.c_str()); return -1; }
                run->npix = npix;
                // owners with long lists (bundle adjustment's cameras: hundreds of observations each, a few thousand owners) get a wave each; 16 instances per owner on
                // average is where a wave's 64 lanes stop being mostly idle
                run->wave = npix > 0 && total >= 16 * npix ? 1 : 0;
                // shorter lists: a few lanes per owner.  Measured (tools/generated_graph_times.py, THALLO_AB=inc_lanes=N): ARAP 102,400 vertices (12 instances per vertex)
                // 47.4 / 37.6 / 39.0 / 48.8 / 57.6 us per applyJTJ at 1 / 2 / 4 / 8 / 16 lanes; bundle adjustment's points (4.3 per point) 67 / 64 / 66 / 75 / 101: the walk
                // is bound by the number of scattered 4-byte gathers, not by its length, and the group's shuffle-adds cost more than they hide beyond 4 lanes.
                run->lanes = run->wave ? 64 : 1;
                if (!run->wave && npix > 0) {
                    const long lanes_env = [] { const char* e = env_switch("THALLO_INC_LANES"); return e ? atol(e) : 0L; }();      // THALLO_AB=inc_lanes=N (tools/generated_graph_times.py sweeps it)
                    const double mean = (double)total / (double)npix;
                    int l = 1; while (l < 4 && mean >= 6.0 * l) l *= 2;
                    if (lanes_env >= 1 && lanes_env <= 64 && (lanes_env & (lanes_env - 1)) == 0) l = (int)lanes_env;
                    run->lanes = l;
                }
                // ... but a handful of owners with enormous lists (a dense residual over W x H x pairs that reads ten camera poses: ten waves for the whole launch) is the
                // case the residual-wise kernels with their wave-aggregated atomics are for: below 256 such owners the residual keeps them
                if (run->wave && npix < 256) { few_owners = true; continue; }
                DeviceBuffer cursor;
                if (run->els.alloc((size_t)total * sizeof(int) + 64) || cursor.alloc((size_t)npix * sizeof(int) + 64) ||
                    thallo_hip_incidence_fill(cp, n, K, slot_base.data(), slot_ch.data(), npix, (const int*)run->ptr.ptr, (int*)cursor.ptr, (int*)run->els.ptr, s) < 0 ||
                    hipStreamSynchronize(s) != hipSuccess) {          // (cursor is released at the end of this scope: the fill must be through)
                    set_error("%s: out of device memory for the instance lists of %s", label.c_str(), P.residuals[(size_t)ir.ri].name.c_str()); return -1;
                }
            }
            if (few_owners) {
                use_inc_[(size_t)ir.ri] = 0;
                for (IncRun* run : inc_runs_) if (run->ri == ir.ri) { run->ptr.release(); run->els.release(); run->npix = 0; }
            }
        }
        inc_ready_ = true;
        return 0;
    }
    int launch_inc(int ri, bool jtj, void* a0, void* a1, hipStream_t s)
    {
        for (IncRun* run : inc_runs_) {
            if (run->ri != ri) continue;
            const int* ip = (const int*)run->ptr.ptr; const int* ie = (const int*)run->els.ptr; long npix = run->npix; int lanes = run->lanes;
            void* args[] = { ctx.data(), a0, a1, &ip, &ie, &npix, &lanes };
            const int rc = launch_fn(jtj ? run->jtj : run->jtf, grid_for(npix * lanes, 4096), args, s); if (rc < 0) return rc;
        }
        return 0;
    }

public:
    GeneratedPlugin(const dsl::Problem& p, const unsigned* dims, bool autoschedule, bool f64) : P(p), f64_(f64)
    {
        if (!f64_) for (auto& in : P.inputs) if (in.fixed_f64) { set_error("%s: %s is declared double: that needs doublePrecision = 1 (the single-precision kernels would read it as floats)", P.file.c_str(), in.name.c_str()); return; }
        if (f64_) {
            // the reference's double mode switches thallo_float (precision.t:3-6); an unknown declared with a fixed float type has no place in double solver vectors
            for (auto& in : P.inputs) if (in.kind == dsl::InputKind::Unknown && in.fixed_f32) { set_error("%s: doublePrecision = 1 and the unknown %s is declared with a fixed float type (use thallo_float)", P.file.c_str(), in.name.c_str()); return; }
            // one schedule in this mode: the generated kernels recompute the partials every PCG iteration (materialized J / J^T J / Jp live in float buffers)
            for (auto& r : P.residuals) { r.mat_J = false; r.mat_JtJ = false; r.mat_Jp = false; }
            P.direct_solve = false;
        }
        label = "generated:" + P.file.substr(P.file.find_last_of('/') == std::string::npos ? 0 : P.file.find_last_of('/') + 1);
        for (size_t d = 0; d < P.dims.size(); ++d) dimv.push_back((long)dims[P.canonical((int)d)]);      // (alias ids: further iteration variables over a declared dimension)
        uoff.assign(P.inputs.size(), -1);
        for (size_t i = 0; i < P.inputs.size(); ++i) {
            const dsl::Input& in = P.inputs[i];
            if (in.kind != dsl::InputKind::Unknown) continue;
            long px = 1; for (int d : in.dims) px *= dimv[d];
            uoff[i] = n_unk; imgs.push_back({ in.slot, px * in.channels }); unknown_input.push_back((int)i); n_unk += px * in.channels;
        }
        std::string err;
        // which residuals the caller wants gathered (they alone join the merged one-kernel-per-domain gather lowering, dsl_codegen.cpp)
        G.want_gather.assign(P.residuals.size(), 0);
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            const dsl::Residual& r = P.residuals[ri];
            G.want_gather[ri] = (r.at_output == 1 || (r.at_output < 0 && autoschedule)) && !r.mat_J && !r.mat_JtJ && !r.mat_Jp;
        }
        if (!dsl::generate_source(P, G, err, f64_)) { set_error("%s: %s", P.file.c_str(), err.c_str()); return; }
        if (compile()) return;
        // Unknown-wise (gather) lowering per residual: asked for with r.<name>:compute_at_output(true), or -- like the reference's autoscheduler, which every
        // example application switches on (thallo.t:5173-5190: residual dims == unknown dims, nothing materialized) -- chosen where it exists.
        gather_.assign(P.residuals.size(), 0);
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            const dsl::Residual& r = P.residuals[ri];
            const bool plain = !r.mat_J && !r.mat_JtJ && !r.mat_Jp;
            const bool want = r.at_output == 1 || (r.at_output < 0 && autoschedule);
            gather_[ri] = want && plain && G.gather_ok[ri];
        }
        // ... and through index maps where the stencil form does not exist (Sparse maps, unknowns over other dimensions): the owners' instance lists
        // (useAutoscheduler = 0 without compute_at_output(true), or compute_at_output(false): the residual-wise scatter, A/B)
        use_inc_.assign(P.residuals.size(), 0);
        {
            const bool off = false;
            for (size_t a = 0; a < G.inc.size(); ++a) {
                const dsl::IncResidual& ir = G.inc[a];
                const dsl::Residual& r = P.residuals[(size_t)ir.ri];
                const bool plain = !r.mat_J && !r.mat_JtJ && !r.mat_Jp;
                const bool want = r.at_output == 1 || (r.at_output < 0 && autoschedule);
                bool fits = n_unk < (1L << 31) && elements(r) < (1L << 31) && ir.K <= 64;
                for (int g : ir.groups) fits = fits && group_pixels(g) < (1L << 31);
                use_inc_[(size_t)ir.ri] = want && plain && !off && fits;
                if (!use_inc_[(size_t)ir.ri]) continue;
                for (size_t k = 0; k < ir.groups.size(); ++k) {
                    IncRun* run = new IncRun(); run->ri = ir.ri; run->g = ir.groups[k];
                    if (hipModuleGetFunction(&run->jtj, mod, ir.jtj[k].c_str()) != hipSuccess || hipModuleGetFunction(&run->jtf, mod, ir.jtf[k].c_str()) != hipSuccess) {
                        delete run; set_error("%s: generated kernel %s missing", label.c_str(), ir.jtj[k].c_str()); return;
                    }
                    inc_runs_.push_back(run);
                }
            }
        }
        want_inc_ = use_inc_;          // (prepare() starts every Init from this; build_incidence may take a residual back for that Init's maps -- ADVICE r4)
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            const dsl::Residual& r = P.residuals[ri];
            const bool plain = !r.mat_J && !r.mat_JtJ && !r.mat_Jp;
            if (r.at_output == 1 && !gather_[ri] && !use_inc_[ri])
                fprintf(stderr, "[thallo] warning: %s: residual %s asks for compute_at_output(true) but %s: its residual-wise kernels run\n", label.c_str(), r.name.c_str(),
                        plain ? "has no unknown-wise lowering (more than 48 unknown accesses per instance: the wide lowering scatters)" : "also materializes J / JtJ / Jp");
        }
        // the merged gather kernels: a group runs when all of its members are gathered (they are: only wanted residuals join a group) -- groups of one included
        group_of_.assign(P.residuals.size(), -1);
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {
            long n = 1; for (int d : G.groups[gi].domain) n *= dimv[d];
            group_nel_.push_back(n);
            for (int ri : G.groups[gi].members) if (gather_[(size_t)ri]) group_of_[(size_t)ri] = (int)gi;
        }
        {   // one group, every residual in it, every unknown channel among its targets: nothing else ever adds to Ap
            size_t nch = 0; for (int ui : unknown_input) nch += (size_t)P.inputs[(size_t)ui].channels;
            whole_ = !f64_ && G.groups.size() == 1 && G.groups[0].members.size() == P.residuals.size() && G.groups[0].targets.size() == nch;
            for (size_t ri = 0; ri < P.residuals.size(); ++ri) whole_ = whole_ && group_of_[ri] == 0;
        }
        // Ctx layout of the generated code: const void* in[NIN]; int dim[max(NDIM,1)]; float prm[NIN]; long uoff[NIN]
        // (doublePrecision = 1: `float` stands for double in the unit, so prm is an 8-byte-aligned array of doubles)
        const size_t nin = P.inputs.size(), nd = P.dims.empty() ? 1 : P.dims.size();
        off_dim = 8 * nin;
        if (f64_) { off_prm = (off_dim + 4 * nd + 7) / 8 * 8; off_uoff = off_prm + 8 * nin; }
        else { off_prm = off_dim + 4 * nd; off_uoff = (off_prm + 4 * nin + 7) / 8 * 8; }
        off_cap = off_uoff + 8 * nin;                          // float* cap[max(NCAP, 1)]: the planes of the materialized computed arrays
        ctx.assign(off_cap + 8 * (size_t)(G.n_planes > 0 ? G.n_planes : 1), 0);
        for (size_t d = 0; d < P.dims.size(); ++d) { const int v = (int)dimv[d]; memcpy(&ctx[off_dim + 4 * d], &v, 4); }
        for (size_t i = 0; i < nin; ++i) memcpy(&ctx[off_uoff + 8 * i], &uoff[i], 8);
        for (auto& gc : G.computed) {
            long n = 1; for (int d : gc.domain) n *= dimv[(size_t)d];
            ca_nel.push_back(n);
            DeviceBuffer* b = new DeviceBuffer(); ca_buf.push_back(b);
            const size_t es = f64_ ? 8 : 4;
            if (b->alloc(es * (size_t)n * (size_t)gc.planes + 256)) { set_error("%s: out of device memory for a computed array (%d planes of %ld elements)", label.c_str(), gc.planes, n); return; }
            for (int q = 0; q < gc.planes; ++q) { char* pl = (char*)b->ptr + es * (size_t)n * (size_t)q; memcpy(&ctx[off_cap + 8 * (size_t)(gc.plane0 + q)], &pl, 8); }
        }
        long rows = 0;
        for (auto& r : P.residuals) { nel.push_back(elements(r)); jp_off.push_back(rows); if (r.mat_Jp) rows += nel.back() * (long)r.exprs.size(); }
        if (rows && jp.alloc(sizeof(float) * (size_t)rows + 256)) { set_error("%s: out of device memory for the materialized Jp", label.c_str()); return; }
        bool all_jtj = true;
        for (auto& r : P.residuals) all_jtj = all_jtj && r.mat_J && r.mat_JtJ;
        long dense_max = 2048; if (const char* e = env_switch("THALLO_DENSE_JTJ_MAX")) dense_max = atol(e);
        dense_ = all_jtj && n_unk <= dense_max;
        if (dense_ && dense.alloc(sizeof(float) * (size_t)n_unk * (size_t)n_unk)) { set_error("%s: out of device memory for the dense JtJ", label.c_str()); return; }
        sparse_jtj_ = all_jtj && !dense_ && !P.residuals.empty() && n_unk < (1L << 31);
        { const char* e = env_switch("THALLO_ENABLE_DIRECT_SOLVE"); direct_ = dense_ && P.direct_solve && e && e[0] == '1'; }
        if (direct_ && info_.alloc(64)) return;
        jval.assign(P.residuals.size(), nullptr); jcol.assign(P.residuals.size(), nullptr); jdest.assign(P.residuals.size(), nullptr);
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (!P.residuals[ri].mat_J && !P.residuals[ri].mat_JtJ) continue;
            const size_t ent = (size_t)rows_of(ri) * (size_t)G.slots_per_row[ri];
            jval[ri] = new DeviceBuffer(); jcol[ri] = new DeviceBuffer();
            if (jval[ri]->alloc(sizeof(float) * ent + 256) || jcol[ri]->alloc(sizeof(int) * ent + 256)) { set_error("%s: out of device memory for the materialized J of %s", label.c_str(), P.residuals[ri].name.c_str()); return; }
        }
        ok_ = true;
    }
    ~GeneratedPlugin() override { if (mod) (void)hipModuleUnload(mod); for (auto b : ca_buf) delete b; for (auto b : jval) delete b; for (auto b : jcol) delete b; for (auto b : jdest) delete b; for (auto r : inc_runs_) delete r; }
    const char* schedule_name() const override
    {
        if (direct_) return "dense direct solve";
        if (dense_) return "dense [JtJ]p";
        if (sparse_jtj_) return "sparse [[Jt][J]]p";
        bool all = !gather_.empty(), any = false;
        for (size_t i = 0; i < gather_.size(); ++i) { const bool g = gather_[i] || use_inc_[i]; all = all && g; any = any || g; }
        const char* base = all ? "per residual, unknown-wise (gather)" : any ? "per residual, some unknown-wise (gather)" : "per residual";
        if (G.computed.empty()) return base;
        int planes = 0; for (auto& gc : G.computed) planes += gc.planes;
        const_cast<GeneratedPlugin*>(this)->sched_ = std::string(base) + "; " + std::to_string(G.computed.size()) + " computed array(s) materialized (" + std::to_string(planes) + " planes, precompute per GN iteration)";
        return sched_.c_str();
    }
    int prepare(LaunchCtx&) override { sp_ready_ = false; inc_ready_ = false; use_inc_ = want_inc_; ca_valid_ = false; return 0; }
    void unknowns_changed() override { ca_valid_ = false; }      // constant inputs (masks, Sparse maps) may differ from the previous Init

    // symbolic phase of the sparse J^T J (the reference: cusparseXcsrgemmNnz, gauss_newton.t:1404-1412): the rows' unknown indices -> CSR pattern
    // + for every product v[i][a] * v[i][b] its position in the values
    int build_sparse_pattern(hipStream_t s)
    {
        const size_t R = P.residuals.size();
        std::vector<std::vector<int>> hc(R);
        size_t products = 0;
        for (size_t ri = 0; ri < R; ++ri) {
            const size_t ent = (size_t)rows_of(ri) * (size_t)G.slots_per_row[ri];
            hc[ri].resize(ent);
            if (ent && hipMemcpyAsync(hc[ri].data(), jcol[ri]->ptr, ent * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
            products += ent * (size_t)G.slots_per_row[ri];
        }
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        std::vector<uint64_t> keys; keys.reserve(products);
        for (size_t ri = 0; ri < R; ++ri) {
            const int K = G.slots_per_row[ri]; const long rows = rows_of(ri);
            for (long i = 0; i < rows; ++i) {
                const int* c = &hc[ri][(size_t)i * K];
                for (int a = 0; a < K; ++a) { if (c[a] < 0) continue; for (int b = 0; b < K; ++b) if (c[b] >= 0) keys.push_back(((uint64_t)(uint32_t)c[a] << 32) | (uint32_t)c[b]); }
            }
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        if (keys.size() >= (size_t)1 << 31) { set_error("%s: J^T J has too many non-zeros for 32-bit positions", label.c_str()); return -1; }
        sp_nnz = (long)keys.size();
        std::vector<int> rowptr((size_t)n_unk + 1, 0), col(keys.size());
        for (size_t q = 0; q < keys.size(); ++q) { rowptr[(size_t)(keys[q] >> 32) + 1]++; col[q] = (int)(uint32_t)keys[q]; }
        for (long i = 0; i < n_unk; ++i) rowptr[i + 1] += rowptr[i];
        auto up = [&](DeviceBuffer& b, const void* h, size_t bytes) { return b.alloc(bytes + 64) || (bytes && hipMemcpy(b.ptr, h, bytes, hipMemcpyHostToDevice) != hipSuccess) ? -1 : 0; };
        if (up(sp_rowptr, rowptr.data(), rowptr.size() * sizeof(int)) || up(sp_col, col.data(), col.size() * sizeof(int)) || sp_val.alloc(keys.size() * sizeof(float) + 64)) {
            set_error("%s: out of device memory for the sparse JtJ (%ld non-zeros)", label.c_str(), sp_nnz); return -1;
        }
        for (size_t ri = 0; ri < R; ++ri) {
            const int K = G.slots_per_row[ri]; const long rows = rows_of(ri);
            std::vector<int> dest((size_t)rows * K * K, -1);
            for (long i = 0; i < rows; ++i) {
                const int* c = &hc[ri][(size_t)i * K];
                for (int a = 0; a < K; ++a) { if (c[a] < 0) continue; for (int b = 0; b < K; ++b) {
                    if (c[b] < 0) continue;
                    const uint64_t key = ((uint64_t)(uint32_t)c[a] << 32) | (uint32_t)c[b];
                    dest[((size_t)i * K + a) * K + b] = (int)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
                } }
            }
            if (!jdest[ri]) jdest[ri] = new DeviceBuffer();
            if (up(*jdest[ri], dest.data(), dest.size() * sizeof(int))) { set_error("%s: out of device memory for the sparse JtJ", label.c_str()); return -1; }
        }
        sp_ready_ = true;
        return 0;
    }
    bool direct_solve() const override { return direct_; }
    int solve_direct(LaunchCtx& c, SolverVectors& v) override
    {   // delta = (J^T J)^-1 r with r = -J^T F as PCGInit1 left it (the reference: LU + inverse + gemv on the same two operands, gauss_newton.t:1290-1326)
        TimedLaunch t(c, "DirectSolve");
        int rc = thallo_hip_dense_cholesky_solve(n_unk, (float*)dense.ptr, v.r, v.delta, (int*)info_.ptr, c.stream);
        if (rc < 0) return rc;
        int info = 0;
        if (hipMemcpyAsync(&info, info_.ptr, sizeof(int), hipMemcpyDeviceToHost, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess) return -1;
        if (info) { set_error("%s: J^T J is not positive definite (pivot %d): the direct solve needs a full-rank J", label.c_str(), info - 1); return -1; }
        return 0;
    }
    bool ok() const { return ok_; }

    int compile()
    {
        // doublePrecision = 1: the same translation unit with `float` standing for double (values, duals, solver vectors, thallo_float arrays, atomics, shuffles);
        // arrays declared with a fixed float type keep reading floats through f32_t
        const std::string f64_prelude = "#define THALLO_F32_T\ntypedef float f32_t;\n#define float double\n#define sqrtf sqrt\n#define sinf sin\n#define cosf cos\n#define fabsf fabs\n"
                                        "#define powf pow\n#define floorf floor\n#define ceilf ceil\n";
        const std::string src = "#include <hip/hip_runtime.h>\n" + (f64_ ? f64_prelude : std::string()) + G.source;
        if (rtc_build(src, G.has_wide, P.file.c_str(), &mod)) return -1;
        for (auto& k : G.kernels) {
            hipFunction_t f = nullptr;
            if (k.name.empty()) { fn.push_back(nullptr); continue; }       // (no gather form for this residual)
            if (hipModuleGetFunction(&f, mod, k.name.c_str()) != hipSuccess) { set_error("%s: generated kernel %s missing", label.c_str(), k.name.c_str()); return -1; }
            fn.push_back(f);
        }
        for (auto& gc : G.computed) {
            hipFunction_t f = nullptr;
            if (hipModuleGetFunction(&f, mod, gc.kernel.c_str()) != hipSuccess) { set_error("%s: generated kernel %s missing", label.c_str(), gc.kernel.c_str()); return -1; }
            ca_fn.push_back(f);
        }
        for (auto& ir : G.inc) {
            hipFunction_t f = nullptr;
            if (hipModuleGetFunction(&f, mod, ir.uidx.c_str()) != hipSuccess) { set_error("%s: generated kernel %s missing", label.c_str(), ir.uidx.c_str()); return -1; }
            inc_uidx_.push_back(f);
        }
        for (auto& g : G.groups) {
            hipFunction_t fa = nullptr, fb = nullptr;
            if (hipModuleGetFunction(&fa, mod, g.jtj.c_str()) != hipSuccess || hipModuleGetFunction(&fb, mod, g.jtf.c_str()) != hipSuccess) { set_error("%s: generated group kernel %s missing", label.c_str(), g.jtj.c_str()); return -1; }
            grp_jtj.push_back(fa); grp_jtf.push_back(fb);
            hipFunction_t fc = nullptr;
            if (!g.cost.empty() && hipModuleGetFunction(&fc, mod, g.cost.c_str()) != hipSuccess) { set_error("%s: generated group kernel %s missing", label.c_str(), g.cost.c_str()); return -1; }
            grp_cost.push_back(fc);
        }
        return 0;
    }

    const char* name() const override { return label.c_str(); }
    long n_unknowns() const override { return n_unk; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return P.use_preconditioner; }
    int bind(void** p) override
    {
        const std::vector<unsigned char> before = ctx;
        struct Stale { GeneratedPlugin* g; const std::vector<unsigned char>& b; ~Stale() { if (g->ctx != b) g->ca_valid_ = false; } } stale{ this, before };      // (a pointer or a scalar parameter moved)
        bound.assign(P.inputs.size(), nullptr);
        for (size_t i = 0; i < P.inputs.size(); ++i) {
            const dsl::Input& in = P.inputs[i];
            void* v = p[in.slot];
            if (!v) { set_error("%s: problem parameter %d (%s) is NULL", label.c_str(), in.slot, in.name.c_str()); return -1; }
            bound[i] = v;
            if (in.kind == dsl::InputKind::Param) {     // host scalar, re-read every Init / Step
                if (f64_) { const double d = in.fixed_f32 ? (double)*(const float*)v : *(const double*)v; memcpy(&ctx[off_prm + 8 * i], &d, 8); }
                else { const float f = *(const float*)v; memcpy(&ctx[off_prm + 4 * i], &f, 4); }
                v = nullptr;
            }
            memcpy(&ctx[8 * i], &v, 8);
        }
        return 0;
    }
    float* unknown_ptr(int k) override { return (float*)bound[unknown_input[k]]; }
    EnergyPlugin64* f64() override { return f64_ ? this : nullptr; }
    double* unknown_ptr64(int k) override { return (double*)bound[unknown_input[k]]; }
    // a merged group whose members all run gathered: ONE cost launch on shared loads (its partials behind `total`; the members are ticked off in `done`)
    template <class T> int cost_groups(LaunchCtx& c, T* out, int cap, int& total, std::vector<char>& done)
    {
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {
            if (gi >= grp_cost.size() || !grp_cost[gi]) continue;
            bool all = true; for (int ri : G.groups[gi].members) all = all && group_of_[(size_t)ri] == (int)gi;
            if (!all) continue;
            const int g = grid_for(group_nel_[gi], cap * (int)G.groups[gi].members.size());
            T* o = out + total; void* args[] = { ctx.data(), &o };
            if (launch_fn(grp_cost[gi], g, args, c.stream) < 0) return -1;
            total += g;
            for (int ri : G.groups[gi].members) done[(size_t)ri] = 1;
        }
        return 0;
    }
    // ---- doublePrecision = 1: the same launches on double vectors (the unit was compiled with float = double), reference-shaped and unfused
    int cost64(LaunchCtx& c, double* out) override
    {
        if (precompute(c) < 0) return -1;
        TimedLaunch t(c, "computeCost");
        const int cap = THALLO_HIP_MAX_PARTIALS / (int)P.residuals.size();
        int total = 0;
        std::vector<char> done(P.residuals.size(), 0);
        if (cost_groups(c, out, cap, total, done) < 0) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (done[ri]) continue;
            const int g = grid_for(nel[ri], cap);
            double* o = out + total; void* args[] = { ctx.data(), &o };
            const int rc = launch(kernel_of((int)ri, 0), g, args, c.stream); if (rc < 0) return rc;
            total += g;
        }
        return total;
    }
    int pcg_init64(LaunchCtx& c, Vectors64& v, double* aN) override
    {
        if (precompute(c) < 0) return -1;
        TimedLaunch t(c, "PCGInit1");
        hipStream_t s = c.stream;
        const size_t bytes = (size_t)v.n_alloc * sizeof(double);
        if (hipMemsetAsync(v.r, 0, bytes, s) != hipSuccess || hipMemsetAsync(v.pre, 0, bytes, s) != hipSuccess || hipMemsetAsync(v.delta, 0, bytes, s) != hipSuccess) return -1;
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {               // merged gather groups first (one launch per iteration domain)
            double *r = v.r, *pre = v.pre; int mode = 0; void* args[] = { ctx.data(), &r, &pre, &mode };
            const int rc = launch_fn(grp_jtf[gi], grid_for(group_nel_[gi], 4096), args, s); if (rc < 0) return rc;
        }
        if (!inc_ready_ && !inc_runs_.empty() && build_incidence(s)) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {            // evalJTF: r = -J^T F, pre = diag(J^T J)   (thallo.t:3898-3902)
            if (group_of_[ri] >= 0) continue;
            double *r = v.r, *pre = v.pre; void* args[] = { ctx.data(), &r, &pre };
            if (use_inc_[ri]) { const int rc = launch_inc((int)ri, false, &r, &pre, s); if (rc < 0) return rc; continue; }
            const int rc = launch(kernel_of((int)ri, gather_[ri] ? 6 : 1), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
        }
        if (v.diag && hipMemcpyAsync(v.diag, v.pre, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;      // LM: the raw diagonal
        return thallo_hip_f64_init_finish(v.r, v.pre, v.z, v.p, v.n, P.use_preconditioner ? 1 : 0, aN, s);           // PCGInit1_Finish (gauss_newton.t:712-731)
    }
    int apply_jtj64(LaunchCtx& c, Vectors64& v, const double* p, double* Ap, double* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        hipStream_t s = c.stream;
        if (hipMemsetAsync(Ap, 0, (size_t)v.n_alloc * sizeof(double), s) != hipSuccess) return -1;                     // Ap_X:clear() (gauss_newton.t:1633-1635)
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {
            int mode = 0; double* none = nullptr; void* args[] = { ctx.data(), &p, &Ap, &mode, &none };
            const int rc = launch_fn(grp_jtj[gi], grid_for(group_nel_[gi], 4096), args, s); if (rc < 0) return rc;
        }
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (group_of_[ri] >= 0) continue;
            void* args[] = { ctx.data(), &p, &Ap };
            if (use_inc_[ri]) { if (!inc_ready_) return -1; const int rc = launch_inc((int)ri, true, &p, &Ap, s); if (rc < 0) return rc; continue; }
            const int rc = launch(kernel_of((int)ri, gather_[ri] ? 7 : 2), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
        }
        return thallo_hip_f64_dot(p, Ap, n_unk, out, s);              // PCGStep1_Finish: alphaD = p . Ap_X
    }

    int cost(LaunchCtx& c, float* out) override
    {
        if (precompute(c) < 0) return -1;
        TimedLaunch t(c, "computeCost");
        const int cap = THALLO_HIP_MAX_PARTIALS / (int)P.residuals.size();
        int total = 0;
        std::vector<char> done(P.residuals.size(), 0);
        if (cost_groups(c, out, cap, total, done) < 0) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (done[ri]) continue;
            const int g = grid_for(nel[ri], cap);
            float* o = out + total; void* args[] = { ctx.data(), &o };
            const int rc = launch(kernel_of((int)ri, 0), g, args, c.stream); if (rc < 0) return rc;
            total += g;
        }
        return total;
    }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        if (precompute(c) < 0) return -1;
        TimedLaunch t(c, "PCGInit1");
        hipStream_t s = c.stream;
        const size_t bytes = (size_t)v.n_alloc * sizeof(float);
        // (one gather kernel writes every unknown: it STORES r and pre -- only the vectors' padding behind the last unknown is cleared, and nothing is read back)
        const size_t head = whole_ ? (size_t)n_unk * sizeof(float) : 0;
        if ((bytes > head && (hipMemsetAsync((char*)v.r + head, 0, bytes - head, s) != hipSuccess || hipMemsetAsync((char*)v.pre + head, 0, bytes - head, s) != hipSuccess)) ||
            hipMemsetAsync(v.p[cur], 0, bytes, s) != hipSuccess || hipMemsetAsync(v.delta, 0, bytes, s) != hipSuccess) return -1;
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {               // merged gather groups first (one launch per iteration domain)
            float *r = v.r, *pre = v.pre; int mode = whole_ ? 1 : 0; void* args[] = { ctx.data(), &r, &pre, &mode };
            const int rc = launch_fn(grp_jtf[gi], grid_for(group_nel_[gi], 4096), args, s); if (rc < 0) return rc;
        }
        if (!inc_ready_ && !inc_runs_.empty() && build_incidence(s)) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {            // evalJTF: r = -J^T F, pre = diag(J^T J)   (thallo.t:3898-3902)
            if (group_of_[ri] >= 0) continue;
            float *r = v.r, *pre = v.pre; void* args[] = { ctx.data(), &r, &pre };
            if (use_inc_[ri]) { const int rc = launch_inc((int)ri, false, &r, &pre, s); if (rc < 0) return rc; continue; }
            const int rc = launch(kernel_of((int)ri, gather_[ri] ? 6 : 1), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
        }
        if (v.diag && hipMemcpyAsync(v.diag, v.pre, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;      // LM: the raw diagonal
        if (dense_ && hipMemsetAsync(dense.ptr, 0, sizeof(float) * (size_t)n_unk * (size_t)n_unk, s) != hipSuccess) return -1;
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {            // precomputeJ (gauss_newton.t:1019-1025): once per GN iteration
            if (!jval[ri]) continue;
            float* jv = (float*)jval[ri]->ptr; int* jc = (int*)jcol[ri]->ptr; void* args[] = { ctx.data(), &jv, &jc };
            int rc = launch(kernel_of((int)ri, 5), grid_for(nel[ri], 4096), args, s); if (rc < 0) return rc;
            if (dense_ && (rc = thallo_hip_dense_jtj_accumulate(rows_of(ri), G.slots_per_row[ri], jv, jc, n_unk, (float*)dense.ptr, s)) < 0) return rc;
        }
        if (sparse_jtj_) {                             // csrgemm (gauss_newton.t:1394-1441): pattern once per Init, values once per GN iteration
            if (!sp_ready_ && build_sparse_pattern(s)) return -1;
            if (hipMemsetAsync(sp_val.ptr, 0, (size_t)sp_nnz * sizeof(float), s) != hipSuccess) return -1;
            for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
                const int rc = thallo_hip_jtj_scatter(rows_of(ri), G.slots_per_row[ri], (const float*)jval[ri]->ptr, (const int*)jdest[ri]->ptr, (float*)sp_val.ptr, s);
                if (rc < 0) return rc;
            }
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
        if (sparse_jtj_)                               // one SpMV; alphaD = p . Ap rides along (the count of its partials is the return value)
            return thallo_hip_csr_spmv((int)n_unk, (const int*)sp_rowptr.ptr, (const int*)sp_col.ptr, (const float*)sp_val.ptr, p, Ap, p, out, s);
        if (whole_) {          // every residual in ONE gather kernel that writes every unknown: Ap is stored, not accumulated (no clear), and alphaD = p . Ap rides along
            const int g = grid_for(group_nel_[0], THALLO_HIP_MAX_PARTIALS);
            int mode = 3; void* args[] = { ctx.data(), &p, &Ap, &mode, &out };
            const int rc = launch_fn(grp_jtj[0], g, args, s);
            return rc < 0 ? rc : g;
        }
        if (hipMemsetAsync(Ap, 0, (size_t)n_alloc * sizeof(float), s) != hipSuccess) return -1;                         // Ap_X:clear() (gauss_newton.t:1633-1635)
        for (size_t gi = 0; gi < G.groups.size(); ++gi) {
            int mode = 0; float* none = nullptr; void* args[] = { ctx.data(), &p, &Ap, &mode, &none };
            const int rc = launch_fn(grp_jtj[gi], grid_for(group_nel_[gi], 4096), args, s); if (rc < 0) return rc;
        }
        for (size_t ri = 0; ri < P.residuals.size(); ++ri) {
            if (group_of_[ri] >= 0) continue;
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
            } else if (use_inc_[ri]) {
                if (!inc_ready_) return -1;
                const int rc = launch_inc((int)ri, true, &p, &Ap, s); if (rc < 0) return rc;
            } else {
                void* args[] = { ctx.data(), &p, &Ap };
                const int rc = launch(kernel_of((int)ri, gather_[ri] ? 7 : 2), g, args, s); if (rc < 0) return rc;
            }
        }
        return thallo_hip_dot(p, Ap, n_unk, out, s);                  // PCGStep1_Finish: alphaD = p . Ap_X
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    { TimedLaunch t(c, "PCGStep1"); return apply(c, p, Ap, out, thallo_hip_vector_elems(n_unk)); }
    // (measured and dropped, round 5: the single-reduction form for a whole-domain gather kernel -- N, S1, S2 in double behind the store of Ap -- made generated
    //  image_warping's two launches 67 + 103 us against 41 + 77 + 43 for the three of the reference's schedule: the flat launches run at HBM rate either way and the
    //  double-precision sums cost the gather kernel more than the third launch does)
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1");
        return apply(c, v.p[cur ^ 1], v.Ap, out, v.n_alloc);
    }
};

}  // namespace

EnergyPlugin* make_generated_plugin(const char* filename, const unsigned* dims, bool autoschedule, bool f64)
{
    dsl::Problem p; std::string err;
    if (!dsl::run_problem_file(filename, p, err, dims)) { set_error("%s", err.c_str()); return nullptr; }
    GeneratedPlugin* g = new GeneratedPlugin(p, dims, autoschedule, f64);
    if (!g->ok()) { delete g; return nullptr; }
    return g;
}

}  // namespace thallo

// Known-answer test of the wave64 primitives the generated kernels use -- the text they are generated with, compiled here with hipRTC (tests/cuda_unit_tests/
// ballot.t:11, get_peers.t:12, reduce_peers.t:14 restated for 64 lanes).  One wave: out_ballot = max over the lanes of ballot(lane != 0) (the reference: 0xFFFFFFFE
// for 32 lanes); out_peers = sum over the lanes of (the mask of the lanes that share the lane's key lane % 4) & 0xFF (the reference: 255 * 32 / 4); sums4[i] = what
// wave_add(sums4, lane % 4, lane) leaves = the sum of the lanes with lane % 4 == i (the reference: 112 + 8 i).  0 on success.
extern "C" int thallo_hip_wave64_selftest(unsigned long long* out_ballot, unsigned* out_peers, float* sums4)
{
    if (!out_ballot || !out_peers || !sums4) return -(int)hipErrorInvalidValue;
    static const char* KAT = R"KAT(
extern "C" __global__ void k_wave64_kat(unsigned long long* ballot_max, unsigned* peers_sum, float* sums)
{
    const int t = threadIdx.x;
    atomicMax(ballot_max, (unsigned long long)__ballot(t != 0));
    unsigned long long mine = 0ull;
    for (int k0 = 0; k0 < 4; ++k0) { const unsigned long long grp = __ballot((t & 3) == k0); if ((t & 3) == k0) mine = grp; }      // wave_add's grouping step
    atomicAdd(peers_sum, (unsigned)(mine & 0xffull));
    wave_add(sums, (long)(t & 3), (float)t);
}
)KAT";
    const std::string src = std::string("#include <hip/hip_runtime.h>\n#define NIN 1\n#define NDIM 1\n#define NCAP 0\n") + thallo::dsl::generated_prelude() + KAT;
    hipModule_t mod = nullptr; hipFunction_t f = nullptr;
    if (thallo::rtc_build(src, false, "wave64 self-test", &mod)) return -1;
    int rc = -1;
    void* dev = nullptr;
    if (hipModuleGetFunction(&f, mod, "k_wave64_kat") == hipSuccess && hipMalloc(&dev, 64) == hipSuccess && hipMemset(dev, 0, 64) == hipSuccess) {
        unsigned long long* b = (unsigned long long*)dev; unsigned* ps = (unsigned*)((char*)dev + 8); float* sm = (float*)((char*)dev + 16);
        void* args[] = { &b, &ps, &sm };
        unsigned char host[64];
        if (hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
            hipMemcpy(host, dev, 64, hipMemcpyDeviceToHost) == hipSuccess) {
            memcpy(out_ballot, host, 8); memcpy(out_peers, host + 8, 4); memcpy(sums4, host + 16, 16); rc = 0;
        }
    }
    if (rc) thallo::set_error("wave64 self-test: launch failed");
    if (dev) hipFree(dev);
    hipModuleUnload(mod);
    return rc;
}

// The front-end without a device (tests, tooling): what = 0 the declarations as text (dsl::describe), 1 the generated HIP translation unit.
// Returns the length of the text (which is truncated to cap - 1), or -1 with ThalloX_LastError() set.
extern "C" int ThalloX_FrontendTextDims(const char* filename, int what, const unsigned* dims, char* out, int cap);
extern "C" int ThalloX_FrontendText(const char* filename, int what, char* out, int cap) { return ThalloX_FrontendTextDims(filename, what, nullptr, out, cap); }
// ... with the problem's dimensions (as Thallo_ProblemPlan gets them: one entry per declared dimension): files that use Sum are expanded for those sizes
extern "C" int ThalloX_FrontendTextDims(const char* filename, int what, const unsigned* dims, char* out, int cap)
{
    if (!filename || !out || cap < 1) return -1;
    thallo::dsl::Problem p; std::string err, text;
    if (!thallo::dsl::run_problem_file(filename, p, err, dims)) { thallo::set_error("%s", err.c_str()); return -1; }
    if (what == 0) text = thallo::dsl::describe(p);
    else {
        thallo::dsl::Generated g;
        if (!thallo::dsl::generate_source(p, g, err)) { thallo::set_error("%s: %s", filename, err.c_str()); return -1; }
        text = "#include <hip/hip_runtime.h>\n" + g.source;
    }
    snprintf(out, (size_t)cap, "%s", text.c_str());
    return (int)text.size();
}
