// plugin.hpp -- host-side seam between the GN/LM driver and one energy's kernels.
//
// Mirrors the reference's per-residual-group function map `fmap`
// (API/src/thallo.t:444-456; consumed at API/src/gauss_newton.t:677,733,979,997-1095):
// cost / evalJTF(+PCGInit fusion) / applyJTJ(+PCGStep1 fusion) / exclude, plus the parameter
// binding of API/src/util.t:609-643.  Each plugin only forwards to the C-ABI shim in
// include/thallo_hip.h, so the same kernels are reachable from a Terra host layer.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include "../../include/thallo_hip.h"

namespace thallo {

void set_error(const char* fmt, ...);
const char* last_error();
// the value of one of the library's environment switches (A/B comparisons for tests and tools; the one table of them is in solver.cpp), or NULL
const char* env_switch(const char* name);

// Sampled / full per-kernel hipEvent timing (reference: util.t:774-790 at timingLevel >= 2).
class KernelTimer {
public:
    struct Stat { std::string name; long launches = 0; long samples = 0; double total_ms = 0; double sq_ms = 0;
                  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
                  // the kernel's OWN duration, where the sampled launch could carry its events (thallo_hip_launch_events_arm): no dispatch gap in it
                  long ksamples = 0; double ktotal_ms = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> kpending; };
    int period = 0;          // 0 = off, 1 = every launch
    bool invasive = false;   // timingLevel 3: device-sync around timed launches
    ~KernelTimer();
    int  begin(const char* name, hipStream_t s);   // returns slot or -1
    void end(int slot, hipStream_t s);
    void collect();                                 // synchronises, folds pending events into totals
    void reset();
    std::vector<Stat> stats;
private:
    std::map<std::string, int> index_;
    std::vector<hipEvent_t> open_;      // start events of the timed scopes that are open (they nest: step_lm's PCGStep2 scope contains an applyJTJ's PCGStep1)
    std::vector<hipEvent_t> pool_;
    hipEvent_t k0_ = nullptr, k1_ = nullptr;        // the armed pair of the outermost open scope
    hipEvent_t get_event();
};

struct LaunchCtx {
    hipStream_t stream = nullptr;
    KernelTimer* timer = nullptr;
    // LM loop only: device word; non-zero = the PCG loop already ended on the device, an applyJTJ launch may return at once
    // (plugins that take it: shape_from_shading, bundle_adjustment -- the two LM configurations; the others simply compute)
    const unsigned* gate = nullptr;
    // LM loop only: when set, an applyJTJ of a plugin with apply_adds_ctc() returns (J^T J + CtC) p and the partials of p . that -- PCGStep1_Finish
    // (gauss_newton.t:774-787) folded into the apply, one launch less per PCG iteration
    const float* lm_ctc = nullptr;
    float *lm_defer_aD_word = nullptr, *lm_defer_bN_word = nullptr;      // pcg_iter_lm (plugins with lm_iter_defers_finish()): non-NULL = `aD` holds the PARTIALS of iteration k-1, which
                                        // this iteration's first launch finishes (words, zeta test) before it updates
    int lm_q_in = 0, lm_q_out = 0;      // ... and the words of the LM state the iteration's own finish (fin.tickets set) reads Q0 from / leaves Q1 in
    float* lm_reset_bn_word = nullptr;  // pcg_iter_lm: non-NULL = the iteration behind a residual reset (delta and r are already this iteration's, betaN_{k-1} arrives as
                                        // partials and is left here as a word; plugins with lm_iter_after_reset())
};

// RAII bracket used by plugins / driver around each shim call.
struct TimedLaunch {
    LaunchCtx& c; int slot;
    TimedLaunch(LaunchCtx& ctx, const char* name) : c(ctx), slot(ctx.timer ? ctx.timer->begin(name, ctx.stream) : -1) {}
    ~TimedLaunch() { if (slot >= 0) c.timer->end(slot, c.stream); }
};

// Solver vectors of PlanData (gauss_newton.t:282-323), flat layout, each `n_alloc` floats.
struct SolverVectors {
    long n = 0, n_alloc = 0;
    float *delta = nullptr, *r = nullptr, *z = nullptr, *Ap = nullptr, *pre = nullptr;
    float *p[2] = { nullptr, nullptr };          // ping-pong: the fused step reads p[cur], writes p[cur^1]
    float *b = nullptr, *Adelta = nullptr, *CtC = nullptr, *SSq = nullptr, *prevX = nullptr, *diag = nullptr;   // LM only
    // one-kernel-per-iteration schedule (plugins with one_kernel_iteration()): r and Ap ping-pong like p; per-workgroup double sums
    float *r2 = nullptr, *Ap2 = nullptr;
    double* s12 = nullptr;                       // 3 * THALLO_HIP_MAX_PARTIALS doubles (N, S1, S2 per workgroup)
    double* s12b = nullptr;                      // its ping-pong partner (deferred finish: a launch reads the previous iteration's sums while it writes its own)
    double* s12buf(int i) { return i ? s12b : s12; }
    unsigned* fin_tickets = nullptr;             // THALLO_HIP_FIN_TICKET_WORDS zeroed words (in-kernel finish of the iteration's scalars)
    float* rbuf(int i) { return i ? r2 : r; }
    float* Abuf(int i) { return i ? Ap2 : Ap; }
};

struct UnknownImage { int param_index; long n_floats; };

// doublePrecision = 1 (precision.t:3-6): what a plugin that also runs on double vectors offers -- the generated plugins, compiled with thallo_float = double.
// The driver is solver_f64.cpp's reference-shaped loop; the hand-written gfx950 kernels are float only.
struct Vectors64 { long n = 0, n_alloc = 0; double *delta = nullptr, *r = nullptr, *z = nullptr, *Ap = nullptr, *pre = nullptr, *p = nullptr;
                   double* diag = nullptr; };      // diag != NULL (LM): pcg_init64 also leaves the RAW diagonal of J^T J there
class EnergyPlugin64 {
public:
    virtual ~EnergyPlugin64() {}
    virtual int cost64(struct LaunchCtx&, double* partials_out) = 0;                         // returns the partial count or < 0
    virtual int pcg_init64(struct LaunchCtx&, Vectors64&, double* alphaN_partials) = 0;       // r = -J^T F, pre = M^-1, z = p = M^-1 r, delta = 0; partials of r . z
    virtual int apply_jtj64(struct LaunchCtx&, Vectors64&, const double* p, double* Ap, double* alphaD_partials) = 0;      // Ap = J^T J p; partials of p . Ap
    virtual double* unknown_ptr64(int k) = 0;
};

class EnergyPlugin {
public:
    virtual ~EnergyPlugin() {}
    virtual const char* name() const = 0;
    virtual EnergyPlugin64* f64() { return nullptr; }                    // non-NULL: built for doublePrecision = 1 (then ONLY this interface is used)
    virtual const char* schedule_name() const { return "matrix-free"; }    // which J^T J p schedule this plugin runs (ThalloX_PlanScheduleName)
    virtual long n_unknowns() const = 0;
    virtual const std::vector<UnknownImage>& unknown_images() const = 0;   // declaration order
    virtual bool use_preconditioner() const = 0;
    // util.t:609-643: read the void** (device pointers for arrays, host pointers for Params).  Called at
    // Init and at every Step, so callers may change weights / buffers between steps.
    virtual int bind(void** params) = 0;
    // once per bind at Init (e.g. graph incidence lists); default nothing
    virtual int prepare(LaunchCtx&) { return 0; }
    // fmap.cost -> partials; returns the partial count or <0
    virtual int cost(LaunchCtx&, float* cost_out) = 0;
    // PCGInit1 (+_Finish): r, pre (already inverted), z = pre*r, p[cur]=0, delta=0, alphaN partials;
    // when v.diag != NULL (LM) also the raw diag(J^T J) into v.diag
    virtual int pcg_init(LaunchCtx&, SolverVectors&, int cur, float* alphaN_out) = 0;
    // plain PCGStep1: Ap = J^T J p, partials of p.Ap (LM branch, computeAdelta, model cost)
    virtual int apply_jtj(LaunchCtx&, const float* p, float* Ap, float* alphaD_out) = 0;
    virtual bool apply_adds_ctc() const { return false; }      // honours LaunchCtx::lm_ctc
    // LM on one GPU: PCGStep3 folded into the apply -- p_out = z + beta p_in (beta = bN_prev / aN_prev, 0 when first), Ap = (J^T J + lm_ctc) p_out, alphaD partials;
    // p_out is written over the owned rows only and differs from p_in.  Only asked for when apply_adds_ctc() and LaunchCtx::lm_ctc is set.
    // LM on one GPU, ONE launch per PCG iteration (round 3; shape_from_shading's marching kernel): vector update + PCGStep3 + (J^T J + CtC) p + alphaD, {N, S1, S2} and the
    // {U, T1, T2} of q's expansion in alpha; the launch's last workgroup finishes alphaD_k, betaN_k, q_{k+1} and the zeta test (thallo_hip_sfs_pcg_iter_lm).
    // r, Ap and p ping-pong as in pcg_iter; SolverVectors::CtC, b, s12b are read / written.
    virtual bool lm_one_kernel() const { return false; }
    virtual bool lm_iter_defers_finish() const { return false; }    // pcg_iter_lm honours LaunchCtx::lm_defer_* / lm_q_* and fin.tickets == NULL (partials only)
    virtual bool lm_iter_after_reset() const { return false; }      // pcg_iter_lm honours LaunchCtx::lm_reset_bn_word and lm_reset_residual exists: the one-reduction LM loop may run past residual_reset_period
    // r = b - (J^T J + CtC) delta (v.CtC, v.b), partials of r . M^-1 r (v.pre) to betaN_out; returns their number
    virtual int lm_reset_residual(LaunchCtx&, SolverVectors&, float* /*betaN_out*/) { return -1; }
    virtual bool lm_one_kernel_slab() const { return false; }       // ... pcg_iter_lm also serves a row slab (ghost rows of r and p kept current; fin without tickets: partials only)
    virtual int pcg_iter_lm(LaunchCtx&, SolverVectors&, int /*cur*/, bool /*first*/, thallo_sum_t /*aN_prev*/, thallo_sum_t /*aD_prev*/, thallo_sum_t /*bN_prev*/, float* /*alphaD_out*/,
                            const thallo_fin_t&, float* /*lm_state*/, int /*k*/, float /*q_tolerance*/) { return -1; }
    // LM on one GPU, round 6 (shape_from_shading on packed planes): PCGFinalizeDiagonal rides in PCGInit1's pass -- r, delta = 0, p[cur] = 0 and, from the raw diagonal formed in the
    // same launch, CtC, pre, b = r, z = pre r, SSq (written when save_ssq), partials of r . z: pcg_init + thallo_hip_lm_finalize_diagonal in one launch
    virtual bool init_folds_lm_diagonal() const { return false; }
    virtual int pcg_init_lm(LaunchCtx&, SolverVectors&, int /*cur*/, float /*radius*/, float /*min_lm_diagonal*/, float /*max_lm_diagonal*/, int /*save_ssq*/, float* /*alphaN_out*/) { return -1; }
    // ... and the step's model cost in ONE launch behind the one-launch LM loop: v.Adelta = v.delta + alpha_kl p_kl (the update of delta the loop owes; kl from lm_state as
    // thallo_hip_lm_owed_delta), partials of that . J^T J that (dJJd_out) and of that . b (db_out); the driver then swaps v.delta and v.Adelta.  Returns the partial count of each.
    virtual bool lm_model_cost_one_launch() const { return false; }
    // update_unknowns: the launch also saves the unknowns to v.prevX and adds the new delta to them (savePreviousUnknowns + PCGLinearUpdate; single-image plugins)
    virtual int lm_model_cost(LaunchCtx&, SolverVectors&, const float* /*alphaN_words*/, const float* /*alphaD_words*/, int /*word_stride*/, const float* /*lm_state*/, int /*L*/,
                              float* /*dJJd_out*/, float* /*db_out*/, bool /*update_unknowns*/) { return -1; }
    virtual bool apply_folds_pupdate() const { return false; }
    virtual int apply_jtj_pupdate(LaunchCtx&, const float* /*z*/, const float* /*p_in*/, float* /*p_out*/, float* /*Ap*/, float* /*alphaD_out*/, bool /*first*/,
                                  thallo_sum_t /*aN_prev*/, thallo_sum_t /*bN_prev*/) { return -1; }
    // fused PCGStep3(k-1) + delta update(k-1) + PCGStep1(k): reads p[cur], writes p[cur^1], Ap, alphaD partials
    virtual int pcg_step1(LaunchCtx&, SolverVectors&, int cur, bool first,
                          thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev, float* alphaD_out) = 0;
    // Plugins whose fused PCGStep1 can defer every other delta update (thallo_hip.h THALLO_IW_STEP1_MODE) return true and
    // implement pcg_step1_mode; the driver then finishes the GN step with up to two pending alpha*p terms.
    virtual bool batches_delta() const { return false; }
    virtual bool takes_any_p_plane() const { return false; }
    // iterations k0 .. k1-1 of the one-kernel GN loop in ONE launch (image_warping's persistent marching loop): planes[k % n] as in the ring schedule, the plan's
    // slot layout passed through (thallo_hip_iw_pcg_march_persist).  Returns the workgroup count or < 0
    virtual bool persist_ok() const { return false; }
    virtual int pcg_persist(struct LaunchCtx&, SolverVectors&, float* const* planes, int n_planes, int k0, int k1, float* parts, int slots, int B, int nb_prev, thallo_sum_t alphaN_prev) { return -1; }            // pcg_iter / pcg_iter_deferred read p[cur] and write p[cur ^ 1] wherever those point, and nothing else of p (the ring of p planes)
    virtual int pcg_step1_mode(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev,
                               thallo_sum_t, thallo_sum_t, float* alphaD_out)
    { return pcg_step1(c, v, cur, (mode & 1) != 0, aN_prev, aD_prev, bN_prev, alphaD_out); }
    // Single-reduction form for gather energies (thallo_hip.h "single-reduction PCG form"): applyJTJ that also writes the three double
    // sums over the unknowns (r = v.r, M^-1 = v.pre or 1) into v.s12; the driver then runs pcg_update + this + scalars_finish per iteration.
    virtual bool apply_returns_sums() const { return false; }
    // fin.tickets != NULL: the two scalar words of the iteration are finished too (by the kernel's last workgroup; no PCGScalars launch)
    virtual int apply_jtj_sums(LaunchCtx&, SolverVectors&, const float* /*p*/, float* /*Ap*/, float* /*alphaD_out*/, const thallo_fin_t& /*fin*/) { return -1; }
    // One kernel per PCG iteration (thallo_hip.h thallo_hip_iw_pcg_iter): reads r/Ap/p[cur], writes r/Ap/p[cur^1], alphaD partials
    // to alphaD_out and the double sums to v.s12; pcg_iter_finish turns them into the two scalar words of the iteration.
    virtual bool one_kernel_iteration() const { return false; }
    // across ranks (solver_dist.cpp): the flat single-image form (applyJTJ with the three sums + one all-gather per iteration) rather than image_warping's
    // one-kernel slab form; by default whatever has no one-kernel iteration
    virtual bool dist_flat_form() const { return !one_kernel_iteration(); }
    // flat slab form, round 3: pcg_iter also serves a row slab (it keeps r and p current on the slab's ghost rows; their A p comes with the exchange) -- then a slab's
    // PCG iteration is pcg_iter + ONE exchange instead of pcg_update + applyJTJ + exchange
    virtual bool one_kernel_slab() const { return false; }
    // aD_word / bN_word non-NULL: the kernel finishes the two scalars itself (no pcg_iter_finish launch)
    virtual int pcg_iter(LaunchCtx&, SolverVectors&, int /*cur*/, int /*mode*/, thallo_sum_t, thallo_sum_t, thallo_sum_t, thallo_sum_t, thallo_sum_t, float* /*alphaD_out*/,
                         float* /*aD_word*/, float* /*bN_word*/) { return -1; }
    virtual int pcg_iter_finish(LaunchCtx&, SolverVectors&, const float* /*alphaD_partials*/, int /*count*/, thallo_sum_t /*alphaN*/, float* /*alphaD_word*/, float* /*betaN_word*/) { return -1; }
    // Deferred finish (thallo_hip.h thallo_prev_t): iteration k's launch adds up iteration k-1's partials itself (prev; ignored for the first iteration
    // of a GN step) and writes its own {N, S1, S2} partials to s12_out (!= prev.s12_partials); pcg_iter_finish_from finishes the LAST iteration.
    virtual bool iter_defers_finish() const { return false; }
    virtual int pcg_iter_deferred(LaunchCtx&, SolverVectors&, int /*cur*/, int /*mode*/, thallo_sum_t /*alphaN_prev*/, thallo_sum_t /*alphaN_prev2*/, thallo_sum_t /*alphaD_prev2*/,
                                  const thallo_prev_t& /*prev*/, float* /*alphaD_out*/, double* /*s12_out*/) { return -1; }
    virtual int pcg_iter_finish_from(LaunchCtx&, const float* /*alphaD_partials*/, const double* /*s12_partials*/, int /*count*/, thallo_sum_t /*alphaN*/,
                                     float* /*alphaD_word*/, float* /*betaN_word*/) { return -1; }
    // The whole PCG loop of a GN step in one launch (thallo_hip_iw_pcg_resident): after pcg_init, L iterations from r = v.rbuf(0), p = v.p[0]; leaves
    // r / Ap / p in buffers (L & 1), delta without its last term, words[2k] = alphaD_k, words[2k + 1] = betaN_k.  resident_ok(): this plan's shape fits.
    virtual bool resident_ok() const { return false; }
    virtual int  pcg_resident(LaunchCtx&, SolverVectors&, int /*L*/, thallo_sum_t /*alphaN0*/, float* /*words*/) { return -1; }
    virtual bool resident_updates_unknowns() const { return false; }                            // PCGLinearUpdate rides in the resident launch (the driver skips its own, and calls unknowns_written())
    // ... an LM step's PCG loop, zeta test, owed delta update, model cost (partials of delta . J^T J delta, delta . b), savePreviousUnknowns and PCGLinearUpdate in one launch
    // (thallo_hip_sfs_pcg_resident_lm): behind pcg_init_lm and a reset state; L within one residual-reset period
    virtual bool resident_lm_ok() const { return false; }
    virtual int  pcg_resident_lm(LaunchCtx&, SolverVectors&, int /*L*/, thallo_sum_t /*alphaN0*/, float* /*words*/, float* /*lm_state*/, float /*q_tol*/, float* /*dJJd_out*/, float* /*db_out*/) { return -1; }
    virtual int  resident_status(LaunchCtx&, int /*clear*/, unsigned* /*pm*/) { return 0; }      // 1: a bounded wait inside the kernel ran out
    virtual void resident_disable() {}                                                          // ... after which the plan stays on one launch per PCG iteration
    // ... for one rank's row slab on the device-side transport (thallo_hip_iw_pcg_resident_dist): boundary rows of A p straight into the neighbours' ghost areas
    // (ghost_off bytes into every rank's mailbox block), the scalars through the mailbox slots slot0 + 7 k ..
    // Row slabs on the device-side transport with the DEFERRED cross-rank finish (round 4): launch k stores this rank's partials only; launch k + 1's designated wave adds
    // them up, trades the rank sums through mailbox slots prev_slot0 .. + 6, leaves the two words of iteration k behind and publishes them to the launch's other waves
    // through `gs`, all while the first rows load; pcg_iter_dist_finish does the same for a GN step's last iteration
    virtual bool dist_defers_finish() const { return false; }
    virtual int  pcg_iter_dist_deferred(LaunchCtx&, SolverVectors&, int /*cur*/, int /*mode*/, thallo_sum_t /*aN*/, thallo_sum_t /*aD*/, thallo_sum_t /*bN*/, thallo_sum_t /*aN2*/, thallo_sum_t /*aD2*/,
                                        const thallo_prev_t& /*prev*/, int /*prev_slot0*/, unsigned long long* /*gs*/, const thallo_dist_t&, float* /*alphaD_out*/, double* /*s12_out*/) { return -1; }
    virtual int  pcg_iter_dist_finish(LaunchCtx&, const thallo_prev_t& /*prev*/, int /*prev_slot0*/, thallo_sum_t /*aN*/, const thallo_dist_t&, unsigned long long* /*gs*/) { return -1; }
    virtual bool dist_batches_delta() const { return false; }          // the device-side transport's iteration kernel has the "apply two delta updates" form too
    virtual bool resident_slab_ok() const { return false; }
    virtual long resident_ghost_bytes() const { return 0; }
    virtual int  pcg_resident_dist(LaunchCtx&, SolverVectors&, int /*L*/, thallo_sum_t /*alphaN0*/, float* /*words*/, const thallo_dist_t&, long /*ghost_off*/, int /*slot0*/) { return -1; }
    // PCGStep2 (r -= alpha Ap, z = M^-1 r, betaN partials); default = the energy-independent flat kernel
    virtual int pcg_step2(LaunchCtx& c, SolverVectors& v, thallo_sum_t aN, thallo_sum_t aD, float* betaN_out)
    {
        TimedLaunch t(c, "PCGStep2");
        return thallo_hip_pcg_step2(v.r, v.Ap, use_preconditioner() ? v.pre : nullptr, v.z, v.n, aN, aD, betaN_out, c.stream);
    }
    // pointer to unknown image k as currently bound
    virtual float* unknown_ptr(int k) = 0;
    // the driver (or an exchange) has just written the unknowns: whatever the plugin derived from them (shape_from_shading's precomputed planes) is stale
    virtual void unknowns_changed() {}
    // ... and the other direction: the SOLVER wrote the unknowns through unknown_ptr (PCGLinearUpdate, an LM revert).  A plugin that keeps the unknowns in its own
    // numbering publishes them to the caller's arrays here; everybody else treats it like unknowns_changed
    virtual void unknowns_written() { unknowns_changed(); }
    // the plan is one shard of several whose shared block of unknowns is all-reduced between ranks: a plugin must keep the caller's numbering there (bundle adjustment's point order)
    virtual void forbid_renumbering() {}
    // Direct solve of the normal equations instead of PCG (gauss_newton.t:1280-1328, 1612-1613): after pcg_init, delta = (J^T J)^-1 r
    virtual bool direct_solve() const { return false; }
    virtual int  solve_direct(LaunchCtx&, SolverVectors&) { return -1; }
    // ---- one row slab of a multi-GPU run (solver_dist.cpp): image-stencil plugins whose kernels take an owned-row range
    virtual bool supports_row_slabs() const { return false; }
    virtual int  set_row_slab(int /*row0*/, int /*row1*/) { return -1; }
    virtual int  slab_ghost_rows() const { return 1; }         // stencil radius = ghost rows towards each neighbour
    virtual int  set_slab_global(int /*global_row0*/, int /*global_rows*/) { return 0; }      // energies whose expressions use global pixel coordinates
    virtual int  slab_width() const { return 0; }
    virtual bool slab_grid_ok() const { return false; }       // after prepare(): the precondition of the one-kernel slab schedule holds on this rank
    virtual unsigned char* slab_flags() { return nullptr; }    // per-pixel byte plane (written by pcg_init) whose ghost rows come from their owner each GN step
    // ---- range partition (graph domains): every rank holds the whole problem and FULL-length vectors, runs the gather kernels for its own contiguous
    // range of units (vertices) only; range_units() = how many units there are (each unknown image has n_floats / units floats per unit)
    virtual long range_units() const { return 0; }
    virtual int  set_owned_range(long /*u0*/, long /*u1*/) { return -1; }
    // ---- shard form (bundle adjustment: camera shards): the unknowns [shared_block_offset, + shared_block_floats) (the points) are held by every rank
    // in full; a rank's J^T F / diag / J^T J p there are partial sums over ITS residuals and are all-reduced; the first shared_split_slots() partial
    // slots of apply_jtj_sums belong to the rank-private unknowns (the cameras)
    virtual long shared_block_offset() const { return 0; }
    virtual long shared_block_floats() const { return 0; }
    virtual int  shared_split_slots() const { return 0; }
    // pcg_iter that also stores its boundary rows of Ap_out into the neighbours' ghost rows and whose last workgroup runs the mailbox exchange
    virtual int  pcg_iter_dist(LaunchCtx&, SolverVectors&, int /*cur*/, int /*mode*/, thallo_sum_t, thallo_sum_t, thallo_sum_t, thallo_sum_t, thallo_sum_t,
                               const thallo_dist_t&, float* /*alphaD_out*/, int /*slot0*/, float* /*aD_word*/, float* /*bN_word*/) { return -1; }
};

struct DeviceBuffer {
    void* ptr = nullptr; size_t bytes = 0;
    int alloc(size_t n);
    void release();
    ~DeviceBuffer() { release(); }
};

// .t front-end: recognise a bundled energy (frontend.cpp)
struct ProblemSpec {
    std::string file;
    std::string energy;            // plugin id, "" if unknown
    std::map<std::string, double> constants;   // literals lifted from the file (e.g. w_fit)
    int  n_dims = 0;
    bool verified_body = false;    // body hash matches a known bundled file
    unsigned long long body_hash = 0;
    std::string diagnostic;
};
bool parse_problem_file(const char* filename, ProblemSpec& out);
bool unit_matches_bundled(const char* filename, const std::string& energy);      // frontend.cpp

EnergyPlugin* make_plugin(const ProblemSpec& spec, const unsigned* dims);
// the mini front-end (dsl.hpp): run the .t, generate its residual-wise kernels, compile them with hipRTC; NULL + set_error on failure
// autoschedule (Thallo_InitializationParameters::useAutoscheduler): residuals that have an unknown-wise (gather) lowering use it unless the file says otherwise
EnergyPlugin* make_generated_plugin(const char* filename, const unsigned* dims, bool autoschedule, bool f64 = false);

}  // namespace thallo
