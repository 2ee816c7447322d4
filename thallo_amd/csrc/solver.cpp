// solver.cpp -- see solver.hpp.
#include "solver.hpp"
#include <stdlib.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace thallo {

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "[thallo] error: %s\n", g_err);
}
const char* last_error() { return g_err; }

#define HIP_OK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) set_error("%s failed: %s", #call, hipGetErrorString(e__)); } while (0)

// ------------------------------------------------------------------ KernelTimer
hipEvent_t KernelTimer::get_event()
{
    if (!pool_.empty()) { hipEvent_t e = pool_.back(); pool_.pop_back(); return e; }
    hipEvent_t e = nullptr; HIP_OK(hipEventCreate(&e)); return e;
}
int KernelTimer::begin(const char* name, hipStream_t s)
{
    auto it = index_.find(name);
    int idx;
    if (it == index_.end()) { idx = (int)stats.size(); stats.emplace_back(); stats.back().name = name; index_[name] = idx; }
    else idx = it->second;
    Stat& st = stats[idx];
    const long n = st.launches++;
    if (period <= 0 || (n % period) != 0) return -1;
    if (invasive) hipDeviceSynchronize();
    if (open_.empty()) { k0_ = get_event(); k1_ = get_event(); thallo_hip_launch_events_arm(k0_, k1_); }
    open_.push_back(get_event());
    hipEventRecord(open_.back(), s);
    return idx;
}
void KernelTimer::end(int slot, hipStream_t s)
{
    if (invasive) hipDeviceSynchronize();
    if (open_.empty()) return;              // (never: every sampled begin() is paired with one end())
    hipEvent_t e = get_event();
    hipEventRecord(e, s);
    // scopes nest (a timed plan-level step around a plugin's own timed launch): the innermost open scope ends first.  A single "current start" slot
    // left the outer scope with a null start event -- hipEventElapsedTime then failed with hipErrorInvalidResourceHandle, and that stale error was
    // reported by whatever launch was checked next (seen as "point order failed" in a later bundle adjustment plan).
    stats[slot].pending.emplace_back(open_.back(), e);
    open_.pop_back();
    if (open_.empty() && k0_) {
        if (thallo_hip_launch_events_take()) stats[slot].kpending.emplace_back(k0_, k1_); else { pool_.push_back(k0_); pool_.push_back(k1_); }
        k0_ = k1_ = nullptr;
    }
}
void KernelTimer::collect()
{
    for (auto& st : stats) {
        for (auto& pr : st.pending) {
            hipEventSynchronize(pr.second);
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { st.samples++; st.total_ms += ms; st.sq_ms += (double)ms * ms; }
            pool_.push_back(pr.first); pool_.push_back(pr.second);
        }
        st.pending.clear();
        for (auto& pr : st.kpending) {
            hipEventSynchronize(pr.second);
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { st.ksamples++; st.ktotal_ms += ms; }
            pool_.push_back(pr.first); pool_.push_back(pr.second);
        }
        st.kpending.clear();
    }
}
void KernelTimer::reset() { collect(); for (auto& st : stats) { st.launches = st.samples = st.ksamples = 0; st.total_ms = st.sq_ms = st.ktotal_ms = 0; } }
KernelTimer::~KernelTimer()
{
    for (auto& st : stats) for (auto& pr : st.pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto& st : stats) for (auto& pr : st.kpending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    if (k0_) { hipEventDestroy(k0_); hipEventDestroy(k1_); }
    for (auto e : pool_) hipEventDestroy(e);
    for (auto e : open_) hipEventDestroy(e);
}

// ------------------------------------------------------------------ CoarseTimer
int CoarseTimer::start(const char* name, hipStream_t s)
{
    if (!enabled) return -1;
    Info i; i.name = name; i.start = i.end = nullptr;
    HIP_OK(hipEventCreate(&i.start)); HIP_OK(hipEventCreate(&i.end));
    hipEventRecord(i.start, s);
    events.push_back(i);
    return (int)events.size() - 1;
}
void CoarseTimer::stop(int idx, hipStream_t s) { if (idx >= 0) hipEventRecord(events[idx].end, s); }
void CoarseTimer::cleanup()
{
    for (auto& i : events) { if (i.start) hipEventDestroy(i.start); if (i.end) hipEventDestroy(i.end); }
    events.clear();
}

namespace {
struct Running { unsigned count = 0; double sum = 0, sq = 0, mn = 0, mx = 0; };
void upd(Running& r, double v)
{   // util.t:428-444 RunningStats
    if (!r.count) { r.mn = r.mx = v; } else { r.mn = std::fmin(r.mn, v); r.mx = std::fmax(r.mx, v); }
    r.count++; r.sum += v; r.sq += v * v;
}
Thallo_PerformanceEntry entry(const Running& r)
{   // util.t:495-508 computeSummary
    Thallo_PerformanceEntry e; memset(&e, 0, sizeof(e));
    if (!r.count) return e;
    const double mean = r.sum / r.count, var = r.sq / r.count - mean * mean;
    e.count = r.count; e.meanMS = mean; e.minMS = r.mn; e.maxMS = r.mx; e.stddevMS = std::sqrt(std::fabs(var));
    return e;
}
}  // namespace

void CoarseTimer::evaluate(Thallo_PerformanceSummary* out, bool print_table, KernelTimer* kt)
{   // util.t:516-593
    hipDeviceSynchronize();
    std::vector<std::string> names; std::vector<Running> stats;
    for (auto& i : events) {
        float ms = 0.0f;
        hipEventSynchronize(i.end);
        if (hipEventElapsedTime(&ms, i.start, i.end) != hipSuccess) continue;
        size_t k = 0; for (; k < names.size(); ++k) if (names[k] == i.name) break;
        if (k == names.size()) { names.push_back(i.name); stats.emplace_back(); }
        upd(stats[k], ms);
    }
    auto get = [&](const char* n) { for (size_t k = 0; k < names.size(); ++k) if (names[k] == n) return entry(stats[k]); return entry(Running()); };
    out->total = get("Total");
    out->nonlinearIteration = get("Nonlinear Iteration");
    out->nonlinearSetup = get("Nonlinear Setup");
    out->linearSolve = get("Linear Solve");
    out->nonlinearResolve = get("Nonlinear Finish");
    if (print_table) {
        printf("\n|        Kernel        |   Count  | Total(ms) | Average(ms) | Std. Dev(ms) |\n");
        printf("|----------------------|----------|-----------|-------------|--------------|\n");
        for (size_t k = 0; k < names.size(); ++k) {
            const Thallo_PerformanceEntry e = entry(stats[k]);
            printf("| %-20s |   %4u   | %8.3f  |   %8.4f  |   %8.4f   |\n", names[k].c_str(), e.count, stats[k].sum, e.meanMS, e.stddevMS);
        }
        if (kt) {
            kt->collect();
            for (auto& st : kt->stats) if (st.samples) {
                const double mean = st.total_ms / st.samples;
                const double sd = std::sqrt(std::fabs(st.sq_ms / st.samples - mean * mean));
                printf("| %-20s |   %4ld   | %8.3f  |   %8.4f  |   %8.4f   |\n", st.name.c_str(), st.samples, st.total_ms, mean, sd);
            }
        }
        printf("|--------------------------------------------------------------------------|\n");
    }
}

// ------------------------------------------------------------------ Plan
Plan::Plan(EnergyPlugin* pl, const Thallo_InitializationParameters& ip_, bool lm, unsigned* dims_)
    : plugin(pl), ip(ip_), dims(dims_), lm_(lm)
{
    memset(&summary, 0, sizeof(summary));
    v_.n = plugin->n_unknowns();
    v_.n_alloc = thallo_hip_vector_elems(v_.n);
    float** vecs[] = { &v_.delta, &v_.r, &v_.z, &v_.Ap, &v_.pre, &v_.p[0], &v_.p[1] };
    for (float** vp : vecs) {     // zero-filled like UnknownType:initGPU (thallo.t:1104-1126)
        DeviceBuffer* b = new DeviceBuffer(); bufs_.push_back(b);
        if (b->alloc((size_t)v_.n_alloc * sizeof(float))) return;
        *vp = (float*)b->ptr;
    }
    if (scratch_.alloc(64 * sizeof(float))) return;
    ctx.timer = &ktimer;
    timer_.enabled = ip.timingLevel >= 1;
    if (ip.timingLevel >= 2) ktimer.period = 1;
    if (ip.timingLevel >= 3) ktimer.invasive = true;
    if (ensure_slots(sp.lIterations)) return;
    ok_ = true;
}

Plan::~Plan()
{
    hipDeviceSynchronize();
    if (host_words_) { hipHostFree(host_words_); host_words_ = nullptr; }
    for (hipEvent_t e : aux_events_) hipEventDestroy(e);
    if (aux_) hipStreamDestroy(aux_);
    dist_release();
    if (rccl_) { rccl_comm_destroy(rccl_); rccl_ = nullptr; }
    for (auto b : bufs_) delete b;
    delete plugin;
}

int Plan::ensure_slots(int L)
{
    const int need = 2 * L + 6;
    if (need <= parts_slots_) return 0;
    hipDeviceSynchronize();
    // `need` partial slots followed by `need` scalar words (scal(j)); 4 KB per slot, so lIterations = 4000
    // (examples/embedded_mesh_deformation) costs 32 MB
    if (parts_.alloc(((size_t)need * THALLO_HIP_MAX_PARTIALS + (size_t)need + 64) * sizeof(float))) { set_error("out of device memory for %d reduction slots (lIterations too large?)", need); return -1; }
    parts_slots_ = need; nb_.assign(need, 1); fin_.assign(need, 0);
    read_ab_switches();
    return 0;
}

// ------------------------------------------------------------------ A/B switches
// Every environment switch of the library, in this one function (the only getenv of a switch in thallo_amd/csrc; dsl_plugin.cpp reads ROCM_PATH / HIP_PATH for hipRTC).
// Eight names (README.md has the table).  Seven are what a user may want to set; the eighth, THALLO_AB, is a comma-separated list of key=value pairs for the A/B
// alternatives that exist because a test or a tool still compares them with the default (round 5: they were seven names of their own) -- callers inside the
// library keep asking for them by their old names.  What lost its comparison in rounds 1-2 is gone (the blocking zeta test, the one-kernel bundle-adjustment gather,
// THALLO_FINISH_SUMS / EXPANDED / DEFER_FINISH / LM_FOLD_CTC / LM_ZETA_IN_STEP2 / DIST_MEM / FRONTEND_DUMP); round 5 replaced THALLO_BATCH_DELTA by THALLO_DELTA_PLANES.
const char* env_switch(const char* name)
{
    static const char* const known[] = {
        "THALLO_RESIDENT",            // 0: image_warping / ARAP / shape_from_shading run one launch per PCG iteration even where the whole PCG loop fits one resident launch; 2: also bundle adjustment's resident
                                      //    PCG loop (bit-identical, measured slower than its three launches per iteration: opt-in)
        "THALLO_MARCH",               // 0: the LDS-tiled one-kernel iteration everywhere; 2: the marching kernel at every size (default: from 0.4 Mpixel up); 3: the marching kernel with the
                                      //    stored A p plane (round 2/3); 4 = 2 + 3
        "THALLO_DELTA_PLANES",        // how often the one-kernel GN loop touches delta.  0: delta += alpha p every iteration (the reference's order of work); 1: every other iteration
                                      //    (round 4); N >= 2: a ring of N p planes, delta updated per half ring next to the loop (where the plugin's kernel takes any p plane);
                                      //    unset: the ring where offered, sized by lIterations and the device's free memory, else 1; "N:W": the ring of N planes with its
                                      //    updates on a second stream NEXT TO the loop, on at most W workgroups (0: one per CU); -N = N
        "THALLO_DIST_P2P",            // 0: never the device-side exchange
        "THALLO_FRONTEND",            // off / generate: see api.cpp
        "THALLO_DENSE_JTJ_MAX",       // largest n for the dense [JtJ]p schedule of generated plugins
        "THALLO_ENABLE_DIRECT_SOLVE", // 1: honour <handle>:set_direct_solve(true) (compiled out in the reference: gauss_newton.t:22)
    };
    static const char* const ab[][2] = {      // THALLO_AB=key=value,...
        { "THALLO_ONE_KERNEL", "one_kernel" },                  // 0: PCGStep1 + PCGStep2 even where the plugin offers the one-kernel iteration
        { "THALLO_FIN_IN_KERNEL", "fin_in_kernel" },            // 0: the iteration's two scalars by a separate one-wave launch; 1: by the last workgroup of the iteration's kernel everywhere;
                                                                //    absent: that, except where the finish of iteration k-1 is folded into the first launch of iteration k
        { "THALLO_LM_FOLD_P", "lm_fold_p" },                    // 0: the reference-shaped LM loop: PCGStep3 / PCGStep2 / the zeta test as launches of their own
        { "THALLO_SFS_FUSED", "sfs_fused" },                    // 0: shape_from_shading's two-pass applyJTJ (round 1)
        { "THALLO_SFS_MARCH", "sfs_march" },                    // 0: shape_from_shading's LDS-tiled kernels instead of the marching ones
        { "THALLO_BA_RENUMBER", "ba_renumber" },                // bundle adjustment's plan-side point order: 0 never, 1 always (default: when the caller's order is far from "by first observing camera"; round 6)
        { "THALLO_LM_FOLD_STEP", "lm_fold_step" },              // 0: the LM step with PCGFinalizeDiagonal and the model cost as launches of their own where a plugin can fold them (round 6)
        { "THALLO_IW_RESIDENT_FOLD", "iw_resident_fold" },      // 0: image_warping's resident PCG loop leaves PCGLinearUpdate a launch of its own (round 6; A/B)
        { "THALLO_SFS_RESIDENT_FOLD", "sfs_resident_fold" },    // 0: shape_from_shading's resident PCG loop leaves PCGLinearUpdate a launch of its own (round 6; A/B)
        { "THALLO_SFS_PAIR", "sfs_pair" },                      // 0: shape_from_shading's one-pixel-per-lane marching kernels on the float4 / float2 / byte planes instead of the pixel-pair kernels on packed planes (round 6)
        { "THALLO_FRONTEND_AGGREGATE", "frontend_aggregate" },  // 0: generated kernels scatter with plain atomics everywhere
        { "THALLO_FRONTEND_COMPUTED", "frontend_computed" },    // 0: computed arrays (expr:get) inlined at every access instead of materialized by a precompute kernel (rounds 1-5; A/B)
        { "THALLO_FRONTEND_PRELOAD", "frontend_preload" },      // 0: every residual instance of a generated merged gather kernel loads for itself (round 4's lowering)
        { "THALLO_INC_LANES", "inc_lanes" },                    // N (a power of two <= 64): lanes per owner in the generated index-map gather kernels (default: by list length and owner count)
        { "THALLO_PERSIST", "persist" },                        // 1: iterations 1 .. L-1 of a GN step of image_warping's marching kernel as persistent launches (bit-identical, measured slower)
    };
    for (const char* k : known) if (!strcmp(k, name)) return getenv(name);
    for (size_t i = 0; i < sizeof(ab) / sizeof(ab[0]); ++i) {
        if (strcmp(ab[i][0], name)) continue;
        static std::string slot[sizeof(ab) / sizeof(ab[0])];
        const char* e = getenv("THALLO_AB");
        if (!e) return nullptr;
        const size_t kl = strlen(ab[i][1]);
        for (const char* t = e; *t; ) {
            while (*t == ',' || *t == ' ') ++t;
            const char* end = t; while (*end && *end != ',' && *end != ' ') ++end;
            if ((size_t)(end - t) > kl && !strncmp(t, ab[i][1], kl) && t[kl] == '=') { slot[i].assign(t + kl + 1, end); return slot[i].c_str(); }
            t = end;
        }
        return nullptr;
    }
    set_error("internal: unknown environment switch %s", name);
    return nullptr;
}

void Plan::read_ab_switches()
{   // read when the reduction slots are (re)sized, i.e. at Plan time and when lIterations grows
    auto off = [](const char* name) { const char* e = env_switch(name); return e && e[0] == '0'; };
    one_kernel_    = !off("THALLO_ONE_KERNEL");
    fin_in_kernel_ = !off("THALLO_FIN_IN_KERNEL");
    { const char* e = env_switch("THALLO_FIN_IN_KERNEL"); fin_deferred_ = !(e && e[0]); }      // unset: deferred where the loop offers it (step_gn_expanded); 1: the in-kernel finish everywhere
    { const char* e = env_switch("THALLO_DELTA_PLANES"); delta_planes_ = e && e[0] ? atoi(e) : -1; if (delta_planes_ > THALLO_HIP_MAX_UPDATE_TERMS + 1) delta_planes_ = THALLO_HIP_MAX_UPDATE_TERMS + 1;
      if (delta_planes_ < -(THALLO_HIP_MAX_UPDATE_TERMS + 1)) delta_planes_ = -(THALLO_HIP_MAX_UPDATE_TERMS + 1);
      const char* c = e ? strchr(e, ':') : nullptr; aux_async_ = c != nullptr; aux_workgroups_ = c && atoi(c + 1) > 0 ? atoi(c + 1) : 0; }      // ("N:W": the update NEXT TO the loop, on at most W workgroups)
    batch_delta_   = delta_planes_ != 0;
    lm_fold_p_     = !off("THALLO_LM_FOLD_P");
    lm_fold_step_  = !off("THALLO_LM_FOLD_STEP");
}

void Plan::set_param(const char* name, const void* value)
{   // gauss_newton.t:1828-1844
#define SETF(f) if (!strcmp(name, #f)) { sp.f = *(const float*)value; return; }
#define SETI(f) if (!strcmp(name, #f)) { sp.f = *(const int*)value; return; }
    SETF(min_relative_decrease) SETF(min_trust_region_radius) SETF(max_trust_region_radius) SETF(q_tolerance)
    SETF(function_tolerance) SETF(trust_region_radius) SETF(radius_decrease_factor) SETF(min_lm_diagonal)
    SETF(max_lm_diagonal) SETF(max_solver_time_in_seconds)
    SETI(residual_reset_period) SETI(nIter) SETI(nIterations) SETI(lIterations)
#undef SETF
#undef SETI
    if (ip.verbosityLevel > 0) printf("Warning: tried to set nonexistent solver parameter %s\n", name);
}
void Plan::get_param(const char* name, void* value)
{   // gauss_newton.t:1845-1862
#define GETF(f) if (!strcmp(name, #f)) { *(float*)value = sp.f; return; }
#define GETI(f) if (!strcmp(name, #f)) { *(int*)value = sp.f; return; }
    GETF(min_relative_decrease) GETF(min_trust_region_radius) GETF(max_trust_region_radius) GETF(q_tolerance)
    GETF(function_tolerance) GETF(trust_region_radius) GETF(radius_decrease_factor) GETF(min_lm_diagonal)
    GETF(max_lm_diagonal) GETF(max_solver_time_in_seconds)
    GETI(residual_reset_period) GETI(nIter) GETI(nIterations) GETI(lIterations)
#undef GETF
#undef GETI
    if (ip.verbosityLevel > 0) printf("Warning: tried to get nonexistent solver parameter %s\n", name);
}

float* Plan::host_words()
{   // lazily: 16 pinned words (NULL if the allocation fails: the callers then copy into their own pageable words, as before)
    if (!host_words_ && hipHostMalloc((void**)&host_words_, 16 * sizeof(float), hipHostMallocDefault) != hipSuccess) { host_words_ = nullptr; (void)hipGetLastError(); }
    return host_words_;
}

float Plan::compute_cost()
{   // gauss_newton.t:1128-1136 -- partials instead of memset + atomics; same blocking 4-byte read-back
    if (dist_) return dist_cost();
    const int nb = plugin->cost(ctx, slot(0));
    if (nb < 0) { set_error("cost kernel launch failed (%d)", nb); return NAN; }
    set_nb(0, nb);
    thallo_hip_finish_sum(sum(0), (float*)scratch_.ptr, ctx.stream);
    float f = 0.0f;
    float* hw = host_words();
    HIP_OK(hipMemcpyAsync(hw ? hw : &f, scratch_.ptr, sizeof(float), hipMemcpyDeviceToHost, ctx.stream));
    HIP_OK(hipStreamSynchronize(ctx.stream));
    if (hw) f = hw[0];
    if (resident_used_) {     // the resident PCG kernel's waits are bounded; one that ran out voids the steps since the last check
        resident_used_ = false;
        unsigned pm[5] = { 0, 0, 0, 0, 0 };
        if (plugin->resident_status(ctx, 1, pm) != 0) {
            // (something kept the kernel's workgroups from being co-resident -- a long-running foreign kernel, a second resident plan on another stream: the launch
            //  itself checks what the device can hold.)  This plan runs one launch per PCG iteration from now on; the voided steps are reported, not silently redone:
            //  the unknowns already carry their update.
            plugin->resident_disable();
            set_error("%s: a bounded wait inside the resident PCG kernel ran out (wait kind %u, workgroup %u, wave %u, index %u, tag %u): the steps since the last cost evaluation are void; "
                      "the plan runs one launch per PCG iteration from now on (Thallo_ProblemInit and solve again)", plugin->name(), pm[0], pm[1], pm[2], pm[3], pm[4]);
            return NAN;
        }
    }
    return f;
}

void Plan::init(void** params)
{   // gauss_newton.t:1166-1198
    if (!ok_) return;
    finalized_ = false;
    timer_.cleanup();
    ev_total_ = timer_.start("Total", ctx.stream);
    ready_ = false;
    // Across ranks Init is collective: a rank that cannot bind or prepare its parameters says so in the agreement below instead of returning in front of the
    // other ranks' first all-gather (they would wait there forever); the same exchange carries the one-kernel slab schedule's precondition.
    bool local_ok = true;
    if (plugin->bind(params)) { set_error("%s: parameter binding failed", plugin->name()); local_ok = false; }
    if (local_ok) {
        plugin->unknowns_changed();            // a new solve: the caller may have rewritten unknowns and inputs behind the same pointers
        if (plugin->prepare(ctx)) { const std::string why = last_error(); set_error("%s: prepare failed: %s", plugin->name(), why.c_str()); local_ok = false; }
    }
    if (!dist_) { if (!local_ok) return; }
    else {
        dist_->failed = false; dist_->stopped = false;      // a new solve
        const bool need_grid = !dist_->flat && !dist_->range;
        bool all = false;
        if (dist_agree(local_ok && (!need_grid || plugin->slab_grid_ok()), all)) return;
        if (!all) {
            if (local_ok && need_grid && !plugin->slab_grid_ok()) set_error("%s: the row-slab schedule needs UrShape on the unit pixel grid on every rank", plugin->name());
            else if (local_ok) set_error("%s: another rank could not start the solve (its parameters, or UrShape off the unit pixel grid); no rank does", plugin->name());
            return;
        }
    }
    ready_ = true;
    if (dist_ && dist_->want_p2p && !dist_->checked && dist_self_check()) { ready_ = false; return; }
    if (dist_ && dist_->checked && dist_->p2p_on && !dist_->flat && !dist_->range && !dist_->shard) {      // (also right behind a self-check that has just passed: ADVICE r5)
        // Every Init agrees anew on what depends on the plugin's state NOW: whether every rank's slab runs the resident kernel (a rank may have switched it
        // off since -- a bounded wait that ran out, THALLO_RESIDENT at re-Init) and whether the marching kernel's cross-rank finish is deferred.  Here, never at the
        // first step: a step may be inside a captured graph.
        bool all = false;
        if (dist_agree(plugin->resident_slab_ok() && dist_->ghost_off > 0, all)) { ready_ = false; return; }
        dist_->resident_all = all;
        const bool mine = plugin->dist_defers_finish() && ensure_iter_buffers() == 0 && (dist_->gs.ptr || dist_->gs.alloc(64) == 0);
        if (dist_agree(mine, all)) { ready_ = false; return; }
        dist_->defer_state = all ? 1 : 0;
    }
    sp.nIter = 0;
    if (!lm_ && one_kernel_ && plugin->one_kernel_iteration() && !plugin->resident_ok()) { ring_prepare(sp.lIterations); ring_L_ = sp.lIterations; }      // (the ring's planes: here, not inside a step)
    prev_cost_ = compute_cost();
    printf("Initial cost: %g\n", prev_cost_);
}

void Plan::finalize()
{   // gauss_newton.t:1200-1212
    prev_cost_ = compute_cost();
    if (ip.verbosityLevel > 0) printf("final cost=%g\n", prev_cost_);
    hipDeviceSynchronize();
    timer_.stop(ev_total_, ctx.stream);
    timer_.evaluate(&summary, ip.verbosityLevel > 0, &ktimer);
    timer_.cleanup();
    finalized_ = true;
}

double Plan::cost()
{   // gauss_newton.t:1787-1793
    if (!ok_ || !ready_) return dist_ && dist_->stopped ? (double)NAN : 0.0;
    if (!finalized_) prev_cost_ = compute_cost();
    return (double)prev_cost_;
}

int Plan::step(void** params)
{   // gauss_newton.t:1545-1785
    if (!ok_ || !ready_) return 0;
    if (plugin->bind(params)) {
        if (dist_) dist_fail("%s: parameter binding failed", plugin->name());      // (stays in step with the other ranks: solver_dist.cpp, DistState::failed)
        else { set_error("%s: parameter binding failed", plugin->name()); return 0; }
    }
    // Gauss-Newton steps derive everything from the unknowns as they are NOW, like the reference (a caller may have rewritten them since the last call).  The LM
    // branch already lives on state carried from the previous step's end -- prev_cost_, as the reference's pd.hd.prevCost (gauss_newton.t:1732,1710) -- so there a
    // plugin may also keep what it derived from the unknowns at that point (shape_from_shading's precomputed planes).
    if (!lm_) plugin->unknowns_changed();
    if (sp.nIter >= sp.nIterations) { if (!finalized_) finalize(); return 0; }
    if (sp.lIterations < 0) { set_error("lIterations = %d is negative", sp.lIterations); if (!finalized_) finalize(); return 0; }
    if (ensure_slots(sp.lIterations)) {
        if (dist_) dist_fail("out of device memory for the reduction slots");
        else { if (!finalized_) finalize(); return 0; }
    }
    const int ev_iter = timer_.start("Nonlinear Iteration", ctx.stream);
    if (dist_ && lm_ && !dist_->shard && (!dist_->flat || dist_->range)) { set_error("distributed: %s runs Gauss-Newton only across ranks", plugin->name()); if (!finalized_) finalize(); return 0; }
    const int rc = lm_ && dist_ && dist_->shard ? step_lm_shard(ev_iter) : lm_ ? step_lm(ev_iter) : dist_ ? step_gn_slab(ev_iter) : step_gn(ev_iter);
    if (rc == 1 && sp.max_solver_time_in_seconds > 0.0f && ev_total_ >= 0) {   // :1767-1779
        hipStream_t s = ctx.stream;
        hipEvent_t q = nullptr; hipEventCreate(&q); hipEventRecord(q, s); hipEventSynchronize(q);
        float ms = 0.0f; hipEventElapsedTime(&ms, timer_.events[ev_total_].start, q); hipEventDestroy(q);
        if (ms / 1000.0f > sp.max_solver_time_in_seconds) { finalize(); return 0; }
    }
    return rc;
}

void Plan::linear_update_tail(int L, bool batched)
{   // PCGLinearUpdate (gauss_newton.t:901-906) at the end of a GN step of any of the fused schedules, including the delta += alpha*p terms
    // still pending: after the last fused PCGStep1 (k = L-1) delta holds the terms up to p_{L-2}; with `batched` (every other update
    // deferred) up to p_{L-2} if L-1 is even and up to p_{L-3} if it is odd.  One row slab of a multi-GPU run touches its owned rows only.
    const int B = 2;                       // slot layout: alphaN_k = B+2k, alphaD_k = B+2k+1
    hipStream_t s = ctx.stream;
    const auto& imgs = plugin->unknown_images();
    const int jN = B + 2 * (L - 1), jD = jN + 1;
    long off = 0;
    for (size_t k = 0; k < imgs.size(); ++k) {
        TimedLaunch t(ctx, "PCGLinearUpdate");
        long lo = 0, len = imgs[k].n_floats;
        if (dist_ && !dist_->range) { const long rowlen = imgs[k].n_floats / dist_->Hl; lo = rowlen * dist_->row0; len = rowlen * (dist_->row1 - dist_->row0); }
        float* X = plugin->unknown_ptr((int)k) + lo;
        const float* dl = v_.delta + off + lo;
        if (L > 1 && batched && ((L - 1) & 1))
            thallo_hip_linear_update2(X, dl, v_.p[cur_ ^ 1] + off + lo, sum(jN - 2), sum(jD - 2), v_.p[cur_] + off + lo, sum(jN), sum(jD), len, s);
        else if (L > 0) thallo_hip_linear_update(X, dl, v_.p[cur_] + off + lo, len, sum(jN), sum(jD), s);
        else            thallo_hip_linear_update(X, dl, nullptr, len, sum(B), sum(B), s);
        off += imgs[k].n_floats;
    }
    plugin->unknowns_written();
}

int Plan::step_gn(int ev_iter)
{   // GN branch, fused schedule (DESIGN.md "PCG schedule")
    if (plugin->direct_solve()) {                     // the linear system solved exactly, no PCG loop (gauss_newton.t:1612-1613)
        hipStream_t s = ctx.stream;
        cur_ = 0;
        const int nb = plugin->pcg_init(ctx, v_, cur_, slot(2));
        if (nb < 0) { set_error("PCGInit1 launch failed (%d)", nb); return 0; }
        set_nb(2, nb);
        const int ev_lin = timer_.start("Linear Solve", s);
        if (plugin->solve_direct(ctx, v_) < 0) { const std::string why = last_error(); set_error("direct solve failed: %s", why.c_str()); return 0; }
        timer_.stop(ev_lin, s);
        last_l_iters = 0;
        linear_update_tail(0, false);
        sp.nIter++;
        timer_.stop(ev_iter, s);
        return 1;
    }
    if (one_kernel_ && sp.lIterations >= 1 && plugin->resident_ok()) return step_gn_resident(ev_iter);
    if (one_kernel_ && plugin->one_kernel_iteration()) return step_gn_one_kernel(ev_iter);
    if (plugin->apply_returns_sums()) return step_gn_expanded(ev_iter);
    const int L = sp.lIterations;
    hipStream_t s = ctx.stream;
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    const int B = 2;                       // slot layout: alphaN_k = B+2k, alphaD_k = B+2k+1, betaN_k = B+2k+2
    cur_ = 0;
    int nb = plugin->pcg_init(ctx, v_, cur_, slot(B));
    if (nb < 0) { set_error("PCGInit1 launch failed (%d)", nb); return 0; }
    set_nb(B, nb); finish(B);
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    const bool batched = plugin->batches_delta() && batch_delta_;      // every other delta update deferred (thallo_hip.h THALLO_IW_STEP1_MODE)
    for (int k = 0; k < L; ++k) {
        const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
        // PCGStep1 (+ previous iteration's PCGStep3 and delta update)
        thallo_sum_t aNp = sum(k ? jN - 2 : jN), aDp = sum(k ? jD - 2 : jD), bNp = sum(jN);
        if (batched) nb = plugin->pcg_step1_mode(ctx, v_, cur_, THALLO_IW_STEP1_MODE(k, 1), aNp, aDp, bNp, sum(k > 1 ? jN - 4 : jN), sum(k > 1 ? jD - 4 : jD), slot(jD));
        else         nb = plugin->pcg_step1(ctx, v_, cur_, k == 0, aNp, aDp, bNp, slot(jD));
        if (nb < 0) { set_error("PCGStep1 launch failed (%d)", nb); return 0; }
        set_nb(jD, nb); finish(jD); cur_ ^= 1;
        nb = plugin->pcg_step2(ctx, v_, sum(jN), sum(jD), slot(jB));      // PCGStep2 (r, z, betaN)
        if (nb < 0) { set_error("PCGStep2 launch failed (%d)", nb); return 0; }
        set_nb(jB, nb); finish(jB);
    }
    last_l_iters = L;
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    linear_update_tail(L, batched);
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    return 1;
}

// ------------------------------------------------------------------ Levenberg-Marquardt branch
void Plan::enable_lm(bool on) { lm_ = on; }
void Plan::unknowns_changed()
{
    if (!ok_ || !ready_) return;
    plugin->unknowns_changed();
    if (lm_ && !finalized_) prev_cost_ = compute_cost();          // (the LM branch carries the cost of the previous step's end)
}

int Plan::ensure_lm_vectors()
{
    if (v_.b) return 0;
    float** vecs[] = { &v_.b, &v_.Adelta, &v_.CtC, &v_.SSq, &v_.prevX, &v_.diag };
    for (float** vp : vecs) {
        DeviceBuffer* b = new DeviceBuffer(); bufs_.push_back(b);
        if (b->alloc((size_t)v_.n_alloc * sizeof(float))) return -1;
        *vp = (float*)b->ptr;
    }
    return 0;
}

int Plan::step_gn_one_kernel(int ev_iter)
{   // GN branch, ONE kernel + one scalar launch per PCG iteration (DESIGN.md "PCG schedule", thallo_hip_iw_pcg_iter)
    if (ensure_iter_buffers()) { set_error("out of device memory for the one-kernel schedule"); return 0; }
    const int L = sp.lIterations;
    hipStream_t s = ctx.stream;
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    const int B = 2;                       // slot layout: alphaN_k = B+2k, alphaD_k = B+2k+1, betaN_k = B+2k+2
    cur_ = 0;
    int nb = plugin->pcg_init(ctx, v_, cur_, slot(B));
    if (nb < 0) { set_error("PCGInit1 launch failed (%d)", nb); return 0; }
    set_nb(B, nb); finish(B);
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    // Deferred finish (where the plugin offers it): the launch of iteration k adds up iteration k-1's partials
    // itself -- alphaD_{k-1} and betaN_{k-1} = N - 2 alpha S1 + alpha^2 S2 -- while its first rows load, instead of iteration k-1's last workgroup
    // reading them back at the very end of its launch; one one-wave launch per GN step finishes the last iteration.
    const bool defer = plugin->iter_defers_finish();
    // Round 5: a RING of p planes instead of the ping-pong pair.  Launch k writes p_k into plane k mod n and carries no delta update at all; delta is touched once
    // per n - 1 iterations by thallo_hip_linear_update_n (every pending alpha_j p_j, oldest first, one fma each: the bits of an update per iteration) and the step's
    // last terms ride in PCGLinearUpdate.  Per iteration and unknown 4 (12 + 24 / (n - 1)) / 12 bytes of delta traffic instead of 24 (every iteration) or 18 (every other).
    const int n_ring = ring_planes(L);
    const bool ring = n_ring >= 2;
    const bool batched = !ring && batch_delta_ && delta_planes_ != 0 && plugin->batches_delta();      // THALLO_IW_STEP1_MODE(k, 1): every other delta update is deferred
    // Where the update runs.  Default: on the loop's own stream, whole rings at a time (up to 32 terms per launch).  THALLO_DELTA_PLANES=N:W: NEXT TO the loop, on a
    // low-priority stream of the plan's own and on at most W workgroups (0: one per CU), in chunks of (n - 1) / 2 terms as soon as their scalars are words; the launch
    // that overwrites a chunk's first plane waits for its event.  Measured (profiles/r05/ring_ab*.txt): next to the loop 0.7-2 % more PCG iterations per second, but the
    // marching launches that share the chip with an update take ~8 us longer each -- the loop streams at the memory system's rate in its steady state, and what the
    // second kernel takes there it does not give back in the ramps and tails.  Not worth a second stream in the default path.
    const bool persist = ring && defer && L >= 2 && plugin->persist_ok();
    const bool async = ring && aux_async_ && !persist && aux_stream();
    const int chunk = !ring ? 0 : async ? (n_ring - 1) / 2 > 0 ? (n_ring - 1) / 2 : 1 : n_ring - 1;
    int flushed = 0;                       // p_0 .. p_{flushed-1} are in delta, or on their way there (async)
    int synced = 0;                        // ... and the loop's stream has waited for the updates of p_0 .. p_{synced-1}
    struct Sent { int upto; hipEvent_t done; };
    std::vector<Sent> sent;
    size_t n_ev = 0;
    SolverVectors vr = v_;                 // (ring: the plugin sees plane k-1 as p[cur], plane k as p[cur ^ 1])
    auto ring_plane = [&](int k) -> float* { return k < 0 ? v_.p[0] : ring_[(size_t)(k % n_ring)]; };
    auto flush_ring = [&](int upto) -> int {      // delta += alpha_j p_j for flushed <= j <= upto (their scalars are words once what is enqueued on s has run)
        while (flushed <= upto) {
            thallo_update_terms_t T; T.count = 0;
            for (; flushed <= upto && T.count < THALLO_HIP_MAX_UPDATE_TERMS; ++flushed) {
                T.p[T.count] = ring_plane(flushed); T.alphaN[T.count] = sum(B + 2 * flushed); T.alphaD[T.count] = sum(B + 2 * flushed + 1); ++T.count;
            }
            if (!async) {
                TimedLaunch t(ctx, "PCGDeltaUpdate");
                if (thallo_hip_linear_update_n(nullptr, v_.delta, T, v_.n_alloc, 0, s) < 0) return -1;
                synced = flushed;
                continue;
            }
            if (n_ev + 2 > aux_events_.size()) { hipEvent_t e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1; aux_events_.push_back(e);
                                                 e = nullptr; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1; aux_events_.push_back(e); }
            hipEvent_t words = aux_events_[n_ev++], done = aux_events_[n_ev++];
            if (hipEventRecord(words, s) != hipSuccess || hipStreamWaitEvent(aux_, words, 0) != hipSuccess) return -1;
            LaunchCtx cx = ctx; cx.stream = aux_;
            {   TimedLaunch t(cx, "PCGDeltaUpdate");
                if (thallo_hip_linear_update_n(nullptr, v_.delta, T, v_.n_alloc, aux_workgroups_ > 0 ? aux_workgroups_ : thallo_hip_device_cu_count(), aux_) < 0) return -1; }
            if (hipEventRecord(done, aux_) != hipSuccess) return -1;
            sent.push_back(Sent{ flushed - 1, done });
        }
        return 0;
    };
    // whichever way this function is left, the loop's stream ends up behind every update that went out on the second stream: the next step's PCGInit1 zeroes delta
    // (on a failed launch too -- the caller may call Step again)
    struct Join { hipStream_t s; const std::vector<Sent>* sent; ~Join() { if (!sent->empty()) (void)hipStreamWaitEvent(s, sent->back().done, 0); } } join_on_exit{ s, &sent };
    auto wait_for = [&](int term) -> int {        // the loop's stream goes on only when the update that took p_term has run
        for (const Sent& q : sent) {
            if (q.upto < synced) continue;
            if (hipStreamWaitEvent(s, q.done, 0) != hipSuccess) return -1;
            synced = q.upto + 1;
            if (q.upto >= term) break;
        }
        return 0;
    };
    // Persistent form (plugins that offer it, on the ring): iteration 0 as a launch of its own, then iterations k .. k + m - 1 per launch, m <= n - 1 (no plane of a
    // launch is written twice, none is overwritten before delta has it)
    int nb_prev = 0;
    for (int k = 0; k < L; ++k) {
        if (persist && k >= 1) {
            const int m = L - k < n_ring - 1 ? L - k : n_ring - 1, k1 = k + m;
            if (k1 - 1 >= n_ring && flushed < k1 - n_ring) { if (flush_ring(k - 2) || wait_for(k - 2)) { set_error("PCGDeltaUpdate launch failed"); return 0; } }
            nb = plugin->pcg_persist(ctx, v_, ring_.data(), n_ring, k, k1, slot(0), parts_slots_, B, nb_prev, sum(B + 2 * (k - 1)));
            if (nb < 0) { set_error("PCGLoopPersistent launch failed (%d)", nb); return 0; }
            resident_used_ = true;
            for (int i = k; i < k1; ++i) { const int jD = B + 2 * i + 1; fin_[jD - 2] = 1; set_nb(jD - 1, 1); fin_[jD - 1] = 1; }      // the words alphaD_{i-1}, betaN_{i-1} = alphaN_i
            set_nb(B + 2 * (k1 - 1) + 1, nb); nb_prev = nb;
            if (m & 1) cur_ ^= 1;
            k = k1 - 1;
            if (k == L - 1) {
                const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
                if (plugin->pcg_iter_finish_from(ctx, slot(jD), v_.s12buf(k & 1), nb, sum(jN), scal(jD), scal(jB)) < 0) { set_error("PCGScalars launch failed"); return 0; }
                fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1;
            }
            continue;
        }
        const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
        const int mode = ring ? (k == 0 ? 1 : 2) : THALLO_IW_STEP1_MODE(k, batched ? 1 : 0);
        if (ring) {
            // plane k mod n still holds p_{k-n}: it has to be in delta before launch k overwrites it.  Terms up to k - 2 can go (launch k - 1 leaves their scalars)
            const bool due = async ? k - 1 - flushed >= chunk : (k >= n_ring && flushed < k - n_ring + 1);
            if (due && flush_ring(k - 2)) { set_error("PCGDeltaUpdate launch failed"); return 0; }
            if (k >= n_ring && synced < k - n_ring + 1 && wait_for(k - n_ring)) { set_error("PCGDeltaUpdate: stream wait failed"); return 0; }
            vr.p[cur_] = ring_plane(k - 1); vr.p[cur_ ^ 1] = ring_plane(k);
        }
        SolverVectors& vv = ring ? vr : v_;
        if (defer) {
            const thallo_prev_t prev = { k ? slot(jD - 2) : nullptr, v_.s12buf((k - 1) & 1), nb_prev, k ? scal(jD - 2) : nullptr, k ? scal(jB - 2) : nullptr };
            nb = plugin->pcg_iter_deferred(ctx, vv, cur_, mode, sum(k ? jN - 2 : jN), sum(k > 1 ? jN - 4 : jN), sum(k > 1 ? jD - 4 : jD), prev, slot(jD), v_.s12buf(k & 1));
            if (nb < 0) { set_error("PCGIteration launch failed (%d)", nb); return 0; }
            if (k) { fin_[jD - 2] = 1; set_nb(jB - 2, 1); fin_[jB - 2] = 1; }      // (that launch's workgroup 0 writes the two words of iteration k-1)
            set_nb(jD, nb); nb_prev = nb; cur_ ^= 1;
            if (k == L - 1) {
                if (plugin->pcg_iter_finish_from(ctx, slot(jD), v_.s12buf(k & 1), nb, sum(jN), scal(jD), scal(jB)) < 0) { set_error("PCGScalars launch failed"); return 0; }
                fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1;
            }
        } else {
            // alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2: by the kernel's last workgroup, or (THALLO_FIN_IN_KERNEL=0) a one-wave launch
            nb = plugin->pcg_iter(ctx, vv, cur_, mode, sum(k ? jN - 2 : jN), sum(k ? jD - 2 : jD), sum(jN),
                                  sum(k > 1 ? jN - 4 : jN), sum(k > 1 ? jD - 4 : jD), slot(jD),
                                  fin_in_kernel_ ? scal(jD) : nullptr, fin_in_kernel_ ? scal(jB) : nullptr);
            if (nb < 0) { set_error("PCGIteration launch failed (%d)", nb); return 0; }
            set_nb(jD, nb); cur_ ^= 1;
            if (!fin_in_kernel_ && plugin->pcg_iter_finish(ctx, v_, slot(jD), nb, sum(jN), scal(jD), scal(jB)) < 0) { set_error("PCGScalars launch failed"); return 0; }
            fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1;
        }
    }
    last_l_iters = L;
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    if (ring && L > 0) {
        // PCGLinearUpdate with every pending term: all but the last THALLO_HIP_MAX_UPDATE_TERMS go into delta first
        if (flush_ring(L - 1 - THALLO_HIP_MAX_UPDATE_TERMS) || wait_for(L)) { set_error("PCGDeltaUpdate launch failed"); return 0; }
        thallo_update_terms_t T; T.count = 0;
        for (int j = flushed; j < L; ++j) { T.p[T.count] = ring_plane(j); T.alphaN[T.count] = sum(B + 2 * j); T.alphaD[T.count] = sum(B + 2 * j + 1); ++T.count; }
        const auto& imgs = plugin->unknown_images();
        long off = 0;
        for (size_t u = 0; u < imgs.size(); ++u) {
            TimedLaunch t(ctx, "PCGLinearUpdate");
            thallo_update_terms_t Tu = T;
            for (int j = 0; j < Tu.count; ++j) Tu.p[j] += off;
            if (thallo_hip_linear_update_n(plugin->unknown_ptr((int)u), v_.delta + off, Tu, imgs[u].n_floats, 0, s) < 0) { set_error("PCGLinearUpdate launch failed"); return 0; }
            off += imgs[u].n_floats;
        }
        plugin->unknowns_written();
    } else linear_update_tail(L, batched);
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    return 1;
}

// The plan's second stream (lowest priority, non-blocking: the loop's stream may be the NULL stream): background updates of delta next to the PCG loop
bool Plan::aux_stream()
{
    if (aux_) return true;
    if (aux_failed_) return false;
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    if (hipStreamCreateWithPriority(&aux_, hipStreamNonBlocking, lo) != hipSuccess) { aux_ = nullptr; aux_failed_ = true; (void)hipGetLastError(); return false; }
    return true;
}

// Planes in the ring of p vectors of the one-kernel GN loop (0: no ring).  v_.p[1], v_.p[0] are its first two; the others are allocated by ring_prepare() -- at Init and
// again only when lIterations asks for more than the last attempt did -- never inside the step's launch sequence (a hipMalloc + NULL-stream memset per plane is
// a device-wide synchronisation, illegal under stream capture, and was retried at every step of a memory-limited plan).
// Footprint: (planes - 2) x one solver vector; by default at most 33 planes, a quarter of the free device memory and 32 GiB (2048^2 image_warping: 31 x 50.3 MB = 1.56 GB;
// 16384 x 8192: 20 planes of 1.6 GB).  THALLO_DELTA_PLANES=N asks for exactly N (no memory test); planes are released with the plan.
bool Plan::ring_possible() const
{   // one GPU, or (round 6) one rank's row slab of the one-kernel schedule: launch k writes p_k into whichever plane it is handed (the ghost rows are kept current there too)
    return plugin->takes_any_p_plane() && (!dist_ || (!dist_->flat && !dist_->range && !dist_->shard && !dist_->part));
}
void Plan::ring_prepare(int L)
{
    if (!ring_possible() || delta_planes_ == 0 || delta_planes_ == 1 || L < 3) return;
    int want = delta_planes_ >= 2 ? delta_planes_ : delta_planes_ <= -2 ? -delta_planes_ : THALLO_HIP_MAX_UPDATE_TERMS + 1;
    if (want > L) want = L;                 // (L planes: delta is never touched inside the loop)
    if (ring_.size() < 2) { ring_.clear(); ring_.push_back(v_.p[1]); ring_.push_back(v_.p[0]); }
    if (want <= (int)ring_.size() || want <= ring_tried_) return;      // enough planes, or an attempt for this many already ran into the memory limit
    ring_tried_ = want;
    const size_t bytes = (size_t)v_.n_alloc * sizeof(float);
    while ((int)ring_.size() < want) {
        size_t free_b = 0, total_b = 0;
        if (delta_planes_ == -1 && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4 / (size_t)(want - (int)ring_.size()) ||
                                  bytes * (ring_.size() - 1) > ((size_t)32 << 30))) break;
        void* ptr = nullptr;
        if (hipMalloc(&ptr, bytes) != hipSuccess) { (void)hipGetLastError(); break; }      // fewer planes, not a failed step: the sticky error goes, no error text is left
        // zeroed once, in stream order on the plan's own stream: a launch writes every unknown of its plane but not the padding behind them, which the flat update reads
        if (hipMemsetAsync(ptr, 0, bytes, ctx.stream) != hipSuccess) { (void)hipGetLastError(); hipFree(ptr); break; }
        DeviceBuffer* b = new DeviceBuffer(); b->ptr = ptr; b->bytes = bytes;
        bufs_.push_back(b); ring_.push_back((float*)ptr);
    }
}
int Plan::ring_planes(int L)
{
    if (!ring_possible() || delta_planes_ == 0 || delta_planes_ == 1 || L < 3) return 0;
    int want = delta_planes_ >= 2 ? delta_planes_ : delta_planes_ <= -2 ? -delta_planes_ : THALLO_HIP_MAX_UPDATE_TERMS + 1;
    if (want > L) want = L;
    if (L != ring_L_) { ring_prepare(L); ring_L_ = L; }      // (lIterations changed since Init: once per change)
    if (ring_.size() < 2) return 0;
    const int n = (int)ring_.size() < want ? (int)ring_.size() : want;
    return n >= 3 || want == 2 ? n : 0;      // (two planes = an update per iteration: only on request)
}

int Plan::step_gn_resident(int ev_iter)
{   // GN branch, the whole PCG loop in ONE launch (plugins whose shape fits the chip's registers: thallo_hip_iw_pcg_resident).  Same slots and words as
    // step_gn_one_kernel leaves behind -- alphaN_k = B+2k (a word from k = 1 on), alphaD_k = B+2k+1, betaN_k = B+2k+2 -- so PCGLinearUpdate, the alpha / beta
    // trace and the cost path do not know which schedule ran.  Replaces the loop of gauss_newton.t:1615-1687.
    if (ensure_iter_buffers()) { set_error("out of device memory for the one-kernel schedule"); return 0; }
    const int L = sp.lIterations, B = 2;
    hipStream_t s = ctx.stream;
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    cur_ = 0;
    int nb = plugin->pcg_init(ctx, v_, cur_, slot(B));
    if (nb < 0) { set_error("PCGInit1 launch failed (%d)", nb); return 0; }
    set_nb(B, nb); finish(B);
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    nb = plugin->pcg_resident(ctx, v_, L, sum(B), scal(B + 1));
    if (nb < 0) { set_error("PCGLoopResident launch failed (%d)", nb); return 0; }
    for (int k = 0; k < L; ++k) { const int jD = B + 2 * k + 1, jB = jD + 1; set_nb(jD, 1); fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1; }
    cur_ = L & 1;
    resident_used_ = true;
    last_l_iters = L;
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    if (plugin->resident_updates_unknowns()) plugin->unknowns_written();      // (PCGLinearUpdate rode in the resident launch)
    else linear_update_tail(L, false);
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    return 1;
}

int Plan::step_gn_expanded(int ev_iter)
{   // GN branch, single-reduction form for gather energies: per PCG iteration pcg_update (flat) + applyJTJ with sums + scalars_finish
    // (thallo_hip.h "single-reduction PCG form") instead of PCGStep3 + applyJTJ + PCGStep2 and their finish launches
    if (ensure_sums_buffer()) { set_error("out of device memory for the single-reduction schedule"); return 0; }
    const int L = sp.lIterations;
    hipStream_t s = ctx.stream;
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    const int B = 2;                       // slot layout: alphaN_k = B+2k, alphaD_k = B+2k+1, betaN_k = B+2k+2
    cur_ = 0;
    int nb = plugin->pcg_init(ctx, v_, cur_, slot(B));
    if (nb < 0) { set_error("PCGInit1 launch failed (%d)", nb); return 0; }
    set_nb(B, nb); finish(B);
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    // THALLO_FIN_IN_KERNEL unset: the finish of iteration k-1 -- alphaD, betaN from the applyJTJ launch's partials -- is folded into the flat update of iteration k
    // (thallo_hip_pcg_update_fin: every workgroup adds the partials up for itself, the same bits), so the applyJTJ launch has no tail; one one-wave launch finishes the last iteration
    const bool defer = fin_in_kernel_ && fin_deferred_;
    int nb_prev = 0;
    for (int k = 0; k < L; ++k) {
        const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
        {   // r -= alpha_{k-1} Ap ; p_k = M^-1 r + beta_{k-1} p_{k-1} ; delta += alpha_{k-1} p_{k-1}      (k = 0: p_0 = M^-1 r_0)
            TimedLaunch t(ctx, "PCGUpdate");
            int rc;
            if (defer && k > 0) rc = thallo_hip_pcg_update_fin(v_.r, v_.Ap, plugin->use_preconditioner() ? v_.pre : nullptr, v_.p[cur_], v_.p[cur_ ^ 1], v_.delta, v_.n, sum(jN - 2),
                                                               slot(jD - 2), v_.s12, nb_prev, scal(jD - 2), scal(jN), s);
            else rc = thallo_hip_pcg_update(v_.r, v_.Ap, plugin->use_preconditioner() ? v_.pre : nullptr, v_.p[cur_], v_.p[cur_ ^ 1], v_.delta, v_.n, k == 0,
                                            sum(k ? jN - 2 : jN), sum(k ? jD - 2 : jD), sum(jN), s);
            if (rc < 0) { set_error("PCGUpdate launch failed"); return 0; }
        }
        cur_ ^= 1;
        // alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2: by the next flat update (default), by the applyJTJ kernel's last workgroup (THALLO_FIN_IN_KERNEL=1), or a one-wave launch (=0)
        const thallo_fin_t fin = { sum(jN), fin_in_kernel_ && !defer ? v_.fin_tickets : nullptr, scal(jD), scal(jB) };
        nb = plugin->apply_jtj_sums(ctx, v_, v_.p[cur_], v_.Ap, slot(jD), fin);
        if (nb < 0) { set_error("PCGStep1 launch failed (%d)", nb); return 0; }
        set_nb(jD, nb); nb_prev = nb;
        if (!fin_in_kernel_ || (defer && k == L - 1)) {
            TimedLaunch t(ctx, "PCGScalars");
            if (thallo_hip_pcg_scalars_finish(slot(jD), v_.s12, nb, sum(jN), scal(jD), scal(jB), s) < 0) { set_error("PCGScalars launch failed"); return 0; }
        }
        fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1;
    }
    const bool batched = false;
    last_l_iters = L;
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    linear_update_tail(L, batched);
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    return 1;
}

int Plan::ensure_sums_buffer()
{   // per-workgroup double sums + the arrival tickets of the in-kernel finish
    auto get = [&](size_t bytes) -> void* { DeviceBuffer* b = new DeviceBuffer(); bufs_.push_back(b); return b->alloc(bytes) ? nullptr : b->ptr; };
    if (!v_.s12 && !(v_.s12 = (double*)get((size_t)3 * THALLO_HIP_MAX_PARTIALS * sizeof(double)))) return -1;
    if (!v_.s12b && !(v_.s12b = (double*)get((size_t)3 * THALLO_HIP_MAX_PARTIALS * sizeof(double)))) return -1;
    if (!v_.fin_tickets && !(v_.fin_tickets = (unsigned*)get(THALLO_HIP_FIN_TICKET_WORDS * sizeof(unsigned)))) return -1;
    return 0;
}

int Plan::ensure_iter_buffers()
{
    if (ensure_sums_buffer()) return -1;
    auto get = [&](size_t bytes) -> void* { DeviceBuffer* b = new DeviceBuffer(); bufs_.push_back(b); return b->alloc(bytes) ? nullptr : b->ptr; };
    if (!v_.r2 && !(v_.r2 = (float*)get((size_t)v_.n_alloc * sizeof(float)))) return -1;            // (a row slab's r' / Ap' live in its exchange block)
    if (!v_.Ap2 && !(v_.Ap2 = (float*)get((size_t)v_.n_alloc * sizeof(float)))) return -1;
    return 0;
}

float Plan::read_sum(int j)
{   // blocking 4-byte read-back, like fetchQ / computeModelCost (gauss_newton.t:1138-1150)
    thallo_hip_finish_sum(sum(j), (float*)scratch_.ptr + 1, ctx.stream);
    float f = 0.0f;
    HIP_OK(hipMemcpyAsync(&f, (float*)scratch_.ptr + 1, sizeof(float), hipMemcpyDeviceToHost, ctx.stream));
    HIP_OK(hipStreamSynchronize(ctx.stream));
    return f;
}

int Plan::step_lm(int ev_iter)
{   // gauss_newton.t:1545-1785 with every UsesLambda() branch taken; unfused (reference-shaped) PCG schedule because
    // q = 0.5 delta.(r+b) needs the current delta inside PCGStep2 and PCGStep1_Finish adds CtC*p.
    // Slots: 0 cost, 1 q, B.. as in GN (alphaN_k = B+2k, alphaD_k = B+2k+1, betaN_k = B+2k+2); last two: scratch dots.
    //
    // The zeta test (:1666-1686) runs ON THE DEVICE: all lIterations iterations are enqueued, a one-wave kernel behind each PCGStep2
    // (or PCGStep2's own last workgroup) applies the test and sets the gate word, after which the remaining launches of the loop return at
    // once; delta, r, z stay as of the break.  One read-back per GN step (iterations done, delta.J^T J delta, delta.b, new cost) instead of
    // the reference's blocking 4-byte copy per PCG iteration (fetchQ :1146-1150); iteration counts and costs are pinned against the oracle's
    // host-side test (tests/test_gpu_parity.py::test_lm_device_side_zeta_matches_the_oracle).
    hipStream_t s = ctx.stream;
    const int L = sp.lIterations, B = 2, QS = 1, T0 = 2 * L + 4, T1 = 2 * L + 5;
    // One row slab of a multi-GPU run (flat form, solver_dist.cpp): the energy-independent kernels run on the owned rows' sub-vectors [o, o + n),
    // p is kept current on the ghost rows too ([oe, oe + ne)), and every reduction is made global where it is produced: per PCG iteration one
    // exchange for alphaD and one for [betaN, q | ghost rows of z]; all ranks see the same scalars, so gate and trust region cannot diverge.
    const bool slab = dist_ != nullptr;
    // A launch of this rank that fails: on one GPU the step ends (return 0).  One slab of several must not leave the collective sequence: it goes on
    // issuing every exchange of the step with poisoned payloads, skips its own launches, and the error becomes everybody's at the cost evaluation
    // at the end of the step (solver_dist.cpp, DistState::failed).
    bool failed = false;
    auto skip = [&]() { return failed || (slab && dist_->failed); };
    auto check = [&](int rc, const char* what) {
        if (rc >= 0 || skip()) return;
        if (slab) dist_fail("%s failed (%d)", what, rc); else { set_error("%s failed (%d)", what, rc); failed = true; }
    };
    if (ensure_lm_vectors()) check(-1, "allocating the LM vectors");
    if (failed) return 0;
    const long o = slab ? dist_->rowlen * dist_->row0 : 0, n = slab ? dist_->rowlen * (dist_->row1 - dist_->row0) : v_.n;
    const long oe = slab ? dist_->rowlen * (dist_->row0 - dist_->top) : 0, ne = slab ? dist_->rowlen * (dist_->row1 + dist_->bot - (dist_->row0 - dist_->top)) : v_.n;
    auto global = [&](int j) { return slab ? dist_sum_slot(j) : 0; };                         // nonzero: the collective itself failed
    auto global_rows = [&](int j, float* vec) { return slab ? dist_sum_and_rows(j, vec) : 0; };
    const bool pc = plugin->use_preconditioner();
    const bool fold_ctc = plugin->apply_adds_ctc();
    // PCGStep3 folded into the apply too (one GPU; plugins that offer it; THALLO_LM_FOLD_P=0: A/B, read_ab_switches)
    const bool fold_p = lm_fold_p_ && fold_ctc && !slab && plugin->apply_folds_pupdate() && v_.p[1] != nullptr;
    // the zeta test by PCGStep2's last workgroup (one GPU): one launch less per iteration.  A slab needs the GLOBAL q first.
    if (!slab && ensure_sums_buffer()) return 0;
    const bool zeta_in_step2 = !slab;
    float* lmst = (float*)scratch_.ptr + 16;                          // 8 words: Q0, gate, iterations done, | dJJd, db, new cost
    const unsigned* gate = reinterpret_cast<const unsigned*>(lmst) + 1;
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    if (sp.nIter == 0) { radius_ = sp.trust_region_radius; decrease_factor_ = sp.radius_decrease_factor; }   // :1185-1186 (copied at init)
    cur_ = 0;
    int nb = 0;
    // (round 6: plugins whose PCGInit1 launch also finalises the diagonal -- shape_from_shading on packed planes, one GPU; THALLO_LM_FOLD_STEP=0: the two launches, A/B)
    const bool fold_init = !slab && lm_fold_step_ && plugin->init_folds_lm_diagonal();
    if (fold_init && !skip()) {
        nb = plugin->pcg_init_lm(ctx, v_, cur_, radius_, sp.min_lm_diagonal, sp.max_lm_diagonal, sp.nIter == 0 ? 1 : 0, slot(B));
        check(nb, "PCGInit1 (+ PCGFinalizeDiagonal) launch");
    }
    if (!fold_init && !skip()) { nb = plugin->pcg_init(ctx, v_, cur_, slot(B)); check(nb, "PCGInit1 launch"); }        // r, raw diag (v_.diag); delta = 0
    if (!fold_init && !skip()) {
        TimedLaunch t(ctx, "PCGFinalizeDiagonal");                    // :1596-1604 (alphaN restarts from 0)
        nb = thallo_hip_lm_finalize_diagonal(v_.diag + o, v_.SSq + o, v_.CtC + o, v_.pre + o, v_.r + o, v_.b + o, v_.z + o, n, radius_, sp.min_lm_diagonal, sp.max_lm_diagonal,
                                             sp.nIter == 0 ? 1 : 0, pc ? 1 : 0, slot(B), s);
        check(nb, "PCGFinalizeDiagonal launch");
    }
    if (failed) return 0;
    if (!skip()) set_nb(B, nb);
    if (global_rows(B, v_.z)) return 0;                               // alphaN_0 over all ranks; ghost rows of z
    if (!skip()) check(thallo_hip_lm_state_reset(lmst, s), "LM state reset");       // (delta = 0 -> q = 0, :965)
    if (failed) return 0;
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    float* p = v_.p[0];
    int k_done = 0;
    thallo_hip_lm_set_gate(gate); ctx.gate = gate;
    bool coll_failed = false;                                         // a collective itself failed: nothing left to stay in step with
    bool model_cost_done = false;                                     // the two sums of the model cost are already in slots T0 / T1 (one launch behind the one-launch loop)
    // One launch per LM iteration (one GPU; plugins that offer it: shape_from_shading's marching kernel; lIterations within one residual-reset period, where the reset
    // -- it falls on the last iteration -- changes nothing that is read afterwards): the vector update, PCGStep3, (J^T J + CtC) p, all sums and the zeta test in
    // pcg_iter_lm; behind the loop the one update of delta it still owes.  THALLO_LM_FOLD_P=0: the reference-shaped loop (A/B).
    const int period = sp.residual_reset_period > 0 ? sp.residual_reset_period : (1 << 30);
    // Plugins whose iteration can start from a reset residual (lm_iter_after_reset; bundle adjustment's three-launch form) run any L: every residual_reset_period-th iteration
    // is followed by the reference's reset (:1653-1657) as launches of its own, below.
    // round 6: the loop, the zeta test, the owed update of delta, the model cost and the update of the unknowns in ONE resident launch (plugins whose state fits the chip's
    // registers: shape_from_shading at the size of the reference's data set); lIterations within one residual-reset period.  THALLO_RESIDENT=0: the launches below (A/B).
    const bool resident_lm = !slab && fold_init && lm_fold_step_ && lm_fold_p_ && fold_ctc && one_kernel_ && plugin->resident_lm_ok() && L >= 1 && L <= period;
    const bool one_kernel_lm = !resident_lm && !slab && lm_fold_p_ && fold_ctc && plugin->lm_one_kernel() && L >= 1 && (L <= period || plugin->lm_iter_after_reset()) && v_.p[1] != nullptr &&
                               ensure_iter_buffers() == 0;
    // ... and on a row slab of a multi-GPU run (device-side transport; plugins whose pcg_iter_lm keeps the ghost rows current): the launch stores partials only, ONE
    // exchange per LM iteration carries the 13 sums and the boundary rows of the new A p, finishes alphaD_k, betaN_k, q_{k+1} and applies the zeta test
    // (thallo_hip_dist_xrows_lm).  Ghost rows of r and M^-1 are fetched once per step.
    const bool one_kernel_lm_slab = slab && dist_->flat && dist_->xrows_now && lm_fold_p_ && fold_ctc && plugin->lm_one_kernel_slab() && L >= 1 && L <= period &&
                                    v_.p[1] != nullptr && v_.r2 != nullptr && v_.Ap2 != nullptr && v_.s12b != nullptr;
    if (one_kernel_lm_slab) {
        if (global_rows(-1, v_.r) || global_rows(-1, v_.pre)) return 0;
        cur_ = 0;
        for (int k = 0; k < L; ++k) {
            const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
            const thallo_fin_t fin = { sum(jN), nullptr, nullptr, nullptr };
            if (!skip()) {
                nb = plugin->pcg_iter_lm(ctx, v_, cur_, k == 0, sum(k ? jN - 2 : jN), sum(k ? jD - 2 : jD), sum(jN), slot(jD), fin, lmst, k, sp.q_tolerance);
                check(nb, "PCGIteration (LM) launch");
            }
            if (!skip()) set_nb(jD, nb);
            float* Ao = v_.Abuf(cur_ ^ 1);
            cur_ ^= 1;
            if (dist_xrows_lm(Ao, jN, jD, jB, nb, lmst, k)) { coll_failed = true; break; }
            k_done = k + 1;
        }
        if (!coll_failed && !skip()) {
            TimedLaunch t(ctx, "PCGUpdate");
            check(thallo_hip_lm_owed_delta(v_.delta + o, v_.p[1] + o, v_.p[0] + o, n, scal(B), scal(B + 1), 2, lmst, L, s), "PCGUpdate (owed delta) launch");
        }
    }
    if (resident_lm) {
        cur_ = 0;
        finish(B);
        nb = plugin->pcg_resident_lm(ctx, v_, L, sum(B), scal(B + 1), lmst, sp.q_tolerance, slot(T0), slot(T1));
        check(nb, "PCGLoopResident (LM) launch");
        if (!failed) {
            for (int k = 0; k < L; ++k) { const int jD = B + 2 * k + 1, jB = jD + 1; set_nb(jD, 1); fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1; }
            set_nb(T0, nb); set_nb(T1, nb);
            model_cost_done = true; k_done = L; resident_used_ = true;
        }
    }
    if (one_kernel_lm) {
        cur_ = 0;
        {   // alphaN_0 as a word (the owed-delta launch reads the scalars of iteration "done - 1" by address)
            check(thallo_hip_finish_sum(partial_sum(B), scal(B), s), "alphaN_0 sum");
            if (!failed) fin_[B] = 1;
        }
        bool after_reset = false; int reset_nb = 0;
        // THALLO_FIN_IN_KERNEL unset and the plugin's iteration is several launches (lm_iter_defers_finish: bundle adjustment): the applyJTJ launches of iteration k leave partials
        // only and the flat update of iteration k + 1 finishes them -- words, q, the zeta test -- in every workgroup (thallo_hip_pcg_update_lm_fin); iterations that a residual
        // reset follows (the reset needs alpha_k first) and the last one finish in their own launch.  Q0 / Q1 alternate between words 0 and 6 of the state by parity.
        const bool defer = fin_deferred_ && plugin->lm_iter_defers_finish();
        auto own_finish = [&](int k) { return !defer || k == L - 1 || (((k + 1) % period) == 0 && k + 1 < L); };
        int nb_prev = 0;
        for (int k = 0; k < L && !failed; ++k) {
            const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
            const thallo_fin_t fin = { sum(jN), own_finish(k) ? v_.fin_tickets : nullptr, scal(jD), scal(jB) };
            const bool finish_prev = defer && k > 0 && !own_finish(k - 1);
            const thallo_sum_t aD_prev_partials = { slot(k ? jD - 2 : jD), nb_prev };
            ctx.lm_defer_aD_word = finish_prev ? scal(jD - 2) : nullptr; ctx.lm_defer_bN_word = finish_prev ? scal(jN) : nullptr;
            ctx.lm_q_in = defer ? ((k & 1) ? 6 : 0) : 0; ctx.lm_q_out = defer ? ((k & 1) ? 0 : 6) : 0;
            // behind a reset the iteration's first launch gets betaN_{k-1} as the reset's PARTIALS and replaces the word (the expansion's value) by their sum; everything
            // later -- this iteration's finish, the next one's alpha, the owed update of delta, alpha_beta_trace -- reads the word, which is valid either way (a loop that the
            // zeta test ended AT the reset iteration never ran the reset: its word is the expansion's)
            const thallo_sum_t bn_partials = { slot(jN), reset_nb };
            ctx.lm_reset_bn_word = after_reset ? scal(jN) : nullptr;
            nb = plugin->pcg_iter_lm(ctx, v_, cur_, k == 0, sum(k ? jN - 2 : jN), finish_prev ? aD_prev_partials : sum(k ? jD - 2 : jD), after_reset ? bn_partials : sum(jN), slot(jD), fin,
                                     lmst, k, sp.q_tolerance);
            ctx.lm_reset_bn_word = nullptr; after_reset = false;
            ctx.lm_defer_aD_word = ctx.lm_defer_bN_word = nullptr; ctx.lm_q_in = ctx.lm_q_out = 0;
            nb_prev = nb;
            check(nb, "PCGIteration (LM) launch");
            if (failed) break;
            set_nb(jD, nb); fin_[jD] = 1; set_nb(jB, 1); fin_[jB] = 1;
            cur_ ^= 1;
            k_done = k + 1;
            if (((k + 1) % period) == 0 && k + 1 < L) {
                // residual reset (:1653-1657; on the last iteration it changes nothing that is read afterwards): delta_{k+1} = delta_k + alpha_k p_k now,
                // r = b - (J^T J + CtC) delta and the partials of betaN_k = r . M^-1 r (the next iteration's first launch adds them up, replaces the expansion's word by
                // the sum and forms p_{k+1} only); q_{k+1} and the zeta test stay the launch's own -- the same quantity, and the gate word it may have set ends these
                // launches too.
                { TimedLaunch t(ctx, "PCGStep2"); check(thallo_hip_lm_step2_first_half(v_.delta, v_.p[cur_], n, sum(jN), sum(jD), s), "PCGStep2 (first half) launch"); }
                if (failed) break;
                nb = plugin->lm_reset_residual(ctx, v_, slot(jB));
                check(nb, "residual reset launch");
                if (failed) break;
                reset_nb = nb;
                after_reset = true;
            }
        }
        // round 6: the owed update of delta, the model cost's applyJTJ and its dot product in ONE launch (plugins that offer it); delta moves to the other buffer
        model_cost_done = !failed && lm_fold_step_ && plugin->lm_model_cost_one_launch();
        if (model_cost_done) {
            nb = plugin->lm_model_cost(ctx, v_, scal(B), scal(B + 1), 2, lmst, L, slot(T0), slot(T1), true);      // (... and savePreviousUnknowns + PCGLinearUpdate)
            check(nb, "PCGModelCost launch");
            if (!failed) { set_nb(T0, nb); set_nb(T1, nb); std::swap(v_.delta, v_.Adelta); }
        } else
        if (!failed) {      // p_k lives in p[1] for even k, p[0] for odd k
            TimedLaunch t(ctx, "PCGUpdate");
            check(thallo_hip_lm_owed_delta(v_.delta, v_.p[1], v_.p[0], n, scal(B), scal(B + 1), 2, lmst, L, s), "PCGUpdate (owed delta) launch");
        }
    }
    for (int k = 0; !resident_lm && !one_kernel_lm && !one_kernel_lm_slab && k < L && !failed && !coll_failed; ++k) {
        const int jN = B + 2 * k, jD = jN + 1, jB = jN + 2;
        if (!skip()) {
            if (fold_p) {                                             // PCGStep3 + PCGStep1 + PCGStep1_Finish in one launch; p ping-pongs between the two buffers
                float* pn = p == v_.p[0] ? v_.p[1] : v_.p[0];
                ctx.lm_ctc = v_.CtC;
                nb = plugin->apply_jtj_pupdate(ctx, v_.z, p, pn, v_.Ap, slot(jD), k == 0, sum(k ? jN - 2 : jN), sum(jN));
                ctx.lm_ctc = nullptr;
                check(nb, "PCGStep1 (+ PCGStep3) launch");
                p = pn;
            } else {
                {   TimedLaunch t(ctx, "PCGStep3");                   // p = z + beta p  (k = 0: p = z)
                    check(thallo_hip_pcg_pupdate(v_.z + oe, p + oe, p + oe, nullptr, ne, k == 0, sum(k ? jN - 2 : jN), sum(k ? jD - 2 : jD), sum(jN), s), "PCGStep3 launch");
                }
                if (fold_ctc) {                                       // PCGStep1 + PCGStep1_Finish in one launch: (J^T J + CtC) p ; alphaD
                    ctx.lm_ctc = v_.CtC;
                    nb = plugin->apply_jtj(ctx, p, v_.Ap, slot(jD));
                    ctx.lm_ctc = nullptr;
                    check(nb, "PCGStep1 launch");
                } else {
                    nb = plugin->apply_jtj(ctx, p, v_.Ap, slot(T0));  // PCGStep1 (J^T J p)
                    check(nb, "PCGStep1 launch");
                    if (!skip()) {
                        TimedLaunch t(ctx, "PCGStep1_Finish");        // + CtC p ; alphaD
                        nb = thallo_hip_lm_step1_finish(v_.Ap + o, v_.CtC + o, p + o, n, slot(jD), s);
                        check(nb, "PCGStep1_Finish launch");
                    }
                }
            }
            if (!skip()) set_nb(jD, nb);
        }
        if (failed) break;
        if (global(jD)) { coll_failed = true; break; }
        int nbq = 0; bool zeta_done = false;
        const bool reset = ((k + 1) % sp.residual_reset_period) == 0;     // :1653-1657
        if (reset) {
            if (!skip()) { TimedLaunch t(ctx, "PCGStep2"); check(thallo_hip_lm_step2_first_half(v_.delta + o, p + o, n, sum(jN), sum(jD), s), "PCGStep2 (first half) launch"); }
            if (global_rows(-1, v_.delta)) { coll_failed = true; break; }          // (slab: applyJTJ reads delta on the ghost rows)
            if (!skip()) {
                TimedLaunch t(ctx, "PCGStep2");
                ctx.lm_ctc = fold_ctc ? v_.CtC : nullptr;
                nb = plugin->apply_jtj(ctx, v_.delta, v_.Adelta, slot(T0));         // computeAdelta (+ CtC delta)
                ctx.lm_ctc = nullptr;
                check(nb, "computeAdelta launch");
                if (!skip() && !fold_ctc) check(thallo_hip_lm_step1_finish(v_.Adelta + o, v_.CtC + o, v_.delta + o, n, slot(T1), s), "computeAdelta (CtC) launch");
                if (!skip()) { nb = thallo_hip_lm_step2_second_half(v_.r + o, v_.b + o, v_.Adelta + o, v_.pre + o, v_.z + o, v_.delta + o, n, slot(jB), slot(QS), s); check(nb, "PCGStep2 (second half) launch"); }
                nbq = nb;
            }
        } else if (!skip()) {
            TimedLaunch t(ctx, "PCGStep2");
            if (zeta_in_step2) nb = thallo_hip_pcg_step2_full_zeta(v_.delta + o, p + o, v_.r + o, v_.Ap + o, v_.pre + o, v_.z + o, v_.b + o, n, sum(jN), sum(jD), slot(jB), slot(QS),
                                                                    v_.fin_tickets, k, sp.q_tolerance, lmst, s);
            else nb = thallo_hip_pcg_step2_full(v_.delta + o, p + o, v_.r + o, v_.Ap + o, v_.pre + o, v_.z + o, v_.b + o, n, sum(jN), sum(jD), slot(jB), slot(QS), 1, s);
            check(nb, "PCGStep2 launch");
            nbq = nb; zeta_done = zeta_in_step2;
        }
        if (failed) break;
        if (!skip()) { set_nb(jB, nb); set_nb(QS, nbq); }
        if (slab) { bool zd = false; if (dist_two_sums_and_rows(QS, jB, v_.z, lmst, k, &zd)) { coll_failed = true; break; } zeta_done = zeta_done || zd; }   // q and betaN over all ranks; ghost rows of z (device-side transport: + the zeta test)
        k_done = k + 1;
        if (!zeta_done && !skip()) {
            TimedLaunch t(ctx, "PCGZeta");
            check(thallo_hip_lm_zeta(sum(QS), k, sp.q_tolerance, lmst, s), "PCGZeta launch");
        }
    }
    thallo_hip_lm_set_gate(nullptr); ctx.gate = nullptr;
    if (failed || coll_failed) return 0;
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    // model_cost_change = cost - 0.5|F + J delta|^2 = delta.b - 0.5 delta.(J^T J delta)   (b = -J^T F; thallo.t:3845-3865
    // expanded algebraically, which also avoids the reference's cancellation between two large sums)
    if (global_rows(-1, v_.delta)) return 0;
    if (!model_cost_done) {
        if (!skip()) { nb = plugin->apply_jtj(ctx, v_.delta, v_.Adelta, slot(T0)); check(nb, "model cost: applyJTJ launch"); }
        if (!skip()) set_nb(T0, nb);
        if (!skip()) { nb = thallo_hip_dot(v_.delta + o, v_.b + o, n, slot(T1), s); check(nb, "model cost: dot launch"); }
        if (!skip()) set_nb(T1, nb);
    }
    if (failed) return 0;
    // the two sums over all ranks, into the step's report.  Device-side transport: ONE exchange carries both and writes them where the report is read from (it was two
    // exchanges and two one-wave launches that copied their words there; three launches fewer, though the step's time did not move: 37.5 us per PCG iteration either way on a
    // 2048 x 256 shape_from_shading slab -- that stretch of the step is bound by the host's launch rate, not by the launches' 4 us each)
    const bool two_in_one = slab && dist_->xrows_now;
    if (two_in_one) {
        const thallo_sum_t dummy = { (const float*)dist_->send.ptr, 1 };
        if (dist_xrows(nullptr, false, 0, skip() ? dummy : partial_sum(T0), skip() ? (const float*)dist_->send.ptr : slot(T1), nullptr, skip() ? 1 : nb_[T1], lmst + 3, lmst + 4, nullptr, 0)) return 0;
    } else if (global(T0) || global(T1)) return 0;
    const auto& imgs = plugin->unknown_images();
    if (!skip()) {
        if (!two_in_one) {
            thallo_hip_finish_sum(sum(T0), lmst + 3, s);
            thallo_hip_finish_sum(sum(T1), lmst + 4, s);
        }
        long off = 0;                                                 // savePreviousUnknowns :915-920
        for (size_t k = 0; !model_cost_done && k < imgs.size(); ++k) {
            HIP_OK(hipMemcpyAsync(v_.prevX + off, plugin->unknown_ptr((int)k), imgs[k].n_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
            off += imgs[k].n_floats;
        }
        if (!model_cost_done) linear_update_tail(0, false);           // PCGLinearUpdate: X += delta (the owned rows of a slab); the one-launch model cost has done both
    }
    if (slab && dist_exchange_unknown_rows()) return 0;
    bool failed_before_cost_exchange = false;
    {   // cost after the step, into the same report: ONE blocking read per GN step
        int nbc = 0;
        if (!skip()) { nbc = plugin->cost(ctx, slot(0)); check(nbc, "cost kernel launch"); }
        if (failed) return 0;
        if (!skip()) set_nb(0, nbc);
        failed_before_cost_exchange = slab && dist_->failed;          // (then this rank's payload in the exchange below is poisoned: every rank's new cost is NaN)
        if (global(0)) return 0;
        if (!skip()) thallo_hip_finish_sum(sum(0), lmst + 5, s);
    }
    // The step's outcome decides the trust region on every rank: agree on whether every rank got through it before anyone acts on its report.  On the host transport that
    // is one more collective per step.  On the device-side transport a failed rank's exchanges are POISONED (dist_xrows: NaN payloads), so the new cost -- a rank-ordered
    // sum, the same bits on every rank -- is non-finite on EVERY rank if any rank failed before the cost exchange: only then do the ranks agree through the host (every rank
    // takes the same branch); a rank that fails behind that exchange poisons the next step's.  (One blocking collective less per LM step: ~35 us of a 455-us step on a
    // 2048 x 256 shape_from_shading slab.)
    auto agree_all = [&]() -> bool {
        bool all = false;
        if (dist_agree(!dist_->failed, all)) return false;
        if (!all) {
            const std::string mine = dist_->failed ? last_error() : "";
            if (dist_->failed) set_error("distributed: this rank failed (%s); every rank stops", mine.c_str()); else set_error("distributed: another rank reported a failure; every rank stops");
            ready_ = false; dist_->stopped = true;
            return false;
        }
        return true;
    };
    const bool late_agree = slab && dist_->xrows_now;
    if (slab && !late_agree && !agree_all()) return 0;
    float rep[8] = { 0 };
    {   float* hw = host_words();
        HIP_OK(hipMemcpyAsync(hw ? hw : rep, lmst, sizeof(rep), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (hw) memcpy(rep, hw, sizeof(rep));
    }
    if (late_agree && (failed_before_cost_exchange || !std::isfinite(rep[5])) && !agree_all()) return 0;      // (a failed rank skipped its own cost launches: its rep[5] says nothing)
    if (resident_used_) {     // (the LM step's resident launch: its waits are bounded; one that ran out voids THIS step -- nothing of its report can be trusted)
        resident_used_ = false;
        unsigned pm[5] = { 0, 0, 0, 0, 0 };
        if (plugin->resident_status(ctx, 1, pm) != 0) {
            plugin->resident_disable();
            set_error("%s: a bounded wait inside the resident PCG kernel ran out (wait kind %u, workgroup %u, wave %u, index %u, tag %u): this step is void; "
                      "the plan runs one launch per PCG iteration from now on (Thallo_ProblemInit and solve again)", plugin->name(), pm[0], pm[1], pm[2], pm[3], pm[4]);
            return 0;
        }
    }
    { int frozen_at; memcpy(&frozen_at, &rep[2], sizeof(int)); unsigned fz; memcpy(&fz, &rep[1], sizeof(fz)); if (fz) k_done = frozen_at; }
    return lm_accept_or_revert(rep[3], rep[4], rep[5], k_done, ev_fin, ev_iter);
}

// The end of an LM step (gauss_newton.t:1707-1753): step quality from the model cost change delta.b - 0.5 delta.(J^T J delta) and the new cost; accept (grow the trust region,
// Ceres' rule) or revert the unknowns and shrink it.  Host logic on scalars that every rank of a multi-GPU run holds bit for bit, so the ranks decide alike.
int Plan::lm_accept_or_revert(float dJJd, float db, float newCost, int k_done, int ev_fin, int ev_iter)
{
    hipStream_t s = ctx.stream;
    const auto& imgs = plugin->unknown_images();
    last_l_iters = k_done;
    const float model_cost_change = db - 0.5f * dJJd;
    const float cost_change = prev_cost_ - newCost;
    const float relative_decrease = cost_change / model_cost_change;
    if (ip.verbosityLevel > 0) printf(" cost=%g new cost=%g model_cost_change=%g rho=%g radius=%g pcg=%d\n", prev_cost_, newCost, model_cost_change, relative_decrease, radius_, k_done);
    bool stop = false;
    if (cost_change >= 0 && relative_decrease > sp.min_relative_decrease) {      // :1715-1732
        if (cost_change <= prev_cost_ * sp.function_tolerance) stop = true;
        else {
            const double tmp_factor = 1.0 - std::pow(2.0 * (double)relative_decrease - 1.0, 3.0);
            radius_ = (float)((double)radius_ / std::fmax(1.0 / 3.0, tmp_factor));
            radius_ = std::fmin(radius_, sp.max_trust_region_radius);
            decrease_factor_ = 2.0f;
            prev_cost_ = newCost;
        }
    } else {                                                                      // :1733-1749
        long off = 0;
        for (size_t k = 0; k < imgs.size(); ++k) {                               // revertUpdate
            HIP_OK(hipMemcpyAsync(plugin->unknown_ptr((int)k), v_.prevX + off, imgs[k].n_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
            off += imgs[k].n_floats;
        }
        plugin->unknowns_written();
        radius_ = radius_ / decrease_factor_;
        decrease_factor_ = 2.0f * decrease_factor_;
        if (radius_ < sp.min_trust_region_radius) { sp.trust_region_radius = 10e4f; stop = true; }
    }
    if (!stop) sp.trust_region_radius = radius_;                                  // :1751
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    if (stop) { finalize(); return 0; }
    sp.nIter++;
    return 1;
}

int Plan::alpha_beta_trace(float* out_pairs, int cap)
{   // the partial slots of the last step are still resident: recompute alpha_k, beta_k from them
    const int L = last_l_iters, B = 2;
    if (!ok_ || L <= 0) return 0;
    if (trace_.bytes < (size_t)L * 2 * sizeof(float) && trace_.alloc((size_t)L * 2 * sizeof(float))) return 0;
    for (int k = 0; k < L; ++k)
        thallo_hip_alpha_beta(sum(B + 2 * k), sum(B + 2 * k + 1), sum(B + 2 * k + 2), (float*)trace_.ptr + 2 * k, ctx.stream);
    const int n = L < cap ? L : cap;
    HIP_OK(hipMemcpyAsync(out_pairs, trace_.ptr, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx.stream));
    HIP_OK(hipStreamSynchronize(ctx.stream));
    return L;
}

}  // namespace thallo
