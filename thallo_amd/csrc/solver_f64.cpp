// solver_f64.cpp -- Thallo_InitializationParameters::doublePrecision = 1: the Gauss-Newton step and its PCG loop on double vectors.
//
// The reference's double mode (precision.t:3-6: thallo_float = double; "switch to double to check for precision issues in the solver") compiles the
// same solver with doubles everywhere.  Here it is the reference-shaped UNFUSED loop (gauss_newton.t:1545-1785, GN branch; the LM branch is dead code in the
// reference as shipped, thallo.t:463; step_lm below runs it when ThalloX_EnableLM asked for it) over the kernels the front-end generates from the .t with thallo_float = double (dsl_plugin.cpp) and the
// energy-independent double kernels of pcg_kernels_f64.hip.  Scalars never leave the device inside a step; the cost is one 8-byte read-back like the
// reference's (gauss_newton.t:1128-1136).  Solver parameters stay floats in both modes (gauss_newton.t:200-216).  Single GPU.
#include "solver.hpp"
#include "solver_f64.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>

namespace thallo {

PlanF64::PlanF64(EnergyPlugin* pl, const Thallo_InitializationParameters& ip_) : plugin(pl), p64(pl->f64()), ip(ip_)
{
    memset(&summary, 0, sizeof(summary));
    if (!p64) { set_error("%s has no double-precision form", plugin->name()); return; }
    v_.n = plugin->n_unknowns();
    v_.n_alloc = thallo_hip_vector_elems(v_.n);
    double** vecs[] = { &v_.delta, &v_.r, &v_.z, &v_.Ap, &v_.pre, &v_.p };
    for (int i = 0; i < 6; ++i) {
        if (bufs_[i].alloc((size_t)v_.n_alloc * sizeof(double))) { set_error("out of device memory for the solver vectors (%ld doubles each)", v_.n_alloc); return; }
        if (hipMemset(bufs_[i].ptr, 0, (size_t)v_.n_alloc * sizeof(double)) != hipSuccess) return;
        *vecs[i] = (double*)bufs_[i].ptr;
    }
    // partial slots: 0 cost, 1 alphaN, 2 alphaD, 3 betaN; words behind them
    if (parts_.alloc((size_t)(8 * THALLO_HIP_MAX_PARTIALS + 16) * sizeof(double))) return;
    ctx.timer = &ktimer;
    timer_.enabled = ip.timingLevel >= 1;
    if (ip.timingLevel >= 2) ktimer.period = 1;
    if (ip.timingLevel >= 3) ktimer.invasive = true;
    ok_ = true;
}

PlanF64::~PlanF64()
{
    hipDeviceSynchronize();
    delete plugin;
}

void PlanF64::set_param(const char* name, const void* value)
{   // gauss_newton.t:1828-1844 (floats and ints in both precisions)
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
void PlanF64::get_param(const char* name, void* value)
{
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

double PlanF64::compute_cost()
{
    const int nb = p64->cost64(ctx, slot(0));
    if (nb < 0) { set_error("cost kernel launch failed (%d)", nb); return NAN; }
    if (thallo_hip_f64_finish(slot(0), nb, word(0), ctx.stream) < 0) { set_error("cost reduction launch failed"); return NAN; }
    double f = 0.0;
    if (hipMemcpyAsync(&f, word(0), sizeof(double), hipMemcpyDeviceToHost, ctx.stream) != hipSuccess || hipStreamSynchronize(ctx.stream) != hipSuccess) { set_error("cost read-back failed"); return NAN; }
    return f;
}

void PlanF64::init(void** params)
{   // gauss_newton.t:1166-1198
    if (!ok_) return;
    finalized_ = false;
    timer_.cleanup();
    ev_total_ = timer_.start("Total", ctx.stream);
    ready_ = false;
    if (plugin->bind(params)) { set_error("%s: parameter binding failed", plugin->name()); return; }
    if (plugin->prepare(ctx)) { const std::string why = last_error(); set_error("%s: prepare failed: %s", plugin->name(), why.c_str()); return; }
    ready_ = true;
    sp.nIter = 0;
    prev_cost_ = compute_cost();
    printf("Initial cost: %g\n", prev_cost_);
}

void PlanF64::finalize()
{   // gauss_newton.t:1200-1212
    prev_cost_ = compute_cost();
    if (ip.verbosityLevel > 0) printf("final cost=%g\n", prev_cost_);
    hipDeviceSynchronize();
    timer_.stop(ev_total_, ctx.stream);
    timer_.evaluate(&summary, ip.verbosityLevel > 0, &ktimer);
    timer_.cleanup();
    finalized_ = true;
}

double PlanF64::cost()
{
    if (!ok_ || !ready_) return 0.0;
    if (!finalized_) prev_cost_ = compute_cost();
    return prev_cost_;
}

int PlanF64::step(void** params)
{   // gauss_newton.t:1545-1785, GN branch, one launch per reference kernel
    if (!ok_ || !ready_) return 0;
    if (plugin->bind(params)) { set_error("%s: parameter binding failed", plugin->name()); return 0; }
    if (!use_lm_) plugin->unknowns_changed();      // (Gauss-Newton steps derive everything from the unknowns as they are now: solver.cpp Plan::step)
    if (sp.nIter >= sp.nIterations) { if (!finalized_) finalize(); return 0; }
    if (sp.lIterations < 0) { set_error("lIterations = %d is negative", sp.lIterations); if (!finalized_) finalize(); return 0; }
    hipStream_t s = ctx.stream;
    const int ev_iter = timer_.start("Nonlinear Iteration", s);
    if (use_lm_) return step_lm(ev_iter);
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    enum { AN = 1, AD = 2, BN = 3 };
    auto fail = [&](const char* what, int rc) { set_error("%s launch failed (%d)", what, rc); if (!finalized_) finalize(); return 0; };
    int nb = p64->pcg_init64(ctx, v_, slot(AN));
    if (nb < 0) return fail("PCGInit1", nb);
    if ((nb = thallo_hip_f64_finish(slot(AN), nb, word(AN), s)) < 0) return fail("PCGInit1_Finish", nb);
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    for (int k = 0; k < sp.lIterations; ++k) {
        nb = p64->apply_jtj64(ctx, v_, v_.p, v_.Ap, slot(AD));                         // PCGStep1 (+ _Finish): Ap = J^T J p, alphaD = p . Ap
        if (nb < 0) return fail("PCGStep1", nb);
        if ((nb = thallo_hip_f64_finish(slot(AD), nb, word(AD), s)) < 0) return fail("PCGStep1_Finish", nb);
        {   TimedLaunch t(ctx, "PCGStep2");                                             // alpha; delta += alpha p; r -= alpha Ap; z = M^-1 r; betaN = z . r
            nb = thallo_hip_f64_step2(v_.delta, v_.r, v_.z, v_.p, v_.Ap, v_.pre, v_.n, word(AN), word(AD), slot(BN), s);
            if (nb < 0) return fail("PCGStep2", nb);
            if ((nb = thallo_hip_f64_finish(slot(BN), nb, word(BN), s)) < 0) return fail("PCGStep2_Finish", nb);
        }
        {   TimedLaunch t(ctx, "PCGStep3");                                             // beta = betaN / alphaN; p = z + beta p; alphaN <- betaN
            if ((nb = thallo_hip_f64_step3(v_.p, v_.z, v_.n, word(BN), word(AN), s)) < 0) return fail("PCGStep3", nb);
            if (hipMemcpyAsync(word(AN), word(BN), sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("PCGStep3", -1);
        }
    }
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    {   TimedLaunch t(ctx, "PCGLinearUpdate");                                          // gauss_newton.t:901-906
        const auto& imgs = plugin->unknown_images();
        long off = 0;
        for (size_t k = 0; k < imgs.size(); ++k) {
            if ((nb = thallo_hip_f64_linear_update(p64->unknown_ptr64((int)k), v_.delta + off, imgs[k].n_floats, s)) < 0) return fail("PCGLinearUpdate", nb);
            off += imgs[k].n_floats;
        }
        plugin->unknowns_written();          // (what the plugin derived from the unknowns -- materialized computed arrays -- is stale)
    }
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    return out_of_time() ? 0 : 1;
}

// gauss_newton.t:1767-1779, behind every step of either branch: the time budget of the solve
bool PlanF64::out_of_time()
{
    if (!(sp.max_solver_time_in_seconds > 0.0f) || ev_total_ < 0) return false;
    hipStream_t s = ctx.stream;
    hipEvent_t q = nullptr; hipEventCreate(&q); hipEventRecord(q, s); hipEventSynchronize(q);
    float ms = 0.0f; hipEventElapsedTime(&ms, timer_.events[ev_total_].start, q); hipEventDestroy(q);
    if (ms / 1000.0f > sp.max_solver_time_in_seconds) { finalize(); return true; }
    return false;
}

int PlanF64::step_lm(int ev_iter)
{   // gauss_newton.t:1545-1785 with every UsesLambda() branch taken, in the reference's own shape: one launch per reference kernel, the zeta test on the host after a blocking
    // read of q per PCG iteration (fetchQ :1146-1150, :1666-1686) -- a diagnostic mode, not a hot path (the float LM loop of solver.cpp keeps the host out of the PCG loop).
    hipStream_t s = ctx.stream;
    enum { AN = 1, AD = 2, BN = 3, Q = 4, T0 = 5, T1 = 6 };
    auto fail = [&](const char* what, int rc) { set_error("%s launch failed (%d)", what, rc); if (!finalized_) finalize(); return 0; };
    if (!v_.diag) {
        double** vecs[] = { &v_.diag, &SSq_, &CtC_, &b_, &Adelta_, &prevX_ };
        for (int i = 0; i < 6; ++i) {
            if (lm_bufs_[i].alloc((size_t)v_.n_alloc * sizeof(double)) || hipMemset(lm_bufs_[i].ptr, 0, (size_t)v_.n_alloc * sizeof(double)) != hipSuccess) {
                set_error("out of device memory for the LM vectors (%ld doubles each)", v_.n_alloc); v_.diag = nullptr; if (!finalized_) finalize(); return 0;
            }
            *vecs[i] = (double*)lm_bufs_[i].ptr;
        }
    }
    const int ev_setup = timer_.start("Nonlinear Setup", s);
    if (sp.nIter == 0) { radius_ = sp.trust_region_radius; decrease_factor_ = sp.radius_decrease_factor; }      // :1185-1186
    int nb = p64->pcg_init64(ctx, v_, slot(AN));                                       // r = -J^T F, the raw diagonal (v_.diag); delta = 0
    if (nb < 0) return fail("PCGInit1", nb);
    {   TimedLaunch t(ctx, "PCGFinalizeDiagonal");                                     // :929-969, :1596-1604
        nb = thallo_hip_f64_lm_finalize_diagonal(v_.diag, SSq_, CtC_, v_.pre, v_.r, b_, v_.z, v_.n, radius_, sp.min_lm_diagonal, sp.max_lm_diagonal, sp.nIter == 0 ? 1 : 0,
                                                 plugin->use_preconditioner() ? 1 : 0, slot(AN), s);
        if (nb < 0) return fail("PCGFinalizeDiagonal", nb);
        if ((nb = thallo_hip_f64_finish(slot(AN), nb, word(AN), s)) < 0) return fail("PCGFinalizeDiagonal", nb);
    }
    timer_.stop(ev_setup, s);
    const int ev_lin = timer_.start("Linear Solve", s);
    const size_t bytes = (size_t)v_.n * sizeof(double);
    double Q0 = 0.0;                                                                   // delta = 0 -> q = 0 (:965)
    int k_done = 0;
    for (int k = 0; k < sp.lIterations; ++k) {
        {   TimedLaunch t(ctx, "PCGStep3");                                             // p = z + beta p (k = 0: p = z); alphaN <- betaN
            if (k == 0) { if (hipMemcpyAsync(v_.p, v_.z, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("PCGStep3", -1); }
            else {
                if ((nb = thallo_hip_f64_lm_step3(v_.p, v_.z, v_.n, word(BN), word(AN), s)) < 0) return fail("PCGStep3", nb);
                if (hipMemcpyAsync(word(AN), word(BN), sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("PCGStep3", -1);
            }
        }
        nb = p64->apply_jtj64(ctx, v_, v_.p, v_.Ap, slot(AD));                         // PCGStep1
        if (nb < 0) return fail("PCGStep1", nb);
        {   TimedLaunch t(ctx, "PCGStep1_Finish");                                      // + CtC p; alphaD
            if ((nb = thallo_hip_f64_lm_step1_finish(v_.Ap, CtC_, v_.p, v_.n, slot(AD), s)) < 0) return fail("PCGStep1_Finish", nb);
            if ((nb = thallo_hip_f64_finish(slot(AD), nb, word(AD), s)) < 0) return fail("PCGStep1_Finish", nb);
        }
        int nq = 0;
        const bool reset = sp.residual_reset_period > 0 && ((k + 1) % sp.residual_reset_period) == 0;      // :1653-1657
        if (reset) {
            TimedLaunch t(ctx, "PCGStep2");
            if ((nb = thallo_hip_f64_lm_step2_first_half(v_.delta, v_.p, v_.n, word(AN), word(AD), s)) < 0) return fail("PCGStep2 (first half)", nb);
            if ((nb = p64->apply_jtj64(ctx, v_, v_.delta, Adelta_, slot(T0))) < 0) return fail("computeAdelta", nb);
            if ((nb = thallo_hip_f64_lm_step1_finish(Adelta_, CtC_, v_.delta, v_.n, slot(T0), s)) < 0) return fail("computeAdelta (CtC)", nb);
            if ((nb = thallo_hip_f64_lm_step2_second_half(v_.r, b_, Adelta_, v_.pre, v_.z, v_.delta, v_.n, slot(BN), slot(Q), s)) < 0) return fail("PCGStep2 (second half)", nb);
            nq = nb;
        } else {
            TimedLaunch t(ctx, "PCGStep2");
            if ((nb = thallo_hip_f64_lm_step2(v_.delta, v_.r, v_.z, v_.p, v_.Ap, v_.pre, b_, v_.n, word(AN), word(AD), slot(BN), slot(Q), s)) < 0) return fail("PCGStep2", nb);
            nq = nb;
        }
        if (thallo_hip_f64_finish(slot(BN), nq, word(BN), s) < 0 || thallo_hip_f64_finish(slot(Q), nq, word(Q), s) < 0) return fail("PCGStep2_Finish", -1);
        k_done = k + 1;
        double Q1 = 0.0;                                                               // fetchQ
        if (hipMemcpyAsync(&Q1, word(Q), sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail("fetchQ", -1);
        const double zeta = (double)(k + 1) * (Q1 - Q0) / Q1;
        if (!std::isfinite(Q1) || !std::isfinite(zeta) || zeta < (double)sp.q_tolerance) break;
        Q0 = Q1;
    }
    timer_.stop(ev_lin, s);
    const int ev_fin = timer_.start("Nonlinear Finish", s);
    // model_cost_change = delta . b - 0.5 delta . (J^T J delta)   (b = -J^T F; thallo.t:3845-3865 expanded algebraically, as solver.cpp does)
    if ((nb = p64->apply_jtj64(ctx, v_, v_.delta, Adelta_, slot(T0))) < 0) return fail("model cost: applyJTJ", nb);
    if (thallo_hip_f64_finish(slot(T0), nb, word(T0), s) < 0) return fail("model cost", -1);
    if ((nb = thallo_hip_f64_dot(v_.delta, b_, v_.n, slot(T1), s)) < 0 || thallo_hip_f64_finish(slot(T1), nb, word(T1), s) < 0) return fail("model cost: dot", nb);
    const auto& imgs = plugin->unknown_images();
    {   TimedLaunch t(ctx, "PCGLinearUpdate");                                          // savePreviousUnknowns :915-920, then X += delta
        long off = 0;
        for (size_t k = 0; k < imgs.size(); ++k) {
            if (hipMemcpyAsync(prevX_ + off, p64->unknown_ptr64((int)k), (size_t)imgs[k].n_floats * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("savePreviousUnknowns", -1);
            if ((nb = thallo_hip_f64_linear_update(p64->unknown_ptr64((int)k), v_.delta + off, imgs[k].n_floats, s)) < 0) return fail("PCGLinearUpdate", nb);
            off += imgs[k].n_floats;
        }
        plugin->unknowns_written();
    }
    double rep[2] = { 0.0, 0.0 };
    if (hipMemcpyAsync(&rep[0], word(T0), sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipMemcpyAsync(&rep[1], word(T1), sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
        return fail("model cost read-back", -1);
    if (hipStreamSynchronize(s) != hipSuccess) return fail("model cost read-back", -1); // rep[] is host stack memory: in place before anything below can return or read it
    const double newCost = compute_cost();
    const double model_cost_change = rep[1] - 0.5 * rep[0];
    const double cost_change = prev_cost_ - newCost;
    const double relative_decrease = cost_change / model_cost_change;
    if (ip.verbosityLevel > 0) printf(" cost=%g new cost=%g model_cost_change=%g rho=%g radius=%g pcg=%d\n", prev_cost_, newCost, model_cost_change, relative_decrease, radius_, k_done);
    bool stop = false;
    if (cost_change >= 0 && relative_decrease > (double)sp.min_relative_decrease) {     // :1715-1732
        if (cost_change <= prev_cost_ * (double)sp.function_tolerance) stop = true;
        else {
            const double tmp_factor = 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3.0);
            radius_ = radius_ / std::fmax(1.0 / 3.0, tmp_factor);
            radius_ = std::fmin(radius_, (double)sp.max_trust_region_radius);
            decrease_factor_ = 2.0;
            prev_cost_ = newCost;
        }
    } else {                                                                            // :1733-1749 revertUpdate
        long off = 0;
        for (size_t k = 0; k < imgs.size(); ++k) {
            if (hipMemcpyAsync(p64->unknown_ptr64((int)k), prevX_ + off, (size_t)imgs[k].n_floats * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("revertUpdate", -1);
            off += imgs[k].n_floats;
        }
        plugin->unknowns_written();
        radius_ = radius_ / decrease_factor_;
        decrease_factor_ = 2.0 * decrease_factor_;
        if (radius_ < (double)sp.min_trust_region_radius) { sp.trust_region_radius = 10e4f; stop = true; }
    }
    if (!stop) sp.trust_region_radius = (float)radius_;                                 // :1751
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    if (stop) { finalize(); return 0; }
    sp.nIter++;
    return out_of_time() ? 0 : 1;
}

}  // namespace thallo
