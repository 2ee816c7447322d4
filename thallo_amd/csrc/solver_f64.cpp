// solver_f64.cpp -- Thallo_InitializationParameters::doublePrecision = 1: the Gauss-Newton step and its PCG loop on double vectors.
//
// The reference's double mode (precision.t:3-6: thallo_float = double; "switch to double to check for precision issues in the solver") compiles the
// same solver with doubles everywhere.  Here it is the reference-shaped UNFUSED loop (gauss_newton.t:1545-1785, GN branch; the LM branch is dead code in the
// reference as shipped, thallo.t:463) over the kernels the front-end generates from the .t with thallo_float = double (dsl_plugin.cpp) and the
// energy-independent double kernels of pcg_kernels_f64.hip.  Scalars never leave the device inside a step; the cost is one 8-byte read-back like the
// reference's (gauss_newton.t:1128-1136).  Solver parameters stay floats in both modes (gauss_newton.t:200-216).  Single GPU, Gauss-Newton only.
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
    if (parts_.alloc((size_t)(4 * THALLO_HIP_MAX_PARTIALS + 16) * sizeof(double))) return;
    ctx.timer = &ktimer;
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
    if (sp.nIter >= sp.nIterations) { if (!finalized_) finalize(); return 0; }
    if (sp.lIterations < 0) { set_error("lIterations = %d is negative", sp.lIterations); if (!finalized_) finalize(); return 0; }
    hipStream_t s = ctx.stream;
    const int ev_iter = timer_.start("Nonlinear Iteration", s);
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
    }
    sp.nIter++;
    timer_.stop(ev_fin, s);
    timer_.stop(ev_iter, s);
    if (sp.max_solver_time_in_seconds > 0.0f && ev_total_ >= 0) {   // gauss_newton.t:1767-1779
        hipEvent_t q = nullptr; hipEventCreate(&q); hipEventRecord(q, s); hipEventSynchronize(q);
        float ms = 0.0f; hipEventElapsedTime(&ms, timer_.events[ev_total_].start, q); hipEventDestroy(q);
        if (ms / 1000.0f > sp.max_solver_time_in_seconds) { finalize(); return 0; }
    }
    return 1;
}

}  // namespace thallo
