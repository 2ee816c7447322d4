// solver_f64.hpp -- the double-precision plan (Thallo_InitializationParameters::doublePrecision = 1); see solver_f64.cpp
#pragma once
#include "solver.hpp"

namespace thallo {

class PlanF64 {
public:
    PlanF64(EnergyPlugin* plugin, const Thallo_InitializationParameters& ip);
    ~PlanF64();
    bool ok() const { return ok_; }
    bool ready() const { return ok_ && ready_; }
    void init(void** params);
    int  step(void** params);
    double cost();
    void enable_lm(bool on) { use_lm_ = on; }      // ThalloX_EnableLM: the LM branch of gauss_newton.t (dead code in the reference as shipped, thallo.t:463)
    void set_param(const char* name, const void* value);
    void get_param(const char* name, void* value);

    EnergyPlugin* plugin;
    EnergyPlugin64* p64;
    SolverParameters sp;
    Thallo_InitializationParameters ip;
    Thallo_PerformanceSummary summary;
    LaunchCtx ctx;
    KernelTimer ktimer;

private:
    bool ok_ = false, ready_ = false, finalized_ = true;
    double prev_cost_ = 0.0;
    Vectors64 v_;
    DeviceBuffer bufs_[6];
    DeviceBuffer lm_bufs_[6];       // LM: raw diagonal, SSq, CtC, b, Adelta, previous unknowns (allocated by the first LM step)
    double *SSq_ = nullptr, *CtC_ = nullptr, *b_ = nullptr, *Adelta_ = nullptr, *prevX_ = nullptr;
    bool use_lm_ = false;
    double radius_ = 0.0, decrease_factor_ = 2.0;
    int step_lm(int ev_iter);
    bool out_of_time();          // max_solver_time_in_seconds behind a step of either branch (finalizes when the budget is spent)
    DeviceBuffer parts_;            // 8 x THALLO_HIP_MAX_PARTIALS partial slots, then 16 scalar words
    CoarseTimer timer_;
    int ev_total_ = -1;
    double* slot(int j) { return (double*)parts_.ptr + (size_t)j * THALLO_HIP_MAX_PARTIALS; }
    double* word(int j) { return (double*)parts_.ptr + (size_t)8 * THALLO_HIP_MAX_PARTIALS + j; }
    double compute_cost();
    void finalize();
};

}  // namespace thallo
