// thallo_harness.hpp -- application-side harness over the public C API (include/Thallo.h), SURVEY.md 8f-1.
//
// Reproduces what every example of the reference does around a solve, in the reference's own artefact formats, so that
// "the cost trajectory matches" is a one-command comparison against a run of the original:
//   finalCosts.json   examples/shared/SolverIteration.h:71-87      (scientific, 20 digits)
//   perf.json         examples/shared/CombinedSolverBase.h:8-92     (scientific, 18 digits; the 5 Thallo_PerformanceSummary entries)
//   results/results_float.csv   SolverIteration.h:30-68            (per-iteration cost and ms of the profiled solve)
//   the profiled solve itself   examples/shared/ThalloUtils.h:75-92 (Init, then Step + device sync + CurrentCost per iteration)
//   solver construction / destruction order   examples/shared/ThalloSolver.h:43-106
// Host C++ only; device memory through the HIP runtime.  Nothing here is on the hot path.
#pragma once
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

extern "C" {
#include "Thallo.h"
}

namespace harness {

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) { std::fprintf(stderr, "HIP error in %s: %s\n", what, hipGetErrorString(e)); std::exit(2); }
}

// A device buffer the application owns (the API takes device pointers for arrays and unknowns: Thallo.h / SURVEY.md 8b).
class DeviceArray {
public:
    DeviceArray() = default;
    explicit DeviceArray(size_t bytes) { resize(bytes); }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    ~DeviceArray() { if (ptr_) (void)hipFree(ptr_); }
    void resize(size_t bytes)
    {
        if (ptr_) (void)hipFree(ptr_);
        ptr_ = nullptr; bytes_ = bytes;
        if (bytes) { hip_check(hipMalloc(&ptr_, bytes), "hipMalloc"); hip_check(hipMemset(ptr_, 0, bytes), "hipMemset"); }
    }
    template <class T> void upload(const std::vector<T>& h)
    {
        if (h.size() * sizeof(T) != bytes_) resize(h.size() * sizeof(T));
        hip_check(hipMemcpy(ptr_, h.data(), bytes_, hipMemcpyHostToDevice), "upload");
    }
    template <class T> std::vector<T> download() const
    {
        std::vector<T> h(bytes_ / sizeof(T));
        hip_check(hipMemcpy(h.data(), ptr_, bytes_, hipMemcpyDeviceToHost), "download");
        return h;
    }
    void zero() { if (ptr_) hip_check(hipMemset(ptr_, 0, bytes_), "hipMemset"); }
    void* data() const { return ptr_; }
    size_t bytes() const { return bytes_; }
private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
};

struct SolverIteration {
    double cost = -std::numeric_limits<double>::infinity();
    double timeInMS = -std::numeric_limits<double>::infinity();
};

// Solver parameters by name; values keep the type the API expects (int for nIterations / lIterations /
// residual_reset_period, float otherwise: gauss_newton.t:200-216).
struct SolverParameters {
    std::map<std::string, unsigned int> ints;
    std::map<std::string, float> floats;
};

class ThalloSolver {
public:
    ThalloSolver(const std::vector<unsigned int>& dims, const std::string& energy_file, const std::string& solver_kind,
                 bool invasive_timing = false, int autoschedule = 1)
    {
        Thallo_InitializationParameters ip;
        std::memset(&ip, 0, sizeof(ip));
        ip.verbosityLevel = 1;
        ip.timingLevel = invasive_timing ? 2 : 1;
        ip.useAutoscheduler = autoschedule;
        std::printf("Thallo Solver Init\n");
        state_ = Thallo_NewState(ip);
        std::printf("Thallo Solver Define\n");
        problem_ = Thallo_ProblemDefine(state_, energy_file.c_str(), solver_kind.c_str());
        std::printf("Thallo Solver Plan\n");
        dims_ = dims;                                   // the plan keeps the pointer (thallo.t:1419): it must outlive the plan
        dims_.resize(10, 0);
        plan_ = problem_ ? Thallo_ProblemPlan(state_, problem_, dims_.data()) : nullptr;
        if (!state_ || !problem_ || !plan_) { std::fprintf(stderr, "could not plan %s\n", energy_file.c_str()); std::exit(3); }
    }
    ThalloSolver(const ThalloSolver&) = delete;
    ThalloSolver& operator=(const ThalloSolver&) = delete;
    ~ThalloSolver()
    {
        if (plan_) Thallo_PlanFree(state_, plan_);
        if (problem_) Thallo_ProblemDelete(state_, problem_);
    }

    // One solve.  profiled: Init + Step loop with a device sync and a cost read after every step, appended to `iters`.
    double solve(const SolverParameters& sp, std::vector<void*>& problem_params, bool profiled, std::vector<SolverIteration>& iters)
    {
        for (auto& kv : sp.ints)   { unsigned int v = kv.second; Thallo_SetSolverParameter(state_, plan_, kv.first.c_str(), &v); }
        for (auto& kv : sp.floats) { float v = kv.second;        Thallo_SetSolverParameter(state_, plan_, kv.first.c_str(), &v); }
        if (profiled) {
            auto t = std::chrono::steady_clock::now();
            auto tick = [&]() { auto n = std::chrono::steady_clock::now(); double ms = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return ms; };
            Thallo_ProblemInit(state_, plan_, problem_params.data());
            hip_check(hipDeviceSynchronize(), "sync");
            SolverIteration it; it.timeInMS = tick(); it.cost = Thallo_ProblemCurrentCost(state_, plan_);
            iters.push_back(it);
            tick();
            while (Thallo_ProblemStep(state_, plan_, problem_params.data())) {
                hip_check(hipDeviceSynchronize(), "sync");
                it.timeInMS = tick(); it.cost = Thallo_ProblemCurrentCost(state_, plan_);
                iters.push_back(it);
                tick();
            }
        } else {
            Thallo_ProblemSolve(state_, plan_, problem_params.data());
        }
        final_cost_ = Thallo_ProblemCurrentCost(state_, plan_);
        Thallo_GetPerformanceSummary(state_, plan_, &summary_);
        return final_cost_;
    }
    double final_cost() const { return final_cost_; }
    const Thallo_PerformanceSummary& summary() const { return summary_; }
    Thallo_Plan* plan() const { return plan_; }
private:
    Thallo_State* state_ = nullptr;
    Thallo_Problem* problem_ = nullptr;
    Thallo_Plan* plan_ = nullptr;
    std::vector<unsigned int> dims_;
    double final_cost_ = std::nan("");
    Thallo_PerformanceSummary summary_{};
};

// ------------------------------------------------------------------------------------------------ artefacts
struct NamedRun {
    std::string name;                       // "ThalloGN" / "ThalloLM"
    double final_cost = std::nan("");
    Thallo_PerformanceSummary perf{};
    std::vector<SolverIteration> iters;
};

inline void write_final_costs(std::ostream& o, const std::string& name, const std::vector<NamedRun>& runs)
{
    o << "{  \"name\" : \"" << name << "\"," << std::endl << "  \"costs\" : {" << std::endl;
    o << std::scientific << std::setprecision(20);
    std::vector<const NamedRun*> ok;
    for (auto& r : runs) if (!std::isnan(r.final_cost)) ok.push_back(&r);
    for (size_t i = 0; i < ok.size(); ++i)
        o << "    \"" << ok[i]->name << "\" : " << ok[i]->final_cost << (i + 1 != ok.size() ? "," : "") << std::endl;
    o << "  }" << std::endl << "}" << std::endl;
}

inline void write_perf_entry(std::ostream& o, const char* label, const Thallo_PerformanceEntry& e, const std::string& ind, bool comma)
{
    auto num = [](double v) { return std::isnan(v) ? 9999999999999999999999.0 : v; };
    o << ind << "\"" << label << "\" : {" << std::endl;
    const std::string in2 = ind + "  ";
    o << in2 << "\"count\" : " << e.count << "," << std::endl;
    o << in2 << "\"minMS\" : " << num(e.minMS) << "," << std::endl;
    o << in2 << "\"maxMS\" : " << num(e.maxMS) << "," << std::endl;
    o << in2 << "\"meanMS\" : " << num(e.meanMS) << "," << std::endl;
    o << in2 << "\"stddevMS\" : " << num(e.stddevMS) << std::endl;
    o << ind << "}" << (comma ? "," : "") << std::endl;
}

inline void write_perf(std::ostream& o, const std::string& name, int autoscheduled, const std::vector<NamedRun>& runs)
{
    o << "{  \"name\" : \"" << name << "\"," << std::endl;
    o << "  \"autoscheduled\" : " << autoscheduled << "," << std::endl << "  \"performance\" : {" << std::endl;
    o << std::scientific << std::setprecision(18);
    for (size_t i = 0; i < runs.size(); ++i) {
        const auto& p = runs[i].perf;
        o << "    \"" << runs[i].name << "\" : {" << std::endl;
        write_perf_entry(o, "total", p.total, "      ", true);
        write_perf_entry(o, "nonlinearIteration", p.nonlinearIteration, "      ", true);
        write_perf_entry(o, "nonlinearSetup", p.nonlinearSetup, "      ", true);
        write_perf_entry(o, "linearSolve", p.linearSolve, "      ", true);
        write_perf_entry(o, "nonlinearResolve", p.nonlinearResolve, "      ", false);
        o << "    }" << (i + 1 != runs.size() ? "," : "") << std::endl;
    }
    o << "  }" << std::endl << "}" << std::endl;
}

// Columns as in the reference (Ceres columns stay, filled with one zero row: there is no Ceres here).
inline void write_results_csv(const std::string& path, const std::vector<SolverIteration>& gn, const std::vector<SolverIteration>& lm)
{
    std::ofstream f(path);
    f << std::scientific << std::setprecision(20);
    const std::string sfx = " (float)";
    f << "Iter, Ceres Error, Thallo(GN) Error" << sfx << ",  Thallo(LM) Error" << sfx << ", Ceres Iter Time(ms), Thallo(GN) Iter Time(ms)" << sfx
      << ", Thallo(LM) Iter Time(ms)" << sfx << ", Total Ceres Time(ms), Total Thallo(GN) Time(ms)" << sfx << ", Total Thallo(LM) Time(ms)" << sfx << std::endl;
    auto padded = [](std::vector<SolverIteration> v) { if (v.empty()) { SolverIteration z; z.cost = 0; z.timeInMS = 0; v.push_back(z); } return v; };
    const auto g = padded(gn), l = padded(lm), c = padded({});
    auto clamp = [](const std::vector<SolverIteration>& v, size_t i) -> const SolverIteration& { return v[i < v.size() ? i : v.size() - 1]; };
    double sg = 0, sl = 0, sc = 0;
    for (size_t i = 0; i < std::max(g.size(), std::max(l.size(), c.size())); ++i) {
        const double tc = i < c.size() ? c[i].timeInMS : 0.0, tg = i < g.size() ? g[i].timeInMS : 0.0, tl = i < l.size() ? l[i].timeInMS : 0.0;
        sc += tc; sg += tg; sl += tl;
        f << i << ", " << clamp(c, i).cost << ", " << clamp(g, i).cost << ", " << clamp(l, i).cost << ", " << tc << ", " << tg << ", " << tl << ", "
          << sc << ", " << sg << ", " << sl << std::endl;
    }
}

inline void save_artefacts(const std::string& name, int autoscheduled, const std::vector<NamedRun>& runs, bool profiled)
{
    { std::ofstream f("finalCosts.json"); write_final_costs(f, name, runs); }
    { std::ofstream f("perf.json"); write_perf(f, name, autoscheduled, runs); }
    write_final_costs(std::cout, name, runs);
    if (profiled) {
        ::mkdir("results", 0755);
        std::vector<SolverIteration> gn, lm;
        for (auto& r : runs) { if (r.name == "ThalloGN") gn = r.iters; if (r.name == "ThalloLM") lm = r.iters; }
        write_results_csv("results/results_float.csv", gn, lm);
    }
}

}  // namespace harness
