// solver.hpp -- Gauss-Newton / Levenberg-Marquardt outer step + PCG inner loop (host driver).
// Restates API/src/gauss_newton.t:1166-1198 (init), :1545-1785 (step), :1200-1212 (finalize),
// :1787-1799 (cost / summary), :1806-1862 (solver parameters), :1963-2071 (makePlan) for the
// fused MI355X schedule documented in DESIGN.md.
#pragma once
#include "plugin.hpp"
#include "rccl_transport.hpp"
#include "../../include/Thallo.h"

namespace thallo {

struct SolverParameters {   // gauss_newton.t:200-216, defaults :41-55
    float min_relative_decrease = 1e-3f;
    float min_trust_region_radius = 1e-32f;
    float max_trust_region_radius = 1e16f;
    float q_tolerance = 0.0001f;
    float function_tolerance = 0.000001f;
    float trust_region_radius = 1e4f;
    float radius_decrease_factor = 2.0f;
    float min_lm_diagonal = 1e-6f;
    float max_lm_diagonal = 1e32f;
    float max_solver_time_in_seconds = 0.0f;
    int residual_reset_period = 10;
    int nIter = 0;
    int nIterations = 10;
    int lIterations = 10;
};

// util.Timer (util.t:446-541) with hipEvents; names as in gauss_newton.t:1173,1563-1564,1611,1690
class CoarseTimer {
public:
    struct Info { std::string name; hipEvent_t start, end; };
    std::vector<Info> events;
    bool enabled = true;      // Thallo_InitializationParameters.timingLevel 0 = "No timing recorded" (Thallo.h): no events at all -- an event record is a barrier packet between two launches
    int  start(const char* name, hipStream_t s);
    void stop(int idx, hipStream_t s);
    void evaluate(Thallo_PerformanceSummary* out, bool print_table, KernelTimer* kt);
    void cleanup();
    ~CoarseTimer() { cleanup(); }
};

// Row-slab multi-GPU state of a Plan (one process per GPU; SURVEY.md 8e -- the reference is single-device, util.t:769-772, so this is
// new design).  The Plan was made for the LOCAL image {W, owned rows + ghost rows}; rows [row0,row1) are owned.  Per PCG iteration ONE
// kernel + ONE exchange: either the caller's all-gather (RCCL) of [alphaD, N, S1, S2 | boundary rows of Ap], or -- after a self-check on
// this very topology -- device mailboxes + peer-to-peer ghost-row stores done by the kernel itself (dist_device.hpp).  solver_dist.cpp.
struct DistState {
    ThalloX_Distributed cfg;
    int W = 0, Hl = 0, row0 = 0, row1 = 0, top = 0, bot = 0, ghost = 1;
    long N = 0, na = 0;
    void* block = nullptr; bool block_ipc = false;      // [r | z | r' | Ap | Ap'] (one allocation peers can map)
    unsigned char handle_block[64], handle_mail[64];
    int mem_kind[2] = { -1, -1 };
    DeviceBuffer send, gath;                             // message buffers (floats), sized for the larger of the two message kinds
    long msg = 0, msg_iter = 0, msg_x = 0;
    thallo_segs_t seg_first_last, seg_top, seg_bot, seg_iter_fl, seg_iter_top, seg_iter_bot;
    // flat form (single-image energies with apply_jtj_sums: shape_from_shading): vectors stay where the Plan allocated them, all-gather transport only
    bool flat = false;
    long rowlen = 0;                                     // floats per image row
    thallo_segs_t seg_rows_fl, seg_rows_top, seg_rows_bot;   // first / last `ghost` owned rows; the ghost rows above / below
    thallo_segs_t seg_rows_first, seg_rows_last;             // ... the first and the last owned rows separately (device-side exchange: one goes up, one goes down)
    thallo_xrows_t xr;                                       // device-side exchange of the flat form (thallo_hip_dist_xrows): inbox geometry, neighbours
    bool xrows_now = false;                                  // the transport dist_sum_slot / dist_sum_and_rows / dist_gn_flat use right now (= p2p_on outside the self-check)
    // range form (graph domains: ARAP): every rank holds the whole problem and FULL-length vectors and owns the contiguous unit range [row0, row1)
    // (units = vertices); equal ranges on all ranks.  pieces = rank 0's owned slice of every plane of the flat vector (rank r's: + r * len)
    bool range = false;
    thallo_segs_t pieces_first, pieces_mine;
    // ... or, partition form (ThalloX_PlanSetGhostExchange): the rank holds its owned units [0, row1) + ghost units; only the boundary units' values travel
    bool part = false;
    DeviceBuffer g_boundary, g_ghost, g_src1, g_src7, g_srcx;   // (g_srcx: the ghosts' sources inside this rank's device-side inbox)
    thallo_units_t u_recvx; long unit_slot = 0;              // device-side exchange of the partition form (thallo_hip_dist_xunits)       // device copies of the index lists; element offsets of the ghosts' sources for the two message kinds
    thallo_units_t u_send, u_recv1, u_recv7;
    long piece_floats = 0;                               // floats a rank owns in a flat vector
    // shard form (bundle adjustment: camera shards): the unknowns [sh_off, sh_off + sh_len) (the points) are replicated; their J^T F / diag / A p are
    // partial sums over the rank's residuals and are all-reduced; sums over them are taken after that, by every rank for itself
    bool shard = false;
    long sh_off = 0, sh_len = 0;
    DeviceBuffer sh_aD, sh_s3;                           // the shared block's partials of an iteration
    DeviceBuffer sh_lm;                                  // ... of the second sum of an LM exchange (q next to betaN, delta.b next to delta.J^T J delta)
    thallo_xreduce_t xa;                                 // device-side all-reduce of the shared block (thallo_hip_dist_allreduce): inbox geometry
    // device-side exchange
    bool want_p2p = false, mapped = false, p2p_on = false, checked = false;
    bool resident_all = false;                           // EVERY rank's slab fits the resident PCG kernel (agreed once per Init: a rank whose last segment cannot be a full one says no)
    void* mail = nullptr; int mail_L = 0;
    long ghost_off = 0;                                  // byte offset of the resident kernel's ghost area inside every rank's mailbox block (0: none)
    DeviceBuffer ctl;
    int defer_state = -1;                                 // the deferred cross-rank finish: -1 not agreed on yet, 0 no, 1 every rank runs it
    DeviceBuffer gs;                                      // two tagged granules: the global words a launch's designated wave publishes for its other waves (deferred cross-rank finish)
    thallo_dist_t d, d_iter[2];
    std::vector<void*> opened;
    std::string info;                                    // JSON: transport, memory kind, self-check outcome
    // A rank-local failure (a launch, a copy, an allocation at Step time) must not end this rank's part of the collective sequence -- the other
    // ranks would wait in the matching all-gather forever.  From the first failure on the rank skips its own launches, keeps issuing every
    // collective of the sequence with a poisoned payload (NaN header: every rank's alpha / beta / cost turn NaN), and the error becomes
    // everybody's at the next cost evaluation (dist_cost carries the flag; Init, Finalize and Thallo_ProblemCurrentCost all end there).
    bool failed = false;
    bool stopped = false;                                // the failure was agreed on: every rank's plan refuses further steps, its cost reads NaN
    int inject = 0;                                      // tests (ThalloX_DistributedControl what = 2): the n-th rank-local launch / copy from now on reports a failure
};

class Plan {
public:
    Plan(EnergyPlugin* plugin, const Thallo_InitializationParameters& ip, bool lm, unsigned* dims);
    ~Plan();
    bool ok() const { return ok_; }
    bool ready() const { return ok_ && ready_; }

    void init(void** params);
    int  step(void** params);
    double cost();
    void set_param(const char* name, const void* value);
    void get_param(const char* name, void* value);
    int  alpha_beta_trace(float* out_pairs, int cap);
    void unknowns_changed();       // the caller rewrote unknowns / inputs in place between two LM steps (ThalloX_UnknownsChanged)
    void enable_lm(bool on);       // extension: run the LM branch the reference text describes (dead as shipped, thallo.t:463)
    bool lm() const { return lm_; }
    // collective over the ranks; before Thallo_ProblemInit.  0 on success (every rank returns the same value)
    int  set_distributed(const ThalloX_Distributed& cfg);
    int  set_ghost_exchange(int n_boundary, const int* boundary_units, int n_ghost, const int* ghost_units, const int* ghost_src_rank, const int* ghost_src_pos);
    // collective; before set_distributed: the plan's own RCCL communicator -- then a NULL all-gather / all-reduce callback means "ncclAllGather / ncclAllReduce on the plan's stream"
    int  use_rccl(const unsigned char* id128, int rank, int world);
    const char* distributed_info() const { return dist_ ? dist_->info.c_str() : ""; }
    void rccl_info(int out[3]);     // what the plan's RCCL communicator says about itself: world, device, rank (-1: no communicator / no answer)
    int  dist_control(int what, int value);
    int  dist_kernel_only(int reps);      // bench: `reps` back-to-back one-kernel iterations on this rank's slab, no exchange

    EnergyPlugin* plugin;
    SolverParameters sp;
    Thallo_InitializationParameters ip;
    Thallo_PerformanceSummary summary;
    LaunchCtx ctx;
    KernelTimer ktimer;
    std::vector<float> ab_trace;   // alpha,beta per PCG iteration of the last step
    int last_l_iters = 0;
    unsigned* dims;

private:
    bool ok_ = false;
    bool ready_ = false;           // Init succeeded (parameters bound, plugin prepared)
    bool lm_ = false;
    bool finalized_ = true;
    float prev_cost_ = 0.0f;
    SolverVectors v_;
    std::vector<DeviceBuffer*> bufs_;
    DeviceBuffer parts_;            // reduction partial slots: (2*L+4) x THALLO_HIP_MAX_PARTIALS floats
    int parts_slots_ = 0;
    DeviceBuffer scratch_;          // a few scalar words (cost, trace)
    DeviceBuffer trace_;
    std::vector<int> nb_;           // partial count per slot
    CoarseTimer timer_;
    float* host_words_ = nullptr;      // 16 pinned host words: the step's read-backs (a cost, the LM report) land here -- a copy into pageable memory is staged and synchronous
    int ev_total_ = -1;
    int cur_ = 0;

    float* slot(int j) { return (float*)parts_.ptr + (size_t)j * THALLO_HIP_MAX_PARTIALS; }
    std::vector<char> fin_;         // slot already reduced to one word (scal(j)) by a 1-wave finish_sum launch
    bool fin_in_kernel_ = true, one_kernel_ = true, batch_delta_ = true, lm_fold_p_ = true, lm_fold_step_ = true;   // A/B switches: read_ab_switches()
    int delta_planes_ = -1;         // THALLO_DELTA_PLANES (-1: unset)
    std::vector<float*> ring_;      // the ring of p planes of the one-kernel GN loop (ring_planes)
    bool ring_possible() const;     // the plugin's iteration takes any p plane, on one GPU or on a row slab of the one-kernel schedule
    int  ring_planes(int L);        // how many planes the loop of L iterations runs on (allocates nothing unless lIterations changed since Init)
    void ring_prepare(int L);       // allocates the ring's planes: at Init, and once per change of lIterations; a memory-limited attempt is cached (ring_tried_)
    int  ring_tried_ = 0, ring_L_ = -1;
    hipStream_t aux_ = nullptr; bool aux_failed_ = false;      // the plan's second stream (background delta updates)
    std::vector<hipEvent_t> aux_events_;
    bool aux_async_ = false;        // THALLO_DELTA_PLANES=N:W: the delta updates of the ring on the second stream, next to the loop
    int aux_workgroups_ = 0;        // share of the chip a background update takes (0: one workgroup per CU -- measured: 64 / 128 too slow, the loop ends up waiting; 320+ put two
                                    // on some CUs, whose marching workgroups then hold everybody's iteration up: 6.4 against 6.0-6.1 ms per GN step)
    bool aux_stream();
    bool fin_deferred_ = true;      // THALLO_FIN_IN_KERNEL unset: the single-reduction GN loop finishes iteration k-1 inside the flat update of iteration k (=1: by the applyJTJ launch's last workgroup; =0: a one-wave launch)
    void read_ab_switches();
    float* scal(int j) { return (float*)parts_.ptr + (size_t)parts_slots_ * THALLO_HIP_MAX_PARTIALS + j; }
    thallo_sum_t partial_sum(int j) { thallo_sum_t s; s.partials = slot(j); s.count = nb_[j]; return s; }
    thallo_sum_t sum(int j) { if (fin_[j]) { thallo_sum_t s; s.partials = scal(j); s.count = 1; return s; } return partial_sum(j); }
    void set_nb(int j, int nb) { nb_[j] = nb; fin_[j] = 0; }
    // Adds the slot's partials once, in the order every consumer would use, so the next kernels read ONE word instead of each of
    // their waves re-adding up to 1024 partials in the prologue (measured: -10 us per PCG iteration at 2048^2)
    // (small launches -- <= 4 partials per lane -- are cheaper to re-add in place than to pay one more launch for)
    void finish(int j) { if (nb_[j] <= 256) return; thallo_hip_finish_sum(partial_sum(j), scal(j), ctx.stream); fin_[j] = 1; }
    int  ensure_slots(int L);
    int  ensure_iter_buffers();
    int  ensure_sums_buffer();
    int  step_gn_expanded(int ev_iter);
    int  step_gn_one_kernel(int ev_iter);
    int  step_gn_resident(int ev_iter);
    bool resident_used_ = false;    // a resident launch ran since the last cost evaluation (its error word is read there)
    float compute_cost();
    float* host_words();
    int   step_gn(int ev_iter);
    int   step_lm(int ev_iter);
    int   lm_accept_or_revert(float dJJd, float db, float newCost, int k_done, int ev_fin, int ev_iter);      // the end of an LM step: accept / revert, trust region (shared by step_lm and the shard form)
    int   step_lm_shard(int ev_iter);                   // solver_dist.cpp: LM on residual shards (bundle adjustment's camera shards)
    int   ensure_lm_vectors();
    float read_sum(int j);
    float radius_ = 1e4f, decrease_factor_ = 2.0f;
    void finalize();
    void linear_update_tail(int L, bool batched);
    // solver_dist.cpp
    DistState* dist_ = nullptr;
    RcclComm* rccl_ = nullptr;
    int  set_distributed_impl(const ThalloX_Distributed& cfg);
    struct GhostSpec { bool given = false; std::vector<int> boundary, ghost, src_rank, src_pos; } ghost_spec_;
    int  dist_ghosts(float* vec, int sum_slot);
    int  dist_allgather(const void* send, void* recv, long bytes);
    int  dist_agree(bool flag, bool& all);
    void dist_fail(const char* fmt, ...);               // first rank-local failure: report it, switch this rank to "collectives only" (DistState::failed)
    bool dist_skip() const { return dist_ && dist_->failed; }
    int  dist_map_peers();
    int  dist_map_peers_flat();
    int  dist_map_mail(long bytes);                     // the mailbox allocation of the flat / shard forms: allocate, exchange handles, map every peer's, agree
    int  dist_xrows_lm(float* vec, int jN, int jD, int jB, int nb, float* lm_state, int k);      // the ONE exchange of a slab's one-launch LM iteration
    int  dist_two_sums_and_rows(int j1, int j2, float* vec, float* zeta_state = nullptr, int zeta_k = 0, bool* zeta_done = nullptr);
    int  dist_xrows(float* vec, bool rows, int mode, thallo_sum_t s, const float* aD_part, const double* s3, int nb, float* out0, float* out1, float* zeta_state = nullptr, int zeta_k = 0);
    int  dist_self_check();
    int  dist_gn(int L, bool p2p);                      // PCGInit + L iterations + linear update + ghost refresh, no bookkeeping
    int  step_gn_slab(int ev_iter);
    int  dist_gn_flat(int L);
    int  dist_gn_range(int L);
    int  dist_gn_shard(int L);                          // shard form: applyJTJ on the rank's residuals, all-reduce of the shared block of A p, one tiny all-gather
    int  dist_allreduce(float* buf, long count);                          // range form: full-length pcg_update + applyJTJ over the owned units + ONE exchange per PCG iteration
    int  dist_replicate(float* vec, int sum_slot);      // every rank's owned pieces of `vec` to every rank (and, sum_slot >= 0, that slot's global sum)                           // flat form: pcg_update + apply_jtj_sums + ONE exchange per PCG iteration
    int  dist_sum_slot(int j);                          // slot j (local partials) -> scal(j) = rank-ordered global sum
    int  dist_sum_and_rows(int j, float* vec);          // ... and the ghost rows of a flat vector from the neighbours' boundary rows (j < 0: rows only)
    int  dist_exchange_unknown_rows();
    float dist_cost();
    void dist_release();
};

}  // namespace thallo
