/*
 * Thallo.h -- public C API of the MI355X-native Thallo solver backend (libThallo.so).
 *
 * ABI-compatible with the reference's API/release/include/Thallo.h:1-105: same 3 opaque
 * handle types, same 6-int initialization struct passed BY VALUE, same 13 entry points and
 * the same performance-summary structs, so an application written against the reference
 * (examples/shared/ThalloSolver.h:43-106, every tests/<x>/main.cpp) relinks unchanged.
 * Callers in C++ wrap the include in extern "C" exactly as they do for the reference; this
 * header also does it itself.
 *
 * Behavioural notes (where this backend differs from the reference are marked DIFF):
 *  - problem specifications: the bundled energies are recognised from the .t file and run on
 *    precompiled gfx950 plugins; an unrecognised .t makes Thallo_ProblemPlan print a
 *    diagnostic and return NULL (the reference prints a Lua traceback and returns NULL,
 *    API/src/thallo.t:1431-1432).
 *  - DIFF: cpuOnly=1 is rejected (NULL state); this build has no CPU fallback on purpose.
 *  - doublePrecision=1 (API/src/precision.t:3-6: thallo_float = double): every problem file goes through the
 *    front-end, whose kernels are generated with thallo_float = double, and the reference-shaped double loop drives
 *    them (Gauss-Newton, and with ThalloX_EnableLM the LM branch; one GPU).  Unknowns / thallo_float arrays are
 *    doubles, arrays and Params declared `float` stay floats, solver parameters stay floats (gauss_newton.t:200-216).
 *    DIFF: the ThalloX_ multi-GPU extensions are not available in this mode.
 *  - DIFF: GPU errors are reported on stderr and surface as NULL / 0 returns; the process is
 *    never exit()ed (reference: API/src/cuda_util.t:103-118).
 */
#ifndef THALLO_H
#define THALLO_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct Thallo_State   Thallo_State;
typedef struct Thallo_Plan    Thallo_Plan;
typedef struct Thallo_Problem Thallo_Problem;

/* Set once per Thallo_NewState; an all-zero struct is the fast default. */
struct Thallo_InitializationParameters {
    int doublePrecision;   /* 1: thallo_float = double (generated kernels, see above) */
    int verbosityLevel;    /* 0 quiet, >=1 prints solver log + timing table at the end of a solve */
    int timingLevel;       /* 0 no timing recorded (no events in the stream: Thallo_GetPerformanceSummary reports zeros), 1 coarse events (eight per solver step -- each one
                              a barrier packet between two launches: ~30 us per step, which small problems notice), 2 per-kernel hipEvents, 3 additionally device-syncs around them */
    int threadsPerBlock;   /* accepted for compatibility; kernels carry their own tuned shapes */
    int useAutoscheduler;  /* bundled energies: their hand-written plugin (a fixed schedule) either way.  Generated plugins (a .t file no plugin recognises, or
                              THALLO_FRONTEND=generate): 1 = the autoscheduler's lowering where it applies -- residuals that live on their unknowns' grid become
                              unknown-wise gather kernels, one merged kernel per (domain, schedule) group, no atomics (API/src/thallo.t:5173-5190,5273-5306);
                              0 = the reference's default residual-wise scatter (thallo.t:4096-4098).  Every example passes 1 (examples/.../main.cpp:37). */
    int cpuOnly;           /* must be 0: Thallo_NewState returns NULL for 1.  The product has no CPU path by design (the reference's needs Terra, API/src/cpu_cuda.t);
                              the CPU restatement under oracle/ is test infrastructure and is never linked or loaded by this library. */
};
typedef struct Thallo_InitializationParameters Thallo_InitializationParameters;

Thallo_State* Thallo_NewState(Thallo_InitializationParameters params);

/* solverkind: "gauss_newton" or "levenberg_marquardt" (anything else: NULL). */
Thallo_Problem* Thallo_ProblemDefine(Thallo_State* state, const char* filename, const char* solverkind);
void            Thallo_ProblemDelete(Thallo_State* state, Thallo_Problem* problem);

/* dimensions[i] is the extent of the i-th Dims() name of the .t; the pointer is retained. */
Thallo_Plan* Thallo_ProblemPlan(Thallo_State* state, Thallo_Problem* problem, unsigned int* dimensions);
void         Thallo_PlanFree(Thallo_State* state, Thallo_Plan* plan);

/* Solver parameters by name; *value has the parameter's own type: int for nIterations,
 * lIterations, residual_reset_period; float for the rest (API/src/gauss_newton.t:200-216). */
void Thallo_SetSolverParameter(Thallo_State* state, Thallo_Plan* plan, const char* name, void* value);
void Thallo_GetSolverParameter(Thallo_State* state, Thallo_Plan* plan, const char* name, void* value);

/* problemparams[i] belongs to input index i of the .t: device pointers for Unknown/Array/Sparse,
 * host pointers to the scalar for Param.  Unknown buffers are updated in place. */
void   Thallo_ProblemSolve(Thallo_State* state, Thallo_Plan* plan, void** problemparams);
void   Thallo_ProblemInit(Thallo_State* state, Thallo_Plan* plan, void** problemparams);
int    Thallo_ProblemStep(Thallo_State* state, Thallo_Plan* plan, void** problemparams);   /* 0 = finished */
double Thallo_ProblemCurrentCost(Thallo_State* state, Thallo_Plan* plan);

struct Thallo_PerformanceEntry {
    unsigned int count;
    double minMS;
    double maxMS;
    double meanMS;
    double stddevMS;
};
typedef struct Thallo_PerformanceEntry Thallo_PerformanceEntry;

struct Thallo_PerformanceSummary {
    Thallo_PerformanceEntry total;
    Thallo_PerformanceEntry nonlinearIteration;
    Thallo_PerformanceEntry nonlinearSetup;
    Thallo_PerformanceEntry linearSolve;
    Thallo_PerformanceEntry nonlinearResolve;
};
typedef struct Thallo_PerformanceSummary Thallo_PerformanceSummary;

void Thallo_GetPerformanceSummary(Thallo_State* state, Thallo_Plan* plan, Thallo_PerformanceSummary* summary);

/* ------------------------------------------------------------------------------------------
 * Extensions (not in the reference; prefixed ThalloX_).  Used by bench.py and the tests.
 * ------------------------------------------------------------------------------------------ */

/* Launch on `stream` (a hipStream_t) instead of the legacy default stream. */
void ThalloX_SetStream(Thallo_Plan* plan, void* stream);

/* Bracket every `period`-th launch of each kernel with hipEvents (0 = off).  Cheap enough to
 * leave on inside a timed region; results via ThalloX_GetKernelStat. */
void ThalloX_SetKernelSampling(Thallo_Plan* plan, int period);
/* Enumerate sampled kernels: returns 0 while index is valid. launches = all launches,
 * samples = timed launches, total_ms = sum over timed launches.  Synchronises the device. */
int  ThalloX_GetKernelStat(Thallo_Plan* plan, int index, const char** name, long* launches, long* samples, double* total_ms);
void ThalloX_ResetKernelStats(Thallo_Plan* plan);
/* ... of the same entry: how many of the sampled launches carried their own start / stop events (the kernel's begin-to-end time, without the dispatch gap that
   events recorded around a launch include) and their total; 0 samples where the kernel's shim does not offer it */
int ThalloX_GetKernelStatOwn(Thallo_Plan* plan, int index, long* samples, double* total_ms);

/* alpha/beta of every PCG iteration of the most recent Step (tests): writes up to cap pairs,
 * returns the number of PCG iterations run. */
int  ThalloX_GetAlphaBetaTrace(Thallo_Plan* plan, float* out_pairs, int cap);

/* Run the Levenberg-Marquardt branch as the reference's text describes it (gauss_newton.t, every UsesLambda()
 * branch).  Off by default even for "levenberg_marquardt": the reference as shipped executes plain GN for that
 * kind (thallo.t:463 matches "LM", which no accepted kind string contains).  Call before Thallo_ProblemInit. */
void ThalloX_EnableLM(Thallo_Plan* plan, int enable);
/* Between two LM steps the branch lives on state of the previous step's end, like the reference (pd.hd.prevCost, gauss_newton.t:1710,1732) -- here also what a
 * plugin derived from the unknowns at that point (shape_from_shading's precomputed planes).  A caller that rewrites the unknowns or the input images IN PLACE
 * between two LM steps (same pointers: parameter binding cannot see it) says so with this call: derived state is dropped and the carried cost re-evaluated.
 * Gauss-Newton steps derive everything from the unknowns as they are at the call and never need it. */
void ThalloX_UnknownsChanged(Thallo_State* state, Thallo_Plan* plan);

/* Name of the plugin a plan runs ("image_warping", "laplacian_image", ...). */
const char* ThalloX_PlanEnergyName(Thallo_Plan* plan);

/* The J^T J p schedule the plan's plugin runs: "matrix-free" (hand-written plugins), or for generated plugins "per residual" (each residual's own
 * schedule lines), "dense [JtJ]p", "sparse [[Jt][J]]p", "dense direct solve". */
const char* ThalloX_PlanScheduleName(Thallo_Plan* plan);

/* 1 if the most recent Thallo_ProblemInit on this plan succeeded (parameters bound, plugin prepared), else 0: Init itself returns void. */
int ThalloX_PlanReady(Thallo_Plan* plan);

/* Last error message of this thread ("" if none). */
const char* ThalloX_LastError(void);

/* What a problem file asks for with its `r.<residual>.J:set_materialize(true)` / `.JtJ:set_materialize(true)` lines
 * (API/src/thallo.t:5661-5690): 0 = matrix-free, 1 = `[Jt][[J]p]` (J and J^T as CSR, two SpMVs per PCG iteration), 2 = `[[Jt][J]]p`
 * (J^T J formed once, one SpMV); -1 = no plugin for the file.  Honoured by the two Laplacian energies (constant J). */
int ThalloX_ProblemFileSchedule(const char* filename);
unsigned long long ThalloX_ProblemFileHash(const char* filename, char* energy_out, int cap);
/* FNV-1a-64 of the translation unit the front-end generates from the file, residual names aside (0: outside the supported subset, ThalloX_LastError).
 * Two files with the same value state the same energy.  A file with a bundled energy's declarations runs on that energy's hand-written plugin only if
 * its text is a known one or this value is the bundled file's; otherwise on generated kernels. */
unsigned long long ThalloX_ProblemFileUnitHash(const char* filename);

/* The mini front-end (thallo_amd/csrc/dsl.hpp) run on a .t file without a device: what = 0 its declarations as text, 1 the HIP translation unit it
 * generates (residual-wise cost / evalJTF / applyJTJ / applyJ / applyJt kernels per named residual).  Returns the text's length (the copy is truncated
 * to cap - 1) or -1 (ThalloX_LastError).  A Plan on a file no hand-written plugin recognises -- or on any file under THALLO_FRONTEND=generate -- compiles
 * that unit with hipRTC and runs it. */
int ThalloX_FrontendText(const char* filename, int what, char* out, int cap);
/* ... given the problem's dimensions (the array Thallo_ProblemPlan takes): needed by files that use Sum, which is expanded for those sizes */
int ThalloX_FrontendTextDims(const char* filename, int what, const unsigned* dims, char* out, int cap);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU, one process per GPU (the reference is single-device, API/src/util.t:769-772).
 *
 * Image-stencil problems are split into contiguous ROW SLABS.  Each process makes its Plan for its LOCAL image -- the rows it owns plus one
 * ghost row towards each neighbour -- passes the local buffers to Init / Step as usual, and declares the split once, before
 * Thallo_ProblemInit, with ThalloX_PlanSetDistributed.  From then on Thallo_ProblemInit / Step / CurrentCost are COLLECTIVE: every rank calls
 * them in the same order; costs, alpha and beta are rank-ordered sums, bit-identical on every rank.  Per PCG iteration each rank runs ONE
 * kernel and ONE exchange:
 *   - the caller's all-gather (RCCL: ncclAllGather on `stream`) of [alphaD, N, S1, S2 | first and last owned row of Ap], or
 *   - when device_exchange != 0 and the self-check at the first Init passes on this topology: no host-visible collective at all -- the
 *     kernel stores its scalars into every rank's mailbox and its boundary rows into the neighbours' ghost rows over xGMI
 *     (hipIpc-mapped fine-grained memory) and its last workgroup waits for the peers' granules.
 * Supported: image_warping (one ghost row per neighbour; UrShape on the unit pixel grid, W % 4 == 0; Gauss-Newton; both transports) and
 * shape_from_shading (TWO ghost rows per neighbour; Gauss-Newton and ThalloX_EnableLM; BOTH transports: per PCG iteration one exchange -- the
 * all-gather carries the GN form's sums and rows, and the LM form's alphaD, then betaN + q + the ghost rows of z, in two; on the device-side
 * transport the LM iteration is one launch + ONE mailbox / peer-to-peer exchange that also finishes alphaD, betaN, q and the zeta test,
 * thallo_hip_dist_xrows_lm).
 * Graph energies (arap_mesh_deformation) are split into contiguous VERTEX RANGES instead: every rank makes its Plan for the WHOLE problem, passes the
 * whole (replicated) buffers, and sets row0 / row1 to the vertex range it owns (equal ranges: N % world == 0); per PCG iteration one all-gather
 * of [alphaD, N, S1, S2 | the owned slice of A p]; the unknowns stay replicated bit for bit.  Gauss-Newton, all-gather transport.
 * bundle_adjustment is split into CAMERA SHARDS: every rank makes its Plan for its SUB-INSTANCE -- its cameras (count padded to a multiple of 4 with
 * unobserved cameras), ALL points, the observations of its cameras with camera indices renumbered locally; row0 / row1 are not used.  The camera block
 * of J^T J p is complete on the rank; the point block is a partial sum and is all-reduced (`allreduce`, 3P floats per PCG iteration), after which
 * every rank updates the (replicated) points identically; the scalars: one all-gather of the ranks' camera sums + the point sums every rank
 * computes for itself.  Gauss-Newton and, after ThalloX_EnableLM, Levenberg-Marquardt (round 6: the element-wise LM kernels run on the camera block and
 * the point block separately; accept / revert and the trust region are decided by every rank on identical scalars).
 * Everything else returns an error.
 * ------------------------------------------------------------------------------------------ */
/* Every rank contributes `bytes_per_rank` bytes at `send` and receives world * bytes_per_rank at `recv`, rank order; DEVICE pointers;
 * enqueued on `stream` (a hipStream_t).  Return 0 on success.  world == 1: may be NULL. */
typedef int (*ThalloX_AllGatherFn)(void* user, const void* send, void* recv, long bytes_per_rank, void* stream);
/* In-place sum over all ranks of `count` floats at `buf` (DEVICE pointer), on `stream` (ncclAllReduce, ncclSum).  Only the shard form needs it. */
typedef int (*ThalloX_AllReduceFn)(void* user, void* buf, long count, void* stream);
typedef struct ThalloX_Distributed {
    int rank, world;                 /* world <= 8 (THALLO_DIST_MAX_WORLD) */
    unsigned int row0, row1;         /* owned rows [row0, row1) of the local image; g ghost rows above iff row0 == g, below iff row1 == H_local - g (g = 1 or 2, see below) */
    ThalloX_AllGatherFn allgather;
    void* user;
    int device_exchange;             /* 1: try the mailbox / peer-to-peer exchange (falls back to the all-gather if its self-check fails) */
    unsigned int global_row0;        /* global index of the local image's row 0 (ghost rows included) and the global image height: energies */
    unsigned int global_rows;        /* whose expressions use pixel coordinates (shape_from_shading) need them; 0 0 = not given */
    ThalloX_AllReduceFn allreduce;   /* bundle adjustment (camera shards) only; NULL otherwise */
} ThalloX_Distributed;
/* Graph energies (arap_mesh_deformation) as a REAL vertex partition (round 3; the replicated "vertex range" form stays): the Plan is made for the rank's LOCAL
 * sub-problem -- its owned vertices first (local ids [0, n_own)), then the GHOSTS: vertices owned elsewhere that an owned vertex shares an edge with; the edges are all
 * directed edges with an owned end.  Call this before ThalloX_PlanSetDistributed (whose row0 / row1 are then 0 / n_own).  boundary_units: local ids of this rank's owned
 * vertices that some other rank holds as ghosts, in an order every rank agrees on (ascending global id); ghost_units[g] (local id, >= n_own), filled from position
 * ghost_src_pos[g] of rank ghost_src_rank[g]'s boundary list.  Host arrays, copied.  Per PCG iteration the ranks exchange [alphaD | N, S1, S2 | A p at the boundary
 * vertices] in one all-gather; vectors, memory and the vector updates are local-sized (thallo_amd/distributed_graph.py builds the lists from the global edge list). */
int ThalloX_PlanSetGhostExchange(Thallo_Plan* plan, int n_boundary, const int* boundary_units, int n_ghost, const int* ghost_units, const int* ghost_src_rank,
                                 const int* ghost_src_pos);
/* Collective.  0 on success, -1 on error (ThalloX_LastError); every rank gets the same answer. */
int ThalloX_PlanSetDistributed(Thallo_Plan* plan, const ThalloX_Distributed* cfg);
/* The collectives INSIDE the library (round 3; no callback, no host-language hop in the PCG loop): rank 0 makes a 128-byte RCCL unique id, the application
 * hands it to every rank by whatever channel it has (MPI, a file, torch.distributed), and each rank gives it to its plan BEFORE ThalloX_PlanSetDistributed
 * (collective: ncclCommInitRank, the rank's GPU current).  A NULL allgather / allreduce in ThalloX_Distributed then means ncclAllGather / ncclAllReduce on the
 * plan's stream.  RCCL is bound at run time (dlopen: the copy already in the process if there is one); -1 + ThalloX_LastError when it is not there.
 * ThalloX_RcclSelfTest: a one-rank communicator moves a small buffer through both collectives (0 = fine). */
int ThalloX_RcclAvailable(void);      /* 1: librccl can be bound in this process (rank-local; all-reduce it before the collective ThalloX_PlanUseRccl) */
int ThalloX_RcclUniqueId(unsigned char* id_out128);
int ThalloX_PlanUseRccl(Thallo_Plan* plan, const unsigned char* id128, int rank, int world);
int ThalloX_RcclSelfTest(void);
/* what the plan's communicator says about itself: out3 = { ncclCommCount, ncclCommCuDevice, ncclCommUserRank }, -1 where there is no communicator or no answer.
 * Diagnostics for the first run on a real node: a count of 1 next to a torch.distributed world of N means the ranks never met. */
void ThalloX_PlanRcclInfo(Thallo_Plan* plan, int out3[3]);
/* JSON text: transport in use ("exchange": "p2p-mailbox" | "allgather"), memory kind of the mapped blocks, self-check outcome. */
const char* ThalloX_PlanDistributedInfo(Thallo_Plan* plan);
/* what = 0: read (and with value != 0 clear) the device-side exchange's error word -- 1 if a bounded mailbox wait timed out since the last clear
 *           (synchronises the stream); what = 1: value 0 switches this plan to the all-gather transport for good (every rank must do the same);
 *           what = 2 (tests): the value-th rank-local launch from now on reports a failure.
 * Returns the word / 0, or -1.
 *
 * Failures across ranks.  Decisions that change the sequence of collectives are unanimous (set-up, Init, the device-side exchange's self-check: a
 * rank that cannot allocate, bind or prepare says so in an agreement, and then EVERY rank returns the error).  A launch or copy that fails on ONE
 * rank in the middle of a step does not end that rank's part of the sequence: it skips its own launches, keeps issuing every collective with a
 * poisoned payload (NaN scalars on every rank) and goes on returning 1 from Thallo_ProblemStep like the others; at the next cost evaluation --
 * the end of Thallo_ProblemSolve, every Thallo_ProblemCurrentCost, every LM step -- all ranks learn of it together, report it, and their plans
 * stop (Step returns 0, the cost is NaN).  What stays rank-local: a failing collective callback itself, and running out of memory for the two
 * message buffers inside ThalloX_PlanSetDistributed before the first collective. */
int ThalloX_DistributedControl(Thallo_Plan* plan, int what, int value);
/* bench: `reps` back-to-back launches of the one-kernel PCG iteration on this rank's slab, without the exchange (kernel time per rank) */
int ThalloX_DistributedKernelOnly(Thallo_Plan* plan, int reps);

#ifdef __cplusplus
}
#endif
#endif
