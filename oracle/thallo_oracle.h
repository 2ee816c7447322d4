/*
 * thallo_oracle.h -- CPU restatement of the Thallo GN/LM + PCG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under thallo_amd/ (the product) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- and there only as the checker / baseline.
 *
 * What it restates (reference paths relative to /root/reference):
 *   driver      API/src/gauss_newton.t:1166-1198 (init), :1545-1785 (step),
 *               :1128-1136 (cost)
 *   kernels     API/src/gauss_newton.t:712-731 (PCGInit1_Finish),
 *               :998-1015 (residual-wise PCGInit1/PCGStep1), :801-843 (PCGStep2),
 *               :889-899 (PCGStep3), :901-906 (PCGLinearUpdate), :929-969 (LM set)
 *   fmap specs  API/src/thallo.t:3536-3569 (applyJTJ), :3867-3908 (evalJTF),
 *               :3911-3937 (computeCtC), :3845-3865 (modelcost), :3939-3949 (cost)
 *   CPU order   API/src/cpu_cuda.t:265-301 (serial, x fastest)
 *
 * Parity pin: tests/golden/minimal_gold.u8 and minimal_graph_gold.u8 (decoded
 * bytes of the reference's tests/minimal/gold.png and tests/minimal_graph/gold.png)
 * are reproduced bit-exactly by this code; see tests/test_oracle_golden.py.
 * Energies without a reference KAT (image_warping, ARAP, SFS, BA) are pinned
 * only through finite-difference Jacobian checks and scipy cross-checks:
 * for those, trajectory parity is "pinned by restatement", stated in DESIGN.md.
 *
 * Formulation: every energy is a *row enumerator* (one residual element ->
 * a few sparse Jacobian rows with hand-derived partials).  cost / evalJTF /
 * applyJTJ / computeCtC / modelcost / CSR export are generic loops over rows,
 * i.e. the reference's residual-wise (scatter) schedule.  The HIP product
 * path uses gather (unknown-wise) kernels, so the two derivations are
 * independent.
 */
#ifndef THALLO_ORACLE_H
#define THALLO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NZ   16
#define ORC_MAX_ROWS 12
#define ORC_MAX_IMG  4

typedef struct {
    int   nnz;
    int   col[ORC_MAX_NZ];   /* flat unknown index = image_offset + C*pix + c (gauss_newton.t:448-451) */
    float val[ORC_MAX_NZ];   /* d r / d unknown */
    float r;                 /* residual value */
} OrcRow;

typedef struct OrcEnergy OrcEnergy;
struct OrcEnergy {
    int      kind;
    unsigned dims[4];
    void**   params;                 /* indexed like the .t Inputs{} */
    int      n_img;                  /* unknown images, declaration order */
    int      img_param[ORC_MAX_IMG];
    int      img_chan[ORC_MAX_IMG];
    long     img_count[ORC_MAX_IMG]; /* elements (pixels / vertices) */
    long     img_off[ORC_MAX_IMG + 1];
    long     n_unknowns;
    long     n_elems;                /* residual elements over all groups */
    int      use_precond;
    float    fconst[8];
    int      iconst[4];
    int    (*rows)(const OrcEnergy*, long elem, OrcRow* out);
    int    (*excluded)(const OrcEnergy*, long flat);
};

enum {
    ORC_LAPLACIAN_IMAGE = 1, /* tests/minimal/laplacian.t            fconst[0]=w_fit, iconst[0]=x-guard variant */
    ORC_LAPLACIAN_GRAPH = 2, /* tests/minimal_graph/laplacian.t      fconst[0]=w_fit */
    ORC_IMAGE_WARPING   = 3, /* examples/image_warping/image_warping.t */
    ORC_ARAP_MESH       = 4, /* examples/arap_mesh_deformation/arap_mesh_deformation.t */
    ORC_BUNDLE_ADJUST   = 5, /* examples/bundle_adjustment/bundle_adjustment.t */
    ORC_SFS             = 6  /* examples/shape_from_shading/shape_from_shading.t */
};

/* solver parameters, names/defaults of gauss_newton.t:41-55 */
typedef struct {
    float min_relative_decrease, min_trust_region_radius, max_trust_region_radius;
    float q_tolerance, function_tolerance, trust_region_radius, radius_decrease_factor;
    float min_lm_diagonal, max_lm_diagonal;
    int   residual_reset_period, nIterations, lIterations;
    int   use_lm;        /* 0: GN (what the reference executes for both kind strings) */
    int   float_sums;    /* 1: accumulate the dot products in float, serial order (cpu_cuda.t);
                            0: accumulate in double (order-independent limit of the GPU's
                               nondeterministic float reduction, util.t:40-50) */
} OrcSolverParams;

void orc_default_params(OrcSolverParams* sp);
/* n > 1: the row loops (cost, evalJTF, applyJTJ, computeCtC, model cost) run on n OpenMP threads (double-accumulator mode only) */
void orc_set_threads(int n);
int orc_get_threads(void);

/* MSVC rand(): s = s*214013+2531011; (s>>16)&0x7fff.  tests/minimal/main.cpp:52-54 */
void orc_msvc_rand_fill(float* out, long n, unsigned seed);

/* Build an energy over host buffers.  fconst/iconst may be NULL. Returns 0 on success. */
int  orc_energy_init(OrcEnergy* e, int kind, const unsigned* dims, void** params,
                     const float* fconst, const int* iconst);

/* fmap pieces (generic over rows) */
double orc_cost(const OrcEnergy* e, int float_sums);
void   orc_eval_jtf(const OrcEnergy* e, float* r /* += -J^T F */, float* pre /* += diag J^T J */);
double orc_apply_jtj(const OrcEnergy* e, const float* p, float* Ap /* += J^T J p */, int float_sums);
void   orc_compute_ctc(const OrcEnergy* e, float inv_radius, float* ctc);
double orc_model_cost(const OrcEnergy* e, const float* delta, int float_sums);
/* CSR export: returns nnz; rowptr has n_rows+1 entries (callers size with orc_count_rows) */
long   orc_count_rows(const OrcEnergy* e, long* nnz_out);
long   orc_export_csr(const OrcEnergy* e, int* rowptr, int* colind, float* vals, float* resid);

/* Full solve: Init + while(Step).  costs[0] = initial cost, costs[k] = cost after GN step k
   (what launchProfiledSolve records, examples/shared/ThalloUtils.h:75-92).
   trace (optional, may be NULL): per PCG iteration alpha,beta appended as pairs, capacity trace_cap pairs.
   Returns number of GN steps taken; unknown buffers in params are updated in place. */
int orc_solve(OrcEnergy* e, const OrcSolverParams* sp, double* costs, int costs_cap,
              float* trace, int trace_cap);

/* Convenience single-call entry for ctypes */
/* PCG iterations per GN / LM step of the last orc_solve (returns the number of steps; fills at most cap) */
int orc_last_pcg_counts(int* out, int cap);
/* LM: the trust region (radius, radius_decrease_factor) as the last orc_solve* call left it */
void orc_last_trust_region(float* radius, float* decrease_factor);
int orc_solve_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst,
                   const OrcSolverParams* sp, double* costs, int costs_cap, float* trace, int trace_cap);

#ifdef __cplusplus
}
#endif
#endif
