/*
 * thallo_hip.h -- thin C-ABI shim over the hand-written gfx950 (CDNA4) kernels of the
 * Gauss-Newton / LM + PCG hot path.
 *
 * This is the seam BASELINE.json's north star describes: a host layer (the reference's
 * API/src/gauss_newton.t driver, or this repo's C++ driver in thallo_amd/csrc/solver.cpp)
 * calls these entry points instead of JIT-compiling Terra kernels to PTX
 * (reference: API/src/util.t:797-927 makeGPUFunctions, API/src/cuda_util.t:470 cudacompile).
 * Plain pointers, sizes and PODs only; no C++/torch types.  INTEGRATION.md shows the
 * `terralib.includec("thallo_hip.h")` binding a Thallo maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what
 *    the reference launches on: util.t:769-772);
 *  - every function returns 0 / the number of partials written (>= 0) on success and a negative
 *    hipError_t on failure; it never exits the process (the reference's cd() macro exits:
 *    cuda_util.t:103-118);
 *  - solver vectors are flat float arrays of n_unknowns elements laid out like the reference's
 *    UnknownType: unknown images concatenated in declaration order, each AoS
 *    (flat = image_offset + channels*element + channel; thallo.t:1104-1125, gauss_newton.t:448-451).
 *    They must be allocated with thallo_hip_vector_elems(n) elements (padding for 16-byte access);
 *  - reductions: a kernel that reduces writes ONE partial per workgroup with a plain store;
 *    consumers take the quantity as a thallo_sum_t {partials, count} and add the partials in a
 *    fixed order.  This replaces the reference's warp-shuffle + red.global.add.f32 into one word
 *    (util.t:40-50, cuda_util.t:287-289,430-439) and its three per-iteration 4-byte memsets and
 *    one 4-byte D2D copy (gauss_newton.t:1528-1541,1665).
 */
#ifndef THALLO_HIP_H
#define THALLO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* thallo_stream_t;

#define THALLO_HIP_MAX_PARTIALS 1024

/* A deferred deterministic sum: sum_{i<count} partials[i]. count==1 is a plain scalar word
   (e.g. the result of a cross-rank all-reduce). */
typedef struct thallo_sum_t {
    const float* partials;
    int          count;
} thallo_sum_t;

/* Up to 8 contiguous pieces [off, off+len) (floats) of a flat solver vector: e.g. one image row of the
   image_warping layout = {2*W*row, 2*W} in the Offset plane + {2*N + W*row, W} in the Angle plane. */
typedef struct thallo_segs_t {
    long off[8];
    long len[8];
    int  n;
} thallo_segs_t;

/* Multi-GPU (one process per GPU) device-side exchange: every rank owns a mailbox of 8-byte granules
   {float value | seq << 32} indexed [slot][source rank]; producers store one granule into EVERY rank's mailbox with
   peer-to-peer stores over xGMI, consumers poll their own.  Replaces a host-driven all-reduce per PCG scalar
   (SURVEY.md 8e).  Pointers in peer_* are the peers' allocations mapped into this process (thallo_hip_ipc_open). */
#define THALLO_DIST_MAX_WORLD 8
#define THALLO_DIST_CTL_WORDS 16
typedef struct thallo_dist_t {
    unsigned long long* mail;                               /* == peer_mail[rank] */
    unsigned long long* peer_mail[THALLO_DIST_MAX_WORLD];
    float*    peer_r[2];         /* image slabs: the r vector of rank-1 / rank+1 (NULL at the image border) */
    long      peer_off_o[2];     /* element offset of my row's landing place (that peer's ghost row), Offset plane */
    long      peer_off_a[2];     /* ... Angle plane */
    unsigned* ctl;               /* THALLO_DIST_CTL_WORDS zeroed device words: [0] sequence number (GN step counter), [1] error
                                    (spin timeout), [4..8] post-mortem of the first timeout */
    int       world, rank;
} thallo_dist_t;

/* In-kernel finish of the single-reduction PCG form (the *_apply_jtj_sums_fin entries): with `tickets` (THALLO_HIP_FIN_TICKET_WORDS zeroed device
   words; zero again when the kernel ends) the launch's last workgroup also writes alphaD_word[0] = alphaD_k and betaN_word[0] = N - 2 alpha_k S1 +
   alpha_k^2 S2 (alpha_k = alphaN / alphaD_k) -- what thallo_hip_pcg_scalars_finish does in a launch of its own, same order, same bits.
   tickets == NULL: partials only. */
typedef struct thallo_fin_t {
    thallo_sum_t alphaN;
    unsigned*    tickets;
    float*       alphaD_word;
    float*       betaN_word;
} thallo_fin_t;

/* DEFERRED finish of the one-kernel PCG iteration (thallo_hip_iw_pcg_iter*_deferred): instead of the launch's last workgroup reading the partials
   back, the NEXT launch receives iteration k-1's raw partials -- alphaD (float) and {N, S1, S2} (double), `count` workgroups -- and every wave adds
   them up itself while its first rows load; its workgroup 0 leaves alphaD_{k-1} / betaN_{k-1} behind in the two words.  Same summation order, same
   bits as thallo_hip_iw_pcg_iter_finish.  The s12 buffer a launch writes must differ from the one it reads (ping-pong).  After the last iteration
   of a GN step, thallo_hip_iw_pcg_iter_finish produces its two words. */
typedef struct thallo_prev_t {
    const float*  alphaD_partials;
    const double* s12_partials;
    int           count;
    float*        alphaD_word;
    float*        betaN_word;
} thallo_prev_t;

long thallo_hip_vector_elems(long n_unknowns);          /* n rounded up to a multiple of 256 */
int  thallo_hip_device_cu_count(void);                  /* multiprocessor count of the current device */

/* ---------------------------------------------------------------- energy-independent PCG chain */

/* Fused-schedule PCGStep2 (reference gauss_newton.t:801-843, minus the delta update which this
 * schedule moves into the next PCGStep1):
 *   alpha = alphaN/alphaD (0 if alphaD==0, GN guard :226-234); r -= alpha*Ap; z = pre*r;
 *   betaN partials = sum z.r     pre==NULL means the identity preconditioner (:821-825). */
int thallo_hip_pcg_step2(float* r, const float* Ap, const float* pre, float* z, long n,
                         thallo_sum_t alphaN, thallo_sum_t alphaD, float* betaN_out, thallo_stream_t stream);

/* Same over two ranges of the flat vector, [off0,off0+len0) U [off1,off1+len1) (floats, multiples of 4):
 * the owned rows of one slab of a row-partitioned image domain (multi-GPU). */
int thallo_hip_pcg_step2_ranges(float* r, const float* Ap, const float* pre, float* z,
                                long off0, long len0, long off1, long len1,
                                thallo_sum_t alphaN, thallo_sum_t alphaD, float* betaN_out, thallo_stream_t stream);

/* Reference-shaped PCGStep2 (gauss_newton.t:801-843) incl. the delta update and, when b != NULL,
 * the LM q term  q = 0.5*delta.(r+b)  (:832-837).  lm selects the unguarded divide. */
int thallo_hip_pcg_step2_full(float* delta, const float* p, float* r, const float* Ap, const float* pre,
                              float* z, const float* b, long n, thallo_sum_t alphaN, thallo_sum_t alphaD,
                              float* betaN_out, float* q_out, int lm, thallo_stream_t stream);

/* PCGStep3 (gauss_newton.t:889-899): beta = betaN/alphaN (guarded unless lm); p = z + beta*p. */
int thallo_hip_pcg_step3(float* p, const float* z, long n, thallo_sum_t betaN, thallo_sum_t alphaN,
                         int lm, thallo_stream_t stream);

/* Stand-alone form of what the fused image kernels do on the fly: PCGStep3 of iteration k-1 (:889-899) and
 * the delta part of PCGStep2 (:814-815):  delta += alpha*p_in ; p_out = z + beta*p_in  (first != 0: p_out = z).
 * Used by plugins whose applyJTJ gathers p through index lists (graph domains). */
int thallo_hip_pcg_pupdate(const float* z, const float* p_in, float* p_out, float* delta, long n, int first,
                           thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, thallo_stream_t stream);
/* (delta == NULL: p update only with the unguarded LM divide; LM keeps the delta update in PCGStep2.) */
/* Same over two ranges of the flat vectors (floats, multiples of 4), cf. thallo_hip_pcg_step2_ranges. */
int thallo_hip_pcg_pupdate_ranges(const float* z, const float* p_in, float* p_out, float* delta,
                                  long off0, long len0, long off1, long len1, int first,
                                  thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, thallo_stream_t stream);

/* ---- Levenberg-Marquardt set.  PCGSaveSSq + PCGComputeCtC + PCGFinalizeDiagonal (gauss_newton.t:929-969,
 * thallo.t:3911-3937) in one pass over the raw diagonal `diag` = diag(J^T J) (pcg_init's diag_out):
 *   SSq = guardedInvert(diag) on the first GN iteration of a solve (save_ssq != 0; 1 without preconditioner);
 *   CtC = clamp(diag/radius, [min,max]_lm_diagonal/(SSq*radius)); pre = 1/(CtC + radius*diag/radius);
 *   b = r; z = pre*r; alphaN partials. */
int thallo_hip_lm_finalize_diagonal(const float* diag, float* SSq, float* CtC, float* pre, const float* r, float* b, float* z, long n,
                                    float radius, float min_lm_diagonal, float max_lm_diagonal, int save_ssq, int use_preconditioner,
                                    float* alphaN_out, thallo_stream_t stream);
/* PCGStep1_Finish, LM form (gauss_newton.t:777-787): Ap += CtC*p; alphaD partials = sum p.Ap */
int thallo_hip_lm_step1_finish(float* Ap, const float* CtC, const float* p, long n, float* alphaD_out, thallo_stream_t stream);
/* PCGStep2_1stHalf / _2ndHalf (gauss_newton.t:845-886), used every residual_reset_period iterations */
int thallo_hip_lm_step2_first_half(float* delta, const float* p, long n, thallo_sum_t alphaN, thallo_sum_t alphaD, thallo_stream_t stream);
int thallo_hip_lm_step2_second_half(float* r, const float* b, const float* Adelta, const float* pre, float* z, const float* delta, long n,
                                    float* betaN_out, float* q_out, thallo_stream_t stream);
/* PCGInit1_Finish (gauss_newton.t:712-731) for callers that assembled r and the RAW diagonal elsewhere (the sharded BA
 * driver all-reduces the point blocks of both first): pre = guardedInvert(diag) (or 1), z = pre*r, alphaN partials. */
int thallo_hip_pcg_init_finish(const float* r, const float* diag, float* pre, float* z, long n, int use_preconditioner,
                               float* alphaN_out, thallo_stream_t stream);
/* partials of sum a.b */
/* LM without the host in the loop (gauss_newton.t:1666-1686 semantics): `state` = 8 device words: [0] Q0, [1] frozen flag (the GATE word),
 * [2] PCG iterations done when the loop froze, [3..7] free for the driver's end-of-step report.  lm_zeta (one wave, behind PCGStep2 of iteration k)
 * applies the zeta test and freezes the loop; while the gate word set with lm_set_gate is non-zero, pcg_pupdate, lm_step1_finish,
 * pcg_step2_full and lm_step2_first / second_half launches return at once (so do applyJTJ kernels given the word: *_apply_jtj_gated).
 * The driver enqueues all lIterations iterations and reads the state back once per GN step. */
void thallo_hip_lm_set_gate(const unsigned* gate);
int thallo_hip_lm_state_reset(float* state, thallo_stream_t stream);
int thallo_hip_lm_zeta(thallo_sum_t q, int k, float q_tolerance, float* state, thallo_stream_t stream);
/* LM PCGStep2 (as thallo_hip_pcg_step2_full with lm = 1, b and q_out given) whose last workgroup also applies thallo_hip_lm_zeta's test for PCG
   iteration k on the q it just produced (tickets: THALLO_HIP_FIN_TICKET_WORDS zeroed device words, zero again when the kernel ends): one launch less
   per LM PCG iteration. */
int thallo_hip_pcg_step2_full_zeta(float* delta, const float* p, float* r, const float* Ap, const float* pre,
                                   float* z, const float* b, long n, thallo_sum_t alphaN, thallo_sum_t alphaD,
                                   float* betaN_out, float* q_out, unsigned* tickets, int k, float q_tolerance, float* lm_state, thallo_stream_t stream);
int thallo_hip_dot(const float* a, const float* b, long n, float* out, thallo_stream_t stream);

/* PCGLinearUpdate (gauss_newton.t:901-906) for one unknown image:  X[i] += delta[i] (+ alpha*p[i]
 * when p != NULL: the fused schedule's last pending delta += alpha*p). */
int thallo_hip_linear_update(float* X, const float* delta, const float* p, long len,
                             thallo_sum_t alphaN, thallo_sum_t alphaD, thallo_stream_t stream);

/* out[0] = sum(partials) -- used for cost / model-cost read-back (gauss_newton.t:1128-1150). */
/* Collective transport of the one-kernel-per-iteration schedule over row slabs: message of a rank = [alphaD_local | N, S1, S2 as (hi, lo)
   words | the listed segments of vec (boundary rows of Ap_out)]; unpack adds the gathered scalars in rank order, writes alphaD_k and
   betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 (alphaN: a one-word sum) and copies the neighbours' rows into the ghost segments.
   ONE all-gather of these messages per PCG iteration. */
int thallo_hip_slab_pack_iter(const float* vec, thallo_segs_t segs, const float* alphaD_partials, const double* s3_partials, int count, float* out, thallo_stream_t stream);
int thallo_hip_slab_unpack_iter(float* vec, thallo_segs_t top, const float* src_top, thallo_segs_t bot, const float* src_bot,
                                const float* gathered, long stride, int world, thallo_sum_t alphaN, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* Shard form (bundle adjustment across ranks): the PCG sums over a block of unknowns every rank holds in full (after the all-reduce completed A p there) --
   alphaD partials (float) and {N, S1, S2} (double) per workgroup, returns their count; pre may be NULL (no preconditioner); n % 4 == 0 */
int thallo_hip_block_sums(const float* p, const float* Ap, const float* r, const float* pre, long n, float* alphaD_out, double* s3_out, thallo_stream_t stream);
/* ... and the two scalars of the iteration from the gathered per-rank messages of the rank-private blocks ([alphaD | N, S1, S2 as (hi, lo) words], rank
   order) plus the shared block's partials.  betaN_word == NULL: only word 0 of each message and the float partials are added into alphaD_word[0]
   (alphaN_0 at PCGInit). */
int thallo_hip_shard_scalars(const float* gathered, long stride, int world, const float* alphaD_partials, const double* s3_partials, int count, thallo_sum_t alphaN,
                             float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* Range partition (graph domains): after an all-gather of messages that carry every rank's owned slice of each plane of a flat vector (equal slices),
   vec[first.off[j] + r * first.len[j] + i] = gathered[r * stride + skip + (pieces before j) + i] for every rank r and piece j (`first` = rank 0's pieces) */
int thallo_hip_range_unpack(float* vec, thallo_segs_t first_rank_pieces, const float* gathered, long stride, long skip, int world, thallo_stream_t stream);
/* X += delta + alpha_older * p_older + alpha * p (in that order): the tail of a GN step whose last two delta updates were
   deferred (THALLO_IW_STEP1_MODE batching with an even number of PCG iterations) */
int thallo_hip_linear_update2(float* X, const float* delta, const float* p_older, thallo_sum_t alphaN_older, thallo_sum_t alphaD_older,
                              const float* p, thallo_sum_t alphaN, thallo_sum_t alphaD, long len, thallo_stream_t stream);
/* delta += alpha_0 p_0, then += alpha_1 p_1, ... (terms.count <= THALLO_HIP_MAX_UPDATE_TERMS pending terms, oldest first; every term one fma on the running
   value, i.e. the bits of one `delta += alpha p` per PCG iteration in that order).  X == NULL: delta is updated in place (the ring of p planes of the
   one-kernel schedule is full; with THALLO_DELTA_PLANES=N:W the library launches it on a stream of its own next to the PCG loop, max_workgroups keeping it to
   a share of the chip).  X != NULL: the tail of a GN step (gauss_newton.t:901-906): X += the updated delta, delta itself is left as it was. */
#define THALLO_HIP_MAX_UPDATE_TERMS 32
typedef struct { const float* p[THALLO_HIP_MAX_UPDATE_TERMS]; thallo_sum_t alphaN[THALLO_HIP_MAX_UPDATE_TERMS], alphaD[THALLO_HIP_MAX_UPDATE_TERMS]; int count; } thallo_update_terms_t;
int thallo_hip_linear_update_n(float* X, float* delta, thallo_update_terms_t terms, long len, int max_workgroups /* 0: as many as the flat kernels use */, thallo_stream_t stream);
/* Known-answer test of the wave64 primitives of the generated kernels (ballot, peer grouping by key, grouped sums: thallo.t:3380-3399 / cuda_util.t:334-427 for 64-wide waves),
   the reference's tests/cuda_unit_tests/{ballot,get_peers,reduce_peers}.t restated for 64 lanes; one wave, compiled from the generated kernels' own prelude with hipRTC:
   out_ballot = max over lanes of ballot(lane != 0) = 0xFFFFFFFFFFFFFFFE; out_peers = sum over lanes of (peer mask of key lane % 4) & 0xFF = 255 * 64 / 4;
   sums4[i] = 480 + 16 i. */
int thallo_hip_wave64_selftest(unsigned long long* out_ballot, unsigned* out_peers, float* sums4);
int thallo_hip_finish_sum(thallo_sum_t s, float* out, thallo_stream_t stream);
/* ... behind a device-side gate word (non-zero: the launch does nothing; the LM loop's gate, see thallo_hip_lm_zeta) */
int thallo_hip_finish_sum_gated(thallo_sum_t s, float* out, const unsigned* gate, thallo_stream_t stream);

/* Row-slab exchange helpers (multi-GPU).  One rank's message per PCG iteration is
 *   out = [ local sum | first owned row | last owned row ]  (pack: out[0] = sum(partials) when sum.count > 0,
 * out[1..] = the listed segments); after an all-gather of those messages, unpack writes
 * sum_out[0] = sum over ranks in rank order (identical bits on every rank) and copies the neighbours' rows
 * (src_top / src_bot point into the gathered buffer, NULL at the domain boundary) into the ghost segments. */
int thallo_hip_slab_pack(const float* vec, thallo_segs_t segs, thallo_sum_t sum, float* out, thallo_stream_t stream);
int thallo_hip_slab_unpack(float* vec, thallo_segs_t top, const float* src_top, thallo_segs_t bot, const float* src_bot,
                           const float* gathered, long stride, int world, float* sum_out, thallo_stream_t stream);

/* alpha/beta trace for tests: out[0]=alphaN/alphaD, out[1]=betaN/alphaN with the GN guards. */
int thallo_hip_alpha_beta(thallo_sum_t alphaN, thallo_sum_t alphaD, thallo_sum_t betaN, float* out2,
                          thallo_stream_t stream);

/* Every <energy>_pcg_init takes `diag_out`: NULL, or a flat vector that receives the RAW diag(J^T J)
 * (= what fmap.evalJTF accumulates into `preconditioner`, thallo.t:3898-3902, before guardedInvert) -- the input of
 * LM's computeCtC (thallo.t:3929-3933).
 *
 * ---------------------------------------------------------------- E5: tests/minimal/laplacian.t
 * X unknown float {W,H} (param 0), A float {W,H} (param 1); fit = w*(X-A), reg = x/y forward
 * differences guarded by InBounds.  xguard: 0 = InBounds(x+1,y+1) as shipped (laplacian.t:11),
 * 1 = InBounds(x+1,y) (the variant gold.png was made with). */
int thallo_hip_lapimg_cost(int W, int H, const float* X, const float* A, float w_fit, int xguard,
                           float* cost_out, thallo_stream_t stream);
/* fused PCGInit1 (+_Finish) (gauss_newton.t:678-731): r=-J^T F, pre (identity: no UsePreconditioner),
 * z = pre*r, p_prev = 0, delta = 0, alphaN partials. */
int thallo_hip_lapimg_pcg_init(int W, int H, const float* X, const float* A, float w_fit, int xguard,
                               float* r, float* z, float* p_prev, float* delta,
                               float* diag_out, float* alphaN_out, thallo_stream_t stream);
/* plain PCGStep1: Ap = J^T J p, alphaD partials */
int thallo_hip_lapimg_apply_jtj(int W, int H, float w_fit, int xguard, const float* p, float* Ap, float* alphaD_out, thallo_stream_t stream);
/* fused PCGStep3(k-1) + delta update(k-1) + PCGStep1(k) (gauss_newton.t:734-752,889-899):
 *   first!=0: p = z (beta=0, no delta update); else beta = betaN/alphaN_prev, alpha = alphaN_prev/alphaD_prev,
 *   delta += alpha*p_in; p_out = z + beta*p_in;  Ap = J^T J p_out; alphaD partials = sum p_out.Ap */
int thallo_hip_lapimg_pcg_step1(int W, int H, float w_fit, int xguard,
                                const float* z, const float* p_in, float* p_out, float* delta, float* Ap,
                                int first, thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                                float* alphaD_out, thallo_stream_t stream);

/* ---------------------------------------------------------------- E1: examples/image_warping/image_warping.t
 * Offset float2 (0), Angle float (1) unknown; UrShape float2 (2), Constraints float2 (3), Mask float (4),
 * w_fitSqrt (5), w_regSqrt (6).  Flat vector layout: [Offset 2*pix+c | Angle 2*N+pix].
 * Per-GN-iteration precomputed planes (allowed by SURVEY.md section 7 step 3):
 *   cs    float2 per pixel = (cos Angle, sin Angle)
 *   flags uint8  per pixel : bit0 = Mask==0 (pixel active, image_warping.t:14-15,23),
 *                            bit1 = fit residual valid (image_warping.t:27),
 *                            bits2-4 = number of valid neighbour pairs (0..4)
 * Row slabs: W x H is the LOCAL image, which may carry one ghost row above and/or below; the kernels
 * produce outputs for the owned rows [row0,row1) only (row0=0,row1=H for a whole image) and read the
 * ghost rows as stencil halo.  pcg_init also fills cs/flags (and p_prev=0) on ghost rows; the fused
 * pcg_step1 also keeps p current on ghost rows (p = z + beta p), so per PCG iteration only the ghost
 * rows of z have to be refreshed from the neighbouring slab.
 * `irregular` (device int, may be NULL): pcg_init stores the number of pixels whose right/down UrShape neighbours are
 * NOT at exact unit offsets; when it is 0 (UrShape = the pixel grid, which is what the reference's harness always
 * passes: CombinedSolver.h:158-176) pcg_step1 / apply_jtj skip the UrShape plane (-8 B/pixel); the bits of the
 * result are the same either way. */
int thallo_hip_iw_cost(int W, int H, int row0, int row1, const float* offset, const float* angle, const float* urshape,
                       const float* constraints, const float* mask, float w_fit, float w_reg,
                       float* cost_out, thallo_stream_t stream);
int thallo_hip_iw_pcg_init(int W, int H, int row0, int row1, const float* offset, const float* angle, const float* urshape,
                           const float* constraints, const float* mask, float w_fit, float w_reg,
                           float* r, float* pre, float* z, float* p_prev, float* delta,
                           float* cs, unsigned char* flags, float* diag_out, int* irregular_out, float* alphaN_out, thallo_stream_t stream);
int thallo_hip_iw_pcg_step1(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg,
                            const float* z, const float* p_in, float* p_out, float* delta, float* Ap,
                            int mode, thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                            thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                            const int* irregular, const float* r, float* alphaD_out, thallo_stream_t stream);
/* mode: bit 0 = first PCG iteration of the GN step (p = z, delta untouched); bits 1-2 = the delta update this launch carries:
 *   0  delta += alpha_{k-1} p_{k-1}                          (every iteration: the plain fused schedule)
 *   1  none -- deferred to the next launch
 *   2  delta += alpha_{k-2} p_{k-2}, then += alpha_{k-1} p_{k-1}   (p_{k-2} is read from p_out just before it is overwritten;
 *      alphaN_prev2 / alphaD_prev2 = the scalars of iteration k-2)
 * Alternating 1, 2 halves delta's read + write traffic (-6 B/pixel/iteration) and produces the same bits as mode 0 throughout.
 * THALLO_IW_STEP1_MODE(k, batched) gives the mode of iteration k. */
#define THALLO_IW_STEP1_MODE(k, batched) ((k) == 0 ? 1 : !(batched) ? 0 : ((k) & 1) ? 2 : 4)
/* One kernel per PCG iteration (replaces pcg_step1 + pcg_step2 of iteration k-1/k): r_out = r_in - alpha_{k-1} Ap_in (first:
 * r_in), z = M^-1 r_out (pixel grid: from the flags byte; else from `pre`), p_out = z + beta_{k-1} p_in, the deferred delta
 * update (mode as in pcg_step1), Ap_out = J^T J p_out, and per workgroup: alphaD partial (float) into alphaD_out[b] and the three
 * double sums N = sum r.M^-1.r, S1 = sum r.M^-1.Ap, S2 = sum Ap.M^-1.Ap into s12_out[3b .. 3b+2] (3 * THALLO_HIP_MAX_PARTIALS
 * doubles).  pcg_iter_finish (one wave) then writes alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2
 * (= r_{k+1}.M^-1 r_{k+1}, evaluated in double from exact products of the float data), i.e. both PCG scalars of the iteration come
 * from ONE reduction point.  r, Ap and p ping-pong (in != out); ghost rows of a slab: r and p are
 * kept current by the kernel, Ap_in's ghost rows have to be refreshed from the neighbour.  81 B/pixel (+18 deferred delta). */
int thallo_hip_iw_pcg_iter(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                           float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                           const float* p_in, float* p_out, float* delta, int mode,
                           thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                           thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                           const int* irregular, float* alphaD_out, double* s12_out,
                           unsigned* fin_tickets, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* fin_tickets / alphaD_word / betaN_word (all or none): THALLO_HIP_FIN_TICKET_WORDS zeroed device words + the two scalar words.  When given,
 * the launch's last workgroup does pcg_iter_finish's job itself (same summation order, same bits) and no separate launch is needed; the
 * tickets are zero again when the kernel ends.  alphaN_k is taken from betaN_prev (which is alphaN_k by definition). */
#define THALLO_HIP_FIN_TICKET_WORDS 528
/* the same over a row slab of a multi-GPU run: additionally stores the first / last owned row of Ap_out into the neighbours' ghost rows
   (d.peer_r[k] + d.peer_off_o/a[k] = that row inside the neighbour's Ap_out buffer), peer-to-peer */
int thallo_hip_iw_pcg_iter_dist(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                                float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                const float* p_in, float* p_out, float* delta, int mode,
                                thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                                thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                                const int* irregular, thallo_dist_t d, float* alphaD_out, double* s12_out,
                                unsigned* fin_tickets, int slot0, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* fin_tickets / alphaD_word / betaN_word (all or none): the launch's last workgroup then IS the exchange (thallo_hip_dist_exchange_iter's job,
   mailbox slots slot0 .. slot0+6, betaN_prev must be a one-word sum): one launch per PCG iteration on every rank. */
/* The same iteration as a barrier-free marching stencil (energy_image_warping_march.hip): a wave owns a 128-pixel column strip, keeps a
 * three-row window in registers, x neighbours through DPP lane shifts, rows prefetched ahead; unit-pixel-grid UrShape ONLY (W even).
 * The caller establishes that property once per Init (thallo_hip_iw_urshape_irregular: *count_out == 0); if the word pcg_init wrote
 * (`irregular`, may be NULL = trust the caller) is nonzero the kernel writes NaN scalars instead of computing.  Same buffers, modes,
 * partial / ticket / scalar-word conventions as thallo_hip_iw_pcg_iter (no urshape / pre arguments: never read). */
int thallo_hip_iw_pcg_iter_march(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                 float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                 const float* p_in, float* p_out, float* delta, int mode,
                                 thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                                 thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                                 const int* irregular, float* alphaD_out, double* s12_out,
                                 unsigned* fin_tickets, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
int thallo_hip_iw_pcg_iter_march_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                      float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                      const float* p_in, float* p_out, float* delta, int mode,
                                      thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev,
                                      thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                                      const int* irregular, thallo_dist_t d, float* alphaD_out, double* s12_out,
                                      unsigned* fin_tickets, int slot0, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* The slab iteration with the DEFERRED cross-rank finish (round 4; energy_image_warping_march_rc.hip): as thallo_hip_iw_pcg_iter_march_rc_dist without tickets -- the launch
   stores this rank's partials only --, and it finishes iteration k-1 itself: `prev` = that iteration's partials of THIS rank and its two words, prev_slot0 = the mailbox
   slots of its exchange, gs = two device words of 8 bytes (zeroed once) through which the launch's designated wave (first segment of strip 0) hands the two GLOBAL words to
   the other waves.  Same granules, slots, rank order and summation order as the exchange at the end of a launch: same bits.  alphaN_prev must be a finished word.
   thallo_hip_iw_dist_finish_deferred: one wave that finishes a GN step's last iteration the same way. */
int thallo_hip_iw_pcg_iter_march_rc_dist_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                                  const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                                  thallo_sum_t alphaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2, thallo_prev_t prev, int prev_slot0,
                                                  unsigned long long* gs, const int* irregular, thallo_dist_t d, float* alphaD_out, double* s12_out, thallo_stream_t stream);
/* 1: the deferred cross-rank finish may run on `rows` owned rows of a W-wide slab (every workgroup of its launch, the eight extra ones included, is resident at once:
   its working waves wait for the last workgroup's granules); 0: the caller runs thallo_hip_iw_pcg_iter_march_rc_dist */
int thallo_hip_iw_march_rc_deferred_fits(int W, int rows);
int thallo_hip_iw_dist_finish_deferred(thallo_prev_t prev, int prev_slot0, thallo_sum_t alphaN_prev, thallo_dist_t d, unsigned long long* gs, thallo_stream_t stream);
/* the two one-kernel iterations with the deferred finish (thallo_prev_t above): alphaN_prev = alphaN_{k-1} (a finished sum), alphaN_prev2 / alphaD_prev2 as
   in the plain forms; `prev` is ignored for mode & 1 (first iteration of a GN step) */
int thallo_hip_iw_pcg_iter_deferred(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                                    float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                    const float* p_in, float* p_out, float* delta, int mode,
                                    thallo_sum_t alphaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2, thallo_prev_t prev,
                                    const int* irregular, float* alphaD_out, double* s12_out, thallo_stream_t stream);
int thallo_hip_iw_pcg_iter_march_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                          float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                          const float* p_in, float* p_out, float* delta, int mode,
                                          thallo_sum_t alphaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2, thallo_prev_t prev,
                                          const int* irregular, float* alphaD_out, double* s12_out, thallo_stream_t stream);
/* The marching iteration WITHOUT an A p plane (round 4; energy_image_warping_march_rc.hip): iteration k >= 1 on the unit pixel grid.
 * r_out = r_in - alpha_{k-1} (J^T J p_in) with J^T J p_{k-1} RECOMPUTED from the rows of p_in the launch loads anyway, p_out = M^-1 r_out + beta_{k-1} p_in,
 * the delta update of `mode` (never mode & 1: a GN step's first iteration has no A p_{k-1}; thallo_hip_iw_pcg_iter_march runs it), and the partials of
 * alphaD_k / N, S1, S2 from J^T J p_out -- which is not stored.  57 B/pixel instead of 81 (+ 18 of deferred delta); r, p, delta and every alpha / beta are
 * bit-identical to thallo_hip_iw_pcg_iter_march with the same rows per segment.  Scalars / partials / tickets as there.
 * Whole images (row0 = 0, row1 = H): Ap_in / Ap_out are not touched (NULL is fine).  A row slab of a multi-GPU run (one ghost row towards each neighbour): the
 * exchange stays the stored-plane kernel's -- rows row0 and row1 - 1 of J^T J p_out are written to Ap_out (the _dist form: into the neighbours' ghost rows,
 * peer-to-peer, and its last workgroup is the scalar exchange), and on a ghost row J^T J p_in is READ from Ap_in, where the exchange put it -- so the two kernels
 * are interchangeable launch by launch.  Replaces PCGStep1 + PCGStep2 + PCGStep3 (gauss_newton.t:734-752,801-843,889-899). */
int thallo_hip_iw_pcg_iter_march_rc(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                    thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                                    const int* irregular, float* alphaD_out, double* s12_out,
                                    unsigned* fin_tickets, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
int thallo_hip_iw_pcg_iter_march_rc_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                             const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                             thallo_sum_t alphaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2, thallo_prev_t prev,
                                             const int* irregular, float* alphaD_out, double* s12_out, thallo_stream_t stream);
int thallo_hip_iw_pcg_iter_march_rc_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                         const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                         thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, thallo_sum_t alphaN_prev2, thallo_sum_t alphaD_prev2,
                                         const int* irregular, thallo_dist_t d, float* alphaD_out, double* s12_out,
                                         unsigned* fin_tickets, int slot0, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
void thallo_hip_march_rc_debug_set(int what, int value);  /* sweep builds (tools/rc_probe.py) only: 0 rows of prefetch (1, 2, 4), 1 register budget (workgroups per CU: 1, 2, 3), 2 cache-policy mask */
/* *count_out (device int) = number of pixels whose right / down UrShape neighbour is not at the exact unit offset (0 = pixel grid) */
int thallo_hip_iw_urshape_irregular(int W, int H, const float* urshape, int* count_out, thallo_stream_t stream);
void thallo_hip_march_debug_set(int what, int value);     /* tools / tests only: 0 rows per wave segment, 6 workgroup budget (sweep builds: 1 prefetch depth, 2 non-temporal mask, 3 occupancy, 4 debug mode, 5 map) */
/* ---- the PCG loop of a whole Gauss-Newton step in ONE launch (energy_image_warping_resident.hip), for images whose solver state fits the chip's
 * registers: a wave keeps r, p, A p of its pixels in registers (delta in LDS) for all L iterations; per iteration the boundary of A p goes to the four
 * neighbouring waves and the workgroup's sums to every workgroup as 8-byte {value | tag} granules, no launch boundary, no grid barrier.  Same geometry,
 * arithmetic and summation order as thallo_hip_iw_pcg_iter_march with the same rows per segment: bit-identical r, p, delta, A p, alpha, beta.
 * Replaces gauss_newton.t:1615-1687 for these shapes. */
/* rows per wave segment (1..6), or 0: does not fit (the caller runs one launch per PCG iteration).  Host logic, no launch. */
int  thallo_hip_iw_resident_rows(int W, int rows);
/* bytes of exchange memory (granule buffers + control words) a plan must hand to the calls below, zeroed once */
long thallo_hip_iw_resident_bytes(int W, int rows);
/* L >= 1 iterations from what thallo_hip_iw_pcg_init left (r_0 in r_in; zeros in p_in and delta; cs, flags; alphaN_0).  Leaves what L launches of
 * the marching kernel leave: r_{L-1}, A p_{L-1}, p_{L-1} in r_out / Ap_out / p_out, delta without its last term (thallo_hip_linear_update adds it), and
 * words[2k] = alphaD_k, words[2k + 1] = betaN_k.  Whole images only (row0 = 0, row1 = H).  Returns the workgroup count, -hipErrorNotSupported when the
 * shape does not fit, another negative hipError_t on failure.  Every wait inside is bounded (2 s by default): see thallo_hip_iw_resident_status. */
int  thallo_hip_iw_pcg_resident(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                                thallo_sum_t alphaN0, float* words, const int* irregular, float* X_offset, float* X_angle, void* xbuf, int L, thallo_stream_t stream);
/* (X_offset / X_angle, round 6: both non-NULL = PCGLinearUpdate rides along, X += delta + alpha_{L-1} p_{L-1} with thallo_hip_linear_update's bits; both NULL: the caller's launch)
   error word of the plan's resident launches (1 = a bounded wait ran out: results void); clear != 0 resets it; spin_ms >= 0 sets the bound in
 * milliseconds (0 = default); pm: 5 words of post-mortem or NULL.  Synchronises the stream. */
int  thallo_hip_iw_resident_status(void* xbuf, int clear, int spin_ms, unsigned* pm, thallo_stream_t stream);
/* One rank's row slab of a multi-GPU run (local image W x H including its ghost rows, owned rows [row0, row1)): the first / last owned row of A p_k goes
 * straight into the neighbouring ranks' ghost areas -- thallo_hip_iw_resident_ghost_bytes(W) bytes at byte offset ghost_off (a multiple of 16) of EVERY rank's
 * mailbox block, d.peer_mail[rank -+ 1] -- workgroup 0 adds this rank's sums up, exchanges them through the mailbox slots slot0 + 7 k .. (the granules, slots
 * and rank order of thallo_hip_iw_pcg_iter_march_dist: same bits) and publishes alphaD_k / betaN_k for the rest of the chip.  alphaN0: the GLOBAL alphaN_0.
 * L <= 4095.  The caller has advanced the GN step counter (thallo_hip_dist_begin_step).  thallo_hip_iw_resident_rows_slab: rows per segment for such a slab
 * (below != 0: a rank below -- the last segment must then be a full one), 0 = does not fit. */
int  thallo_hip_iw_resident_rows_slab(int W, int rows, int below);
long thallo_hip_iw_resident_ghost_bytes(int W);
int  thallo_hip_iw_pcg_resident_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                     const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                                     thallo_sum_t alphaN0, float* words, const int* irregular, void* xbuf, thallo_dist_t d, long ghost_off, int slot0, int L, thallo_stream_t stream);
void thallo_hip_resident_debug_set(int what, int value);    /* tools / tests only: 0 rows per wave segment, 1 workgroup budget, 2 fault injection (one workgroup withholds its sums of iteration 2), 3 the waits' bound in ms for the next launches (-1: leave) */
/* rows per wave segment the marching kernels use on `rows` owned rows of a W-wide image; 0 = more column strips than the device has workgroup
 * slots: the marching entry points return -hipErrorNotSupported, the caller stays on thallo_hip_iw_pcg_iter (host logic, no launch) */
int thallo_hip_iw_march_rows(int W, int rows);
int thallo_hip_iw_pcg_iter_finish(const float* alphaD_partials, const double* s12_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream);

/* image_warping's PCGStep2 (gauss_newton.t:801-843 minus delta): r -= alpha*Ap, betaN partials = sum (M^-1 r).r over the owned
 * rows.  When *irregular == 0 (UrShape = unit pixel grid) M^-1 is recomputed from the flags byte and z is NOT written
 * (37 B/pixel instead of 60) -- pcg_step1 then forms z = M^-1 r from `r` on the fly; otherwise pre is read and z written. */
int thallo_hip_iw_pcg_step2(int W, int H, int row0, int row1, const unsigned char* flags, float w_fit, float w_reg,
                            float* r, const float* Ap, const float* pre, float* z,
                            thallo_sum_t alphaN, thallo_sum_t alphaD, const int* irregular, float* betaN_out, thallo_stream_t stream);

/* ---------------------------------------------------------------- graph-edge domains
 * Incidence lists built by the host from the Sparse maps V0/V1 (device int32 arrays, thallo.t:136) once per
 * Init -- legal because the maps are constant during a solve; the reference instead scatters with atomics
 * every iteration (thallo.t:3352-3403):
 *   out_ptr[N+1], out_v1[E] : edges sorted by source vertex; the position in this order is the edge id e'
 *   in_ptr[N+1], in_edge[E], in_src[E] : for each vertex the ids / source vertices of the edges ending there
 *
 * E6: tests/minimal_graph/laplacian.t  (X float unknown (0), A float (1), v0 (2), v1 (3); w_fit literal) */
int thallo_hip_lapgraph_cost(int N, const int* out_ptr, const int* out_v1, const float* X, const float* A, float w_fit,
                             float* cost_out, thallo_stream_t stream);
int thallo_hip_lapgraph_pcg_init(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                 const float* X, const float* A, float w_fit, float* r, float* z, float* p_prev, float* delta,
                                 float* diag_out, float* alphaN_out, thallo_stream_t stream);
int thallo_hip_lapgraph_apply_jtj(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                  float w_fit, const float* p, float* Ap, float* alphaD_out, thallo_stream_t stream);

/* E2: examples/arap_mesh_deformation/arap_mesh_deformation.t  (w_fitSqrt (0), w_regSqrt (1), Position float3 (2) and
 * Angle float3 (3) unknown, Original float3 (4), Constraints float3 (5), V0 (6), V1 (7)).
 * Flat vector layout [Position 3n+c | Angle 3N+3n+c].  precompute (once per GN iteration) fills, per edge in
 * out-CSR order, F[3E] = the reg residual and G[9E] = d(R(Angle) dv)/d(Angle) (three float3 columns).
 * cost / pcg_init / apply_jtj produce outputs for the vertex range [n0,n1) only (0,N = everything): the vertex-partitioned
 * multi-GPU driver (thallo_amd/distributed_graph.py) keeps the vectors replicated and all-gathers p.
 * ell_stride selects the layout of the per-edge arrays (out_v1, F, G; in_edge holds positions in that layout):
 *   0      out-CSR order, array of structs: edge k of the CSR at k, its 3 / 9 floats contiguous;
 *   S > 0  "ELL", structure of arrays: the j-th edge of vertex n at position j*N + n, component c of F / G at c*S + position,
 *          S = maxdeg * N (out_v1 has S entries, F 3*S, G 9*S); in_edge / in_src likewise hold the j-th incoming edge of vertex n at
 *          j*N + n (max in-degree * N entries; in_ptr still gives the degrees).  A wave of consecutive vertices then reads consecutive addresses in
 *          every per-edge array (2 cache lines per wave instruction instead of up to 64): what the single-GPU plugin uses when
 *          the padding is bounded (maxdeg <= 32, maxdeg*N <= 3E + N). */
int thallo_hip_arap_cost(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                         const float* original, const float* constraints, float w_fit, float w_reg, float* cost_out, long ell_stride, thallo_stream_t stream);
int thallo_hip_arap_precompute(int N, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                               const float* original, float w_reg, float* F, float* G, long ell_stride, thallo_stream_t stream);
/* ---- ARAP: the PCG loop of a whole Gauss-Newton step in ONE launch (round 4; energy_graph.hip k_arap_resident).  One thread keeps its vertex's r, p, A p, M^-1, delta in
 * registers for all L iterations (and the G matrices of its edges, which a Gauss-Newton step does not change); a workgroup keeps p_k of its vertices and of the vertices
 * they share an edge with ("ghosts": a host-built ascending list per workgroup) in LDS and updates the ghosts' r and p itself, so an iteration has ONE hand-over: A p_k of every
 * vertex as {value | tag} granules for the workgroups that have it as a ghost, together with the workgroup's {alphaD | N, S1, S2} record for everybody.  Same vertex ->
 * workgroup map, expressions and summation order as thallo_hip_pcg_update + thallo_hip_arap_apply_jtj_rc per iteration: bit-identical r, p, delta, A p, alpha_k, beta_k.
 * Every wait is bounded (thallo_hip_arap_resident_status).
 * xbuf: thallo_hip_arap_resident_bytes(N) zero-filled bytes; per workgroup (= vertex / 256) 4 + max_ghosts ints {count, 0, 0, 0, the vertices of OTHER workgroups that one of
 * its vertices shares an edge with (either direction), ascending} at thallo_hip_arap_resident_lists_offset(N); count <= thallo_hip_arap_resident_max_ghosts().  fits: the ELL
 * layout (ell_stride = out_slots x N, in-lists of in_slots x N entries; <= 32 slots) and every workgroup resident at once (<= 2 per CU, <= 512).  A vertex's first 6 out- and
 * in-edges live in registers; the others go through `overflow` (thallo_hip_arap_resident_overflow_floats(N, out_slots, in_slots) floats, written by the launch itself).  Replaces the loop of gauss_newton.t:1615-1687. */
/* tools / tests: 0 = the ARAP plugin keeps the caller's vertex numbering (default 1: it renumbers by recursive coordinate bisection of Original when that leaves its
   workgroups fewer ghost vertices: plugins.cpp ArapPlugin) */
void thallo_hip_arap_debug_reorder(int on);
/* *out_device += an order-sensitive checksum of n ints (out_device: 8 zeroed bytes of device memory) */
int  thallo_hip_checksum_i32(long n, const int* v, unsigned long long* out_device, thallo_stream_t stream);
/* N float3 between two numberings: dst[i] = src[idx[i]] (scatter = 0) or dst[idx[i]] = src[i] (scatter != 0); src != dst */
int  thallo_hip_permute3(int N, const int* idx, const float* src, float* dst, int scatter, thallo_stream_t stream);
long thallo_hip_arap_resident_bytes(int N);
long thallo_hip_arap_resident_lists_offset(int N);
long thallo_hip_arap_resident_overflow_floats(int N, int out_slots, int in_slots);
int  thallo_hip_arap_resident_max_ghosts(void);
int  thallo_hip_arap_resident_fits(int N, long ell_stride);
int  thallo_hip_arap_pcg_resident(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                  const float* constraints, const float* original, const float* SC, float w_fit, float w_reg, long ell_stride,
                                  float* r, float* Ap, const float* pre, float* p0, float* p1, float* delta, thallo_sum_t alphaN0, float* words,
                                  void* xbuf, float* overflow, int in_slots, int L, thallo_stream_t stream);
int  thallo_hip_arap_resident_status(void* xbuf, int clear, unsigned* pm, thallo_stream_t stream);
int thallo_hip_arap_pcg_init(int N, int n0, int n1, const int* out_ptr, const int* in_ptr, const int* in_edge, const float* position,
                             const float* constraints, const float* F, const float* G, float w_fit, float w_reg,
                             float* r, float* pre, float* z, float* p_prev, float* delta, float* diag_out, float* alphaN_out, long ell_stride, thallo_stream_t stream);
int thallo_hip_arap_apply_jtj(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                              const float* constraints, const float* G, float w_fit, float w_reg,
                              const float* p, float* Ap, float* alphaD_out, long ell_stride, thallo_stream_t stream);

/* ---------------------------------------------------------------- E4: examples/bundle_adjustment/bundle_adjustment.t
 * on the materialized sparse-J path (reference: precomputeJ + cuSPARSE csrsort/csr2csc/csrmv x2,
 * gauss_newton.t:327-487,1332-1525).  cameras float9 (0) and points float3 (1) unknown, observations float2 (2),
 * oToC (3), oToP (4).  Flat layout [cameras 9c+k | points 9C+3p+k].
 * Host-built incidence (once per Init), q = position of an observation in camera-sorted order:
 *   cam_ptr[C+1], cam_obs[O] (original observation id of q), q_cam[O], q_pt[O] (camera / point of q),
 *   pt_ptr[P+1], pt_pos[O] (the q's of each point's observations)
 * compute_j (once per GN iteration) materialises J as one 24-float block per observation,
 *   Jb[24q..] = { dr0/dcam[9], dr0/dpt[3], dr1/dcam[9], dr1/dpt[3] }, and the residuals F[2q..]. */
int thallo_hip_ba_cost(int C, int P, int O, const float* cameras, const float* points, const float* observations,
                       const int* oToC, const int* oToP, float* cost_out, thallo_stream_t stream);
int thallo_hip_ba_compute_j(int O, const float* cameras, const float* points, const float* observations,
                            const int* cam_obs, const int* q_cam, const int* q_pt, float* Jb, float* F, thallo_stream_t stream);
/* the same blocks from forward-mode dual numbers over the residual's expression (what rounds 1-3 stored): the reference of the closed form in the tests */
int thallo_hip_ba_compute_j_ad(int O, const float* cameras, const float* points, const float* observations,
                            const int* cam_obs, const int* q_cam, const int* q_pt, float* Jb, float* F, thallo_stream_t stream);
int thallo_hip_ba_pcg_init(int C, int P, const int* cam_ptr, const int* q_pt, const int* pt_ptr, const int* pt_pos, const int* q_cam,
                           const float* Jb, const float* F, float* r, float* pre, float* z, float* p_prev, float* delta,
                           float* diag_out, float* alphaN_out, thallo_stream_t stream);

/* ---------------------------------------------------------------- E3: examples/shape_from_shading/shape_from_shading.t
 * params 0-15 host scalars (w_p, w_s, w_g = SQUARED weights, f_x, f_y, u_x, u_y, L_1..L_9), X float unknown (16), D_i (17),
 * Im (18), edgeMaskR/C uint8 (19, 20).  host_params = those 16 floats copied into one host array.
 * precompute (the reference's `precompute` kernels for the computed array B_I_comp and its gradient images,
 * gauss_newton.t:979-986, thallo.t:4046-4094; once per GN iteration and before every cost evaluation) fills
 *   G  float4/pixel = (dBI/dX(c), dBI/dX(c-ex), dBI/dX(c-ey), BI(c)),  Wt float2/pixel = shading row weights,
 *   fl uint8/pixel  = bit0 D_i>0, bit1 reg row valid.
 * U (float2/pixel) and R (3 floats/pixel, planar) are scratch for the row pass of J^T(Jv).
 * Row slabs: W x H is the LOCAL image (2 ghost rows per interior side: the chain B_I -> row -> gather has radius 2);
 * yoff = global row of local row 0, Hg = global image height (pixel coordinates and the border guard of the shading
 * rows are global); precompute visits local rows [ra,rb), the others produce outputs for the owned rows [row0,row1).
 * Whole image: ra=row0=0, rb=row1=H, yoff=0, Hg=H. */
int thallo_hip_sfs_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                              const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl,
                              thallo_stream_t stream);
/* precompute over the rows [ra, rb) and computeCost over the rows [c0, c1) of them in ONE launch (the marching precompute kernel with k_cost's terms riding one row
 * behind); returns the number of cost partials, or -hipErrorNotSupported where the marching kernel does not run (the caller then launches the two separately) */
int thallo_hip_sfs_precompute_cost(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                                   const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl,
                                   int c0, int c1, float* cost_out, thallo_stream_t stream);
int thallo_hip_sfs_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                        const unsigned char* fl, float* cost_out, thallo_stream_t stream);
int thallo_hip_sfs_pcg_init(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                            const unsigned char* fl, float* U, float* R, float* r, float* z, float* p_prev, float* delta,
                            float* diag_out, float* alphaN_out, thallo_stream_t stream);
int thallo_hip_sfs_apply_jtj(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                             float* U, float* R, const float* p, float* Ap, float* alphaD_out, thallo_stream_t stream);
/* LM on one GPU, PCGStep3 folded into the apply (marching kernel only; thallo_hip_sfs_lm_pupdate_supported() says whether this build / environment
 * runs it): p_out = z + beta p_in with beta = betaN_prev / alphaN_prev (0 when first), Ap = (J^T J + CtC) p_out over the rows [row0, row1),
 * alphaD partials; p_out is written on those rows only, p_in != p_out.  Replaces thallo_hip_pcg_pupdate + thallo_hip_sfs_apply_jtj_lm
 * (gauss_newton.t:889-899 + :734-787). */
/* GN on one GPU: ONE launch per PCG iteration (marching kernel only) = thallo_hip_pcg_update + thallo_hip_sfs_apply_jtj_sums_fin:
 * r_out = r_in - alpha Ap_in, p_out = r_out + beta p_in, delta += alpha p_in (alpha = alphaN_prev / alphaD_prev, beta = betaN_prev / alphaN_prev; first:
 * r_out = r_in, p_out = r_in, delta untouched; delta == NULL: never touched -- the caller keeps every p_k (a ring of planes) and adds alpha_k p_k with
 * thallo_hip_linear_update_n), Ap_out = J^T J p_out, alphaD partials, the three double sums {N, S1, S2} of r_out / Ap_out, and with
 * fin.tickets the two scalar words of the iteration.  The in / out planes of r, Ap, p are different buffers (gauss_newton.t:734-752,801-843,889-899). */
int thallo_hip_sfs_pcg_iter(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                            const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                            thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, float* alphaD_out, double* s3_out, thallo_fin_t fin, thallo_stream_t stream);
/* ... with the finish of iteration k-1 deferred into the launch of iteration k (round 4; thallo_prev_t above): every workgroup adds iteration k-1's partials up for itself at its
 * start (the same order, the same bits), workgroup 0 leaves the two words; the launch writes partials only (s3_out != prev.s12_partials); thallo_hip_pcg_scalars_finish behind the
 * loop finishes the last iteration.  The launch loses the tail of the in-kernel finish (write-through slots, tickets, the last workgroup's read-back). */
int thallo_hip_sfs_pcg_iter_deferred(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                     const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                                     thallo_sum_t alphaN_prev, thallo_prev_t prev, float* alphaD_out, double* s3_out, thallo_stream_t stream);
/* Round 3 -- LM on one GPU: ONE launch per PCG iteration.  As thallo_hip_sfs_pcg_iter with A = J^T J + CtC, the LM preconditioner (z = pre r: thallo_hip_lm_finalize_diagonal's M^-1), the scalars divided blindly (gauss_newton.t:226-234), and
 * besides alphaD / {N, S1, S2} the three sums of q's expansion in alpha: q_{k+1} = 0.5 [U + alpha (T1 - T2) - alpha^2 alphaD], U = delta_k.(r_k + b), T1 = p_k.(r_k + b),
 * T2 = delta_k.(A p_k) (q3_out: 3 * THALLO_HIP_MAX_PARTIALS doubles; the reference forms q after the update, :801-843, 965).  The launch's last workgroup (fin.tickets is
 * required) finishes alphaD_k, betaN_k, q_{k+1} and applies thallo_hip_lm_zeta's test to lm_state; once lm_state[1] (the gate) is set later launches return at once, and
 * thallo_hip_lm_owed_delta adds the one update of delta the loop still owes.  Replaces thallo_hip_sfs_apply_jtj_lm_pupdate + thallo_hip_pcg_step2_full_zeta. */
int thallo_hip_sfs_pcg_iter_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                               const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, const float* CtC, const float* b,
                               const float* pre, int first, thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, float* alphaD_out, double* s3_out, double* q3_out,
                               thallo_fin_t fin, float* lm_state, int k, float q_tolerance, thallo_stream_t stream);
/* delta += alpha_kl p_kl with kl = (lm_state[1] ? lm_state[2] : L) - 1: the update of delta the one-launch LM loop owes when it ends (by the gate or after L iterations):
 * the delta half of the last PCGStep2 (gauss_newton.t:801-843).
 * p_even / p_odd: the buffers holding p_k for even / odd k; alphaN_words / alphaD_words: pointers to the scalar words of iteration 0, `word_stride` floats apart per iteration. */
int thallo_hip_lm_owed_delta(float* delta, const float* p_even, const float* p_odd, long n, const float* alphaN_words, const float* alphaD_words, int word_stride,
                             const float* lm_state, int L, thallo_stream_t stream);
int thallo_hip_sfs_lm_pupdate_supported(void);
/* ... and for a W-wide image: no more 60-pixel column strips than the device has workgroup slots (otherwise the LDS-tiled kernels run) */
int thallo_hip_sfs_march_fits(int W);
int thallo_hip_sfs_apply_jtj_lm_pupdate(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                        const float* z, const float* p_in, float* p_out, const float* CtC, float* Ap, float* alphaD_out, int first,
                                        thallo_sum_t alphaN_prev, thallo_sum_t betaN_prev, const unsigned* gate, thallo_stream_t stream);
void thallo_hip_arap_debug_set(int what, int value);     /* tools / tests only: 0 = the unrolled ELL form of the ARAP applyJTJ on (1, default) / off */
/* tools / tests only: 0 = rows per wave segment of the marching J^T(J v) kernel, 1 = workgroups per CU its grid is sized for (0 = automatic),
 * 2 = kernel choice (1 marching, 0 LDS-tiled, -1 the environment's THALLO_SFS_MARCH; default marching) */
void thallo_hip_sfs_march_debug_set(int what, int value);
/* Round 6 -- the layout of the precomputed planes every thallo_hip_sfs_* entry point reads / writes for a W x H (local) image: 0 = G float4 / Wt float2 / fl byte per pixel
 * as described above; 1 = PACKED (images of even width; csrc/sfs_pair.hpp): the G buffer holds four planes of N floats Gx | Gy | Gz | BI, the first 4 N bytes of the Wt
 * buffer one dword per pixel = flags | edgeMaskR << 8 | edgeMaskC << 16 (mask bytes zeroed outside the inner image), fl is not used; the marching kernels then work on
 * pixel PAIRS (energy_sfs_pair.hip: 40 instead of 49 bytes per pixel and GN iteration).  Buffer sizes are the same in both.  thallo_hip_sfs_cost returns
 * -hipErrorNotSupported on packed planes (thallo_hip_sfs_precompute_cost serves).  thallo_hip_sfs_march_debug_set(6, 0 / 1 / -1) forces the layout off / on / back to the
 * environment's (THALLO_SFS_PAIR, default on); what = 7: rows of prefetch of the pair kernels (3 / 6; 0 = automatic). */
int thallo_hip_sfs_planes_layout(int W, int H);
/* Round 6 -- the whole PCG loop of a Gauss-Newton step in ONE launch for images whose solver state fits the chip's registers (the reference's 640 x 480 data set; packed
 * planes, whole image on one GPU; csrc/energy_sfs_resident.hip): L iterations from what thallo_hip_sfs_pcg_init left (r_0 in r_in, zeros in p_in and delta, alphaN_0), leaving
 * what L launches of thallo_hip_sfs_pcg_iter_deferred leave -- r_{L-1}, A p_{L-1}, p_{L-1} in the *_out planes (which may be the *_in planes), delta without its last term,
 * words[2k] = alphaD_k, words[2k + 1] = betaN_k -- bit for bit when both run with the same rows per wave; with X != NULL PCGLinearUpdate rides along (X += delta + alpha_{L-1} p_{L-1},
 * thallo_hip_linear_update's bits; gauss_newton.t:901-906).  thallo_hip_sfs_resident_rows: rows per wave segment, 0 = the shape
 * does not fit (the caller runs one launch per PCG iteration).  xbuf: thallo_hip_sfs_resident_bytes() bytes, zeroed once by the caller, private to the plan.  Returns the number of
 * workgroups, -hipErrorNotSupported when the shape does not fit.  thallo_hip_sfs_resident_status: 1 = a bounded wait inside the kernel ran out (the steps since the last
 * check are void; pm: 5 words of post-mortem, may be NULL); clear != 0 resets; spin_ms >= 0 sets the bound (0 = the 2 s default); synchronises the stream.
 * thallo_hip_sfs_resident_debug_set (tools / tests): 0 = rows per segment, 1 = workgroup budget (0 = automatic), 2 = A/B bits of the exchange layout (4: fault injection -- one
 * workgroup withholds its sums of iteration 2), 3 = the bound of the kernel's waits in ms for the next launches (-1: leave).  Replaces gauss_newton.t:1615-1687 for these shapes. */
int thallo_hip_sfs_resident_rows(int W, int H);
long thallo_hip_sfs_resident_bytes(int W, int H);
int thallo_hip_sfs_pcg_resident(int W, int H, int yoff, const float* host_params, const float* G, const float* Fw,
                                const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                                thallo_sum_t alphaN0, float* words, float* X, void* xbuf, int L, thallo_stream_t stream);
/* ... and a Levenberg-Marquardt step's loop WITH its tail: from what thallo_hip_sfs_pcg_init_lm left (r = b, M^-1 in pre, CtC, zeros in p_prev and delta, alphaN_0) and a reset
 * state (thallo_hip_lm_state_reset), at most L iterations of thallo_hip_sfs_pcg_iter_lm -- the zeta test ends the loop on the device, in every workgroup alike; lm_state[1] /
 * [2] = gate / iterations done as the launches leave them -- then thallo_hip_sfs_lm_model_cost's launch: the owed update of delta (into `delta`), per-workgroup partials of
 * delta . J^T J delta and delta . b (the return value says how many), prevX = X, X += delta.  L <= the residual reset period (gauss_newton.t:1653-1657).
 * thallo_hip_sfs_resident_rows_lm: rows per wave, 0 = does not fit (more registers per row: at most 7 rows per wave; Gauss-Newton: 12). */
int thallo_hip_sfs_resident_rows_lm(int W, int H);
int thallo_hip_sfs_pcg_resident_lm(int W, int H, int yoff, const float* host_params, const float* G, const float* Fw,
                                   const float* r_in, const float* p_in, const float* pre, const float* CtC, float* delta,
                                   thallo_sum_t alphaN0, float* words, float* lm_state, float q_tolerance, float* dJJd_out, float* db_out,
                                   float* X, float* prevX, void* xbuf, int L, thallo_stream_t stream);
int thallo_hip_sfs_resident_status(void* xbuf, int clear, int spin_ms, unsigned* pm, thallo_stream_t stream);
void thallo_hip_sfs_resident_debug_set(int what, int value);
/* Packed planes only (-hipErrorNotSupported elsewhere: the caller runs the launches they replace).
 * thallo_hip_sfs_pcg_init_lm: PCGInit1's J^T F pass with PCGFinalizeDiagonal (gauss_newton.t:936-969) riding along: r = -J^T F, delta = 0, p_prev = 0 and, from the raw diagonal
 * of J^T J formed in the same pass, CtC, pre = M^-1, b = r, z = M^-1 r, SSq (written when save_ssq, else read), partials of r . z -- thallo_hip_sfs_pcg_init +
 * thallo_hip_lm_finalize_diagonal in one launch.
 * thallo_hip_sfs_lm_model_cost: delta_out = delta + alpha_kl p_kl (the update the one-launch LM loop owes: thallo_hip_lm_owed_delta's rule; delta_out != delta) and the partials of
 * delta_out . (J^T J delta_out) and delta_out . b -- thallo_hip_lm_owed_delta + thallo_hip_sfs_apply_jtj + thallo_hip_dot in one launch (whole image on one GPU); with X and
 * prevX (both or neither) also savePreviousUnknowns and PCGLinearUpdate of the step: prevX = X, X = X + delta_out (gauss_newton.t:901-906,915-920).  Returns the number of
 * partials in each of dJJd_out / db_out. */
int thallo_hip_sfs_pcg_init_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                               const unsigned char* fl, float* r, float* z, float* p_prev, float* delta, float* SSq, float* CtC, float* pre, float* b,
                               float radius, float min_lm_diagonal, float max_lm_diagonal, int save_ssq, float* alphaN_out, thallo_stream_t stream);
int thallo_hip_sfs_lm_model_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                 const float* delta, float* delta_out, const float* p_even, const float* p_odd, const float* b, const float* alphaN_words, const float* alphaD_words,
                                 int word_stride, const float* lm_state, int L, float* dJJd_out, float* db_out, float* X /* may be NULL */, float* prevX /* with X */, thallo_stream_t stream);

/* Plain PCGStep1 (gauss_newton.t:734-752): Ap = J^T J p, alphaD partials = sum p.Ap -- the reference-shaped
 * kernel whose algorithmic traffic is SURVEY.md 8d's 48 B/pixel; used by the unfused schedule and by bench.py's
 * stand-alone applyJTJ roofline measurement. */
int thallo_hip_iw_apply_jtj(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg, const float* p, float* Ap, const int* irregular, float* alphaD_out, thallo_stream_t stream);

/* ---------------------------------------------------------------- tuning / diagnostic hooks (tools/microbench.py, tools/sweep_*.sh)
 * Not part of the contract a host layer needs.  thallo_hip_debug_set(what, value): 0 = diagnostic mode of the image_warping
 * step kernel (1 skip arithmetic, 2 skip loads), 3 = its cache-policy bits, 4 = ignore the regular-grid fast path,
 * 5 = workgroups per CU, 6 = threads per workgroup (256 | 512).  thallo_hip_debug_set2(bits): cache policy of PCGStep2. */
void thallo_hip_debug_set(int what, int value);
void thallo_hip_debug_set2(int value);

/* ---------------------------------------------------------------- single-reduction PCG form (any energy whose applyJTJ also returns N, S1, S2)
 * An apply_jtj entry point that takes (r, pre, s3_out) additionally writes, per workgroup b, the doubles
 *   s3_out[3b] = sum r.M^-1.r,  s3_out[3b+1] = sum r.M^-1.Ap,  s3_out[3b+2] = sum Ap.M^-1.Ap     (M^-1 = pre, or 1 if pre == NULL)
 * over the unknowns it produced.  thallo_hip_pcg_scalars_finish (one wave) turns them and the alphaD partials into the two scalar
 * words alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 = r_{k+1}.M^-1 r_{k+1}; thallo_hip_pcg_update is then the only other
 * launch of the iteration: r -= alpha_{k-1} Ap, p_out = M^-1 r + beta_{k-1} p_in, delta += alpha_{k-1} p_in (first: p_out = M^-1 r).
 * Two kernels + one scalar launch per PCG iteration instead of three + two, one reduction point instead of two. */
int thallo_hip_pcg_update(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, int first,
                          thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, thallo_stream_t stream);
int thallo_hip_pcg_scalars_finish(const float* alphaD_partials, const double* s3_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* thallo_hip_pcg_update of iteration k (not the first) with thallo_hip_pcg_scalars_finish of iteration k-1 folded into it (round 4): every workgroup adds the partials of the
 * previous applyJTJ up for itself -- the same order, the same bits -- forms alpha_{k-1}, betaN_{k-1}, beta_{k-1} and updates its share of the vectors; workgroup 0 leaves the two
 * words.  The applyJTJ launch then needs neither tickets nor a read-back at its end (its ~3-us tail on bundle adjustment's 12-us point launch); one thallo_hip_pcg_scalars_finish
 * behind the loop finishes the last iteration. */
int thallo_hip_pcg_update_fin(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, thallo_sum_t alphaN_prev,
                              const float* alphaD_partials, const double* s3_partials, int count, float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* the applyJTJ entry points of E2 / E3 / E4 (argument meaning as in the plain forms below) that also return the three sums */
/* Round 3: applyJTJ with the per-edge G block RECOMPUTED from the source vertex's sines / cosines (SC: [sin a, sin b, sin g] per vertex, then the cosines; written by
 * thallo_hip_arap_precompute2) and dv = Original differences, instead of 2 x 36 streamed bytes per edge; ELL layout with at most 8 edge slots per vertex
 * (thallo_hip_arap_recompute_supported).  s3_out == NULL: alphaD partials only; else the three sums of the single-reduction form and, with fin.tickets, the scalars.
 * Same terms in the same order as thallo_hip_arap_apply_jtj(_sums_fin) (thallo.t:3536-3569's sums, gathered per vertex). */
int thallo_hip_arap_precompute2(int N, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                                const float* original, float w_reg, float* F, float* G, float* SC, long ell_stride, thallo_stream_t stream);
int thallo_hip_arap_recompute_supported(int N, long ell_stride);
int thallo_hip_arap_apply_jtj_rc(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                 const float* constraints, const float* original, const float* SC, float w_fit, float w_reg,
                                 const float* p, float* Ap, float* alphaD_out, long ell_stride, const float* r, const float* pre, double* s3_out,
                                 thallo_fin_t fin, thallo_stream_t stream);
int thallo_hip_arap_apply_jtj_sums(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                                   const float* constraints, const float* G, float w_fit, float w_reg,
                                   const float* p, float* Ap, float* alphaD_out, long ell_stride, const float* r, const float* pre, double* s3_out, thallo_stream_t stream);
/* the same three with the in-kernel finish of the iteration's scalars (thallo_fin_t above); fin.tickets == NULL: identical to the plain forms */
int thallo_hip_arap_apply_jtj_sums_fin(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                                       const float* constraints, const float* G, float w_fit, float w_reg,
                                       const float* p, float* Ap, float* alphaD_out, long ell_stride, const float* r, const float* pre, double* s3_out,
                                       thallo_fin_t fin, thallo_stream_t stream);
int thallo_hip_sfs_apply_jtj_sums_fin(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                      float* U, float* R, const float* p, float* Ap, float* alphaD_out, const float* r, double* s3_out, thallo_fin_t fin, thallo_stream_t stream);
int thallo_hip_ba_apply_jtj2_fin(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                 const float* cameras, const float* points, const float* JP, float* JpC, const float* p, float* Ap, float* alphaD_out,
                                 const float* r, const float* pre, double* s3_out, const unsigned* gate, thallo_fin_t fin, thallo_stream_t stream);
int thallo_hip_ba_apply2_camera_slots(int C_, int P_);      /* how many of thallo_hip_ba_apply_jtj2*'s partial slots (the first ones) are the camera launch's */
/* LM: applyJTJ with PCGStep1_Finish folded in (gauss_newton.t:774-787): Ap = (J^T J + CtC) p, partials of p . Ap; gate as below (may be NULL) */
int thallo_hip_sfs_apply_jtj_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                float* U, float* R, const float* p, const float* CtC, float* Ap, float* alphaD_out, const unsigned* gate, thallo_stream_t stream);
int thallo_hip_ba_apply_jtj2_lm(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                const float* cameras, const float* points, const float* JP, float* JpC, const float* p, const float* CtC, float* Ap, float* alphaD_out,
                                const unsigned* gate, thallo_stream_t stream);
/* LM in the single-reduction form (round 4): an iteration is thallo_hip_pcg_update_lm (r -= alpha_{k-1} A p_{k-1}, delta += alpha_{k-1} p_{k-1}, p_k = M^-1 r + beta_{k-1} p_{k-1};
 * the LM branch's unguarded divides; first = 1: p_0 = M^-1 r; first = 2, behind a residual reset: p_k = M^-1 r + beta_{k-1} p_{k-1} only) and thallo_hip_ba_pcg_apply_lm: A p_k = (J^T J + CtC) p_k, the alphaD partials, {N, S1, S2} and the {U, T1, T2} of
 * q_{k+1} = 0.5 [U + alpha (T1 - T2) - alpha^2 alphaD] (U = delta_k.(r_k + b), T1 = p_k.(r_k + b), T2 = delta_k.A p_k) in double; the point launch's last workgroup finishes
 * alphaD_k, betaN_k, q_{k+1} and applies the zeta test of gauss_newton.t:1666-1686 to lm_state (thallo_hip_lm_zeta's words; word 1 gates both launches).  fin.tickets must be
 * set.  Three launches per LM iteration where thallo_hip_pcg_pupdate + thallo_hip_ba_apply_jtj2_lm + thallo_hip_pcg_step2_full_zeta were four, one reduction point instead of
 * two; thallo_hip_lm_owed_delta applies the last alpha p behind the loop.  Replaces gauss_newton.t:734-787,801-843,889-899 on the LM branch. */
int thallo_hip_pcg_update_lm(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, int first,
                             thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, float* betaN_word, const float* lm_state, thallo_stream_t stream);
/* The residual reset of the LM loop (gauss_newton.t:1653-1657) for that form, behind thallo_hip_lm_step2_first_half (delta += alpha_k p_k): the two gather launches with
 * p = delta, r = b - (J^T J + CtC) delta and the partials of betaN_k = r . M^-1 r as their epilogue (returns the number of partials); the next thallo_hip_pcg_update_lm
 * (first = 2) adds them up, forms p_{k+1} and leaves betaN_k in betaN_word (may be NULL otherwise).  Three launches per reset where the reference-shaped loop has five. */
int thallo_hip_ba_lm_reset_residual(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                    const float* cameras, const float* points, const float* JP, float* JpC, const float* delta, const float* CtC, const float* b, const float* pre,
                                    float* r, float* betaN_out, const unsigned* gate, thallo_stream_t stream);
int thallo_hip_ba_pcg_apply_lm(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                               const float* cameras, const float* points, const float* JP, float* JpC, const float* p, const float* CtC, float* Ap, float* alphaD_out,
                               const float* r, const float* pre, const float* delta, const float* b, double* s3_out, double* q3_out, thallo_fin_t fin,
                               float* lm_state, int k, float q_tolerance, int q_in, int q_out, thallo_stream_t stream);
/* ... and with the finish deferred (round 4): thallo_hip_ba_pcg_apply_lm with fin.tickets = NULL leaves partials only, and the flat update of the NEXT iteration
 * (thallo_hip_pcg_update_lm_fin) finishes alphaD, betaN, q and the zeta test in every workgroup before it updates -- the same arithmetic and order.  q_in / q_out: the words of
 * lm_state Q0 is read from / Q1 is left in (0 and 6 by the iteration's parity in that loop, so that no launch reads a word one of its workgroups writes; 0, 0 in the loop that
 * finishes in the applyJTJ launch).  Iterations that a residual reset follows, and the last one, finish in their own launch (fin.tickets set). */
int thallo_hip_pcg_update_lm_fin(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, thallo_sum_t alphaN_prev,
                                 const float* alphaD_partials, const double* s3_partials, const double* q3_partials, int count, float* alphaD_word, float* betaN_word,
                                 float* lm_state, int k_prev, float q_tolerance, int q_in, int q_out, thallo_stream_t stream);
/* shape_from_shading applyJTJ with a device-side gate word (may be NULL): non-zero = the launch does nothing (the LM branch ends its PCG loop on
 * the device without a host round trip per iteration, solver.cpp) */
int thallo_hip_sfs_apply_jtj_gated(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                   float* U, float* R, const float* p, float* Ap, float* alphaD_out, const unsigned* gate, thallo_stream_t stream);
/* J^T (J p) with J p formed once (energy_ba.hip): q_ptk[q] = position of observation q in its point's list (inverse of pt_pos), JP = the
 * 6 point partials of every observation packed in that order once per GN iteration, JpC = 2 floats per observation of workspace (J p in camera order: written coalesced by
 * the camera kernel, gathered through pt_pos by the point kernel -- round 4; rounds 2-3 scattered it into point order).  The camera kernel REBUILDS an observation's block
 * from `cameras` / `points` (the unknowns as they were when thallo_hip_ba_compute_j ran: they do not change inside a PCG loop) in closed form instead of loading its 96 bytes.
 * r / pre / s3_out (all or none): also the three double sums of the single-reduction PCG form; gate: see thallo_hip_lm_set_gate. */
int thallo_hip_ba_point_order(int O, const int* pt_pos, int* q_ptk, thallo_stream_t stream);
int thallo_hip_ba_pack_point_blocks(int O, const float* Jb, const int* q_ptk, float* JP, thallo_stream_t stream);
int thallo_hip_ba_apply_jtj2(int C, int P, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                             const float* cameras, const float* points, const float* JP, float* JpC, const float* p, float* Ap, float* alphaD_out,
                             const float* r, const float* pre, double* s3_out, const unsigned* gate, thallo_stream_t stream);
int thallo_hip_sfs_apply_jtj_sums(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                  float* U, float* R, const float* p, float* Ap, float* alphaD_out, const float* r, double* s3_out, thallo_stream_t stream);

/* ---------------------------------------------------------------- materialized schedules (CSR) */
/* y = A x for a CSR matrix (rows+1 row pointers, int32 columns, float values); with dot_with / dot_out (both or neither) it also writes
   the per-workgroup partials of dot_with . y.  Replaces the cuSPARSE csrmv calls of gauss_newton.t:1470-1517: `[Jt][[J]p]` = two calls
   (J, then J^T with dot_with = p), `[[Jt][J]]p` = one call on the pre-multiplied J^T J. */
int thallo_hip_csr_spmv(int rows, const int* rowptr, const int* col, const float* val, const float* x, float* y,
                        const float* dot_with, float* dot_out, thallo_stream_t stream);
/* Materialized J of a generated plugin in ELL form ([rows][K] values and unknown indices, -1 = no unknown), applied without a transpose:
 * mode 0: Ap += J^T (J p) in one pass ([Jt][[J]p]); mode 1: Jp = J p; mode 2: Ap += J^T Jp (the Jt[Jp] pair on a materialized J). */
int thallo_hip_ell_apply(int mode, long rows, int K, const float* val, const int* col, const float* p, float* Jp, float* Ap, thallo_stream_t stream);
/* dense J^T J (n x n, zeroed by the caller) from materialized rows, and y = M x: the dense [JtJ]p schedule for small n (gauss_newton.t:560-622) */
int thallo_hip_dense_jtj_accumulate(long rows, int K, const float* val, const int* col, long n, float* JtJ, thallo_stream_t stream);
int thallo_hip_dense_gemv(long n, const float* M, const float* x, float* y, thallo_stream_t stream);
/* Numeric phase of the sparse J^T J of a non-constant J ([[Jt][J]]p, gauss_newton.t:1394-1441 csrgemm): out[dest[(i*K + a)*K + b]] += val[i*K + a] * val[i*K + b]
 * over the ELL rows of J; `dest` (positions in the CSR values, -1 = none) comes from the symbolic phase the host runs once per Init. */
int thallo_hip_jtj_scatter(long rows, int K, const float* val, const int* dest, float* out, thallo_stream_t stream);
/* Per-owner instance lists of a generated plugin's gather through index maps (round 5: built on the device; the reference maps a residual group at its output when
 * it can, thallo.t:5273-5306).  col[el * K + q]: the flat unknown index slot q of residual instance el touches (-1: none), from the residual's own index evaluation
 * (the generated uidx kernel).  slot_base[q] >= 0: slot q's image belongs to the owner group, its flat offset; slot_ch[q]: its channels; owner = (col - base) / ch in
 * [0, npix).  count: ptr[0 .. npix] (device) becomes the CSR row pointer of the lists (an instance counts once per distinct owner), *total_dev (device, 8 bytes) their
 * total length.  fill: els[ptr[px] .. ptr[px + 1]) = the instances of owner px in ascending order; cursor: npix ints of scratch.  K <= THALLO_HIP_INC_MAX_SLOTS. */
#define THALLO_HIP_INC_MAX_SLOTS 48
/* Measurement (the library's per-kernel timer, bench.py's roofline figure): arm -- the NEXT marching PCG iteration this thread launches (thallo_hip_iw_pcg_iter_march_rc*)
 * carries the two events as hipExtLaunchKernelGGL's start / stop events, i.e. they take the kernel's own begin / end timestamps (what rocprofv3 reports as its duration;
 * events recorded around a launch also contain the dispatch gap in front of it).  take -- disarm; 1 if a launch used them (then, and only then, they are recorded). */
void thallo_hip_launch_events_arm(void* start_event, void* stop_event);
int thallo_hip_launch_events_take(void);
int thallo_hip_incidence_count(const int* col, long n, int K, const long* slot_base, const int* slot_ch, long npix, int* ptr, long* total_dev, thallo_stream_t stream);
int thallo_hip_incidence_fill(const int* col, long n, int K, const long* slot_base, const int* slot_ch, long npix, const int* ptr, int* cursor, int* els, thallo_stream_t stream);
/* ---- doublePrecision = 1 (precision.t:3-6: thallo_float = double): the energy-independent PCG kernels on double vectors, reference-shaped and unfused
 * (gauss_newton.t:712-731, 774-787, 801-843, 889-899, 901-906).  Reductions leave per-workgroup partials (the return value = how many, <= THALLO_HIP_MAX_PARTIALS),
 * thallo_hip_f64_finish adds them in index order into one device word, and the kernels that need a scalar read such words: nothing goes through the host.
 *   init_finish: pre = guardedInvert(pre) (or 1), z = pre r, p = z, partials of r . z (alphaN_0)
 *   dot:         partials of a . b (PCGStep1_Finish: alphaD = p . Ap)
 *   step2:       alpha = alphaN / alphaD (0 when alphaD = 0); delta += alpha p; r -= alpha Ap; z = pre r; partials of z . r (betaN)
 *   step3:       beta = betaN / alphaN (0 when alphaN = 0); p = z + beta p
 *   linear_update: X += delta */
int thallo_hip_f64_init_finish(const double* r, double* pre, double* z, double* p, long n, int use_preconditioner, double* partials_out, thallo_stream_t stream);
int thallo_hip_f64_dot(const double* a, const double* b, long n, double* partials_out, thallo_stream_t stream);
int thallo_hip_f64_step2(double* delta, double* r, double* z, const double* p, const double* Ap, const double* pre, long n, const double* alphaN_word, const double* alphaD_word,
                         double* partials_out, thallo_stream_t stream);
int thallo_hip_f64_step3(double* p, const double* z, long n, const double* betaN_word, const double* alphaN_word, thallo_stream_t stream);
int thallo_hip_f64_linear_update(double* X, const double* delta, long n, thallo_stream_t stream);
int thallo_hip_f64_finish(const double* partials, int count, double* word, thallo_stream_t stream);
/* ... and the Levenberg-Marquardt set in double (round 4; the float forms: thallo_hip_lm_finalize_diagonal, thallo_hip_lm_step1_finish, thallo_hip_pcg_step2_full,
 * thallo_hip_lm_step2_first_half / _second_half; gauss_newton.t:929-969, 774-787, 801-886).  LM divides blindly (gauss_newton.t:226-234).
 *   lm_finalize_diagonal: from the RAW diagonal d: SSq (first GN iteration only) = guardedInvert(d) or 1; CtC = clamp(d / radius, min / (SSq radius), max / (SSq radius));
 *                         pre = 1 / (CtC + d); b = r; z = pre r; partials of r . z
 *   lm_step1_finish:      Ap += CtC p; partials of p . Ap
 *   lm_step2:             delta += alpha p; r -= alpha Ap; z = pre r; partials of z . r and of q = 0.5 delta . (r + b)
 *   lm_step2_first_half / _second_half: the residual reset -- delta += alpha p; then (with Adelta = (J^T J + CtC) delta) r = b - Adelta, z, the two partial sets
 *   lm_step3:             p = z + (betaN / alphaN) p */
int thallo_hip_f64_lm_finalize_diagonal(const double* diag, double* SSq, double* CtC, double* pre, const double* r, double* b, double* z, long n, double radius, double min_lm_diagonal,
                                        double max_lm_diagonal, int save_ssq, int use_preconditioner, double* partials_out, thallo_stream_t stream);
int thallo_hip_f64_lm_step1_finish(double* Ap, const double* CtC, const double* p, long n, double* partials_out, thallo_stream_t stream);
int thallo_hip_f64_lm_step2(double* delta, double* r, double* z, const double* p, const double* Ap, const double* pre, const double* b, long n, const double* alphaN_word,
                            const double* alphaD_word, double* betaN_out, double* q_out, thallo_stream_t stream);
int thallo_hip_f64_lm_step2_first_half(double* delta, const double* p, long n, const double* alphaN_word, const double* alphaD_word, thallo_stream_t stream);
int thallo_hip_f64_lm_step2_second_half(double* r, const double* b, const double* Adelta, const double* pre, double* z, const double* delta, long n, double* betaN_out, double* q_out,
                                        thallo_stream_t stream);
int thallo_hip_f64_lm_step3(double* p, const double* z, long n, const double* betaN_word, const double* alphaN_word, thallo_stream_t stream);
/* Direct solve of the dense normal equations (gauss_newton.t:1280-1328; compiled out there, opt-in here): A (n x n row-major, symmetric positive
 * definite, OVERWRITTEN by its Cholesky factor) x = b.  info[0] (device int) = 0, or 1 + the row of the first non-positive pivot. n <= 8192. */
int thallo_hip_dense_cholesky_solve(long n, float* A, const float* b, float* x, int* info, thallo_stream_t stream);

/* ---------------------------------------------------------------- multi-GPU device-side exchange (one process per GPU) */
/* Device memory that other processes can map: *ptr = hipMalloc(bytes) (zeroed), handle_out = 64-byte hipIpcMemHandle_t. */
int thallo_hip_ipc_alloc(long bytes, void** ptr, void* handle_out64);
/* the same, reporting the memory kind it got: 1 fine-grained device memory (the default: what peers write and running kernels poll),
 * 0 plain coarse-grained hipMalloc (THALLO_DIST_MEM=coarse, or the fine-grained allocation / its IPC export was refused) */
int thallo_hip_ipc_alloc2(long bytes, void** ptr, void* handle_out64, int* kind_out);
int thallo_hip_ipc_open(const void* handle64, void** ptr);      /* maps a peer's allocation (enables peer access lazily) */
int thallo_hip_ipc_close(void* ptr);
int thallo_hip_ipc_free(void* ptr);
/* seq += 1 (start of a GN step); error word untouched */
int thallo_hip_dist_begin_step(thallo_dist_t d, thallo_stream_t stream);
/* The exchange, one single-wave launch behind the producing kernel: local = that kernel's per-workgroup partials; adds them in
   the single-GPU order, stores the sum as one granule into EVERY rank's mailbox slot, waits (bounded) until every rank's
   granule of the slot carries the current seq, out[0] = their rank-ordered sum.  The kernels after it on the stream read out[0]
   as a thallo_sum_t of count 1; the kernel boundaries also order the neighbours' ghost rows (stored by their PCGStep2 before
   their granule) ahead of every later read. */
int thallo_hip_dist_exchange(thallo_dist_t d, int slot, thallo_sum_t local, float* out, thallo_stream_t stream);
/* The exchange of the one-kernel-per-iteration schedule: local fixed-order sums of the alphaD partials (float) and of the N, S1, S2
   partials (double) of thallo_hip_iw_pcg_iter(_dist), 7 granules to every rank (slots slot0 .. slot0+6), bounded wait, rank-ordered
   sums, then alphaD_word[0] = alphaD_k and betaN_word[0] = N - 2 alpha_k S1 + alpha_k^2 S2 with alpha_k = alphaN / alphaD_k
   (alphaN: a one-word sum).  ONE exchange per PCG iteration. */
int thallo_hip_dist_exchange_iter(thallo_dist_t d, int slot0, const float* alphaD_partials, const double* s12_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* out[j] = rank-ordered sum of slot slot0+j for j < nslots (waits for each); diagnostics */
int thallo_hip_dist_collect(thallo_dist_t d, int slot0, int nslots, float* out, thallo_stream_t stream);
/* Rows + scalars of a flat solver vector between row slabs in ONE launch, device side (round 3; replaces slab_pack + all-gather + slab_unpack of the
   single-image slab form, i.e. shape_from_shading's Gauss-Newton and Levenberg-Marquardt exchanges; reference: none -- single device, util.t:769-772).
   Every rank's mailbox allocation carries, behind a ring of 4 x 16 scalar slots (slots ring0 ..), an INBOX of 2 (parity) x 2 (from above, from below) areas of
   inbox_half floats at byte offset inbox_off (the same numbers on every rank).  The launch (one workgroup for rows up to 32 K floats, else 8 workgroups and a ticket):
     1. stores the `first` segments of vec into the upper neighbour's inbox (its "from below" area) and the `last` segments into the lower neighbour's
        ("from above"), peer-to-peer, fences, takes a ticket;
     2. its last workgroup adds the local partials in the fixed single-GPU order, sends the scalar granules {value | tag} to every rank, waits (bounded, as
        everywhere: ctl[1] error word + post-mortem) for every rank's, adds them in rank order -- the same bits as the all-gather path on every rank --
        and writes the result words:  mode 0: out0[0] = sum over ranks of sum(local) (local.count == 0: no scalar, out0 ignored), and optionally a second one:
        out1[0] = sum over ranks of sum(alphaD_partials[0 .. count)) (count == 0: none);
        mode 1 (iteration form): out0[0] = alphaD_k, out1[0] = betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 from alphaD_partials / s3_partials[count]
        and the one-word alphaN, exactly thallo_hip_slab_unpack_iter's arithmetic;
     3. then copies its own inbox areas into the `top` / `bot` ghost segments of vec (a neighbour's granule is sent after its rows are fenced).
   tag = ctl[10] + 1, a device-side counter of these exchanges (every rank issues the same sequence of them); parity = tag & 1: a rank can be at most one
   exchange ahead of a neighbour, which still reads the other parity.  poison != 0: this rank failed earlier -- it sends NaN scalars and no rows so that
   nobody waits for it and every rank sees the failure in its sums. */
typedef struct thallo_xrows_t {
    long inbox_off;              /* bytes from the start of a rank's mailbox allocation */
    long inbox_half;             /* floats per (parity, direction) area; >= the total length of `first` / `last` */
    int  above, below;           /* neighbour ranks; -1: image border */
    int  ring0;                  /* first scalar slot of the ring (4 x 16 slots) */
} thallo_xrows_t;
int thallo_hip_dist_xrows(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                          int mode, thallo_sum_t local_or_alphaN, const float* alphaD_partials, const double* s3_partials, int count, int poison,
                          float* out0, float* out1, thallo_stream_t stream);
/* ... mode 0 with the LM zeta test on the FIRST sum (q): the wave that holds the global q applies thallo_hip_lm_zeta's rule to lm_state right there (one launch
   less per LM iteration of a slab) */
int thallo_hip_dist_xrows_zeta(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                               thallo_sum_t q_local, const float* second_partials, int second_count, int poison, float* q_out, float* second_out,
                               float* lm_state, int k, float q_tolerance, thallo_stream_t stream);
/* Ghost units of a PARTITIONED graph problem (round 3; SURVEY.md 8e row 2: ARAP with ghost vertices; the reference has no analogue -- single device, API/src/util.t:769-772;
   the vectors exchanged are those of gauss_newton.t:282-323).  A rank's local problem = its owned units (vertices) first, then
   the ghosts: units owned elsewhere that its owned ones touch.  Per exchange a rank sends, behind the scalars, the values of its BOUNDARY units (owned here, ghost
   somewhere) -- `per` floats per unit, taken from up to 8 planes of the flat vector (plane k: base[k] + unit * len[k], len[k] floats) -- and fills each of its ghosts from
   (source rank, position in that rank's boundary list).  thallo_hip_units_pack(_iter): message = [1 (or 7) scalar words as thallo_hip_slab_pack(_iter) writes them |
   boundary unit 0's floats | unit 1's | ...]; thallo_hip_units_unpack(_iter): the rank-ordered scalar sums exactly as thallo_hip_slab_unpack(_iter), and
   vec[plane k of ghost g] <- gathered[ghost_src[g] ...] with ghost_src[g] = source rank * stride + header + position * per (element offset, precomputed by the host). */
typedef struct thallo_units_t {
    const int*  units;           /* DEVICE: pack: the boundary units (local ids); unpack: the ghost units (local ids) */
    const long* src;             /* DEVICE, unpack only: element offset of each ghost's floats inside the gathered buffer */
    int  n;                      /* how many */
    int  nplanes;                /* <= 8 */
    long base[8];                /* plane k starts at this element of the flat vector */
    int  len[8];                 /* floats per unit in plane k */
} thallo_units_t;
int thallo_hip_units_pack(const float* vec, thallo_units_t u, thallo_sum_t sum, float* out, thallo_stream_t stream);
int thallo_hip_units_unpack(float* vec, thallo_units_t u, const float* gathered, long stride, int world, float* sum_out, thallo_stream_t stream);
int thallo_hip_units_pack_iter(const float* vec, thallo_units_t u, const float* alphaD_partials, const double* s3_partials, int count, float* out, thallo_stream_t stream);
int thallo_hip_units_unpack_iter(float* vec, thallo_units_t u, const float* gathered, long stride, int world, thallo_sum_t alphaN,
                                 float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* ... the exchange of a slab's one-launch LM iteration (thallo_hip_sfs_pcg_iter_lm without tickets): the ranks' alphaD / {N, S1, S2} / {U, T1, T2} partials travel as 13
   granules, are added in rank order, and the wave that holds them writes alphaD_k, betaN_k, forms q_{k+1} = 0.5 [U + alpha (T1 - T2) - alpha^2 alphaD] and applies the zeta
   test to lm_state; the boundary rows of the new A p travel as in thallo_hip_dist_xrows.  ONE exchange per LM iteration. */
int thallo_hip_dist_xrows_lm(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                             thallo_sum_t alphaN, const float* alphaD_partials, const double* s3_partials, const double* q3_partials, int count, int poison,
                             float* alphaD_word, float* betaN_word, float* lm_state, int k, float q_tolerance, thallo_stream_t stream);
/* thallo_hip_dist_xrows for a PARTITIONED graph (thallo_units_t): the boundary units' values go into every other rank's inbox (area [parity][this rank] of
   unit_slot_floats floats at x.inbox_off), the scalars travel as in thallo_hip_dist_xrows (mode 0 / 1), and the ghost units are filled from the rank's own inbox --
   recv.src[g] = source rank * unit_slot_floats + position * floats per unit.  One launch, no collective. */
int thallo_hip_dist_xunits(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_units_t send, thallo_units_t recv, long unit_slot_floats,
                           int mode, thallo_sum_t local_or_alphaN, const float* alphaD_partials, const double* s3_partials, int count, int poison,
                           float* out0, float* out1, thallo_stream_t stream);
/* The scalars of a PCG iteration in the shard form (bundle adjustment across camera shards) without a collective: every rank's OWN partials (its cameras':
   alphaD float, {N, S1, S2} double) travel as 7 granules and are added in rank order; the sums of the SHARED block (the points, identical on every rank after the
   all-reduce, thallo_hip_block_sums) join behind them; then alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 -- thallo_hip_shard_scalars' arithmetic and order.
   x: only ring0 is used (no rows travel). */
int thallo_hip_dist_xscalars_shard(thallo_dist_t d, thallo_xrows_t x, thallo_sum_t alphaN, const float* own_alphaD_partials, const double* own_s3_partials, int own_count,
                                   const float* shared_alphaD_partials, const double* shared_s3_partials, int shared_count, int poison,
                                   float* alphaD_word, float* betaN_word, thallo_stream_t stream);
/* In-place sum over all ranks of `len` floats at `buf`, by peer stores only (round 3; the reference has no analogue -- single device, API/src/util.t:769-772; bundle adjustment's point block across camera shards: SURVEY.md 8e row 3, "prefer
   the direct form" -- replaces ncclAllReduce in the PCG loop).  Reduce-scatter + all-gather in ONE launch: rank r owns chunk r (x.chunk floats); every rank stores its part
   of chunk c into rank c's inbox, the owner adds the `world` contributions IN RANK ORDER (every run and every rank gets the same bits) and stores the sums into every rank's
   second inbox, from which they are copied into place.  Two waits on tagged granules (bounded, error word + post-mortem as everywhere), parity double-buffering: a rank can
   be at most one all-reduce ahead.  Mailbox allocation of every rank: [scalar ring | at inbox_off: A[2][world][chunk], then B[2][world][chunk]].  len % 4 == 0,
   chunk % 4 == 0, chunk * world >= len; all workgroups of the launch are resident at once (<= 64).  tag = ctl[12] + 1.  poison: this rank failed earlier; it takes part
   (nobody waits for it) and turns the first element of its chunk into NaN. */
typedef struct thallo_xreduce_t {
    long inbox_off;              /* bytes from the start of a rank's mailbox allocation */
    long chunk;                  /* floats per rank */
    int  ring0;                  /* first of 8 scalar slots (2 granule kinds x 4 ring positions) */
} thallo_xreduce_t;
int thallo_hip_dist_allreduce(thallo_dist_t d, thallo_xreduce_t x, float* buf, long len, int poison, thallo_stream_t stream);
/* host-side read / clear of the error word (synchronises the stream) */
int thallo_hip_dist_error(thallo_dist_t d, int clear, thallo_stream_t stream);
/* image_warping PCGStep2 over a row slab (z-free schedule only: UrShape must be the unit pixel grid on every rank) that also
   stores its first / last owned row of r into the neighbours' ghost rows (d.peer_r) with peer-to-peer stores. */
int thallo_hip_iw_pcg_step2_dist(int W, int H, int row0, int row1, const unsigned char* flags, float w_fit, float w_reg,
                                 float* r, const float* Ap, thallo_sum_t alphaN, thallo_sum_t alphaD,
                                 thallo_dist_t d, float* betaN_out, thallo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
