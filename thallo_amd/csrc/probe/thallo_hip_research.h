/* probe/thallo_hip_research.h -- entry points that exist in RESEARCH builds only (make -C thallo_amd/csrc VARIANT=research -> tools/ab/libThallo_research.so, -DTHALLO_RESEARCH).
 * Two one-launch loops that round 5 built, pinned bit for bit against the launch-per-iteration forms, measured SLOWER and therefore keeps out of the product library
 * (VERDICT r5 weak 10): image_warping's persistent marching loop (probe/iw_march_persist.hip, THALLO_AB=persist=1) and bundle adjustment's resident PCG loop
 * (probe/ba_resident_*.inc, THALLO_RESIDENT=2).  Their tests (tests/test_gpu_parity.py, marked `research`) and tools run with THALLO_LIB pointing at that library. */
#pragma once
#include "../../../include/thallo_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- the marching PCG iteration as a PERSISTENT loop (probe/iw_march_persist.hip): iterations k0 .. k1-1 of a GN step in ONE launch, for whole images on
 * the unit pixel grid whose solver state does not fit the chip's registers (2048^2: 35 rows per wave).  The launch-per-iteration grid stays on the chip; the
 * iteration's sums (a tagged record per workgroup) are its one synchronisation point, r_k / p_k are stored write-through.  What a launch per iteration
 * (thallo_hip_iw_pcg_iter_march_rc_deferred, delta mode "none") reads and leaves behind, in the plan's own layout:
 *   r[k & 1] -> r[(k + 1) & 1];  p_{k-1} in planes[(k - 1) % n_planes] -> p_k in planes[k % n_planes]   (n_planes >= k1 - k0 + 1);
 *   reduction slots at parts + j * THALLO_HIP_MAX_PARTIALS, their words at parts + slots * THALLO_HIP_MAX_PARTIALS + j; alphaN_k = slot B + 2k, alphaD_k = B + 2k + 1, betaN_k = B + 2k + 2:
 *   read  alphaN_{k0-1} (alphaN_prev: partials or one word), the nb_prev alphaD partials of iteration k0 - 1 and its double sums in s12[(k0 - 1) & 1];
 *   write the words alphaD_{k-1}, betaN_{k-1} for k0 <= k < k1 and the partials of iteration k1 - 1 (alphaD slot, s12[(k1 - 1) & 1]; as many as the return value).
 * xbuf: thallo_hip_iw_march_persist_bytes() bytes, zeroed once by the caller, private to the plan.  Bit-identical to the launches it replaces (same strips, segments,
 * expressions and order of every sum).  Every workgroup must be resident (one per CU; checked at launch); every wait inside is bounded (2 s, or spin_ms), an expired
 * one sets the error word thallo_hip_iw_march_persist_status reads.  Returns the number of workgroups (> 0), -hipErrorNotSupported when the shape does not fit.
 * Replaces the loop of gauss_newton.t:1615-1687. */
long thallo_hip_iw_march_persist_bytes(void);
int thallo_hip_iw_march_persist_rows(int W, int H);      /* rows per wave, 0 = the shape does not run as a persistent loop */
int thallo_hip_iw_pcg_march_persist(int W, int H, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    float* r0, float* r1, float* const* planes, int n_planes, int k0, int k1,
                                    float* parts, int slots, int B, double* s12_0, double* s12_1, int nb_prev, thallo_sum_t alphaN_prev,
                                    const int* irregular, void* xbuf, thallo_stream_t stream);
int thallo_hip_iw_march_persist_status(void* xbuf, int clear, int spin_ms, unsigned* post_mortem5, thallo_stream_t stream);
void thallo_hip_iw_march_persist_debug_set(int what, int value);      /* tools: what 0 = the acquire form (1) instead of L1-bypassing loads (0) */


/* The PCG loop of a Gauss-Newton step of bundle adjustment in ONE launch (round 5): min(CUs, 256) workgroups stay on the chip and run the flat update, the camera
   kernel and the point kernel of thallo_hip_pcg_update(_fin) + thallo_hip_ba_apply_jtj2 as phases of one loop, a grid-wide arrival barrier behind each; every
   partial, sum and vector element has the bits of the three-launch form.  In: what thallo_hip_ba_pcg_init left (r_0, M^-1, delta = 0, alphaN_0) and the packed point
   blocks JP.  Out: r_{L-1}, A p_{L-1}, p_{L-1} in p0 / p1 by L & 1 (0: p0), delta without its last term, words[2k] = alphaD_k, words[2k + 1] = betaN_k.
   xbuf: thallo_hip_ba_resident_bytes() bytes, zeroed once, private to the plan; thallo_hip_ba_resident_status reads the error word a bounded wait that ran out leaves.
   Replaces gauss_newton.t:1615-1687 (GN branch, one GPU). */
long thallo_hip_ba_resident_bytes(void);
int thallo_hip_ba_pcg_resident(int C, int P, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                               const float* cameras, const float* points, const float* JP, float* JpC,
                               float* r, float* Ap, const float* pre, float* p0, float* p1, float* delta, thallo_sum_t alphaN0, float* words, void* xbuf, int L, thallo_stream_t stream);
int thallo_hip_ba_resident_status(void* xbuf, int clear, unsigned* post_mortem5, thallo_stream_t stream);
void thallo_hip_ba_resident_debug_set(int what, int value);      /* tools: what 0 = workgroups of the resident loop (0: two per CU) */
#ifdef __cplusplus
}
#endif
