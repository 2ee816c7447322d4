// sfs_pair.hpp -- host-side interface between energy_sfs.hip (the C-ABI entry points of shape_from_shading) and energy_sfs_pair.hip (round 6: the marching kernels on
// PIXEL PAIRS and the PACKED plane layout they run on).  Not part of the C ABI: thallo_hip_sfs_* keep their signatures and dispatch here when sfs_pair_ok() says the image
// runs in this form (the caller's G / Wt / fl buffers then hold the packed planes below; thallo_hip_sfs_planes_layout() tells).
//
// Packed planes (written by sfs_pair_precompute, read by every other function here):
//   G buffer (16 N bytes):  Gx | Gy | Gz | BI   four planes of N floats:  dBI/dX(c), dBI/dX(c-ex), dBI/dX(c-ey), BI(c)   -- planar, so that a lane's two pixels arrive as
//                           one aligned register pair per plane (8-byte loads) and the iteration never touches BI (4 of the legacy layout's 16 bytes)
//   Wt buffer (first 4 N bytes):  one dword per pixel = flags (bit 0 D > 0, bit 1 reg row valid) | edgeMaskR << 8 | edgeMaskC << 16, the mask bytes zeroed outside
//                           the inner image -- the row weights h = w_g * maskR, k = w_g * maskC are two conversions and two multiplications, not 8 bytes
//   fl buffer: not used.
// Per pixel and PCG iteration (GN, delta left to the ring of p planes): Gx, Gy, Gz 12 + flags / masks 4 + r, A p, p read 12 and written 12 = 40 bytes
// (legacy layout: 49).  Replaces the same reference functions as energy_sfs.hip (gauss_newton.t:734-752,801-843,889-899,936-969,979-986).
#pragma once
#include "../../include/thallo_hip.h"

namespace thallo {

struct SfsTune { int rows = 0, wgcu = 0, cap = 0, depth = 0; };      // tools / tests: rows per wave segment, workgroups per CU the grid is sized for, workgroup budget, rows of prefetch (0 = automatic)

// the image runs on pixel pairs: even width, no more 124-pixel strips than workgroup slots, every plane inside one buffer descriptor
bool sfs_pair_ok(int W, int H, const SfsTune& t);

// what one launch needs of the LM step (PCGFinalizeDiagonal folded into the J^T F pass, gauss_newton.t:936-969): NULL pointers = not folded
struct SfsFinDiag { float* SSq = nullptr; float* CtC = nullptr; float* pre = nullptr; float* b = nullptr; float radius = 0.f, min_lm = 0.f, max_lm = 0.f; int save_ssq = 0; };

int sfs_pair_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* hp, const float* X, const float* D, const float* Im, const unsigned char* mR, const unsigned char* mC,
                        float* G, float* Fw, float* cost_out, int c0, int c1, const SfsTune& t, thallo_stream_t stream);
int sfs_pair_init(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* X, const float* D, const float* G, const float* Fw,
                  float* r, float* z, float* p_prev, float* delta, float* diag_out, float* aN_out, const SfsFinDiag& fd, const SfsTune& t, thallo_stream_t stream);
int sfs_pair_apply(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* p, float* Ap, float* aD_out,
                   const float* r, double* s3_out, const unsigned* gate, thallo_fin_t fin, const float* ctc, const SfsTune& t, thallo_stream_t stream);
int sfs_pair_apply_pupdate(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* z, const float* p_in, float* p_out,
                           const float* ctc, float* Ap, float* aD_out, int first, thallo_sum_t aN_prev, thallo_sum_t bN_prev, const unsigned* gate, const SfsTune& t, thallo_stream_t stream);
// GN iteration (prev.count > 0: the finish of iteration k-1 deferred into this launch; else aD_prev / bN_prev are sums and fin may carry tickets)
int sfs_pair_iter(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw,
                  const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                  thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev, const thallo_prev_t* prev, float* aD_out, double* s3_out, thallo_fin_t fin,
                  const SfsTune& t, thallo_stream_t stream);
int sfs_pair_iter_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw,
                     const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, const float* CtC, const float* b,
                     const float* pre, int first, thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev, float* aD_out, double* s3_out, double* q3_out,
                     thallo_fin_t fin, float* lm_state, int k, float q_tol, const SfsTune& t, thallo_stream_t stream);
// LM model cost in ONE launch (round 6): delta_out = delta + alpha_kl p_kl (the update the one-launch LM loop owes, kl from lm_state as thallo_hip_lm_owed_delta; delta_out != delta:
// the halo rows of neighbouring segments read the old plane), J^T J delta_out and the partials of delta_out . J^T J delta_out (dJJd_out) and delta_out . b (db_out) -- replaces
// thallo_hip_lm_owed_delta + thallo_hip_sfs_apply_jtj + thallo_hip_dot
int sfs_pair_model_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* delta, float* delta_out, const float* p_even, const float* p_odd,
                        const float* b, const float* alphaN_words, const float* alphaD_words, int word_stride, const float* lm_state, int L, float* dJJd_out, float* db_out,
                        float* X, float* prevX /* both or neither: prevX = X, X += delta_out ride along (savePreviousUnknowns + PCGLinearUpdate) */, const SfsTune& t, thallo_stream_t stream);

}  // namespace thallo
