// pcg_kernels.hip -- energy-independent kernels of the PCG chain (gfx950).
//
// All are pure streaming kernels over the flat unknown vector (HBM-bound; roofline = bytes/8 TB/s):
// 16-byte accesses per lane, persistent grid of <= 1024 workgroups, grid-stride loop, one
// partial per workgroup.  Reference kernels replaced: gauss_newton.t:801-843 (PCGStep2),
// :889-899 (PCGStep3), :901-906 (PCGLinearUpdate); launch shapes util.t:715-765.
//
// Exclusion (fmap.exclude, gauss_newton.t:805): excluded unknowns hold r = p = z = delta = Ap = 0
// (the init kernels write those zeros), so the flat kernels need no mask test: every update
// leaves the zeros in place, exactly like the reference's skipped lanes over zero-initialised
// vectors (thallo.t:1104-1126 initGPU memset).
#include "device_common.hpp"
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {

constexpr int BLOCK = 256;

inline int flat_grid(long n4, int cus)
{
    long want = (n4 + BLOCK - 1) / BLOCK;
    long cap = (long)cus * 4;                       // 4 x 256 threads per CU = 16 waves/CU
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    if (cap >= 8) cap -= cap % 8;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

int g_cus = 0;
int g_nt2 = 3;   // cache policy: bit0 r load/store nt, bit1 pre load nt (measured best in round 1, docs/history), bit2 Ap load nt, bit3 z store nt
int cu_count()
{
    if (!g_cus) {
        int dev = 0; hipGetDevice(&dev);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cus = prop.multiProcessorCount;
        if (g_cus <= 0) g_cus = 256;
    }
    return g_cus;
}

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

// r -= alpha*Ap ; z = pre*r ; betaN partial
template <bool HAS_PRE>
__global__ __launch_bounds__(BLOCK) void k_step2(float4* __restrict__ r, const float4* __restrict__ Ap,
                                                  const float4* __restrict__ pre, float4* __restrict__ z,
                                                  long off0, long len0, long off1, long len1,
                                                  thallo_sum_t aN, thallo_sum_t aD, float* __restrict__ bN_out, int ntm)
{   // two float4 ranges [off0,off0+len0) U [off1,off1+len1): the owned rows of a slab in the flat layout
    __shared__ float red[16];
    const bool nt_r = ntm & 1, nt_pre = ntm & 2, nt_ap = ntm & 4, nt_z = ntm & 8;
    const float alpha = safe_div<false>(sum_partials(aN.partials, aN.count), sum_partials(aD.partials, aD.count));
    float acc = 0.0f;
    const long n4 = len0 + len1;
    for (long j = (long)blockIdx.x * BLOCK + threadIdx.x; j < n4; j += (long)gridDim.x * BLOCK) {
        const long i = j < len0 ? off0 + j : off1 + (j - len0);
        float4 rv = ldf4(r + i, nt_r); const float4 av = ldf4(Ap + i, nt_ap);
        float4 pv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (HAS_PRE) pv = ldf4(pre + i, nt_pre);
        rv.x -= alpha * av.x; rv.y -= alpha * av.y; rv.z -= alpha * av.z; rv.w -= alpha * av.w;
        float4 zv = make_float4(pv.x * rv.x, pv.y * rv.y, pv.z * rv.z, pv.w * rv.w);
        stf4(r + i, rv, nt_r); stf4(z + i, zv, nt_z);
        acc += zv.x * rv.x + zv.y * rv.y + zv.z * rv.z + zv.w * rv.w;
    }
    block_store_partial(acc, bN_out, red);
}

// LM: the zeta test of gauss_newton.t:1666-1686 done by the last workgroup of PCGStep2 itself (tickets != NULL) instead of a one-wave launch
// behind it: state = the 8-word LM state of thallo_hip_lm_zeta (Q0, gate, iterations done), k = this PCG iteration.
struct ZetaArgs { unsigned* tickets; float* state; int k; float q_tolerance; };

// reference-shaped PCGStep2 incl. delta update and optional LM q
template <bool HAS_PRE, bool LM, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_step2_full(float4* __restrict__ delta, const float4* __restrict__ p,
                                                       float4* __restrict__ r, const float4* __restrict__ Ap,
                                                       const float4* __restrict__ pre, float4* __restrict__ z,
                                                       const float4* __restrict__ b, long n4,
                                                       thallo_sum_t aN, thallo_sum_t aD,
                                                       float* __restrict__ bN_out, float* __restrict__ q_out, const unsigned* gate, ZetaArgs zeta)
{
    __shared__ float red[32];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device
    const float alpha = safe_div<LM>(sum_partials(aN.partials, aN.count), sum_partials(aD.partials, aD.count));
    float acc[2] = { 0.0f, 0.0f };
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 dv = delta[i]; const float4 pp = p[i];
        float4 rv = r[i]; const float4 av = Ap[i];
        float4 pv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (HAS_PRE) pv = pre[i];
        dv.x += alpha * pp.x; dv.y += alpha * pp.y; dv.z += alpha * pp.z; dv.w += alpha * pp.w;
        rv.x -= alpha * av.x; rv.y -= alpha * av.y; rv.z -= alpha * av.z; rv.w -= alpha * av.w;
        float4 zv = make_float4(pv.x * rv.x, pv.y * rv.y, pv.z * rv.z, pv.w * rv.w);
        delta[i] = dv; r[i] = rv; z[i] = zv;
        acc[0] += zv.x * rv.x + zv.y * rv.y + zv.z * rv.z + zv.w * rv.w;
        if (HAS_B) {
            const float4 bv = b[i];
            acc[1] += 0.5f * (dv.x * (rv.x + bv.x) + dv.y * (rv.y + bv.y) + dv.z * (rv.z + bv.z) + dv.w * (rv.w + bv.w));
        }
    }
    if (HAS_B) {
        float* __restrict__ const outs[2] = { bN_out, q_out };
        if (!zeta.tickets) block_store_partials<2>(acc, outs, red);
        else {
            // block_store_partials<2> with WRITE-THROUGH stores of the two partials (agent scope: the last workgroup may sit on another XCD, behind
            // another L2; a full __threadfence() here would write every workgroup's dirty L2 lines back -- measured: the kernel takes twice as long),
            // then an arrival ticket (two levels, as in block_finish_sums); the last workgroup adds the q partials in sum_partials' order and
            // applies the test exactly as k_lm_zeta does
            const int wlane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE, nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
#pragma unroll
            for (int q = 0; q < 2; ++q) { const float w = wave_sum_all(acc[q]); if (wlane == 0) red[q * 16 + wave] = w; }
            __syncthreads();
            if (threadIdx.x < 2) {
                float w = 0.0f;
                for (int k = 0; k < nw; ++k) w += red[threadIdx.x * 16 + k];
                __hip_atomic_store(outs[threadIdx.x] + blockIdx.x, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (threadIdx.x < THALLO_WAVE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                const unsigned grp = blockIdx.x % 32, members = (gridDim.x - grp + 31) / 32, groups = gridDim.x < 32 ? gridDim.x : 32;
                unsigned* sub = zeta.tickets + 16 + 16 * grp;
                bool last = false;
                if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                    __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = __hip_atomic_fetch_add(zeta.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
                }
                red[31] = last ? 1.0f : 0.0f;
            }
            __syncthreads();
            if (red[31] != 0.0f && threadIdx.x < THALLO_WAVE) {
                const int lane = threadIdx.x, nb = gridDim.x;
                float t[THALLO_MAX_PARTIALS / THALLO_WAVE];
#pragma unroll
                for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) {
                    const int i = lane + k * THALLO_WAVE;
                    t[k] = i < nb ? __hip_atomic_load(q_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
                }
                float Q1 = 0.0f;
#pragma unroll
                for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) Q1 += t[k];
                Q1 = nb == 1 ? t[0] : wave_sum_all(Q1);
                Q1 = __shfl(Q1, 0, THALLO_WAVE);                                          // (nb == 1: lane 0 holds it)
                const float Q0 = zeta.state[0];
                const float zt = (float)(zeta.k + 1) * (Q1 - Q0) / Q1;
                const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < zeta.q_tolerance;
                if (lane == 0) {
                    __hip_atomic_store(zeta.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (stop) { reinterpret_cast<unsigned*>(zeta.state)[1] = 1u; reinterpret_cast<int*>(zeta.state)[2] = zeta.k + 1; }
                    else zeta.state[0] = Q1;
                }
            }
        }
    } else {
        block_store_partial(acc[0], bN_out, red);
    }
}

template <bool LM>
__global__ __launch_bounds__(BLOCK) void k_step3(float4* __restrict__ p, const float4* __restrict__ z, long n4,
                                                  thallo_sum_t bN, thallo_sum_t aN)
{
    const float beta = safe_div<LM>(sum_partials(bN.partials, bN.count), sum_partials(aN.partials, aN.count));
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 pv = p[i]; const float4 zv = z[i];
        pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y; pv.z = zv.z + beta * pv.z; pv.w = zv.w + beta * pv.w;
        p[i] = pv;
    }
}

// unfused PCGStep3 + delta update for energies whose applyJTJ gathers p through indices (graph domains):
//   delta += alpha*p_in ; p_out = z + beta*p_in    (first: p_out = z)
__global__ __launch_bounds__(BLOCK) void k_pupdate(const float4* __restrict__ z, const float4* __restrict__ p_in, float4* __restrict__ p_out,
                                                    float4* __restrict__ delta, long off0, long len0, long off1, long len1, int first,
                                                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, const unsigned* __restrict__ gate)
{   // delta == NULL: p update only (LM: PCGStep3 with the unguarded divide, delta lives in PCGStep2)
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device
    float alpha = 0.0f, beta = 0.0f;
    const bool lm = delta == nullptr;
    if (!first) {
        const float an = sum_partials(aNp.partials, aNp.count);
        if (lm) beta = safe_div<true>(sum_partials(bNp.partials, bNp.count), an);
        else {
            alpha = safe_div<false>(an, sum_partials(aDp.partials, aDp.count));
            beta  = safe_div<false>(sum_partials(bNp.partials, bNp.count), an);
        }
    }
    const long n4 = len0 + len1;
    for (long j = (long)blockIdx.x * BLOCK + threadIdx.x; j < n4; j += (long)gridDim.x * BLOCK) {
        const long i = j < len0 ? off0 + j : off1 + (j - len0);
        const float4 pv = p_in[i], zv = z[i];
        if (!first && !lm) {
            float4 dv = delta[i];
            dv.x += alpha * pv.x; dv.y += alpha * pv.y; dv.z += alpha * pv.z; dv.w += alpha * pv.w;
            delta[i] = dv;
        }
        p_out[i] = make_float4(zv.x + beta * pv.x, zv.y + beta * pv.y, zv.z + beta * pv.z, zv.w + beta * pv.w);
    }
}

// ------------------------------------------------------------------ single-reduction form, energy-independent half
// One flat pass = PCGStep2 of iteration k-1 + PCGStep3 + the delta update of iteration k (gauss_newton.t:801-843,889-899):
//   r -= alpha_{k-1} Ap (fma) ;  z = M^-1 r (not stored) ;  p_out = z + beta_{k-1} p_in ;  delta += alpha_{k-1} p_in      [first: p_out = M^-1 r only]
// possible because beta_{k-1} is already known: the applyJTJ kernel of iteration k-1 also produced N, S1, S2 (Sums3) and
// k_scalars_finish expanded betaN_{k-1} = N - 2 alpha S1 + alpha^2 S2 from them.  No reduction in here.
// LM (thallo_hip_pcg_update_lm): the unguarded divides of the LM branch (gauss_newton.t:226-234) and the loop's gate word; first == 2: the iteration behind a residual
// reset (gauss_newton.t:1653-1657: delta and r are already those of this iteration) -- p_out = M^-1 r + beta p_in only.
template <bool HAS_PRE, bool LM>
__global__ __launch_bounds__(BLOCK) void k_pcg_update(float4* __restrict__ r, const float4* __restrict__ Ap, const float4* __restrict__ pre,
                                                       const float4* __restrict__ p_in, float4* __restrict__ p_out, float4* __restrict__ delta, long n4, int first,
                                                       thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, const unsigned* __restrict__ gate, float* __restrict__ bn_word)
{
    if (LM && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;                    // the PCG loop already ended on the device
    float alpha = 0.0f, beta = 0.0f;
    if (first != 1) {
        const float an = sum_partials(aNp.partials, aNp.count);
        alpha = safe_div<LM>(an, sum_partials(aDp.partials, aDp.count));
        const float bn = sum_partials(bNp.partials, bNp.count);
        beta  = safe_div<LM>(bn, an);
        if (LM && bn_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) bn_word[0] = bn;     // (behind a residual reset: betaN_k exists as partials only)
    }
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 rv = r[i];
        const float4 pv = p_in[i];
        if (first == 0) {
            const float4 av = Ap[i];
            rv.x = __builtin_fmaf(-alpha, av.x, rv.x); rv.y = __builtin_fmaf(-alpha, av.y, rv.y); rv.z = __builtin_fmaf(-alpha, av.z, rv.z); rv.w = __builtin_fmaf(-alpha, av.w, rv.w);
            r[i] = rv;
            float4 dv = delta[i];
            dv.x = __builtin_fmaf(alpha, pv.x, dv.x); dv.y = __builtin_fmaf(alpha, pv.y, dv.y); dv.z = __builtin_fmaf(alpha, pv.z, dv.z); dv.w = __builtin_fmaf(alpha, pv.w, dv.w);
            delta[i] = dv;
        }
        float4 zv = rv;
        if (HAS_PRE) { const float4 m = pre[i]; zv.x *= m.x; zv.y *= m.y; zv.z *= m.z; zv.w *= m.w; }
        p_out[i] = make_float4(zv.x + beta * pv.x, zv.y + beta * pv.y, zv.z + beta * pv.z, zv.w + beta * pv.w);
    }
}

// ... with the FINISH of iteration k-1 deferred into it (thallo_hip_pcg_update_fin): the applyJTJ launch of iteration k-1 left per-workgroup partials only -- no write-through
// slots, no arrival tickets, no last workgroup reading everything back at the end of a 12-us launch (a ~3-us tail, profiles/r04/ba_lm_loops.json) -- and EVERY workgroup of this
// launch adds them up for itself (all four waves, last_workgroup_totals' order: the same bits as the in-kernel finish and as k_scalars_finish; <= 1024 slots x 28 bytes from L2,
// behind the workgroup's own first vector loads), forms alpha_{k-1}, betaN_{k-1} and beta_{k-1} from them and goes on as k_pcg_update; workgroup 0 leaves the two words.
template <bool HAS_PRE>
__global__ __launch_bounds__(BLOCK) void k_pcg_update_fin(float4* __restrict__ r, const float4* __restrict__ Ap, const float4* __restrict__ pre,
                                                           const float4* __restrict__ p_in, float4* __restrict__ p_out, float4* __restrict__ delta, long n4,
                                                           thallo_sum_t aNp, const float* __restrict__ aD_part, const double* __restrict__ s3_part, int nb,
                                                           float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    __shared__ float red[16];
    __shared__ double redd[8];
    const long i0 = (long)blockIdx.x * BLOCK + threadIdx.x;
    const bool have = i0 < n4;
    float4 rv0 = make_float4(0.f, 0.f, 0.f, 0.f), av0 = rv0, pv0 = rv0, dv0 = rv0, mv0 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (have) { rv0 = r[i0]; av0 = Ap[i0]; pv0 = p_in[i0]; dv0 = delta[i0]; if (HAS_PRE) mv0 = pre[i0]; }      // (in flight while the partials are added up)
    float ad, an; double t3[3];
    last_workgroup_totals<3, true>(aD_part, s3_part, nullptr, nb, aNp, red, redd, ad, an, t3);
    const float alpha = safe_div<false>(an, ad);
    double bn = t3[0] - 2.0 * (double)alpha * t3[1] + (double)alpha * (double)alpha * t3[2];
    if (!(bn > 0.0)) bn = 0.0;
    const float bnf = (float)bn, beta = safe_div<false>(bnf, an);
    if (blockIdx.x == 0 && threadIdx.x == 0) { aD_word[0] = ad; bN_word[0] = bnf; }
    for (long i = i0; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 rv, av, pv, dv, m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (i == i0) { rv = rv0; av = av0; pv = pv0; dv = dv0; m = mv0; }
        else { rv = r[i]; av = Ap[i]; pv = p_in[i]; dv = delta[i]; if (HAS_PRE) m = pre[i]; }
        rv.x = __builtin_fmaf(-alpha, av.x, rv.x); rv.y = __builtin_fmaf(-alpha, av.y, rv.y); rv.z = __builtin_fmaf(-alpha, av.z, rv.z); rv.w = __builtin_fmaf(-alpha, av.w, rv.w);
        r[i] = rv;
        dv.x = __builtin_fmaf(alpha, pv.x, dv.x); dv.y = __builtin_fmaf(alpha, pv.y, dv.y); dv.z = __builtin_fmaf(alpha, pv.z, dv.z); dv.w = __builtin_fmaf(alpha, pv.w, dv.w);
        delta[i] = dv;
        float4 zv = rv;
        if (HAS_PRE) { zv.x *= m.x; zv.y *= m.y; zv.z *= m.z; zv.w *= m.w; }
        p_out[i] = make_float4(zv.x + beta * pv.x, zv.y + beta * pv.y, zv.z + beta * pv.z, zv.w + beta * pv.w);
    }
}

// The LM form of k_pcg_update_fin (thallo_hip_pcg_update_lm_fin): the finish of LM iteration k-1 -- alphaD, betaN, q_k = 0.5 [U + alpha (T1 - T2) - alpha^2 alphaD] and the zeta
// test (device_common.hpp block_finish_sums_lm's arithmetic and order) -- by EVERY workgroup of iteration k's flat update, from the partials the applyJTJ launches of iteration
// k-1 left.  All workgroups reach the same verdict; workgroup 0 records it (the two words; Q1 in state[q_out], or the gate and the iteration count) -- Q0 is read from another
// word than Q1 is written to.  A stop ends this launch too: delta, r, p stay as of the break, like everywhere in the LM loop.
__global__ __launch_bounds__(BLOCK) void k_pcg_update_lm_fin(float4* __restrict__ r, const float4* __restrict__ Ap, const float4* __restrict__ pre,
                                                              const float4* __restrict__ p_in, float4* __restrict__ p_out, float4* __restrict__ delta, long n4,
                                                              thallo_sum_t aNp, const float* __restrict__ aD_part, const double* __restrict__ s3_part, const double* __restrict__ q3_part, int nb,
                                                              float* __restrict__ aD_word, float* __restrict__ bN_word, float* __restrict__ state, int kprev, float q_tol, int q_in, int q_out)
{
    __shared__ float red[16];
    __shared__ double redd[8];
    if (__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(state)[1]) != 0) return;      // the PCG loop already ended on the device
    const long i0 = (long)blockIdx.x * BLOCK + threadIdx.x;
    const bool have = i0 < n4;
    float4 rv0 = make_float4(0.f, 0.f, 0.f, 0.f), av0 = rv0, pv0 = rv0, dv0 = rv0, mv0 = rv0;
    if (have) { rv0 = r[i0]; av0 = Ap[i0]; pv0 = p_in[i0]; dv0 = delta[i0]; mv0 = pre[i0]; }
    float ad, an; double t6[6];
    last_workgroup_totals<6, true>(aD_part, s3_part, q3_part, nb, aNp, red, redd, ad, an, t6);
    const float alpha = safe_div<true>(an, ad);                  // LM divides blindly (gauss_newton.t:226-234)
    double bn = t6[0] - 2.0 * (double)alpha * t6[1] + (double)alpha * (double)alpha * t6[2];
    if (!(bn > 0.0)) bn = 0.0;
    const float bnf = (float)bn;
    const float Q1 = (float)(0.5 * (t6[3] + (double)alpha * (t6[4] - t6[5]) - (double)alpha * (double)alpha * (double)ad));
    const float Q0 = state[q_in];
    const float zt = (float)(kprev + 1) * (Q1 - Q0) / Q1;
    const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < q_tol;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        aD_word[0] = ad; bN_word[0] = bnf;
        if (stop) { reinterpret_cast<unsigned*>(state)[1] = 1u; reinterpret_cast<int*>(state)[2] = kprev + 1; }
        else state[q_out] = Q1;
    }
    if (stop) return;
    const float beta = safe_div<true>(bnf, an);
    for (long i = i0; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 rv, av, pv, dv, m;
        if (i == i0) { rv = rv0; av = av0; pv = pv0; dv = dv0; m = mv0; }
        else { rv = r[i]; av = Ap[i]; pv = p_in[i]; dv = delta[i]; m = pre[i]; }
        rv.x = __builtin_fmaf(-alpha, av.x, rv.x); rv.y = __builtin_fmaf(-alpha, av.y, rv.y); rv.z = __builtin_fmaf(-alpha, av.z, rv.z); rv.w = __builtin_fmaf(-alpha, av.w, rv.w);
        r[i] = rv;
        dv.x = __builtin_fmaf(alpha, pv.x, dv.x); dv.y = __builtin_fmaf(alpha, pv.y, dv.y); dv.z = __builtin_fmaf(alpha, pv.z, dv.z); dv.w = __builtin_fmaf(alpha, pv.w, dv.w);
        delta[i] = dv;
        const float4 zv = make_float4(rv.x * m.x, rv.y * m.y, rv.z * m.z, rv.w * m.w);
        p_out[i] = make_float4(zv.x + beta * pv.x, zv.y + beta * pv.y, zv.z + beta * pv.z, zv.w + beta * pv.w);
    }
}

// the update of delta the one-launch LM loop owes at its end (thallo_hip.h thallo_hip_lm_owed_delta)
__global__ __launch_bounds__(BLOCK) void k_lm_owed_delta(float4* __restrict__ delta, const float4* __restrict__ p_even, const float4* __restrict__ p_odd, long n4,
                                                          const float* __restrict__ aN_words, const float* __restrict__ aD_words, int stride, const float* __restrict__ state, int L)
{
    const unsigned gate = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(state)[1]);
    const int done = gate ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(state)[2]) : L;
    const int kl = done - 1;
    if (kl < 0) return;
    const float alpha = safe_div<true>(aN_words[(long)kl * stride], aD_words[(long)kl * stride]);
    const float4* __restrict__ p = (kl & 1) ? p_odd : p_even;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 d = delta[i]; const float4 pv = p[i];
        d.x = __builtin_fmaf(alpha, pv.x, d.x); d.y = __builtin_fmaf(alpha, pv.y, d.y); d.z = __builtin_fmaf(alpha, pv.z, d.z); d.w = __builtin_fmaf(alpha, pv.w, d.w);
        delta[i] = d;
    }
}

// one wave: alphaD_k (float partials, the usual order), N, S1, S2 (double partials, lane-strided then butterfly), then
// betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 with alpha_k = alphaN_k / alphaD_k exactly as every consumer forms it
__global__ __launch_bounds__(64) void k_scalars_finish(const float* __restrict__ aD_part, const double* __restrict__ s3, int nb, thallo_sum_t aN,
                                                       float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    const int lane = threadIdx.x;
    const float ad = sum_partials(aD_part, nb);
    double n = 0.0, a = 0.0, b = 0.0;
    for (int i = lane; i < nb; i += THALLO_WAVE) { n += s3[3 * i]; a += s3[3 * i + 1]; b += s3[3 * i + 2]; }
    n = wave_sum_all_f64(n); a = wave_sum_all_f64(a); b = wave_sum_all_f64(b);
    const float an = sum_partials(aN.partials, aN.count);
    const float alpha = safe_div<false>(an, ad);
    double bn = n - 2.0 * (double)alpha * a + (double)alpha * (double)alpha * b;
    if (!(bn > 0.0)) bn = 0.0;                       // r . M^-1 r is a sum of squares; guards the last bits at convergence
    if (lane == 0) { aD_word[0] = ad; bN_word[0] = (float)bn; }
}

// ------------------------------------------------------------------ Levenberg-Marquardt set (gauss_newton.t:929-969,774-787,845-886)
// PCGSaveSSq + PCGComputeCtC + PCGFinalizeDiagonal in one pass over the raw diagonal d = diag(J^T J):
//   SSq (first GN iteration only) = guardedInvert(d) (or 1 without preconditioner)   -- Jacobi scale^2, ONCE_PER_SOLVE
//   unclamped = d / radius ; CtC = clamp(unclamped, min_lm/(SSq*radius), max_lm/(SSq*radius))
//   M^-1 = 1 / (CtC + radius*unclamped) ; b = r ; z = M^-1 r ; alphaN partials = sum r.z
__global__ __launch_bounds__(BLOCK) void k_lm_finalize(const float4* __restrict__ diag, float4* __restrict__ SSq, float4* __restrict__ CtC,
                                                        float4* __restrict__ pre, const float4* __restrict__ r, float4* __restrict__ b,
                                                        float4* __restrict__ z, long n4, float radius, float min_lm, float max_lm,
                                                        int save_ssq, int use_precond, float* __restrict__ aN_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    const float inv_radius = 1.0f / radius;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        const float4 d4 = diag[i], r4 = r[i];
        float4 s4;
        if (save_ssq) {
            if (use_precond) s4 = make_float4(guarded_invert(d4.x), guarded_invert(d4.y), guarded_invert(d4.z), guarded_invert(d4.w));
            else s4 = make_float4(1.f, 1.f, 1.f, 1.f);
            SSq[i] = s4;
        } else s4 = SSq[i];
        const float dd[4] = { d4.x, d4.y, d4.z, d4.w }, ss[4] = { s4.x, s4.y, s4.z, s4.w }, rr[4] = { r4.x, r4.y, r4.z, r4.w };
        float cc[4], mm[4], zz[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float unclamped = dd[k] * inv_radius;
            const float cm = (1.0f / ss[k]) / radius;
            const float c = fminf(fmaxf(unclamped, min_lm * cm), max_lm * cm);
            cc[k] = c;
            mm[k] = 1.0f / (c + radius * unclamped);
            zz[k] = mm[k] * rr[k];
            acc += rr[k] * zz[k];
        }
        CtC[i] = make_float4(cc[0], cc[1], cc[2], cc[3]); pre[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        b[i] = r4; z[i] = make_float4(zz[0], zz[1], zz[2], zz[3]);
    }
    block_store_partial(acc, aN_out, red);
}

// PCGStep1_Finish (LM): Ap += CtC*p ; alphaD partials = sum p.Ap
__global__ __launch_bounds__(BLOCK) void k_lm_step1_finish(float4* __restrict__ Ap, const float4* __restrict__ CtC, const float4* __restrict__ p,
                                                            long n4, float* __restrict__ aD_out, const unsigned* __restrict__ gate)
{
    __shared__ float red[16];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 a = Ap[i]; const float4 c = CtC[i], pv = p[i];
        a.x += c.x * pv.x; a.y += c.y * pv.y; a.z += c.z * pv.z; a.w += c.w * pv.w;
        Ap[i] = a;
        acc += pv.x * a.x + pv.y * a.y + pv.z * a.z + pv.w * a.w;
    }
    block_store_partial(acc, aD_out, red);
}

// PCGStep2_1stHalf: delta += alpha*p
__global__ __launch_bounds__(BLOCK) void k_lm_step2_first(float4* __restrict__ delta, const float4* __restrict__ p, long n4,
                                                           thallo_sum_t aN, thallo_sum_t aD, const unsigned* __restrict__ gate)
{
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device
    const float alpha = safe_div<true>(sum_partials(aN.partials, aN.count), sum_partials(aD.partials, aD.count));
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        float4 d = delta[i]; const float4 pv = p[i];
        d.x += alpha * pv.x; d.y += alpha * pv.y; d.z += alpha * pv.z; d.w += alpha * pv.w;
        delta[i] = d;
    }
}

// PCGStep2_2ndHalf: r = b - Adelta ; z = M^-1 r ; betaN ; q = 0.5 delta.(r+b)
__global__ __launch_bounds__(BLOCK) void k_lm_step2_second(float4* __restrict__ r, const float4* __restrict__ b, const float4* __restrict__ Ad,
                                                            const float4* __restrict__ pre, float4* __restrict__ z, const float4* __restrict__ delta,
                                                            long n4, float* __restrict__ bN_out, float* __restrict__ q_out, const unsigned* __restrict__ gate)
{
    __shared__ float red[32];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device
    float acc[2] = { 0.0f, 0.0f };
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        const float4 bv = b[i], av = Ad[i], mv = pre[i], dv = delta[i];
        const float4 rv = make_float4(bv.x - av.x, bv.y - av.y, bv.z - av.z, bv.w - av.w);
        const float4 zv = make_float4(mv.x * rv.x, mv.y * rv.y, mv.z * rv.z, mv.w * rv.w);
        r[i] = rv; z[i] = zv;
        acc[0] += zv.x * rv.x + zv.y * rv.y + zv.z * rv.z + zv.w * rv.w;
        acc[1] += 0.5f * (dv.x * (rv.x + bv.x) + dv.y * (rv.y + bv.y) + dv.z * (rv.z + bv.z) + dv.w * (rv.w + bv.w));
    }
    float* __restrict__ const outs[2] = { bN_out, q_out };
    block_store_partials<2>(acc, outs, red);
}

// The zeta test of the LM branch on the device (gauss_newton.t:1666-1686: Q1 = q of this iteration; break if it or zeta = (k+1)(Q1-Q0)/Q1
// is not finite or zeta < q_tolerance; else Q0 = Q1).  state: [0] Q0, [1] frozen (uint, the gate word of the loop's kernels), [2] number of
// PCG iterations done when the loop froze (int).  One wave; runs behind PCGStep2 of iteration k.
__global__ __launch_bounds__(64) void k_lm_zeta(thallo_sum_t q, int k, float q_tolerance, float* __restrict__ state)
{
    unsigned* su = reinterpret_cast<unsigned*>(state);
    if (__builtin_amdgcn_readfirstlane((int)su[1]) != 0) return;
    const float Q1 = sum_partials(q.partials, q.count), Q0 = state[0];
    const float zeta = (float)(k + 1) * (Q1 - Q0) / Q1;
    const bool stop = !isfinite(Q1) || !isfinite(zeta) || zeta < q_tolerance;
    if (threadIdx.x == 0) {
        if (stop) { su[1] = 1u; reinterpret_cast<int*>(state)[2] = k + 1; }
        else state[0] = Q1;
    }
}

// PCGInit1_Finish (gauss_newton.t:712-731) for callers that assembled r and the RAW diagonal elsewhere (e.g. after a
// cross-rank reduction): pre = guardedInvert(diag) (or 1), z = pre*r, alphaN partials
__global__ __launch_bounds__(BLOCK) void k_init_finish(const float4* __restrict__ r, const float4* __restrict__ diag, float4* __restrict__ pre,
                                                        float4* __restrict__ z, long n4, int use_precond, float* __restrict__ aN_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        const float4 rv = r[i];
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (use_precond) { const float4 d = diag[i]; m = make_float4(guarded_invert(d.x), guarded_invert(d.y), guarded_invert(d.z), guarded_invert(d.w)); }
        const float4 zv = make_float4(m.x * rv.x, m.y * rv.y, m.z * rv.z, m.w * rv.w);
        pre[i] = m; z[i] = zv;
        acc += rv.x * zv.x + rv.y * zv.y + rv.z * zv.z + rv.w * zv.w;
    }
    block_store_partial(acc, aN_out, red);
}

// sum a.b partials
__global__ __launch_bounds__(BLOCK) void k_dot(const float4* __restrict__ a, const float4* __restrict__ b, long n4, float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        const float4 x = a[i], y = b[i];
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    block_store_partial(acc, out, red);
}

// X += delta (+ alpha*p).  X is a caller buffer of exactly `len` floats (not padded): scalar tail.
template <bool HAS_P>
__global__ __launch_bounds__(BLOCK) void k_linear_update(float* __restrict__ X, const float* __restrict__ delta,
                                                          const float* __restrict__ p, long len,
                                                          thallo_sum_t aN, thallo_sum_t aD)
{
    float alpha = 0.0f;
    if (HAS_P) alpha = safe_div<false>(sum_partials(aN.partials, aN.count), sum_partials(aD.partials, aD.count));
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < len; i += (long)gridDim.x * BLOCK) {
        float d = delta[i];
        if (HAS_P) d = __builtin_fmaf(alpha, p[i], d);      // explicit fma: the same rounding wherever a delta update is applied
        X[i] = X[i] + d;
    }
}

__global__ __launch_bounds__(BLOCK) void k_linear_update2(float* __restrict__ X, const float* __restrict__ delta,
                                                           const float* __restrict__ p0, thallo_sum_t aN0, thallo_sum_t aD0,
                                                           const float* __restrict__ p1, thallo_sum_t aN1, thallo_sum_t aD1, long len)
{
    const float a0 = safe_div<false>(sum_partials(aN0.partials, aN0.count), sum_partials(aD0.partials, aD0.count));
    const float a1 = safe_div<false>(sum_partials(aN1.partials, aN1.count), sum_partials(aD1.partials, aD1.count));
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < len; i += (long)gridDim.x * BLOCK) {
        float d = delta[i];
        d = __builtin_fmaf(a0, p0[i], d);
        d = __builtin_fmaf(a1, p1[i], d);
        X[i] = X[i] + d;
    }
}

// delta += sum_j alpha_j p_j over up to THALLO_HIP_MAX_UPDATE_TERMS pending terms, oldest first, each as ONE explicit fma on the running value -- the roundings
// of `delta += alpha p` applied once per iteration in that order (round 5: the one-kernel schedule keeps a RING of p planes and touches delta once per ring
// instead of once per iteration or two).  X != NULL: the end of a GN step, X += (the updated delta) and delta itself is not written back.
// V4: every pointer 16-byte aligned and len a multiple of 4 (solver vectors always; caller buffers usually).
template <bool V4>
__global__ __launch_bounds__(BLOCK) void k_linear_update_n(float* __restrict__ X, float* __restrict__ delta, thallo_update_terms_t T, long len)
{
    __shared__ float al[THALLO_HIP_MAX_UPDATE_TERMS];
    for (int j = 0; j < T.count; ++j) {     // (wave-cooperative sums: every wave computes every alpha, in the order every other consumer uses)
        const float a = safe_div<false>(sum_partials(T.alphaN[j].partials, T.alphaN[j].count), sum_partials(T.alphaD[j].partials, T.alphaD[j].count));
        if (threadIdx.x == 0) al[j] = a;
    }
    __syncthreads();
    if (V4) {
        const long n4 = len >> 2;
        for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
            float4 d = reinterpret_cast<const float4*>(delta)[i];
            int j = 0;
            for (; j + 8 <= T.count; j += 8) {      // eight planes' loads in flight per lane (the terms are still applied one after the other): a background launch on 256
                float4 q[8];                        // workgroups is latency-bound -- 9 x 16 B x 65,536 threads in flight against 5 x with four
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = ldf4(reinterpret_cast<const float4*>(T.p[j + u]) + i, true);      // read once, never again
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a = al[j + u];
                    d.x = __builtin_fmaf(a, q[u].x, d.x); d.y = __builtin_fmaf(a, q[u].y, d.y); d.z = __builtin_fmaf(a, q[u].z, d.z); d.w = __builtin_fmaf(a, q[u].w, d.w);
                }
            }
            for (; j + 4 <= T.count; j += 4) {
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = ldf4(reinterpret_cast<const float4*>(T.p[j + u]) + i, true);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float a = al[j + u];
                    d.x = __builtin_fmaf(a, q[u].x, d.x); d.y = __builtin_fmaf(a, q[u].y, d.y); d.z = __builtin_fmaf(a, q[u].z, d.z); d.w = __builtin_fmaf(a, q[u].w, d.w);
                }
            }
            for (; j < T.count; ++j) {
                const float4 q = ldf4(reinterpret_cast<const float4*>(T.p[j]) + i, true);
                const float a = al[j];
                d.x = __builtin_fmaf(a, q.x, d.x); d.y = __builtin_fmaf(a, q.y, d.y); d.z = __builtin_fmaf(a, q.z, d.z); d.w = __builtin_fmaf(a, q.w, d.w);
            }
            if (X) { float4 x = reinterpret_cast<const float4*>(X)[i]; x.x = x.x + d.x; x.y = x.y + d.y; x.z = x.z + d.z; x.w = x.w + d.w; reinterpret_cast<float4*>(X)[i] = x; }
            else reinterpret_cast<float4*>(delta)[i] = d;
        }
    } else {
        for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < len; i += (long)gridDim.x * BLOCK) {
            float d = delta[i];
            for (int j = 0; j < T.count; ++j) d = __builtin_fmaf(al[j], T.p[j][i], d);
            if (X) X[i] = X[i] + d; else delta[i] = d;
        }
    }
}

// out[0] = sum(partials) (if any), out[1..] = the listed segments of `vec`, concatenated (slab boundary rows)
__global__ __launch_bounds__(BLOCK) void k_slab_pack(const float* __restrict__ vec, thallo_segs_t segs, thallo_sum_t s, float* __restrict__ out)
{
    if (blockIdx.x == 0 && s.count > 0) {
        const float v = sum_partials(s.partials, s.count);
        if (threadIdx.x == 0) out[0] = v;
    }
    long base = 1;
    for (int k = 0; k < segs.n; ++k) {
        for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < segs.len[k]; i += (long)gridDim.x * BLOCK) out[base + i] = vec[segs.off[k] + i];
        base += segs.len[k];
    }
}

// sum_out[0] = sum_r gathered[r*stride] in rank order (bit-identical on every rank); ghost segments of `vec`
// <- the neighbours' packed boundary rows
__global__ __launch_bounds__(BLOCK) void k_slab_unpack(float* __restrict__ vec, thallo_segs_t top, const float* __restrict__ src_top,
                                                        thallo_segs_t bot, const float* __restrict__ src_bot,
                                                        const float* __restrict__ gathered, long stride, int world, float* __restrict__ sum_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && sum_out) {
        float v = 0.0f;
        for (int r = 0; r < world; ++r) v += gathered[(long)r * stride];
        sum_out[0] = v;
    }
    if (src_top) {
        long base = 0;
        for (int k = 0; k < top.n; ++k) {
            for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < top.len[k]; i += (long)gridDim.x * BLOCK) vec[top.off[k] + i] = src_top[base + i];
            base += top.len[k];
        }
    }
    if (src_bot) {
        long base = 0;
        for (int k = 0; k < bot.n; ++k) {
            for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < bot.len[k]; i += (long)gridDim.x * BLOCK) vec[bot.off[k] + i] = src_bot[base + i];
            base += bot.len[k];
        }
    }
}

// Shard form (bundle adjustment across ranks, solver_dist.cpp): the sums of a PCG iteration over a block of unknowns that every rank holds in
// full (the points), taken AFTER the cross-rank all-reduce completed A p there: alphaD partial (float) and {N, S1, S2} (double) per workgroup
template <bool HAS_PRE>
__global__ __launch_bounds__(BLOCK) void k_block_sums(const float4* __restrict__ p, const float4* __restrict__ Ap, const float4* __restrict__ r, const float4* __restrict__ pre,
                                                       long n4, float* __restrict__ aD_out, double* __restrict__ s3_out)
{
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    float acc = 0.0f; Sums3 sm;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
        const float4 pv = p[i], av = Ap[i], rv = r[i];
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (HAS_PRE) m = pre[i];
        acc += pv.x * av.x + pv.y * av.y + pv.z * av.z + pv.w * av.w;
        sm.add(m.x, rv.x, av.x); sm.add(m.y, rv.y, av.y); sm.add(m.z, rv.z, av.z); sm.add(m.w, rv.w, av.w);
    }
    block_store_partial(acc, aD_out, red);
    block_store_sums3(sm, s3_out, redd);
}
// ... and the iteration's two scalars from the gathered per-rank sums of the rank-private blocks (cameras: [alphaD | N, S1, S2 as (hi, lo)] per
// rank, added in rank order) plus the shared block's partials above (identical on every rank): alphaD_k, betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2.
// count_only_first != 0: only word 0 of every rank's message and the float partials are added (alphaN_0 at PCGInit): out_a[0] = that sum.
__global__ __launch_bounds__(64) void k_shard_scalars(const float* __restrict__ gathered, long stride, int world, const float* __restrict__ aD_part, const double* __restrict__ s3_part,
                                                      int nb, thallo_sum_t aN, int count_only_first, float* __restrict__ out_a, float* __restrict__ out_b)
{
    const int lane = threadIdx.x;
    float ad = 0.0f; double q[3] = { 0.0, 0.0, 0.0 };
    for (int r = 0; r < world; ++r) {
        const float* m = gathered + (long)r * stride;
        ad += m[0];
        if (!count_only_first)
            for (int j = 0; j < 3; ++j) {
                const unsigned long long b = ((unsigned long long)__float_as_uint(m[1 + 2 * j]) << 32) | (unsigned long long)__float_as_uint(m[2 + 2 * j]);
                q[j] += __longlong_as_double((long long)b);
            }
    }
    ad += sum_partials(aD_part, nb);
    if (count_only_first) { if (lane == 0) out_a[0] = ad; return; }
    double n = 0.0, a = 0.0, b = 0.0;
    for (int i = lane; i < nb; i += THALLO_WAVE) { n += s3_part[3 * i]; a += s3_part[3 * i + 1]; b += s3_part[3 * i + 2]; }
    n = q[0] + wave_sum_all_f64(n); a = q[1] + wave_sum_all_f64(a); b = q[2] + wave_sum_all_f64(b);
    const float an = sum_partials(aN.partials, aN.count);
    const float alpha = safe_div<false>(an, ad);
    double bn = n - 2.0 * (double)alpha * a + (double)alpha * (double)alpha * b;
    if (!(bn > 0.0)) bn = 0.0;
    if (lane == 0) { out_a[0] = ad; out_b[0] = (float)bn; }
}

// Range partition (graph domains, solver_dist.cpp): every rank's message carries its owned slice of each plane of a flat vector; rank r's slice j
// goes to vec[first.off[j] + r * first.len[j] ...] (equal slices; `first` = the pieces of rank 0), read from gathered[r * stride + skip + ...]
__global__ __launch_bounds__(BLOCK) void k_range_unpack(float* __restrict__ vec, thallo_segs_t first, const float* __restrict__ gathered, long stride, long skip, int world)
{
    for (int r = 0; r < world; ++r) {
        long base = (long)r * stride + skip;
        for (int j = 0; j < first.n; ++j) {
            float* __restrict__ dst = vec + first.off[j] + (long)r * first.len[j];
            for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < first.len[j]; i += (long)gridDim.x * BLOCK) dst[i] = gathered[base + i];
            base += first.len[j];
        }
    }
}

// One-kernel-per-iteration schedule over row slabs, collective transport: the message of a rank =
//   [ alphaD_local | N, S1, S2 as (hi, lo) words | first owned row of Ap_out | last owned row of Ap_out ]
// (7 scalars words, then the listed segments).  One all-gather of these per PCG iteration replaces the all-reduce + all-gather of
// the two-kernel form.
__global__ __launch_bounds__(BLOCK) void k_slab_pack_iter(const float* __restrict__ vec, thallo_segs_t segs, const float* __restrict__ aD_part,
                                                           const double* __restrict__ s3, int nb, float* __restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x < THALLO_WAVE) {
        const int lane = threadIdx.x;
        const float ad = sum_partials(aD_part, nb);
        double q[3] = { 0.0, 0.0, 0.0 };
        for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s3[3 * i]; q[1] += s3[3 * i + 1]; q[2] += s3[3 * i + 2]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) q[j] = wave_sum_all_f64(q[j]);
        if (lane == 0) {
            out[0] = ad;
            for (int j = 0; j < 3; ++j) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(q[j]);
                out[1 + 2 * j] = __uint_as_float((unsigned)(b >> 32)); out[2 + 2 * j] = __uint_as_float((unsigned)b);
            }
        }
    }
    long base = 7;
    for (int k = 0; k < segs.n; ++k) {
        for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < segs.len[k]; i += (long)gridDim.x * BLOCK) out[base + i] = vec[segs.off[k] + i];
        base += segs.len[k];
    }
}

// rank-ordered sums of the gathered scalars -> alphaD_k, betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 (as k_scalars_finish); ghost rows of
// Ap_out <- the neighbours' boundary rows
__global__ __launch_bounds__(BLOCK) void k_slab_unpack_iter(float* __restrict__ vec, thallo_segs_t top, const float* __restrict__ src_top,
                                                             thallo_segs_t bot, const float* __restrict__ src_bot,
                                                             const float* __restrict__ gathered, long stride, int world, thallo_sum_t aN,
                                                             float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gad = 0.0f; double gq[3] = { 0.0, 0.0, 0.0 };
        for (int r = 0; r < world; ++r) {
            const float* m = gathered + (long)r * stride;
            gad += m[0];
            for (int j = 0; j < 3; ++j) {
                const unsigned long long b = ((unsigned long long)__float_as_uint(m[1 + 2 * j]) << 32) | (unsigned long long)__float_as_uint(m[2 + 2 * j]);
                gq[j] += __longlong_as_double((long long)b);
            }
        }
        const float an = aN.count == 1 ? aN.partials[0] : 0.0f;
        const float alpha = safe_div<false>(an, gad);
        double bn = gq[0] - 2.0 * (double)alpha * gq[1] + (double)alpha * (double)alpha * gq[2];
        if (!(bn > 0.0)) bn = 0.0;
        aD_word[0] = gad; bN_word[0] = (float)bn;
    }
    if (src_top) {
        long base = 0;
        for (int k = 0; k < top.n; ++k) {
            for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < top.len[k]; i += (long)gridDim.x * BLOCK) vec[top.off[k] + i] = src_top[base + i];
            base += top.len[k];
        }
    }
    if (src_bot) {
        long base = 0;
        for (int k = 0; k < bot.n; ++k) {
            for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < bot.len[k]; i += (long)gridDim.x * BLOCK) vec[bot.off[k] + i] = src_bot[base + i];
            base += bot.len[k];
        }
    }
}

// ---- ghost units of a partitioned graph problem (thallo_hip.h thallo_units_t): the row parts of k_slab_pack / _unpack(_iter) with index lists
__device__ __forceinline__ int units_per(const thallo_units_t& u) { int p = 0; for (int k = 0; k < u.nplanes; ++k) p += u.len[k]; return p; }
__device__ __forceinline__ void units_gather(const float* __restrict__ vec, const thallo_units_t& u, float* __restrict__ out)
{   // one thread per (unit, float of the unit)
    const int per = units_per(u);
    const long total = (long)u.n * per;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
        const int unit = (int)(i / per); int c = (int)(i - (long)unit * per), k = 0;
        while (c >= u.len[k]) { c -= u.len[k]; ++k; }
        out[i] = vec[u.base[k] + (long)u.units[unit] * u.len[k] + c];
    }
}
__device__ __forceinline__ void units_scatter(float* __restrict__ vec, const thallo_units_t& u, const float* __restrict__ gathered)
{
    const int per = units_per(u);
    const long total = (long)u.n * per;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
        const int g = (int)(i / per); const int within = (int)(i - (long)g * per); int c = within, k = 0;
        while (c >= u.len[k]) { c -= u.len[k]; ++k; }
        vec[u.base[k] + (long)u.units[g] * u.len[k] + c] = gathered[u.src[g] + within];
    }
}
__global__ __launch_bounds__(BLOCK) void k_units_pack(const float* __restrict__ vec, thallo_units_t u, thallo_sum_t s, float* __restrict__ out)
{
    if (blockIdx.x == 0 && s.count > 0) {
        const float v = sum_partials(s.partials, s.count);
        if (threadIdx.x == 0) out[0] = v;
    }
    units_gather(vec, u, out + 1);
}
__global__ __launch_bounds__(BLOCK) void k_units_unpack(float* __restrict__ vec, thallo_units_t u, const float* __restrict__ gathered, long stride, int world, float* __restrict__ sum_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && sum_out) {
        float v = 0.0f;
        for (int r = 0; r < world; ++r) v += gathered[(long)r * stride];
        sum_out[0] = v;
    }
    units_scatter(vec, u, gathered);
}
__global__ __launch_bounds__(BLOCK) void k_units_pack_iter(const float* __restrict__ vec, thallo_units_t u, const float* __restrict__ aD_part, const double* __restrict__ s3, int nb,
                                                            float* __restrict__ out)
{   // header: k_slab_pack_iter's
    if (blockIdx.x == 0 && threadIdx.x < THALLO_WAVE) {
        const int lane = threadIdx.x;
        const float ad = sum_partials(aD_part, nb);
        double q[3] = { 0.0, 0.0, 0.0 };
        for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s3[3 * i]; q[1] += s3[3 * i + 1]; q[2] += s3[3 * i + 2]; }
#pragma unroll
        for (int j = 0; j < 3; ++j) q[j] = wave_sum_all_f64(q[j]);
        if (lane == 0) {
            out[0] = ad;
            for (int j = 0; j < 3; ++j) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(q[j]);
                out[1 + 2 * j] = __uint_as_float((unsigned)(b >> 32)); out[2 + 2 * j] = __uint_as_float((unsigned)b);
            }
        }
    }
    units_gather(vec, u, out + 7);
}
__global__ __launch_bounds__(BLOCK) void k_units_unpack_iter(float* __restrict__ vec, thallo_units_t u, const float* __restrict__ gathered, long stride, int world, thallo_sum_t aN,
                                                              float* __restrict__ aD_word, float* __restrict__ bN_word)
{   // scalars: k_slab_unpack_iter's
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float gad = 0.0f; double gq[3] = { 0.0, 0.0, 0.0 };
        for (int r = 0; r < world; ++r) {
            const float* m = gathered + (long)r * stride;
            gad += m[0];
            for (int j = 0; j < 3; ++j) {
                const unsigned long long b = ((unsigned long long)__float_as_uint(m[1 + 2 * j]) << 32) | (unsigned long long)__float_as_uint(m[2 + 2 * j]);
                gq[j] += __longlong_as_double((long long)b);
            }
        }
        const float an = aN.count == 1 ? aN.partials[0] : 0.0f;
        const float alpha = safe_div<false>(an, gad);
        double bn = gq[0] - 2.0 * (double)alpha * gq[1] + (double)alpha * (double)alpha * gq[2];
        if (!(bn > 0.0)) bn = 0.0;
        aD_word[0] = gad; bN_word[0] = (float)bn;
    }
    units_scatter(vec, u, gathered);
}

__global__ void k_finish_sum(thallo_sum_t s, float* __restrict__ out)
{
    const float v = sum_partials(s.partials, s.count);
    if (threadIdx.x == 0) out[0] = v;
}
__global__ void k_finish_sum_gated(thallo_sum_t s, float* __restrict__ out, const unsigned* __restrict__ gate)
{
    if (__builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;      // LM: the PCG loop already ended on the device, the partials were not written
    const float v = sum_partials(s.partials, s.count);
    if (threadIdx.x == 0) out[0] = v;
}

__global__ void k_alpha_beta(thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* __restrict__ out)
{
    const float n = sum_partials(aN.partials, aN.count);
    const float alpha = safe_div<false>(n, sum_partials(aD.partials, aD.count));
    const float beta = safe_div<false>(sum_partials(bN.partials, bN.count), n);
    if (threadIdx.x == 0) { out[0] = alpha; out[1] = beta; }
}

}  // namespace

extern "C" {

void thallo_hip_debug_set2(int value) { g_nt2 = value; }

long thallo_hip_vector_elems(long n) { return (n + 255) / 256 * 256; }
int thallo_hip_device_cu_count(void) { return cu_count(); }

int thallo_hip_pcg_step2_ranges(float* r, const float* Ap, const float* pre, float* z,
                                long off0, long len0, long off1, long len1,
                                thallo_sum_t aN, thallo_sum_t aD, float* bN_out, thallo_stream_t stream)
{
    if ((off0 | len0 | off1 | len1) & 3) return -(int)hipErrorInvalidValue;     // 16-byte granules
    const long o0 = off0 / 4, l0 = len0 / 4, o1 = off1 / 4, l1 = len1 / 4;
    const int grid = flat_grid(l0 + l1, cu_count());
    hipStream_t s = (hipStream_t)stream;
    if (pre) hipLaunchKernelGGL(k_step2<true>, dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (float4*)z, o0, l0, o1, l1, aN, aD, bN_out, g_nt2);
    else     hipLaunchKernelGGL(k_step2<false>, dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (float4*)z, o0, l0, o1, l1, aN, aD, bN_out, g_nt2);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_pcg_step2(float* r, const float* Ap, const float* pre, float* z, long n,
                         thallo_sum_t aN, thallo_sum_t aD, float* bN_out, thallo_stream_t stream)
{
    return thallo_hip_pcg_step2_ranges(r, Ap, pre, z, 0, (n + 3) / 4 * 4, 0, 0, aN, aD, bN_out, stream);
}

// Ambient gate word of the LM loop (thallo_hip_lm_set_gate): passed to the loop's energy-independent kernels (pcg_pupdate, lm_step1_finish,
// pcg_step2_full, lm_step2_first / second_half), which do nothing once it is non-zero.  NULL outside an LM loop.  Per THREAD: the word is set and reset
// around one plan's LM loop by the thread that runs it, so two plans stepped from two threads never see each other's (or a freed plan's) word.
static thread_local const unsigned* g_gate = nullptr;
void thallo_hip_lm_set_gate(const unsigned* gate) { g_gate = gate; }

int thallo_hip_lm_state_reset(float* state, thallo_stream_t stream)
{
    if (!state) return -(int)hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(state, 0, 8 * sizeof(float), (hipStream_t)stream);
    return e == hipSuccess ? 0 : -(int)e;
}

int thallo_hip_lm_zeta(thallo_sum_t q, int k, float q_tolerance, float* state, thallo_stream_t stream)
{
    if (!q.partials || q.count < 1 || !state) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lm_zeta, dim3(1), dim3(64), 0, (hipStream_t)stream, q, k, q_tolerance, state);
    return check_launch();
}

static int step2_full(float* delta, const float* p, float* r, const float* Ap, const float* pre,
                      float* z, const float* b, long n, thallo_sum_t aN, thallo_sum_t aD,
                      float* bN_out, float* q_out, int lm, ZetaArgs zeta, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4;
    const int grid = flat_grid(n4, cu_count());
    hipStream_t s = (hipStream_t)stream;
#define L2F(PRE, LMF, HB) hipLaunchKernelGGL((k_step2_full<PRE, LMF, HB>), dim3(grid), dim3(BLOCK), 0, s, \
        (float4*)delta, (const float4*)p, (float4*)r, (const float4*)Ap, (const float4*)pre, (float4*)z, (const float4*)b, n4, aN, aD, bN_out, q_out, g_gate, zeta)
    const bool hp = pre != nullptr, hb = (b != nullptr && q_out != nullptr);
    if (hp) { if (lm) { if (hb) L2F(true, true, true); else L2F(true, true, false); } else { if (hb) L2F(true, false, true); else L2F(true, false, false); } }
    else    { if (lm) { if (hb) L2F(false, true, true); else L2F(false, true, false); } else { if (hb) L2F(false, false, true); else L2F(false, false, false); } }
#undef L2F
    int e = check_launch();
    return e ? e : grid;
}
int thallo_hip_pcg_step2_full(float* delta, const float* p, float* r, const float* Ap, const float* pre,
                              float* z, const float* b, long n, thallo_sum_t aN, thallo_sum_t aD,
                              float* bN_out, float* q_out, int lm, thallo_stream_t stream)
{ return step2_full(delta, p, r, Ap, pre, z, b, n, aN, aD, bN_out, q_out, lm, ZetaArgs{ nullptr, nullptr, 0, 0.0f }, stream); }
int thallo_hip_pcg_step2_full_zeta(float* delta, const float* p, float* r, const float* Ap, const float* pre,
                                   float* z, const float* b, long n, thallo_sum_t aN, thallo_sum_t aD,
                                   float* bN_out, float* q_out, unsigned* tickets, int k, float q_tolerance, float* lm_state, thallo_stream_t stream)
{
    if (!b || !q_out || !tickets || !lm_state) return -(int)hipErrorInvalidValue;
    return step2_full(delta, p, r, Ap, pre, z, b, n, aN, aD, bN_out, q_out, 1, ZetaArgs{ tickets, lm_state, k, q_tolerance }, stream);
}

int thallo_hip_pcg_step3(float* p, const float* z, long n, thallo_sum_t bN, thallo_sum_t aN, int lm, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4;
    const int grid = flat_grid(n4, cu_count());
    hipStream_t s = (hipStream_t)stream;
    if (lm) hipLaunchKernelGGL(k_step3<true>, dim3(grid), dim3(BLOCK), 0, s, (float4*)p, (const float4*)z, n4, bN, aN);
    else    hipLaunchKernelGGL(k_step3<false>, dim3(grid), dim3(BLOCK), 0, s, (float4*)p, (const float4*)z, n4, bN, aN);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_pcg_pupdate_ranges(const float* z, const float* p_in, float* p_out, float* delta,
                                  long off0, long len0, long off1, long len1, int first,
                                  thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_stream_t stream)
{
    if ((off0 | len0 | off1 | len1) & 3) return -(int)hipErrorInvalidValue;
    const int grid = flat_grid((len0 + len1) / 4, cu_count());
    hipLaunchKernelGGL(k_pupdate, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)z, (const float4*)p_in, (float4*)p_out, (float4*)delta,
                       off0 / 4, len0 / 4, off1 / 4, len1 / 4, first, aNp, aDp, bNp, g_gate);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_pcg_pupdate(const float* z, const float* p_in, float* p_out, float* delta, long n, int first,
                           thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_stream_t stream)
{
    return thallo_hip_pcg_pupdate_ranges(z, p_in, p_out, delta, 0, (n + 3) / 4 * 4, 0, 0, first, aNp, aDp, bNp, stream);
}

int thallo_hip_pcg_update(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, int first,
                          thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_stream_t stream)
{
    if (!r || !p_in || !p_out || (!first && (!Ap || !delta))) return -(int)hipErrorInvalidValue;
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipStream_t s = (hipStream_t)stream;
    if (pre) hipLaunchKernelGGL((k_pcg_update<true, false>), dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out, (float4*)delta, n4, first, aNp, aDp, bNp, (const unsigned*)nullptr, (float*)nullptr);
    else     hipLaunchKernelGGL((k_pcg_update<false, false>), dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out, (float4*)delta, n4, first, aNp, aDp, bNp, (const unsigned*)nullptr, (float*)nullptr);
    return check_launch();
}

int thallo_hip_pcg_update_fin(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, thallo_sum_t alphaN_prev,
                              const float* alphaD_partials, const double* s3_partials, int count, float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!r || !Ap || !p_in || !p_out || !delta || !alphaN_prev.partials || !alphaD_partials || !s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || !alphaD_word || !betaN_word)
        return -(int)hipErrorInvalidValue;
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipStream_t s = (hipStream_t)stream;
    if (pre) hipLaunchKernelGGL((k_pcg_update_fin<true>), dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out, (float4*)delta, n4,
                                alphaN_prev, alphaD_partials, s3_partials, count, alphaD_word, betaN_word);
    else     hipLaunchKernelGGL((k_pcg_update_fin<false>), dim3(grid), dim3(BLOCK), 0, s, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out, (float4*)delta, n4,
                                alphaN_prev, alphaD_partials, s3_partials, count, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_pcg_update_lm_fin(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, thallo_sum_t alphaN_prev,
                                 const float* alphaD_partials, const double* s3_partials, const double* q3_partials, int count, float* alphaD_word, float* betaN_word,
                                 float* lm_state, int k_prev, float q_tolerance, int q_in, int q_out, thallo_stream_t stream)
{
    if (!r || !Ap || !pre || !p_in || !p_out || !delta || !alphaN_prev.partials || !alphaD_partials || !s3_partials || !q3_partials || count < 1 || count > THALLO_MAX_PARTIALS ||
        !alphaD_word || !betaN_word || !lm_state || k_prev < 0 || q_in < 0 || q_in > 7 || q_out < 0 || q_out > 7 || q_in == q_out) return -(int)hipErrorInvalidValue;
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_pcg_update_lm_fin, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out, (float4*)delta, n4,
                       alphaN_prev, alphaD_partials, s3_partials, q3_partials, count, alphaD_word, betaN_word, lm_state, k_prev, q_tolerance, q_in, q_out);
    return check_launch();
}

int thallo_hip_pcg_update_lm(float* r, const float* Ap, const float* pre, const float* p_in, float* p_out, float* delta, long n, int first,
                             thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, float* betaN_word, const float* lm_state, thallo_stream_t stream)
{
    if (!r || !pre || !p_in || !p_out || !lm_state || first < 0 || first > 2 || (!first && (!Ap || !delta))) return -(int)hipErrorInvalidValue;
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL((k_pcg_update<true, true>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)r, (const float4*)Ap, (const float4*)pre, (const float4*)p_in, (float4*)p_out,
                       (float4*)delta, n4, first, aNp, aDp, bNp, reinterpret_cast<const unsigned*>(lm_state) + 1, betaN_word);
    return check_launch();
}

int thallo_hip_lm_owed_delta(float* delta, const float* p_even, const float* p_odd, long n, const float* alphaN_words, const float* alphaD_words, int word_stride,
                             const float* lm_state, int L, thallo_stream_t stream)
{
    if (!delta || !p_even || !p_odd || !alphaN_words || !alphaD_words || word_stride < 1 || !lm_state || L < 0) return -(int)hipErrorInvalidValue;
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_lm_owed_delta, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)delta, (const float4*)p_even, (const float4*)p_odd, n4, alphaN_words, alphaD_words,
                       word_stride, lm_state, L);
    return check_launch();
}

int thallo_hip_pcg_scalars_finish(const float* aD_partials, const double* s3_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!aD_partials || !s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || !alphaD_word || !betaN_word) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_scalars_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, aD_partials, s3_partials, count, alphaN, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_lm_finalize_diagonal(const float* diag, float* SSq, float* CtC, float* pre, const float* r, float* b, float* z, long n,
                                    float radius, float min_lm_diagonal, float max_lm_diagonal, int save_ssq, int use_preconditioner,
                                    float* aN_out, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_lm_finalize, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)diag, (float4*)SSq, (float4*)CtC, (float4*)pre,
                       (const float4*)r, (float4*)b, (float4*)z, n4, radius, min_lm_diagonal, max_lm_diagonal, save_ssq, use_preconditioner, aN_out);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_lm_step1_finish(float* Ap, const float* CtC, const float* p, long n, float* aD_out, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_lm_step1_finish, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)Ap, (const float4*)CtC, (const float4*)p, n4, aD_out, g_gate);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_lm_step2_first_half(float* delta, const float* p, long n, thallo_sum_t aN, thallo_sum_t aD, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_lm_step2_first, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)delta, (const float4*)p, n4, aN, aD, g_gate);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_lm_step2_second_half(float* r, const float* b, const float* Adelta, const float* pre, float* z, const float* delta, long n,
                                    float* bN_out, float* q_out, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_lm_step2_second, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)r, (const float4*)b, (const float4*)Adelta,
                       (const float4*)pre, (float4*)z, (const float4*)delta, n4, bN_out, q_out, g_gate);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_pcg_init_finish(const float* r, const float* diag, float* pre, float* z, long n, int use_preconditioner,
                               float* aN_out, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_init_finish, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)r, (const float4*)diag, (float4*)pre, (float4*)z, n4,
                       use_preconditioner, aN_out);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_dot(const float* a, const float* b, long n, float* out, thallo_stream_t stream)
{
    const long n4 = (n + 3) / 4; const int grid = flat_grid(n4, cu_count());
    hipLaunchKernelGGL(k_dot, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b, n4, out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_linear_update(float* X, const float* delta, const float* p, long len,
                             thallo_sum_t aN, thallo_sum_t aD, thallo_stream_t stream)
{
    const int grid = flat_grid(len, cu_count());
    hipStream_t s = (hipStream_t)stream;
    if (p) hipLaunchKernelGGL(k_linear_update<true>, dim3(grid), dim3(BLOCK), 0, s, X, delta, p, len, aN, aD);
    else   hipLaunchKernelGGL(k_linear_update<false>, dim3(grid), dim3(BLOCK), 0, s, X, delta, p, len, aN, aD);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_linear_update2(float* X, const float* delta, const float* p_older, thallo_sum_t aN0, thallo_sum_t aD0,
                              const float* p, thallo_sum_t aN1, thallo_sum_t aD1, long len, thallo_stream_t stream)
{
    if (!p_older || !p) return -(int)hipErrorInvalidValue;
    const int grid = flat_grid(len, cu_count());
    hipLaunchKernelGGL(k_linear_update2, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, X, delta, p_older, aN0, aD0, p, aN1, aD1, len);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_linear_update_n(float* X, float* delta, thallo_update_terms_t terms, long len, int max_workgroups, thallo_stream_t stream)
{
    if (!delta || len < 0 || terms.count < 0 || terms.count > THALLO_HIP_MAX_UPDATE_TERMS) return -(int)hipErrorInvalidValue;
    bool v4 = !(len & 3) && !(((uintptr_t)X | (uintptr_t)delta) & 15);
    for (int j = 0; j < terms.count; ++j) {
        if (!terms.p[j] || !terms.alphaN[j].partials || !terms.alphaD[j].partials || terms.alphaN[j].count < 1 || terms.alphaD[j].count < 1) return -(int)hipErrorInvalidValue;
        if ((uintptr_t)terms.p[j] & 15) v4 = false;
    }
    if (len == 0 || (terms.count == 0 && !X)) return 0;
    int grid = flat_grid(v4 ? len >> 2 : len, 2 * cu_count());
    if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;      // (a background update next to the PCG loop's launches: a share of the bandwidth, not all of it)
    if (v4) hipLaunchKernelGGL(k_linear_update_n<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, X, delta, terms, len);
    else    hipLaunchKernelGGL(k_linear_update_n<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, X, delta, terms, len);
    int e = check_launch();
    return e ? e : grid;
}

int thallo_hip_finish_sum(thallo_sum_t sum, float* out, thallo_stream_t stream)
{
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, sum, out);
    return check_launch();
}

int thallo_hip_finish_sum_gated(thallo_sum_t sum, float* out, const unsigned* gate, thallo_stream_t stream)
{
    if (!gate || !out || !sum.partials) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_finish_sum_gated, dim3(1), dim3(64), 0, (hipStream_t)stream, sum, out, gate);
    return check_launch();
}

int thallo_hip_slab_pack(const float* vec, thallo_segs_t segs, thallo_sum_t sum, float* out, thallo_stream_t stream)
{
    if (segs.n < 0 || segs.n > 8) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_slab_pack, dim3(8), dim3(BLOCK), 0, (hipStream_t)stream, vec, segs, sum, out);
    return check_launch();
}

int thallo_hip_slab_unpack(float* vec, thallo_segs_t top, const float* src_top, thallo_segs_t bot, const float* src_bot,
                           const float* gathered, long stride, int world, float* sum_out, thallo_stream_t stream)
{
    if (top.n < 0 || top.n > 8 || bot.n < 0 || bot.n > 8) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_slab_unpack, dim3(8), dim3(BLOCK), 0, (hipStream_t)stream, vec, top, src_top, bot, src_bot, gathered, stride, world, sum_out);
    return check_launch();
}

int thallo_hip_block_sums(const float* p, const float* Ap, const float* r, const float* pre, long n, float* aD_out, double* s3_out, thallo_stream_t stream)
{
    if (!p || !Ap || !r || !aD_out || !s3_out || n < 0 || (n & 3)) return -(int)hipErrorInvalidValue;
    const long n4 = n / 4; const int grid = flat_grid(n4 ? n4 : 1, cu_count());
    if (pre) hipLaunchKernelGGL(k_block_sums<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)p, (const float4*)Ap, (const float4*)r, (const float4*)pre, n4, aD_out, s3_out);
    else     hipLaunchKernelGGL(k_block_sums<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (const float4*)p, (const float4*)Ap, (const float4*)r, (const float4*)pre, n4, aD_out, s3_out);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_shard_scalars(const float* gathered, long stride, int world, const float* aD_partials, const double* s3_partials, int count, thallo_sum_t alphaN,
                             float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!gathered || world < 1 || stride < 1 || !aD_partials || count < 0 || count > THALLO_MAX_PARTIALS || !alphaD_word) return -(int)hipErrorInvalidValue;      // (count 0: the shared block contributes nothing -- a sum that is linear in the ranks' parts)
    const int only_first = betaN_word == nullptr;
    if (!only_first && (!s3_partials || alphaN.count < 1)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_shard_scalars, dim3(1), dim3(64), 0, (hipStream_t)stream, gathered, stride, world, aD_partials, s3_partials, count, alphaN, only_first, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_range_unpack(float* vec, thallo_segs_t first_rank_pieces, const float* gathered, long stride, long skip, int world, thallo_stream_t stream)
{
    if (!vec || !gathered || first_rank_pieces.n < 1 || first_rank_pieces.n > 8 || world < 1 || stride < 1 || skip < 0) return -(int)hipErrorInvalidValue;
    long len = 0; for (int j = 0; j < first_rank_pieces.n; ++j) len += first_rank_pieces.len[j];
    int grid = (int)((len + BLOCK - 1) / BLOCK); if (grid > 64) grid = 64; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_range_unpack, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, vec, first_rank_pieces, gathered, stride, skip, world);
    return check_launch();
}

static bool units_ok(const thallo_units_t& u, bool unpack)
{
    if (u.n < 0 || u.nplanes < 1 || u.nplanes > 8 || (u.n > 0 && (!u.units || (unpack && !u.src)))) return false;
    for (int k = 0; k < u.nplanes; ++k) if (u.len[k] < 1 || u.base[k] < 0) return false;
    return true;
}
static int units_grid(const thallo_units_t& u) { long t = 0; for (int k = 0; k < u.nplanes; ++k) t += u.len[k]; t *= u.n; int g = (int)((t + BLOCK - 1) / BLOCK); return g > 64 ? 64 : g < 1 ? 1 : g; }
int thallo_hip_units_pack(const float* vec, thallo_units_t u, thallo_sum_t sum, float* out, thallo_stream_t stream)
{
    if (!units_ok(u, false) || !out || (u.n > 0 && !vec) || sum.count < 0 || (sum.count > 0 && !sum.partials)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_units_pack, dim3(units_grid(u)), dim3(BLOCK), 0, (hipStream_t)stream, vec, u, sum, out);
    return check_launch();
}
int thallo_hip_units_unpack(float* vec, thallo_units_t u, const float* gathered, long stride, int world, float* sum_out, thallo_stream_t stream)
{
    if (!units_ok(u, true) || !gathered || (u.n > 0 && !vec) || stride < 1 || world < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_units_unpack, dim3(units_grid(u)), dim3(BLOCK), 0, (hipStream_t)stream, vec, u, gathered, stride, world, sum_out);
    return check_launch();
}
int thallo_hip_units_pack_iter(const float* vec, thallo_units_t u, const float* aD_partials, const double* s3_partials, int count, float* out, thallo_stream_t stream)
{
    if (!units_ok(u, false) || !aD_partials || !s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || !out || (u.n > 0 && !vec)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_units_pack_iter, dim3(units_grid(u)), dim3(BLOCK), 0, (hipStream_t)stream, vec, u, aD_partials, s3_partials, count, out);
    return check_launch();
}
int thallo_hip_units_unpack_iter(float* vec, thallo_units_t u, const float* gathered, long stride, int world, thallo_sum_t alphaN,
                                 float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!units_ok(u, true) || !gathered || stride < 7 || world < 1 || !alphaD_word || !betaN_word || (u.n > 0 && !vec)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_units_unpack_iter, dim3(units_grid(u)), dim3(BLOCK), 0, (hipStream_t)stream, vec, u, gathered, stride, world, alphaN, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_slab_pack_iter(const float* vec, thallo_segs_t segs, const float* aD_partials, const double* s3_partials, int count, float* out, thallo_stream_t stream)
{
    if (segs.n < 0 || segs.n > 8 || !aD_partials || !s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || !out) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_slab_pack_iter, dim3(8), dim3(BLOCK), 0, (hipStream_t)stream, vec, segs, aD_partials, s3_partials, count, out);
    return check_launch();
}

int thallo_hip_slab_unpack_iter(float* vec, thallo_segs_t top, const float* src_top, thallo_segs_t bot, const float* src_bot,
                                const float* gathered, long stride, int world, thallo_sum_t alphaN, float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (top.n < 0 || top.n > 8 || bot.n < 0 || bot.n > 8 || !gathered || world < 1 || alphaN.count != 1 || !alphaD_word || !betaN_word) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_slab_unpack_iter, dim3(8), dim3(BLOCK), 0, (hipStream_t)stream, vec, top, src_top, bot, src_bot, gathered, stride, world, alphaN, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_alpha_beta(thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out2, thallo_stream_t stream)
{
    hipLaunchKernelGGL(k_alpha_beta, dim3(1), dim3(64), 0, (hipStream_t)stream, aN, aD, bN, out2);
    return check_launch();
}

}  // extern "C"
