// energy_image_warping_march_rc.hip -- the marching one-kernel PCG iteration of image_warping WITHOUT an A p plane (round 4).
//
// energy_image_warping_march.hip moves 81 B/pixel per iteration (+ 18 of deferred delta): launch k-1 writes A p_{k-1} (12 B/pixel) only so that launch k
// can read it (12 B/pixel) to form r_k = r_{k-1} - alpha_{k-1} A p_{k-1}.  Here launch k RECOMPUTES A p_{k-1} from the rows of p_{k-1} it loads anyway
// (it needs p_{k-1} for p_k = M^-1 r_k + beta_{k-1} p_{k-1}): the stencil runs twice per pixel and launch -- once on p_{k-1} for the residual update, once
// on p_k for the iteration's sums -- and the plane never exists.  Per pixel: read r 12, p 12, cs 8, flags 1; write r 12, p 12 = 57 B (+ 18 of deferred
// delta on average = 75 B against 99).  The kernel is bandwidth- and latency-bound at one wave per SIMD; the second stencil fills issue slots the wave
// spent waiting.  Replaces PCGStep1 + PCGStep2 + PCGStep3 of gauss_newton.t:734-752,801-843,889-899.
//
// Same expressions on the same inputs as the stored-plane kernel (jtjp_pair / iter_sums_pixel of iw_march.hpp, -ffp-contract=on in both files), same
// strips, segments and summation order: r, p, delta and every alpha_k / beta_k are BIT-identical to it (tests/test_gpu_parity.py).
//
// Shape: as the stored-plane kernel -- a wave owns a column strip of 128 pixels (lane l: pixels x0+2l, x0+2l+1; lanes 1..62 are outputs) and marches down
// its R rows; x neighbours through DPP wave shifts, y neighbours from the lane's own registers.  What changes is the depth of the pipeline: at the step
// that takes row t (p_{k-1}, cos / sin, flags of row t; r_{k-1}, delta of row t-1) the lane forms
//     A p_{k-1}(t-1)  from p_{k-1}(t-2 .. t)            ->  r_k(t-1), p_k(t-1)   (stored for the segment's own rows)
//     A p_k(t-2)      from p_k(t-3 .. t-1)              ->  the sums of row t-2
// so a segment [ya, yb) takes rows ya-2 .. yb+1 of p / cs / flags (4 halo rows instead of 2) and rows ya-1 .. yb of r.  Row state lives in rings of four
// indexed by the row modulo 4, four rows per loop trip (compile-time indices, no register shifts); rows are prefetched DEPTH steps ahead into registers.
#include "iw_march.hpp"
#include <hip/hip_ext.h>

using namespace thallo;

namespace {

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

// Raw-buffer addressing: ONE descriptor per solver vector / plane (4 SGPRs), the lane's position in a row as a constant VGPR byte offset, the row as an SGPR byte
// offset -- a row step needs no vector address arithmetic at all (the 64-bit global pointers of the first version cost ~10 v_lshl_add_u64 per row and ~20 VGPRs).
// A vector is [Offset part: N x float2 | Angle part: N x float] = 12 N bytes < 4 GiB up to N = 357 Mpixel (16384^2 = 268 M: checked by the host).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000); }
template <bool NT> __device__ __forceinline__ u32x4 bld4(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, NT ? 2 : 0); }
template <bool NT> __device__ __forceinline__ u32x2 bld2(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, NT ? 2 : 0); }
template <bool NT> __device__ __forceinline__ unsigned bld1(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, NT ? 2 : 0); }
template <bool NT> __device__ __forceinline__ void bst4(rsrc_t r, unsigned vo, unsigned so, float a, float b, float c, float d)
{ u32x4 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); v.z = __float_as_uint(c); v.w = __float_as_uint(d); __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, NT ? 2 : 0); }
template <bool NT> __device__ __forceinline__ void bst2(rsrc_t r, unsigned vo, unsigned so, float a, float b)
{ u32x2 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, NT ? 2 : 0); }
__device__ __forceinline__ float uf(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float to_sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// iteration_scalars (iw_device.hpp) in a lean form: sums are added up by a ROLLED loop with one load in flight (the same additions in the same order as
// sum_partials / k_iter_finish; finished sums are one word; alphaN_0 and iteration k-1's raw partials are at most a few per lane).  The general routine keeps 16
// loads in flight per sum behind per-lane predicates: inside the row loop, with every ring register live, that cost ~20 VGPRs and ~90 spilled SGPRs.
__device__ __forceinline__ float rc_sum(thallo_sum_t s)
{
    if (s.count == 1) return s.partials[0];
    float t = 0.0f;
    for (int i = threadIdx.x & (THALLO_WAVE - 1); i < s.count; i += THALLO_WAVE) t += s.partials[i];
    return wave_sum_all(t);
}
__device__ __forceinline__ void rc_iteration_scalars(thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, const PrevSums& prev, float& alpha, float& beta, bool writer)
{
    const float an = rc_sum(aNp);
    float ad, bn;
    if (prev.count > 0) {
        const int lane = threadIdx.x & (THALLO_WAVE - 1), nb = prev.count;
        float t = 0.0f; double n = 0.0, a1 = 0.0, b1 = 0.0;
        for (int i = lane; i < nb; i += THALLO_WAVE) { t += prev.aD_part[i]; n += prev.s12_part[3 * i]; a1 += prev.s12_part[3 * i + 1]; b1 += prev.s12_part[3 * i + 2]; }
        ad = nb == 1 ? prev.aD_part[0] : wave_sum_all(t);
        n = wave_sum_all_d(n); a1 = wave_sum_all_d(a1); b1 = wave_sum_all_d(b1);
        alpha = safe_div<false>(an, ad);
        double bd = n - 2.0 * (double)alpha * a1 + (double)alpha * (double)alpha * b1;
        if (!(bd > 0.0)) bd = 0.0;
        bn = (float)bd;
        if (writer) { prev.aD_word[0] = ad; prev.bN_word[0] = bn; }      // writer: exactly ONE thread of the launch, one that certainly gets here
    } else {
        ad = rc_sum(aDp); bn = rc_sum(bNp);
        alpha = safe_div<false>(an, ad);
    }
    beta = safe_div<false>(bn, an);
}

// ... and on a row slab of a multi-GPU run with the DEFERRED cross-rank finish (round 4): the partials of iteration k-1 are this rank's only.  ONE wave of the launch that
// has no rows (the launch carries eight extra workgroups for it: its last workgroup's wave 0) adds them up -- load_iteration_sums' order, every load in flight at once --,
// trades the rank's four sums with the other ranks through the mailbox (dist_exchange_iter_wave_seq: the granules, slots and rank order of the exchange that used to
// sit at the END of launch k-1, where nothing could hide its ~8 us), leaves the two words behind and publishes them as two tagged granules; every working wave polls
// those (agent scope, one line; bounded like every wait of the exchange) at the start of its second trip, i.e. behind the round trip of its first row loads.
// A ghost row's A p_{k-1} is read after that point: the peers' rows were stored before their sums went out.
__device__ __forceinline__ void rc_exchange_prev(thallo_sum_t aNp, const PrevSums& prev, const thallo_dist_t& dd)
{
    typedef unsigned long long u64_t;
    const unsigned seq = ld_agent(dd.ctl + DIST_SEQ);
    const unsigned tag = (seq << 12) | (unsigned)(prev.xslot / 7 + 1);
    const IterationSums S = load_iteration_sums(prev.aD_part, prev.s12_part, prev.count, aNp);
    float ad = 0.0f, al = 0.0f, bn = 0.0f;
    dist_exchange_iter_wave_seq(dd, seq, prev.xslot, S.ad, S.n, S.s1, S.s2, S.an, prev.aD_word, prev.bN_word, ExtraSums{ false, 0.0f, 0.0, 0.0, 0.0 }, &ad, &al, &bn);
    if ((threadIdx.x & (THALLO_WAVE - 1)) == 0) {
        __hip_atomic_store(prev.gs, ((u64_t)tag << 32) | (u64_t)__float_as_uint(ad), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(prev.gs + 1, ((u64_t)tag << 32) | (u64_t)__float_as_uint(bn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void rc_iteration_scalars_x(thallo_sum_t aNp, const PrevSums& prev, const thallo_dist_t& dd, float& alpha, float& beta)
{
    typedef unsigned long long u64_t;
    const float an = rc_sum(aNp);
    const unsigned seq = ld_agent(dd.ctl + DIST_SEQ);
    const unsigned tag = (seq << 12) | (unsigned)(prev.xslot / 7 + 1);
    u64_t v0 = __hip_atomic_load(prev.gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), v1 = __hip_atomic_load(prev.gs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int it = 0; long long t0 = 0;
    const long long bound = dist_spin_ticks(dd);
    while ((unsigned)(v0 >> 32) != tag || (unsigned)(v1 >> 32) != tag) {
        if ((it & 255) == 0) { if (ld_agent(dd.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
        ++it;
        if ((it & 255) == 0 && wall_clock64() - t0 > bound) {
            if (__hip_atomic_exchange(dd.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                unsigned* pm = dd.ctl + DIST_POST_MORTEM;
                pm[0] = (unsigned)prev.xslot; pm[1] = 0xffffu; pm[2] = tag; pm[3] = (unsigned)(v0 >> 32); pm[4] = (unsigned)(v1 >> 32);
            }
            break;
        }
        __builtin_amdgcn_s_sleep(4);
        v0 = __hip_atomic_load(prev.gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v1 = __hip_atomic_load(prev.gs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float ad = __uint_as_float((unsigned)v0), bn = __uint_as_float((unsigned)v1);
    alpha = safe_div<false>(an, ad);
    beta = safe_div<false>(bn, an);
}
// the last iteration of a GN step has no next launch: one wave does what the exchange wave of a launch would have done
__global__ __launch_bounds__(64) void k_rc_dist_finish(thallo_sum_t aNp, PrevSums prev, thallo_dist_t dd) { rc_exchange_prev(aNp, prev, dd); }

template <int DMODE>
struct RawRc {                              // what one step takes, for one lane (2 pixels)
    u32x4 po, cs; u32x2 pa; unsigned f;     // row t:   p_{k-1} (Offset part x0,y0,x1,y1 | Angle part a0,a1), (c0,s0,c1,s1), the dword holding the pair's flags bytes
    u32x4 ro; u32x2 ra;                     // row t-1: r_{k-1}
    u32x4 dlo, ppo; u32x2 dla, ppa;         // row t-1: delta (DMODE 0, 2) and p_{k-2} (DMODE 2)
};
// (pairs move as pairs: one v_mov_b64 per two words, and {x, y} / {c, s} stay in an aligned register pair for the packed arithmetic)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void take4u(u32x4& d, const u32x4& s) { const u64x2 sv = __builtin_bit_cast(u64x2, s); unsigned long long a, b; take_pair(a, sv.x); take_pair(b, sv.y); u64x2 dv; dv.x = a; dv.y = b; d = __builtin_bit_cast(u32x4, dv); }
__device__ __forceinline__ void take2u(u32x2& d, const u32x2& s) { unsigned long long a; take_pair(a, __builtin_bit_cast(unsigned long long, s)); d = __builtin_bit_cast(u32x2, a); }
template <int DMODE>
__device__ __forceinline__ void take(RawRc<DMODE>& d, const RawRc<DMODE>& s)
{
    take4u(d.po, s.po); take2u(d.pa, s.pa); take4u(d.cs, s.cs); take1(d.f, s.f); take4u(d.ro, s.ro); take2u(d.ra, s.ra);
    if (DMODE != 1) { take4u(d.dlo, s.dlo); take2u(d.dla, s.dla); }
    if (DMODE == 2) { take4u(d.ppo, s.ppo); take2u(d.ppa, s.ppa); }
}

struct PRow { v2f xy[2]; float pa[2]; };                         // p of a lane's pixel pair in one row: (x, y) of each pixel as a register pair (iw_march.hpp jtjp_pair_xy)
struct GRow { v2f cs[2], gx[2]; float a[2]; unsigned f; };       // (cos, sin) of Angle and (sin, -cos), the active bits as 0 / 1, the two flags bytes
struct RRow { v2f xy[2]; float ra[2]; };                         // r_k
__device__ __forceinline__ v2f uf2(unsigned a, unsigned b) { return v2f{ uf(a), uf(b) }; }

// DMODE: the delta update this launch carries (THALLO_IW_STEP1_MODE): 0 delta += alpha p_{k-1}; 1 none; 2 delta += alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}
// SLAB: 0 a whole image; 1 / 2 one rank's row slab of a multi-GPU run (local image = owned rows [row0, row1) + one ghost row towards each neighbour).  The exchange is
// the stored-plane kernel's -- per iteration the first / last owned row of A p_k and the iteration's sums -- so the two kernels are interchangeable per launch:
//   * the rows of A p_k the neighbours need exist anyway (the second stencil forms A p_k on every owned row for the sums): rows row0 and row1 - 1 go into A_out
//     (1: this rank's plane, from where thallo_hip_slab_pack_iter takes them; 2: straight into the neighbours' ghost rows, peer-to-peer, and the launch's last
//     workgroup is the scalar exchange);
//   * on a GHOST row A p_{k-1} cannot be recomputed (it needs p_{k-1} two rows into the neighbour): it is read from A_in, where the exchange put it, and r_k, p_k of
//     the ghost row follow locally -- same inputs, same bits as the owner's -- and are stored (the ghost rows of r, p stay current, as in the stored-plane kernel).
template <int DMODE, int DEPTH, int NTM, int OCC, int SLAB>
__global__ __launch_bounds__(MARCH_NT, OCC) void k_iter_march_rc(MarchGeo g, const float* __restrict__ cs, const unsigned char* __restrict__ flags, float wf2, float wr2,
                                                            const float* __restrict__ r_in, float* __restrict__ r_out, const float* __restrict__ A_in, float* __restrict__ A_out,
                                                            const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta,
                                                            thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                                            float* __restrict__ aD_out, double* __restrict__ s12_out, const int* __restrict__ irregular,
                                                            thallo_dist_t dd, unsigned* __restrict__ fin_tickets, float* __restrict__ aD_word, float* __restrict__ bN_word, int xslot,
                                                            PrevSums prev)
{
    static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "the prefetch slots rotate inside a trip of four rows");
    __shared__ float4 lut[32];              // by the 5-bit flags value: M^-1 of the Offset channels, of the Angle channel, w_fit^2 where the fit residual is valid
    __shared__ float red[16];
    __shared__ double redd[48];
    constexpr bool nt_delta = NTM & 1, nt_ra = NTM & 2, nt_out = NTM & 4, nt_pin = NTM & 8, nt_pout = NTM & 16, nt_const = NTM & 32;
    // unit-pixel-grid form only; should the word pcg_init wrote this GN step say otherwise, poison the scalars (NaN cost downstream)
    if (irregular != nullptr && __builtin_amdgcn_readfirstlane(irregular[0]) != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && aD_word) { aD_word[0] = __builtin_nanf(""); bN_word[0] = __builtin_nanf(""); }
        if (blockIdx.x == 0 && threadIdx.x == 0 && prev.count > 0) { prev.aD_word[0] = __builtin_nanf(""); prev.bN_word[0] = __builtin_nanf(""); }
        if (threadIdx.x == 0) { aD_out[blockIdx.x] = __builtin_nanf(""); }
        return;
    }
    if (threadIdx.x < 32) { float mo, ma; pre_from_flags((unsigned char)threadIdx.x, wf2, wr2, mo, ma); lut[threadIdx.x] = make_float4(mo, ma, (threadIdx.x & 2) ? wf2 : 0.f, 0.f); }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned N = (unsigned)g.W * (unsigned)g.H;         // (12 N < 2^32: host-checked)
    const int W2 = g.W >> 1;                                  // pixel pairs per row
    int strip, ya, yb;
    march_place(g, wave, strip, ya, yb);
    const bool work = ya < yb;
    const int x0 = strip * MARCH_USE - 2 + 2 * lane;          // first of this lane's two pixels
    const bool xin = x0 >= 0 && x0 < g.W;                     // W even: both pixels exist or neither
    const bool xout = xin && lane >= 1 && lane <= 62;         // this lane's pixels are outputs of this wave

    const rsrc_t RS_P = make_rsrc(p_in), RS_R = make_rsrc(r_in), RS_CS = make_rsrc(cs), RS_F = make_rsrc(flags);
    const rsrc_t RS_Q = make_rsrc(p_out), RS_RO = make_rsrc(r_out), RS_D = make_rsrc(delta);
    const rsrc_t RS_AI = make_rsrc(SLAB ? (const void*)A_in : (const void*)r_in), RS_AO = make_rsrc(SLAB ? (void*)A_out : (void*)r_out);      // (slabs only: the exchanged rows of A p)
    // Loads are UNCONDITIONAL (rows clamped into the image / the segment, columns into the row; validity applied when the row enters the rings): a load under a
    // branch is merged with the slot's old value right behind the branch, i.e. waited for at once (energy_image_warping_march.hip)
    const int xc = x0 < 0 ? 0 : x0 > g.W - 2 ? g.W - 2 : x0;
    const unsigned h = (unsigned)xc >> 1;                     // the lane's pixel pair in its row; output lanes: xc == x0, so loads and stores share the offsets
    const unsigned vo16 = h * 16u, vo8 = h * 8u;              // byte offsets inside a row: Offset part / cs (float4 per pair), Angle part (float2 per pair)
    // flags: the aligned dword that holds the pair's two bytes.  Pair index i = t W2 + h, dword i >> 1, bytes at bit 16 (i & 1).  With p = (t W2) & 1:
    // p = 0: dword t W2 / 2 + (h >> 1); p = 1: (t W2 - 1) / 2 + ((h + 1) >> 1) -- the row part is scalar, the lane part one of two constants
    const unsigned vf0 = (h >> 1) * 4u, vf1 = ((h + 1u) >> 1) * 4u, sh0 = (h & 1u) * 16u;
    const unsigned angle0 = 8u * N;                           // byte offset of a vector's Angle part

    float alpha = 0.0f, beta = 0.0f, alpha2 = 0.0f;
    // the words of iteration k-1 are left behind by the one wave that owns the first segment of strip 0
    const bool scal_writer = work && strip == 0 && ya == g.row0 && lane == 0;

    typedef RawRc<DMODE> RawT;
    RawT slot[DEPTH];
    auto issue = [&](RawT& s, int t) {
        const unsigned tc = (unsigned)(t < 0 ? 0 : t > g.H - 1 ? g.H - 1 : t), row = tc * (unsigned)W2;
        s.po = bld4<nt_pin>(RS_P, vo16, row * 16u); s.pa = bld2<nt_pin>(RS_P, vo8, angle0 + row * 8u);
        s.cs = bld4<nt_const>(RS_CS, vo16, row * 16u);
        const unsigned par = row & 1u;
        s.f = bld1<nt_const>(RS_F, par ? vf1 : vf0, (row - par) * 2u);
        const unsigned tr = (unsigned)(t - 1 < 0 ? 0 : t - 1 > g.H - 1 ? g.H - 1 : t - 1), rrow = tr * (unsigned)W2;
        s.ro = bld4<nt_ra>(RS_R, vo16, rrow * 16u); s.ra = bld2<nt_ra>(RS_R, vo8, angle0 + rrow * 8u);
        if (DMODE != 1) {       // delta (and p_{k-2}) of the segment's own rows only (the halo rows re-read a row of the segment, unused)
            const unsigned td = (unsigned)(t - 1 < ya ? ya : t - 1 > yb - 1 ? yb - 1 : t - 1), drow = td * (unsigned)W2;
            s.dlo = bld4<nt_delta>(RS_D, vo16, drow * 16u); s.dla = bld2<nt_delta>(RS_D, vo8, angle0 + drow * 8u);
            if (DMODE == 2) { s.ppo = bld4<false>(RS_Q, vo16, drow * 16u); s.ppa = bld2<false>(RS_Q, vo8, angle0 + drow * 8u); }
        }
    };

    // rings of four rows, indexed by the row modulo 4 (compile-time inside a trip of four rows)
    PRow pp[4], pk[4]; GRow gg[4]; RRow rr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { pp[i].xy[q] = v2f{ 0.f, 0.f }; pp[i].pa[q] = 0.f; pk[i].xy[q] = v2f{ 0.f, 0.f }; pk[i].pa[q] = 0.f;
                                      gg[i].cs[q] = v2f{ 1.f, 0.f }; gg[i].gx[q] = v2f{ 0.f, -1.f }; gg[i].a[q] = 0.f; }
        gg[i].f = 0u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) { rr[i].xy[q] = v2f{ 0.f, 0.f }; rr[i].ra[q] = 0.f; }

    float acc = 0.0f; double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    const unsigned mxin = xin ? 0xffffu : 0u;
    // deferred cross-rank finish: the launch's last workgroup has no rows (the host added eight workgroups); its wave 0 is the exchange of iteration k-1
    if (SLAB == 2 && prev.gs != nullptr && prev.count > 0 && blockIdx.x == gridDim.x - 1 && wave == 0 && !work) rc_exchange_prev(aNp, prev, dd);

    if (work) {
        const int t_first = ya - 2, t_last = yb + 1;          // rows of p_{k-1} / cs / flags to take
        // No prologue (a second path into the loop header makes its wait the conservative merge of both): the loop starts DEPTH rows early with empty slots.
        // The rings are indexed by j, the position inside the trip, so any starting row works.
        // The row step itself is BRANCH-FREE: rows outside [t_first, t_last] (lead-in: empty slots; rounding-up: clamped re-reads) and lanes outside the image go
        // through the same arithmetic on zeros / finite garbage that nothing consumes -- pixels outside the image are INACTIVE (flags 0: every term that would read
        // them is multiplied by 0), and a row's p_k is only used by the sums of rows that are themselves in range -- and only the stores (exec mask) and the sums
        // (a 0 / 1 multiplicand, M^-1 = 0) are predicated.  A uniform branch around the arithmetic is a merge point with a phi per ring register.
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) slot[j] = RawT{};
        const int t_begin = t_first - DEPTH;
        for (int t0 = t_begin; t0 <= t_last; t0 += 4) {
            // The iteration's scalars, at the start of the SECOND trip: the first trip issued the loads of the first rows and entered (at most) rows ya-2, ya-1,
            // which need neither alpha nor beta; the partial loads queue up behind those row loads and the additions run while the rows arrive.
            // (DEPTH 1 reaches row ya in its first trip: in front of the loop.)  Wave-uniform: kept in SGPRs.
            if (DEPTH == 1 ? t0 == t_begin : t0 == t_begin + 4) {
                if (SLAB == 2 && prev.gs != nullptr && prev.count > 0) rc_iteration_scalars_x(aNp, prev, dd, alpha, beta);
                else rc_iteration_scalars(aNp, aDp, bNp, prev, alpha, beta, scal_writer);
                alpha = to_sgpr(alpha); beta = to_sgpr(beta);
                if (DMODE == 2) alpha2 = to_sgpr(safe_div<false>(rc_sum(aNpp), rc_sum(aDpp)));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j;
                // ring roles at this step: row t -> index j, t-1 -> j+3, t-2 -> j+2, t-3 -> j+1 (mod 4)
                PRow& p0 = pp[j % 4]; const PRow& p1 = pp[(j + 3) % 4]; const PRow& p2 = pp[(j + 2) % 4];
                GRow& g0 = gg[j % 4]; const GRow& g1 = gg[(j + 3) % 4]; const GRow& g2 = gg[(j + 2) % 4]; const GRow& g3 = gg[(j + 1) % 4];
                PRow& k1 = pk[(j + 3) % 4]; const PRow& k2 = pk[(j + 2) % 4]; const PRow& k3 = pk[(j + 1) % 4];
                RRow& r1 = rr[(j + 1) % 2]; const RRow& r2 = rr[j % 2];
                RawT cur;
                take(cur, slot[j % DEPTH]);                  // (the only place that waits for memory)
                fence_order();                               // the refill stays behind the moves ...
                issue(slot[j % DEPTH], t + DEPTH > t_last ? t_last : t + DEPTH);
                fence_order();                               // ... and in front of the arithmetic
                // ---- row t enters the p_{k-1} / geometry rings (outside the image: inactive)
                {
                    const bool rowok = t >= 0 && t < g.H;
                    const unsigned par = ((unsigned)t * (unsigned)W2) & 1u;
                    const unsigned fl = (cur.f >> (par ? 16u - sh0 : sh0)) & (rowok ? mxin : 0u);
                    p0.xy[0] = uf2(cur.po.x, cur.po.y); p0.xy[1] = uf2(cur.po.z, cur.po.w); p0.pa[0] = uf(cur.pa.x); p0.pa[1] = uf(cur.pa.y);
                    g0.cs[0] = uf2(cur.cs.x, cur.cs.y); g0.cs[1] = uf2(cur.cs.z, cur.cs.w); g0.f = fl;
                    g0.gx[0] = v2f{ g0.cs[0].y, -g0.cs[0].x }; g0.gx[1] = v2f{ g0.cs[1].y, -g0.cs[1].x };
                    g0.a[0] = (float)(fl & 1u); g0.a[1] = (float)((fl >> 8) & 1u);
                }
                // ---- row u = t-1: A p_{k-1}(u) -> r_k(u), p_k(u)
                const int u = t - 1;
                {
                    const float4 m0 = lut[g1.f & 31u], m1 = lut[(g1.f >> 8) & 31u];      // (M^-1 offsets, M^-1 angle, w_fit^2 or 0)
                    const float wfit[2] = { m0.z, m1.z };
                    v2f axy[2]; float av[2];
                    jtjp_pair_xy(p2, p1, p0, g2, g1, g0, g2.a, g1.a, g0.a, wfit, wr2, axy, av);
                    // a ghost row of the slab: the row above the strip's first segment / below its last one (wave-uniform).  Only THAT wave keeps it current: the
                    // rounding-up steps of other segments pass by the same row index with clamped re-reads in their slots
                    const bool ghost_row = SLAB && ((u == ya - 1 && ya == g.row0 && u >= 0) || (u == yb && yb == g.row1 && u < g.H));
                    if (ghost_row) {            // A p_{k-1} of a ghost row: what the exchange delivered (a blocking load, twice per boundary wave and launch)
                        const unsigned row = (unsigned)u * (unsigned)W2;
                        const u32x4 ao = bld4<false>(RS_AI, vo16, row * 16u); const u32x2 aa = bld2<false>(RS_AI, vo8, angle0 + row * 8u);
                        axy[0] = uf2(ao.x, ao.y); axy[1] = uf2(ao.z, ao.w); av[0] = uf(aa.x); av[1] = uf(aa.y);
                    }
                    v2f rxy[2] = { uf2(cur.ro.x, cur.ro.y), uf2(cur.ro.z, cur.ro.w) }; float rq[2] = { uf(cur.ra.x), uf(cur.ra.y) };
                    rxy[0] = fma2(-alpha, axy[0], rxy[0]); rxy[1] = fma2(-alpha, axy[1], rxy[1]);
                    rq[0] = __builtin_fmaf(-alpha, av[0], rq[0]); rq[1] = __builtin_fmaf(-alpha, av[1], rq[1]);
                    const float mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        k1.xy[q] = mo[q] * rxy[q] + beta * p1.xy[q]; k1.pa[q] = ma[q] * rq[q] + beta * p1.pa[q];
                        r1.xy[q] = rxy[q]; r1.ra[q] = rq[q];
                    }
                    const bool mine = u >= ya && u < yb;
                    if (xout && (mine || ghost_row)) {      // this wave's own rows, or a ghost row of the slab (kept current here)
                        const unsigned row = (unsigned)u * (unsigned)W2;
                        bst4<nt_out>(RS_RO, vo16, row * 16u, rxy[0].x, rxy[0].y, rxy[1].x, rxy[1].y); bst2<nt_out>(RS_RO, vo8, angle0 + row * 8u, rq[0], rq[1]);
                        bst4<nt_pout>(RS_Q, vo16, row * 16u, k1.xy[0].x, k1.xy[0].y, k1.xy[1].x, k1.xy[1].y); bst2<nt_pout>(RS_Q, vo8, angle0 + row * 8u, k1.pa[0], k1.pa[1]);
                        if (DMODE != 1 && mine) {
                            float d[4] = { uf(cur.dlo.x), uf(cur.dlo.y), uf(cur.dlo.z), uf(cur.dlo.w) }, da[2] = { uf(cur.dla.x), uf(cur.dla.y) };
                            if (DMODE == 2) {
                                d[0] = __builtin_fmaf(alpha2, uf(cur.ppo.x), d[0]); d[1] = __builtin_fmaf(alpha2, uf(cur.ppo.y), d[1]);
                                d[2] = __builtin_fmaf(alpha2, uf(cur.ppo.z), d[2]); d[3] = __builtin_fmaf(alpha2, uf(cur.ppo.w), d[3]);
                                da[0] = __builtin_fmaf(alpha2, uf(cur.ppa.x), da[0]); da[1] = __builtin_fmaf(alpha2, uf(cur.ppa.y), da[1]);
                            }
                            d[0] = __builtin_fmaf(alpha, p1.xy[0].x, d[0]); d[1] = __builtin_fmaf(alpha, p1.xy[0].y, d[1]);
                            d[2] = __builtin_fmaf(alpha, p1.xy[1].x, d[2]); d[3] = __builtin_fmaf(alpha, p1.xy[1].y, d[3]);
                            da[0] = __builtin_fmaf(alpha, p1.pa[0], da[0]); da[1] = __builtin_fmaf(alpha, p1.pa[1], da[1]);
                            bst4<nt_delta>(RS_D, vo16, row * 16u, d[0], d[1], d[2], d[3]); bst2<nt_delta>(RS_D, vo8, angle0 + row * 8u, da[0], da[1]);
                        }
                    }
                }
                // (phases in sequence: letting the scheduler interleave the two stencils and the two pixels' double sums for ILP costs ~60 registers -- the
                //  difference between one and two waves per SIMD; at two waves per SIMD the other wave fills the issue slots)
                __builtin_amdgcn_sched_barrier(0);
                // ---- row v = t-2: A p_k(v) and the iteration's sums (the wave's own rows and output lanes only: elsewhere the table's entry 0 -- M^-1 = 0 -- and a 0 multiplicand)
                {
                    const int v = t - 2;
                    const bool on = xout && v >= ya && v < yb;
                    const float4 m0 = lut[on ? g2.f & 31u : 0u], m1 = lut[on ? (g2.f >> 8) & 31u : 0u];
                    const float wfit[2] = { m0.z, m1.z }, mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
                    v2f axy[2]; float av[2];
                    jtjp_pair_xy(k3, k2, k1, g3, g2, g1, g3.a, g2.a, g1.a, wfit, wr2, axy, av);
                    if (SLAB && xout && v >= ya && v < yb && (v == g.row0 || v == g.row1 - 1)) {      // the rows of A p_k the neighbouring ranks' ghost rows need
                        if (SLAB == 1) {
                            const unsigned row = (unsigned)v * (unsigned)W2;
                            bst4<false>(RS_AO, vo16, row * 16u, axy[0].x, axy[0].y, axy[1].x, axy[1].y); bst2<false>(RS_AO, vo8, angle0 + row * 8u, av[0], av[1]);
                        } else {            // peer-to-peer, write-through; drained by every wave before the arrival ticket (iter_tail)
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                if (v == (k == 0 ? g.row0 : g.row1 - 1) && dd.peer_r[k]) {
                                    float* d2 = dd.peer_r[k] + dd.peer_off_o[k] + 2 * x0;
                                    st_sys(d2, axy[0].x); st_sys(d2 + 1, axy[0].y); st_sys(d2 + 2, axy[1].x); st_sys(d2 + 3, axy[1].y);
                                    float* d1 = dd.peer_r[k] + dd.peer_off_a[k] + x0;
                                    st_sys(d1, av[0]); st_sys(d1 + 1, av[1]);
                                }
                            }
                        }
                    }
                    const float msum = on ? 1.0f : 0.0f;
                    __builtin_amdgcn_sched_barrier(0);
                    iter_sums_pixel_masked(msum, k2.xy[0].x, k2.xy[0].y, k2.pa[0], axy[0].x, axy[0].y, av[0], r2.xy[0].x, r2.xy[0].y, r2.ra[0], mo[0], ma[0], acc, s0, s1, s2);
                    __builtin_amdgcn_sched_barrier(0);
                    iter_sums_pixel_masked(msum, k2.xy[1].x, k2.xy[1].y, k2.pa[1], axy[1].x, axy[1].y, av[1], r2.xy[1].x, r2.xy[1].y, r2.ra[1], mo[1], ma[1], acc, s0, s1, s2);
                }
                // a row's arithmetic stays inside its step: without the pin the double sums of all four rows of a trip sink behind the fourth row's take (their only
                // consumers are the next sums), with the 14 values each of them reads kept alive until then -- 99 live registers at a trip's first take, 256 at its end
                asm volatile("" : "+v"(acc), "+v"(s0), "+v"(s1), "+v"(s2));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    iter_tail<MARCH_NT, SLAB == 2>(acc, s0, s1, s2, red, redd, aD_out, s12_out, bNp, &dd, fin_tickets, aD_word, bN_word, xslot);
}

}  // namespace

// product configuration (tools/rc_probe.py, profiles/r04): two rows of prefetch, compiled for two workgroups of 4 waves per CU (<= 256 registers; the kernel needs
// 160-200), the grid sized for one.  Depth 1 / 2 / 4 and one, two or three waves per SIMD all run within 2 % of each other: the launch moves its bytes at the
// 5.4-5.7 TB/s this access pattern gets out of the memory system, plus ~7 us of launch ramp and tail.
[[maybe_unused]] constexpr int MARCH_RC_DEPTH = 4;      // (sweep build: the depth of the cache-policy variants)
constexpr int MARCH_RC_OCC = 2;
// Rows of prefetch: four where a wave has many rows (since the arithmetic went onto register pairs the row step is short enough for memory latency to show: 2048^2,
// 35 rows per wave, 63.3 -> 59.5 us per PCG iteration), two where it has few -- the loop starts DEPTH rows early with empty slots, and at 5 rows per wave (2048 x 256)
// four lead-in steps cost 14 %.  tools/rc_depth_by_size.py: 26 rows a tie, 22 / 18 / 9 / 5 rows 1-14 % for two.
[[maybe_unused]] constexpr int MARCH_RC_DEEP_ROWS = 24;
#ifdef THALLO_MARCH_SWEEP
namespace thallo {
int g_march_rc_depth = MARCH_RC_DEPTH;      // rows of prefetch (1, 2, 4)
int g_march_rc_occ = MARCH_RC_OCC;          // register budget: workgroups of 4 waves per CU the kernel is compiled for (1: <= 512 registers, 2: <= 256, 3: <= 168)
int g_march_rc_nt = MARCH_NTM;              // cache-policy mask (iw_march.hpp)
}
#endif

namespace {
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; int armed = 0, used = 0; };
thread_local LaunchEvents g_launch_ev;

template <int SLAB>
int launch_march_rc(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                    const float* r_in, float* r_out, const float* A_in, float* A_out, const float* p_in, float* p_out, float* delta, int mode,
                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp, const int* irregular, thallo_dist_t d,
                    float* aD_out, double* s12_out, unsigned* fin_tickets, float* aD_word, float* bN_word, int xslot, hipStream_t stream, PrevSums prev)
{
    const int R = march_pick_rows(W, row1 - row0);
    if (R <= 0) return -(int)hipErrorNotSupported;
    const MarchGeo g = make_march_geo(W, H, row0, row1, R);
    const int grid = (g.total + 7) / 8 * 8 + ((SLAB == 2 && prev.gs != nullptr) ? 8 : 0);      // (deferred cross-rank finish: eight workgroups without rows; wave 0 of the last one is the exchange)
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const int dmode = (mode >> 1) & 3;
    const float wf2 = w_fit * w_fit, wr2 = w_reg * w_reg;
    // (g_launch_ev armed -- the library's kernel timer sampling this launch, bench.py's roofline figure -- : the launch carries two events that take the kernel's OWN
    //  begin / end timestamps, what rocprofv3 reports as its duration; events recorded around the launch also contain the dispatch gap in front of it)
    const bool ev_on = g_launch_ev.armed != 0;
    if (ev_on) g_launch_ev.used = 1;
#define RC_LAUNCH(DM, DP, OCC, NTM) do { if (ev_on) hipExtLaunchKernelGGL((k_iter_march_rc<DM, DP, NTM, OCC, SLAB>), dim3(grid), dim3(MARCH_NT), 0, stream, g_launch_ev.start, g_launch_ev.stop, 0, g, cs, flags, wf2, wr2, \
        r_in, r_out, A_in, A_out, p_in, p_out, delta, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, irregular, d, fin_tickets, aD_word, bN_word, xslot, prev); \
    else hipLaunchKernelGGL((k_iter_march_rc<DM, DP, NTM, OCC, SLAB>), dim3(grid), dim3(MARCH_NT), 0, stream, g, cs, flags, wf2, wr2, \
        r_in, r_out, A_in, A_out, p_in, p_out, delta, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, irregular, d, fin_tickets, aD_word, bN_word, xslot, prev); } while (0)
#ifdef THALLO_MARCH_SWEEP      // tools/rc_probe.py: prefetch depth x register budget at the product's cache policy, and the cache-policy masks at the product's depth / budget
#define RC_BY_DEPTH(DM) do { if constexpr (SLAB != 0) RC_LAUNCH(DM, MARCH_RC_DEPTH, MARCH_RC_OCC, MARCH_NTM); else { const int dp = g_march_rc_depth, oc = g_march_rc_occ, nt = g_march_rc_nt; \
        if (nt != MARCH_NTM) { if (nt == 0) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 0); else if (nt == 1) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 1); else if (nt == 4) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 4); else if (nt == 7) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 7); \
                                 else if (nt == 13) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 13); else if (nt == 21) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 21); else if (nt == 37) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 37); else if (nt == 63) RC_LAUNCH(DM, MARCH_RC_DEPTH, 2, 63); else return -(int)hipErrorInvalidValue; } \
        else if (oc == 1) { if (dp == 4) RC_LAUNCH(DM, 4, 1, MARCH_NTM); else RC_LAUNCH(DM, 2, 1, MARCH_NTM); } \
        else if (oc == 3) { if (dp == 1) RC_LAUNCH(DM, 1, 3, MARCH_NTM); else RC_LAUNCH(DM, 2, 3, MARCH_NTM); } \
        else { if (dp == 1) RC_LAUNCH(DM, 1, 2, MARCH_NTM); else if (dp == 4) RC_LAUNCH(DM, 4, 2, MARCH_NTM); else RC_LAUNCH(DM, 2, 2, MARCH_NTM); } } } while (0)
#else
#define RC_BY_DEPTH(DM) do { if (R >= MARCH_RC_DEEP_ROWS) RC_LAUNCH(DM, 4, MARCH_RC_OCC, MARCH_NTM); else RC_LAUNCH(DM, 2, MARCH_RC_OCC, MARCH_NTM); } while (0)
#endif
    if (dmode == 1) RC_BY_DEPTH(1); else if (dmode == 2) RC_BY_DEPTH(2); else RC_BY_DEPTH(0);
#undef RC_BY_DEPTH
#undef RC_LAUNCH
    int e = check_launch(); return e ? e : grid;
}

// what every entry checks: the shape (a whole image, or a slab with at most one ghost row towards each neighbour), the planes, the sums this iteration starts from
int rc_check(int W, int H, int row0, int row1, const void* cs, const void* flags, const void* r_in, const void* r_out, const void* A_in, const void* A_out,
             const void* p_in, const void* p_out, const void* delta, int mode, thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, const void* aD_out, const void* s12_out)
{
    if ((W & 1) || W < 2 || H < 1 || row0 < 0 || row1 > H || row0 >= row1 || row0 > 1 || H - row1 > 1) return -(int)hipErrorInvalidValue;
    if (mode & 1) return -(int)hipErrorInvalidValue;          // (the first iteration of a GN step has no A p_{k-1}: the stored-plane kernel runs it)
    if (!cs || !flags || !r_in || !r_out || !p_in || !p_out || !delta || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if ((row0 > 0 || row1 < H) && (!A_in || !A_out)) return -(int)hipErrorInvalidValue;          // slabs: the exchanged rows of A p
    if (aNp.count < 1 || !aNp.partials) return -(int)hipErrorInvalidValue;
    if (((mode >> 1) & 3) == 2 && (aNpp.count < 1 || aDpp.count < 1 || !aNpp.partials || !aDpp.partials)) return -(int)hipErrorInvalidValue;
    if (12.0 * (double)W * (double)H >= 4294967296.0) return -(int)hipErrorNotSupported;       // a vector must fit one buffer descriptor
    return 0;
}
}  // namespace

extern "C" {

int thallo_hip_iw_pcg_iter_march_rc(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                    const int* irregular, float* aD_out, double* s12_out,
                                    unsigned* fin_tickets, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (int e = rc_check(W, H, row0, row1, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aNpp, aDpp, aD_out, s12_out)) return e;
    if (aDp.count < 1 || bNp.count < 1 || !aDp.partials || !bNp.partials) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    const PrevSums none = { nullptr, nullptr, 0, nullptr, nullptr };
    if (row0 == 0 && row1 == H)
        return launch_march_rc<0>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, nullptr, nullptr, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, irregular, thallo_dist_t{},
                                  aD_out, s12_out, fin_tickets, aD_word, bN_word, 0, (hipStream_t)stream, none);
    return launch_march_rc<1>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, irregular, thallo_dist_t{},
                              aD_out, s12_out, fin_tickets, aD_word, bN_word, 0, (hipStream_t)stream, none);
}

int thallo_hip_iw_pcg_iter_march_rc_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                             const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                             thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, thallo_prev_t prev,
                                             const int* irregular, float* aD_out, double* s12_out, thallo_stream_t stream)
{
    if (int e = rc_check(W, H, row0, row1, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aNpp, aDpp, aD_out, s12_out)) return e;
    if (prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word || !prev.betaN_word ||
        prev.s12_partials == s12_out) return -(int)hipErrorInvalidValue;
    const thallo_sum_t none = { nullptr, 0 };
    const PrevSums ps = { prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word };
    if (row0 == 0 && row1 == H)
        return launch_march_rc<0>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, nullptr, nullptr, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp, irregular, thallo_dist_t{},
                                  aD_out, s12_out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, ps);
    return launch_march_rc<1>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp, irregular, thallo_dist_t{},
                              aD_out, s12_out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, ps);
}

int thallo_hip_iw_pcg_iter_march_rc_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                         const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                         thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                         const int* irregular, thallo_dist_t d, float* aD_out, double* s12_out,
                                         unsigned* fin_tickets, int slot0, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (int e = rc_check(W, H, row0, row1, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aNpp, aDpp, aD_out, s12_out)) return e;
    if (!Ap_in || !Ap_out || aDp.count < 1 || bNp.count < 1 || !aDp.partials || !bNp.partials || d.world < 1 || d.world > THALLO_DIST_MAX_WORLD) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    if (fin_tickets && (slot0 < 0 || !d.mail || !d.ctl || 7 * d.world > 64 || bNp.count != 1)) return -(int)hipErrorInvalidValue;
    for (int k = 0; k < 2; ++k) if (d.peer_r[k] && ((d.peer_off_o[k] | d.peer_off_a[k]) & 1)) return -(int)hipErrorInvalidValue;
    return launch_march_rc<2>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, irregular, d,
                              aD_out, s12_out, fin_tickets, aD_word, bN_word, slot0, (hipStream_t)stream, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
}

/* The deferred cross-rank finish makes every working wave wait for granules that wave 0 of the launch's LAST workgroup publishes: the whole grid (its eight extra
 * workgroups included) must be resident at once -- the kernel is built for MARCH_RC_OCC workgroups per CU, while march_pick_rows may size a wide slab's grid for up to
 * four.  1: it is; 0: run the launch-end exchange (thallo_hip_iw_pcg_iter_march_rc_dist), which has no such requirement. */
int thallo_hip_iw_march_rc_deferred_fits(int W, int rows)
{
    if (W < 2 || (W & 1) || rows < 1) return 0;
    const int R = march_pick_rows(W, rows);
    if (R <= 0) return 0;
    const MarchGeo g = make_march_geo(W, rows, 0, rows, R);
    const long grid = (g.total + 7) / 8 * 8 + 8;
    return grid <= march_cap(MARCH_RC_OCC) && grid <= THALLO_MAX_PARTIALS ? 1 : 0;
}

/* ... with the deferred cross-rank finish (thallo_hip.h): `prev` = iteration k-1's partials of THIS rank, its two words, and where the exchange happens */
int thallo_hip_iw_pcg_iter_march_rc_dist_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                                  const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                                                  thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, thallo_prev_t prev, int prev_slot0, unsigned long long* gs,
                                                  const int* irregular, thallo_dist_t d, float* aD_out, double* s12_out, thallo_stream_t stream)
{
    if (int e = rc_check(W, H, row0, row1, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aNpp, aDpp, aD_out, s12_out)) return e;
    if (!Ap_in || !Ap_out || d.world < 1 || d.world > THALLO_DIST_MAX_WORLD || !d.mail || !d.ctl || 7 * d.world > 64 || prev_slot0 < 0 || !gs || aNp.count != 1) return -(int)hipErrorInvalidValue;
    if (!thallo_hip_iw_march_rc_deferred_fits(W, row1 - row0)) return -(int)hipErrorNotSupported;      // (its waits need every workgroup resident)
    if (prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word || !prev.betaN_word ||
        prev.s12_partials == s12_out) return -(int)hipErrorInvalidValue;
    for (int k = 0; k < 2; ++k) if (d.peer_r[k] && ((d.peer_off_o[k] | d.peer_off_a[k]) & 1)) return -(int)hipErrorInvalidValue;
    const thallo_sum_t none = { nullptr, 0 };
    const PrevSums ps = { prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word, gs, prev_slot0 };
    return launch_march_rc<2>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp, irregular, d,
                              aD_out, s12_out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, ps);
}
int thallo_hip_iw_dist_finish_deferred(thallo_prev_t prev, int prev_slot0, thallo_sum_t aNp, thallo_dist_t d, unsigned long long* gs, thallo_stream_t stream)
{
    if (prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word || !prev.betaN_word || prev_slot0 < 0 || !gs ||
        aNp.count != 1 || !aNp.partials || d.world < 1 || d.world > THALLO_DIST_MAX_WORLD || !d.mail || !d.ctl || 7 * d.world > 64) return -(int)hipErrorInvalidValue;
    const PrevSums ps = { prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word, gs, prev_slot0 };
    hipLaunchKernelGGL(k_rc_dist_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, aNp, ps, d);
    return check_launch();
}

void thallo_hip_march_rc_debug_set(int what, int value)
{
#ifdef THALLO_MARCH_SWEEP
    if (what == 0) g_march_rc_depth = value;
    if (what == 1) g_march_rc_occ = value;
    if (what == 2) g_march_rc_nt = value;
#else
    (void)what; (void)value;
#endif
}

void thallo_hip_launch_events_arm(void* start, void* stop) { g_launch_ev.start = (hipEvent_t)start; g_launch_ev.stop = (hipEvent_t)stop; g_launch_ev.armed = start && stop ? 1 : 0; g_launch_ev.used = 0; }
int thallo_hip_launch_events_take(void) { const int u = g_launch_ev.used; g_launch_ev = LaunchEvents(); return u; }

}  // extern "C"
