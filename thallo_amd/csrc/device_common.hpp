// device_common.hpp -- wave64 / workgroup primitives shared by every kernel (gfx950 only).
//
// Replaces the reference's warp-32 PTX primitives (API/src/cuda_util.t:287-449, util.t:40-50):
//   reference: 5-step shfl.down, then ONE red.global.add.f32 per warp into a single word
//              -> order-nondeterministic, 131k same-address atomics per reduction at 2048^2.
//   here:      6-step wave64 butterfly, LDS across waves, ONE plain store per workgroup into
//              partial[blockIdx]; the consumer kernel re-sums the <=1024 partials in a fixed
//              order.  No atomics, no memsets, bitwise reproducible for a fixed launch shape
//              (MI355X memory-side float atomics serialise at ~12 ns per same-address add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/thallo_hip.h"

#define THALLO_WAVE 64
#define THALLO_MAX_PARTIALS 1024   // upper bound on producer grid size for reduced quantities

namespace thallo {

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not -- as __syncthreads() does with its
// s_waitcnt vmcnt(0) -- for every global load and STORE it has in flight.  In a kernel's reduction tail that drain is 1-2 us during which
// nothing else happens (the stores the wave issued a moment ago travel to L2 while the wave could already add its partials up); in the
// tile kernel's loop it would serialise the prefetch of the next tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The wave64 butterfly (v += value of lane ^ 32, ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: every lane ends with the same total, fixed association order) WITHOUT
// the LDS crossbar: __shfl_xor is a ds_bpermute_b32 per 32-bit word and step (address arithmetic + ~100 cycles of LDS round trip, six of them
// in a dependent chain = ~0.5 us per reduction point, which is 5 % of a whole PCG iteration on the small working sets).  gfx950 has the lane
// exchanges as plain VALU operations: v_permlane32_swap / v_permlane16_swap for the two cross-row steps, DPP row_ror:8, row_shl:4 + row_shr:4
// (bank-masked halves), quad_perm for the steps inside a row of 16.  Same partners, same order, IEEE addition commutes: bit-identical to the
// __shfl_xor form (tests/test_gpu_parity.py pins the reduce primitive against a numpy restatement).  All 64 lanes must be active.
template <int M> __device__ __forceinline__ unsigned lane_xor_u32(unsigned v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "in-row partners only");
    const int x = (int)v;
    if constexpr (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm:[1,0,3,2]
    else if constexpr (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    else if constexpr (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);  // row_ror:8
    else {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);                            // row_shl:4 into banks 0, 2 (lane i <- i + 4)
        return (unsigned)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);                         // row_shr:4 into banks 1, 3 (lane i <- i - 4)
    }
}
// own + partner across the two halves of the wave (M = 32) or across neighbouring rows of 16 (M = 16).  After the swap of two copies of v:
// a = {low, low} / {row0, row0, row2, row2}, b = {high, high} / {row1, row1, row3, row3}; a + b is own + partner in the first of each pair and
// partner + own in the second.  The second copy is made opaque to the optimiser: it folds a swap of two IDENTICAL values into a no-op
// (true for wave-uniform values only; seen as v_add v, v, v in the ISA).
template <int M> __device__ __forceinline__ void swap_halves_u32(unsigned v, unsigned& a, unsigned& b)
{
    static_assert(M == 16 || M == 32, "cross-row partners only");
    unsigned w = v;
    asm volatile("" : "+v"(w));
    if constexpr (M == 32) { const auto r = __builtin_amdgcn_permlane32_swap(v, w, false, false); a = r[0]; b = r[1]; }
    else { const auto r = __builtin_amdgcn_permlane16_swap(v, w, false, false); a = r[0]; b = r[1]; }
}
__device__ __forceinline__ float wave_sum_all(float v)
{
    unsigned a, b;
    swap_halves_u32<32>(__builtin_bit_cast(unsigned, v), a, b); v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    swap_halves_u32<16>(__builtin_bit_cast(unsigned, v), a, b); v = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    v += __builtin_bit_cast(float, lane_xor_u32<8>(__builtin_bit_cast(unsigned, v)));
    v += __builtin_bit_cast(float, lane_xor_u32<4>(__builtin_bit_cast(unsigned, v)));
    v += __builtin_bit_cast(float, lane_xor_u32<2>(__builtin_bit_cast(unsigned, v)));
    v += __builtin_bit_cast(float, lane_xor_u32<1>(__builtin_bit_cast(unsigned, v)));
    return v;
}

// Sum of `nb` per-workgroup partials written by the previous kernel.  Every wave computes it
// itself (<= 16 L2-resident dwords per lane), so no barrier is needed and every thread in the
// grid holds the bit-identical value.
__device__ __forceinline__ float sum_partials(const float* __restrict__ part, int nb)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    if (nb == 1) return part[0];                           // already reduced (finish_sum / cross-rank exchange)
    // all <= 16 loads of a lane are issued before the first add (independent), then added in index order: the same value as
    // the rolled loop `for (i = lane; i < nb; i += 64) s += part[i]` (missing entries add +0.0f)
    float v[THALLO_MAX_PARTIALS / THALLO_WAVE];
#pragma unroll
    for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) {
        const int i = lane + k * THALLO_WAVE;
        v[k] = i < nb ? part[i] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) s += v[k];
    return wave_sum_all(s);
}

// Workgroup reduction -> partial[blockIdx.x].  `red` is >= blockDim.x/64 floats of LDS.
// Must be called by every thread of the workgroup.
template <int NQ>
__device__ __forceinline__ void block_store_partials(float (&v)[NQ], float* __restrict__ const (&out)[NQ], float* red)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    const int wave = threadIdx.x / THALLO_WAVE;
    const int nw   = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float s = wave_sum_all(v[q]);
        if (lane == 0) red[q * 16 + wave] = s;
    }
    lds_barrier();
    if (threadIdx.x < NQ) {
        float s = 0.0f;
        for (int w = 0; w < nw; ++w) s += red[threadIdx.x * 16 + w];
        out[threadIdx.x][blockIdx.x] = s;
    }
}

__device__ __forceinline__ void block_store_partial(float v, float* __restrict__ out, float* red)
{
    float vv[1] = { v };
    float* __restrict__ const oo[1] = { out };
    block_store_partials<1>(vv, oo, red);
}

// --- the three double sums of the single-reduction PCG form (DESIGN.md "one reduction point per iteration"):
//   N = sum r.M^-1.r,  S1 = sum r.M^-1.Ap,  S2 = sum Ap.M^-1.Ap   accumulated from exact products of the float data, so that
//   betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 (= r_{k+1}.M^-1 r_{k+1}) survives its cancellation
struct Sums3 {
    double n = 0.0, s1 = 0.0, s2 = 0.0;
    __device__ __forceinline__ void add(float m, float r, float a)
    {
        const double dm = m, dr = r, da = a;
        n += dm * (dr * dr); s1 += dm * (dr * da); s2 += dm * (da * da);
    }
};
// the double butterfly: the two 32-bit halves travel separately (see wave_sum_all)
template <int M> __device__ __forceinline__ double lane_xor_f64(double v)
{
    return __hiloint2double((int)lane_xor_u32<M>((unsigned)__double2hiint(v)), (int)lane_xor_u32<M>((unsigned)__double2loint(v)));
}
template <int M> __device__ __forceinline__ double swap_add_f64(double v)
{
    unsigned la, lb, ha, hb;
    swap_halves_u32<M>((unsigned)__double2loint(v), la, lb); swap_halves_u32<M>((unsigned)__double2hiint(v), ha, hb);
    return __hiloint2double((int)ha, (int)la) + __hiloint2double((int)hb, (int)lb);
}
__device__ __forceinline__ double wave_sum_all_f64(double v)
{
    v = swap_add_f64<32>(v); v = swap_add_f64<16>(v);
    v += lane_xor_f64<8>(v); v += lane_xor_f64<4>(v); v += lane_xor_f64<2>(v); v += lane_xor_f64<1>(v);
    return v;
}
// one {N, S1, S2} triple per workgroup into out[3*blockIdx.x ..]; every thread calls it; redd >= 3 * (blockDim.x / 64) doubles of LDS
__device__ __forceinline__ void block_store_sums3(const Sums3& s, double* __restrict__ out, double* redd)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE;
    const int nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
    const double a = wave_sum_all_f64(s.n), b = wave_sum_all_f64(s.s1), c = wave_sum_all_f64(s.s2);
    if (lane == 0) { redd[3 * wave] = a; redd[3 * wave + 1] = b; redd[3 * wave + 2] = c; }
    lds_barrier();
    if (threadIdx.x == 0) {
        double x = 0.0, y = 0.0, z = 0.0;
        for (int w = 0; w < nw; ++w) { x += redd[3 * w]; y += redd[3 * w + 1]; z += redd[3 * w + 2]; }
        out[3 * blockIdx.x] = x; out[3 * blockIdx.x + 1] = y; out[3 * blockIdx.x + 2] = z;
    }
}

// safeDivideIfNotLM (gauss_newton.t:226-234): GN guards the zero denominator, LM divides blindly.
template <bool LM>
__device__ __forceinline__ float safe_div(float num, float den)
{
    if (LM) return num / den;
    return den != 0.0f ? num / den : 0.0f;
}

// The sums of one PCG iteration as the LAST workgroup of the producing launch reads them back (one wave, every lane gets the totals): the slots were
// written by other workgroups of the same launch (write-through, agent scope), so they are read with agent-scope loads.  Everything that can be in
// flight together is issued before the first addition -- the <= 16 float slots of a lane, its first four {N, S1, S2} slots and alphaN -- because this
// read-back sits on the critical path between two dependent launches; the additions keep k_scalars_finish's order (absent slots add +0.0).
struct IterationSums { float ad, an; double n, s1, s2; };
__device__ __forceinline__ IterationSums load_iteration_sums(const float* aD_out, const double* s3_out, int nb, thallo_sum_t aN)
{
    typedef unsigned long long u64_t;
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    const u64_t* sp = reinterpret_cast<const u64_t*>(s3_out);
    float t[THALLO_MAX_PARTIALS / THALLO_WAVE];
#pragma unroll
    for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) {
        const int i = lane + k * THALLO_WAVE;
        t[k] = i < nb ? __hip_atomic_load(aD_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
    }
    auto load4 = [&](int i0, double (&v)[4][3]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * THALLO_WAVE;
            const bool ok = i < nb;
#pragma unroll
            for (int q = 0; q < 3; ++q) v[u][q] = ok ? __longlong_as_double((long long)__hip_atomic_load(sp + 3 * i + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
        }
    };
    double v[4][3];
    load4(lane, v);
    IterationSums S;
    S.an = sum_partials(aN.partials, aN.count);
    float ad = 0.0f;
#pragma unroll
    for (int k = 0; k < THALLO_MAX_PARTIALS / THALLO_WAVE; ++k) ad += t[k];
    S.ad = wave_sum_all(ad);                                                         // == sum_partials(aD_out, nb) for nb > 1
    double n = 0.0, a1 = 0.0, b1 = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { n += v[u][0]; a1 += v[u][1]; b1 += v[u][2]; }
    for (int i0 = lane + 4 * THALLO_WAVE; i0 < nb; i0 += 4 * THALLO_WAVE) {          // more than 256 slots: four per lane and round, 12 loads in flight
        load4(i0, v);
#pragma unroll
        for (int u = 0; u < 4; ++u) { n += v[u][0]; a1 += v[u][1]; b1 += v[u][2]; }
    }
    S.n = wave_sum_all_f64(n); S.s1 = wave_sum_all_f64(a1); S.s2 = wave_sum_all_f64(b1);
    return S;
}

// The same read-back by ALL waves of the last workgroup: the 1 + NQD quantities (alphaD in float with alphaN beside it; N, S1, S2 from s3; with NQD = 6 also U, T1, T2 from
// q3) are dealt over the waves, a wave issues every load of its (at most two) quantities before the first addition -- one round of fabric latency for the whole read-back
// where the one-wave form takes a round per 256 slots and quantity group (bundle adjustment, 943 slots: the LM point launch from 1.33 to 1.05 times the camera launch of the same run, profiles/r04/ba_lm_loops.json) -- and adds them in
// load_iteration_sums' order (lane l: slots l, l + 64, ... ascending from zero, then the butterfly): bit-identical totals.  Must be reached by every wave of the workgroup
// (two barriers inside); the totals come back in every lane of every wave.  red >= 16 floats (words 13, 14 used), redd >= NQD doubles.
// PLAIN: the slots were written by an EARLIER launch (the finish deferred into the next launch, pcg_kernels.hip k_pcg_update_fin): ordinary loads, which every workgroup of
// that launch can take from its L2.
template <int NQD, bool PLAIN = false>
__device__ __forceinline__ void last_workgroup_totals(const float* aD_out, const double* s3_out, const double* q3_out, int nb, thallo_sum_t aN, float* red, double* redd,
                                                      float& ad_out, float& an_out, double (&tot)[NQD])
{
    typedef unsigned long long u64_t;
    constexpr int PER = THALLO_MAX_PARTIALS / THALLO_WAVE, NT = 1 + NQD;
    const int lane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE;
    const int nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
    for (int q0 = wave; q0 < NT; q0 += 2 * nw) {
        const int qs[2] = { q0, q0 + nw };
        float t[PER]; double v[2][PER];
        if (q0 == 0) {
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int i = lane + k * THALLO_WAVE; t[k] = i < nb ? (PLAIN ? aD_out[i] : __hip_atomic_load(aD_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0f; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = qs[h];
            if (q < 1 || q >= NT) continue;
            const u64_t* sp = reinterpret_cast<const u64_t*>(q <= 3 ? s3_out : q3_out) + (q - 1) % 3;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = lane + k * THALLO_WAVE;
                v[h][k] = i < nb ? (PLAIN ? __longlong_as_double((long long)sp[3 * i]) : __longlong_as_double((long long)__hip_atomic_load(sp + 3 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) : 0.0;
            }
        }
        if (q0 == 0) {
            const float an = sum_partials(aN.partials, aN.count);
            float ad = 0.0f;
#pragma unroll
            for (int k = 0; k < PER; ++k) ad += t[k];
            ad = wave_sum_all(ad);
            if (lane == 0) { red[13] = ad; red[14] = an; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = qs[h];
            if (q < 1 || q >= NT) continue;
            double x = 0.0;
#pragma unroll
            for (int k = 0; k < PER; ++k) x += v[h][k];
            x = wave_sum_all_f64(x);
            if (lane == 0) redd[q - 1] = x;
        }
    }
    lds_barrier();
    ad_out = red[13]; an_out = red[14];
#pragma unroll
    for (int q = 0; q < NQD; ++q) tot[q] = redd[q];
}

// End of a single-reduction applyJTJ (alphaD partial + {N, S1, S2}), called by EVERY thread of the workgroup with its private terms: one
// partial set per workgroup into slot blk_off + blockIdx.x.  With fin.tickets the launch's last workgroup to arrive also finishes the two
// scalars of the PCG iteration over all fin.nb_total slots -- alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 -- in exactly
// k_scalars_finish's order (pcg_kernels.hip), so the separate one-wave launch (5-7 us per iteration on the small working sets) is not needed.
// Tickets: two levels (workgroup b -> group b % 32, each group word on its own 64-byte line: same-address atomics serialise at ~12 ns each);
// they are zero again when the kernel ends.  red >= 16 floats, redd >= 3 * (blockDim.x / 64) doubles of LDS.
struct FinArgs {
    thallo_sum_t alphaN;          // alphaN_k (a one-word sum or partials)
    unsigned* tickets;            // THALLO_HIP_FIN_TICKET_WORDS zeroed words, or NULL: partials only
    float *aD_word, *bN_word;
    int blk_off, nb_total;        // slot offset of this launch's workgroups; slots to add up (>= blk_off + gridDim.x when several launches share the slots)
};
__device__ __forceinline__ void block_finish_sums(float acc, const Sums3& sm, float* __restrict__ aD_out, double* __restrict__ s3_out, const FinArgs& fin,
                                                  float* red, double* redd)
{
    typedef unsigned long long u64_t;
    const int lane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE;
    const int nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
    const float wa = wave_sum_all(acc);
    const double w0 = wave_sum_all_f64(sm.n), w1 = wave_sum_all_f64(sm.s1), w2 = wave_sum_all_f64(sm.s2);
    if (lane == 0) { red[wave] = wa; redd[3 * wave] = w0; redd[3 * wave + 1] = w1; redd[3 * wave + 2] = w2; }
    lds_barrier();
    const int slot = fin.blk_off + blockIdx.x;
    if (threadIdx.x == 0) {
        float a = 0.0f; double b0 = 0.0, b1 = 0.0, b2 = 0.0;
        for (int w = 0; w < nw; ++w) { a += red[w]; b0 += redd[3 * w]; b1 += redd[3 * w + 1]; b2 += redd[3 * w + 2]; }
        if (!fin.tickets) { aD_out[slot] = a; s3_out[3 * slot] = b0; s3_out[3 * slot + 1] = b1; s3_out[3 * slot + 2] = b2; }
        else {
            __hip_atomic_store(aD_out + slot, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u64_t* sp = reinterpret_cast<u64_t*>(s3_out) + 3 * slot;
            __hip_atomic_store(sp, (u64_t)__double_as_longlong(b0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sp + 1, (u64_t)__double_as_longlong(b1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sp + 2, (u64_t)__double_as_longlong(b2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned grp = blockIdx.x % 32, members = (gridDim.x - grp + 31) / 32, groups = gridDim.x < 32 ? gridDim.x : 32;
            unsigned* sub = fin.tickets + 16 + 16 * grp;
            bool last = false;
            if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = __hip_atomic_fetch_add(fin.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
            }
            red[15] = last ? 1.0f : 0.0f;
        }
    }
    if (!fin.tickets) return;
    lds_barrier();
    if (red[15] == 0.0f) return;                                   // (the same for every thread of the workgroup)
    float ad, an; double t3[3];
    last_workgroup_totals<3>(aD_out, s3_out, nullptr, fin.nb_total, fin.alphaN, red, redd, ad, an, t3);
    if (threadIdx.x != 0) return;
    const double n = t3[0], a1 = t3[1], b1 = t3[2];
    __hip_atomic_store(fin.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float al = safe_div<false>(an, ad);
    double bn = n - 2.0 * (double)al * a1 + (double)al * (double)al * b1;
    if (!(bn > 0.0)) bn = 0.0;
    fin.aD_word[0] = ad; fin.bN_word[0] = (float)bn;
}

// ---- Levenberg-Marquardt iteration in ONE launch (round 3; shape_from_shading).  The reference's PCGStep2 forms q_{k+1} = 0.5 delta_{k+1} . (r_{k+1} + b) AFTER the
// vector update (gauss_newton.t:801-843, 965); with delta_{k+1} = delta_k + alpha p_k and r_{k+1} = r_k - alpha A p_k that is
//   q_{k+1} = 0.5 [ U + alpha (T1 - T2) - alpha^2 alphaD ],   U = delta_k.(r_k + b),  T1 = p_k.(r_k + b),  T2 = delta_k.A p_k
// -- three more sums over what a kernel that carries the vector update already holds (delta_k, r_k, p_k, A p_k) plus b, taken in double like N, S1, S2.  The launch's
// last workgroup then has alpha_k, betaN_k AND q_{k+1}, applies the zeta test (k_lm_zeta's rule, :1666-1686) and sets the gate.
struct SumsQ {
    double u = 0.0, t1 = 0.0, t2 = 0.0;
    __device__ __forceinline__ void add(float d, float r, float b, float p, float a)
    {
        const double rb = (double)r + (double)b;
        u += (double)d * rb; t1 += (double)p * rb; t2 += (double)d * (double)a;
    }
};
struct LmFin {
    const float* b;                         // b == NULL: not an LM launch
    double* q3_out;                         // {U, T1, T2} per workgroup slot (3 * THALLO_MAX_PARTIALS doubles)
    float* state;                           // lm state words: [0] Q0, [1] gate, [2] iterations done at the stop
    int k; float q_tol;
    int q_in = 0, q_out = 0;                // words of `state` the zeta test reads Q0 from / leaves Q1 in (0, 0: one word; the loop whose finishes alternate between launches
                                            // of different kinds -- thallo_hip_pcg_update_lm_fin -- uses words 0 and 6 by the iteration's parity, so that a launch never reads
                                            // the word another of its workgroups writes)
};
__device__ __forceinline__ void load_sums3_wave(const double* s3, int nb, double& a, double& b, double& c)
{   // one wave; four slots per lane and round with all twelve loads in flight (the read-back sits between two dependent launches), then the butterfly
    typedef unsigned long long u64_t;
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    const u64_t* sp = reinterpret_cast<const u64_t*>(s3);
    double x = 0.0, y = 0.0, z = 0.0;
    for (int i0 = lane; i0 < nb; i0 += 4 * THALLO_WAVE) {
        double v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * THALLO_WAVE;
            const bool ok = i < nb;
#pragma unroll
            for (int q = 0; q < 3; ++q) v[u][q] = ok ? __longlong_as_double((long long)__hip_atomic_load(sp + 3 * i + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { x += v[u][0]; y += v[u][1]; z += v[u][2]; }
    }
    a = wave_sum_all_f64(x); b = wave_sum_all_f64(y); c = wave_sum_all_f64(z);
}
// block_finish_sums for an LM launch: also the {U, T1, T2} slot, and the last workgroup finishes alphaD_k, betaN_k, q_{k+1} and the zeta test.
// red >= 16 floats, redd >= 6 * (blockDim.x / 64) doubles of LDS.  fin.tickets must be set.
__device__ __forceinline__ void block_finish_sums_lm(float acc, const Sums3& sm, const SumsQ& sq, float* __restrict__ aD_out, double* __restrict__ s3_out, const FinArgs& fin,
                                                     const LmFin& lm, float* red, double* redd)
{
    typedef unsigned long long u64_t;
    const int lane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE;
    const int nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
    const float wa = wave_sum_all(acc);
    const double w0 = wave_sum_all_f64(sm.n), w1 = wave_sum_all_f64(sm.s1), w2 = wave_sum_all_f64(sm.s2);
    const double w3 = wave_sum_all_f64(sq.u), w4 = wave_sum_all_f64(sq.t1), w5 = wave_sum_all_f64(sq.t2);
    if (lane == 0) { red[wave] = wa; double* d = redd + 6 * wave; d[0] = w0; d[1] = w1; d[2] = w2; d[3] = w3; d[4] = w4; d[5] = w5; }
    lds_barrier();
    const int slot = fin.blk_off + blockIdx.x;
    if (threadIdx.x == 0) {
        float a = 0.0f; double b[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        for (int w = 0; w < nw; ++w) { a += red[w]; for (int q = 0; q < 6; ++q) b[q] += redd[6 * w + q]; }
        if (!fin.tickets) {      // partials only (a row slab: the cross-rank exchange finishes the scalars and applies the zeta test)
            aD_out[slot] = a;
            for (int q = 0; q < 3; ++q) { s3_out[3 * slot + q] = b[q]; lm.q3_out[3 * slot + q] = b[3 + q]; }
            red[15] = 0.0f;
        } else {
        __hip_atomic_store(aD_out + slot, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64_t* sp = reinterpret_cast<u64_t*>(s3_out) + 3 * slot; u64_t* qp = reinterpret_cast<u64_t*>(lm.q3_out) + 3 * slot;
        for (int q = 0; q < 3; ++q) {
            __hip_atomic_store(sp + q, (u64_t)__double_as_longlong(b[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(qp + q, (u64_t)__double_as_longlong(b[3 + q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned grp = blockIdx.x % 32, members = (gridDim.x - grp + 31) / 32, groups = gridDim.x < 32 ? gridDim.x : 32;
        unsigned* sub = fin.tickets + 16 + 16 * grp;
        bool last = false;
        if (__hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(fin.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1;
        }
        red[15] = last ? 1.0f : 0.0f;
        }
    }
    if (!fin.tickets) return;
    lds_barrier();
    if (red[15] == 0.0f) return;                                   // (the same for every thread of the workgroup)
    struct { float ad, an; double n, s1, s2; } S; double t6[6];
    last_workgroup_totals<6>(aD_out, s3_out, lm.q3_out, fin.nb_total, fin.alphaN, red, redd, S.ad, S.an, t6);
    if (wave != 0) return;
    S.n = t6[0]; S.s1 = t6[1]; S.s2 = t6[2];
    const double U = t6[3], T1 = t6[4], T2 = t6[5];
    if (lane == 0) __hip_atomic_store(fin.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float al = safe_div<true>(S.an, S.ad);                 // LM divides blindly (gauss_newton.t:226-234)
    double bn = S.n - 2.0 * (double)al * S.s1 + (double)al * (double)al * S.s2;
    if (!(bn > 0.0)) bn = 0.0;
    const float Q1 = (float)(0.5 * (U + (double)al * (T1 - T2) - (double)al * (double)al * (double)S.ad));
    if (lane == 0) {
        fin.aD_word[0] = S.ad; fin.bN_word[0] = (float)bn;
        const float Q0 = lm.state[lm.q_in];                       // k_lm_zeta's rule
        const float zt = (float)(lm.k + 1) * (Q1 - Q0) / Q1;
        const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < lm.q_tol;
        if (stop) { reinterpret_cast<unsigned*>(lm.state)[1] = 1u; reinterpret_cast<int*>(lm.state)[2] = lm.k + 1; }
        else lm.state[lm.q_out] = Q1;
    }
}

// guardedInvert, CERES flavour (gauss_newton.t:638-648)
__device__ __forceinline__ float guarded_invert(float d)
{
    const float s = 1.0f + sqrtf(d);
    return 1.0f / (s * s);
}

// XCD-aware persistent tile schedule.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 labels the XCD group), so group g owns the contiguous tile range
// [g*T/8, (g+1)*T/8) and its members sweep it together: x/y-neighbouring tiles -- whose halos
// overlap -- are in flight on the same XCD's L2 at about the same time.  Purely a speed
// heuristic; correctness never depends on placement.
struct TileSweep {
    int first, last, stride, cur;
    __device__ __forceinline__ TileSweep(int n_tiles)
    {
        const int G = (gridDim.x >= 8 && (gridDim.x % 8) == 0) ? 8 : 1;
        const int g = blockIdx.x % G, l = blockIdx.x / G, per = gridDim.x / G;
        const long lo = (long)n_tiles * g / G, hi = (long)n_tiles * (g + 1) / G;
        first = (int)lo + l; last = (int)hi; stride = per; cur = first;
    }
    __device__ __forceinline__ bool valid() const { return cur < last; }
    __device__ __forceinline__ void next() { cur += stride; }
};


// Host: rows per wave segment of a marching kernel -- the smallest R >= rmin such that nstrips * ceil(ceil(rows / R) / waves_per_wg) workgroups
// (rounded up to a multiple of 8) fit under `cap` (clamped to [8, THALLO_MAX_PARTIALS]); 0 when not even ONE segment per strip fits (the image is
// wider than `cap` strips): the caller then falls back to its tile kernel or reports "unsupported".  Bounded: R never passes `rows`.
inline int march_rows_per_segment(int rows, int nstrips, int waves_per_wg, long cap, int rmin = 4)
{
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    cap -= cap % 8;
    if (cap < 8) cap = 8;
    const int rmax = rows > rmin ? rows : rmin;
    for (int R = rmin; R <= rmax; ++R) {
        const long nseg = (rows + R - 1) / R, total = (long)nstrips * ((nseg + waves_per_wg - 1) / waves_per_wg);
        if ((total + 7) / 8 * 8 <= cap) return R;
    }
    return 0;
}

// ... and whether any R can: one segment per strip (R = rows) is the smallest grid
inline bool march_strips_fit(int nstrips, long cap)
{
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    cap -= cap % 8;
    if (cap < 8) cap = 8;
    return ((long)nstrips + 7) / 8 * 8 <= cap;
}

// Cache-policy helpers.  `nt` selects the non-temporal (streaming) form: data that is touched once per
// sweep and not again before ~600 MB of other traffic (delta, r, pre, the per-GN constant planes)
// should not displace the vectors that ARE re-read by the next kernel (Ap, z, p) from the 256 MiB
// Infinity Cache.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ldf(const float* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float2 ldf2(const float2* p, bool nt)
{
    if (nt) { const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)); return make_float2(v.x, v.y); }
    return *p;
}
__device__ __forceinline__ float4 ldf4(const float4* p, bool nt)
{
    if (nt) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *p;
}
__device__ __forceinline__ unsigned char ldb(const unsigned char* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void stf(float* p, float v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void stf2(float2* p, float2 v, bool nt)
{
    if (nt) { v2f w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<v2f*>(p)); } else *p = v;
}
__device__ __forceinline__ void stf4(float4* p, float4 v, bool nt)
{
    if (nt) { v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; __builtin_nontemporal_store(w, reinterpret_cast<v4f*>(p)); } else *p = v;
}

}  // namespace thallo
