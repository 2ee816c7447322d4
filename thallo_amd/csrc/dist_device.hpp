// dist_device.hpp -- device side of the multi-GPU scalar/ghost-row exchange ("8-rank mailbox all-gather of
// 4-8 B, summed inside the consumer kernel"; reference single-GPU scalars: gauss_newton.t:1994-2000).
//
// One process per GPU.  Every rank owns a MAILBOX in its own HBM: granules of 8 bytes {float value | uint32 seq << 32},
// indexed [slot][source rank].  The exchange is ONE wave (k_exchange, dist_p2p.hip) launched behind the producing kernel:
// it adds that kernel's per-workgroup partials in the fixed single-GPU order, stores ONE granule into every rank's mailbox
// (peer-to-peer stores over xGMI; write-through, system scope), polls its own mailbox until the `world` granules of the slot
// carry the current sequence number and adds them in rank order into the scalar word the next kernels read: every rank
// holds bit-identical alpha/beta without a host round trip or a collective.  Measured alternatives that lost: polling from
// every workgroup of the consumer kernel (1,536 system-scope loads of one line serialise to 5-15 us per kernel) and
// last-workgroup tickets inside the producer (+1.2 us on 512 workgroups, +5.6 us on 1,024).  seq = Gauss-Newton step counter kept in device memory (so a captured hipGraph can be replayed), a
// granule of an earlier step never matches.  Every spin is bounded: on timeout the error word is set, later waits return
// immediately and the host raises after the step.
#pragma once
#include "device_common.hpp"
#include "thallo_hip.h"

namespace thallo {

typedef unsigned long long u64;
constexpr long long DIST_SPIN_TICKS = 20LL * 100000000LL;     // default bound: 20 s of the 100 MHz wall clock: covers host-side skew between ranks (a peer may not have launched yet)

__device__ __forceinline__ u64  ld_sys(const u64* p)   { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64* p, u64 v)  { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ctl words
enum { DIST_SEQ = 0, DIST_ERR = 1, DIST_SPIN_MS = 2, DIST_POST_MORTEM = 4, DIST_XSEQ = 10, DIST_XTICKET = 11, DIST_ASEQ = 12, DIST_ATICKET = 13 /* .. 15 */, DIST_CTL_WORDS = 16 };     // [10], [11]: thallo_hip_dist_xrows' exchange counter and ticket
// ctl[DIST_SPIN_MS] != 0: spin bound in milliseconds instead of the default (the set-up's self-check runs with 500 ms, so that a
// topology on which granules never become visible costs half a second, not 20 s, before every rank falls back to the collectives)
__device__ __forceinline__ long long dist_spin_ticks(const thallo_dist_t& d)
{
    const unsigned ms = ld_agent(d.ctl + DIST_SPIN_MS);
    return ms ? (long long)ms * 100000LL : DIST_SPIN_TICKS;
}

// Wait for NS slots and return their rank-ordered sums.  Every thread of the workgroup calls it (contains a barrier);
// `vals` = NS*8 floats of LDS.  Needs blockDim.x >= NS*world.
template <int NS>
__device__ __forceinline__ void dist_fetch(const thallo_dist_t& d, const int (&slots)[NS], float (&out)[NS], float* vals)
{
    if ((int)threadIdx.x < NS * d.world) {
        const unsigned seq = ld_agent(d.ctl + DIST_SEQ);
        int r = threadIdx.x, slot = slots[0];            // lane -> (slot, source rank), without dynamic indexing of `slots`
#pragma unroll
        for (int q = 1; q < NS; ++q) if ((int)threadIdx.x >= q * d.world) { r = threadIdx.x - q * d.world; slot = slots[q]; }
        const u64* g = d.mail + (long)slot * d.world + r;
        u64 v = ld_sys(g);
        int it = 0;
        long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(v >> 32) != seq) {
            if ((it & 1023) == 0) {
                if (ld_agent(d.ctl + DIST_ERR) != 0) break;                          // somebody already timed out: do not pile up
                if (it == 0) t0 = wall_clock64();
            }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    // first timeout of this rank: leave a post-mortem (slot, source rank, expected seq, granule as found)
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)slot; pm[1] = (unsigned)r; pm[2] = seq; pm[3] = (unsigned)(v >> 32); pm[4] = (unsigned)v;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            v = ld_sys(g);
        }
        vals[threadIdx.x] = __uint_as_float((unsigned)v);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float t = 0.0f;
        for (int r = 0; r < d.world; ++r) t += vals[s * d.world + r];
        out[s] = t;
    }
}

// The exchange of the one-kernel-per-iteration schedule, executed by ONE full wave (all 64 lanes, converged; no LDS, no barrier):
// this rank's sums (alphaD float; N, S1, S2 double) go out as 7 granules to every rank's mailbox slots slot0 .. slot0+6 (the doubles
// as hi / lo words), the wave waits (bounded) for everybody's, adds them in rank order and lane 0 writes alphaD_k and
// betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2 with alpha_k = alphaN / alphaD_k.  Identical bits on every rank.
struct ExtraSums { bool on; float ad; double q0, q1, q2; };     // added behind the rank-ordered sums (shard form: the replicated block's own sums, k_shard_scalars' order)
__device__ __forceinline__ void dist_exchange_iter_wave_seq(const thallo_dist_t& d, unsigned seq, int slot0, float ad, double q0, double q1, double q2, float an,
                                                            float* __restrict__ aD_word, float* __restrict__ bN_word, ExtraSums ex = ExtraSums{ false, 0.0f, 0.0, 0.0, 0.0 },
                                                            float* gad_out = nullptr, float* alpha_out = nullptr, float* bn_out = nullptr);
__device__ __forceinline__ void dist_exchange_iter_wave(const thallo_dist_t& d, int slot0, float ad, double q0, double q1, double q2, float an,
                                                        float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    dist_exchange_iter_wave_seq(d, ld_agent(d.ctl + DIST_SEQ), slot0, ad, q0, q1, q2, an, aD_word, bN_word);
}
// (seq given by the caller: thallo_hip_dist_xrows tags with its own exchange counter)
__device__ __forceinline__ void dist_exchange_iter_wave_seq(const thallo_dist_t& d, const unsigned seq, int slot0, float ad, double q0, double q1, double q2, float an,
                                                            float* __restrict__ aD_word, float* __restrict__ bN_word, ExtraSums ex, float* gad_out, float* alpha_out, float* bn_out)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    unsigned w[7];
    w[0] = __float_as_uint(ad);
    { const u64 b = (u64)__double_as_longlong(q0); w[1] = (unsigned)(b >> 32); w[2] = (unsigned)b; }
    { const u64 b = (u64)__double_as_longlong(q1); w[3] = (unsigned)(b >> 32); w[4] = (unsigned)b; }
    { const u64 b = (u64)__double_as_longlong(q2); w[5] = (unsigned)(b >> 32); w[6] = (unsigned)b; }
    if (lane < d.world) {
#pragma unroll
        for (int j = 0; j < 7; ++j) st_sys(d.peer_mail[lane] + (long)(slot0 + j) * d.world + d.rank, ((u64)seq << 32) | (u64)w[j]);
    }
    unsigned mine = 0;
    if (lane < 7 * d.world) {                       // lane -> (granule j, source rank r)
        const int j = lane / d.world, r = lane - j * d.world;
        const u64* g = d.mail + (long)(slot0 + j) * d.world + r;
        u64 v = ld_sys(g);
        int it = 0; long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(v >> 32) != seq) {
            if ((it & 1023) == 0) { if (ld_agent(d.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)(slot0 + j); pm[1] = (unsigned)r; pm[2] = seq; pm[3] = (unsigned)(v >> 32); pm[4] = (unsigned)v;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            v = ld_sys(g);
        }
        mine = (unsigned)v;
    }
    float gad = 0.0f; double gq[3] = { 0.0, 0.0, 0.0 };
    for (int r = 0; r < d.world; ++r) {             // every lane runs the same shuffles; lane 0 keeps the result
        gad += __uint_as_float(__shfl(mine, r, THALLO_WAVE));
        for (int j = 0; j < 3; ++j) {
            const unsigned hi = __shfl(mine, (1 + 2 * j) * d.world + r, THALLO_WAVE), lo = __shfl(mine, (2 + 2 * j) * d.world + r, THALLO_WAVE);
            gq[j] += __longlong_as_double((long long)(((u64)hi << 32) | (u64)lo));
        }
    }
    if (ex.on) { gad += ex.ad; gq[0] += ex.q0; gq[1] += ex.q1; gq[2] += ex.q2; }
    const float alpha = safe_div<false>(an, gad);
    double bn = gq[0] - 2.0 * (double)alpha * gq[1] + (double)alpha * (double)alpha * gq[2];
    if (!(bn > 0.0)) bn = 0.0;
    if (lane == 0) { aD_word[0] = gad; bN_word[0] = (float)bn; }
    if (gad_out) *gad_out = gad;
    if (alpha_out) *alpha_out = alpha;
    if (bn_out) *bn_out = (float)bn;
}

// ND doubles per rank (as hi / lo words: 2 * ND granules), by ONE full wave (2 * ND * world <= 64): bounded wait, rank-ordered double sums (every lane returns them)
template <int ND>
__device__ __forceinline__ void dist_exchange_doubles_wave_seq(const thallo_dist_t& d, const unsigned seq, int slot0, const double (&v)[ND], double (&out)[ND])
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    if (lane < d.world) {
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const u64 b = (u64)__double_as_longlong(v[j]);
            st_sys(d.peer_mail[lane] + (long)(slot0 + 2 * j) * d.world + d.rank, ((u64)seq << 32) | (b >> 32));
            st_sys(d.peer_mail[lane] + (long)(slot0 + 2 * j + 1) * d.world + d.rank, ((u64)seq << 32) | (b & 0xffffffffull));
        }
    }
    unsigned got = 0;
    if (lane < 2 * ND * d.world) {
        const int j = lane / d.world, r = lane - j * d.world;
        const u64* g = d.mail + (long)(slot0 + j) * d.world + r;
        u64 w = ld_sys(g);
        int it = 0; long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(w >> 32) != seq) {
            if ((it & 1023) == 0) { if (ld_agent(d.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)(slot0 + j); pm[1] = (unsigned)r; pm[2] = seq; pm[3] = (unsigned)(w >> 32); pm[4] = (unsigned)w;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            w = ld_sys(g);
        }
        got = (unsigned)w;
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        double t = 0.0;
        for (int r = 0; r < d.world; ++r) {
            const unsigned hi = __shfl(got, (2 * j) * d.world + r, THALLO_WAVE), lo = __shfl(got, (2 * j + 1) * d.world + r, THALLO_WAVE);
            t += __longlong_as_double((long long)(((u64)hi << 32) | (u64)lo));
        }
        out[j] = t;
    }
}

// NS floats per rank, by ONE full wave (NS * world <= 64): granules to every rank's slots slot0 .. slot0+NS-1, bounded wait, rank-ordered sums (every lane returns them)
template <int NS>
__device__ __forceinline__ void dist_exchange_words_wave_seq(const thallo_dist_t& d, const unsigned seq, int slot0, const float (&w)[NS], float (&out)[NS])
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    if (lane < d.world) {
#pragma unroll
        for (int j = 0; j < NS; ++j) st_sys(d.peer_mail[lane] + (long)(slot0 + j) * d.world + d.rank, ((u64)seq << 32) | (u64)__float_as_uint(w[j]));
    }
    unsigned got = 0;
    if (lane < NS * d.world) {
        const int j = lane / d.world, r = lane - j * d.world;
        const u64* g = d.mail + (long)(slot0 + j) * d.world + r;
        u64 v = ld_sys(g);
        int it = 0; long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(v >> 32) != seq) {
            if ((it & 1023) == 0) { if (ld_agent(d.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)(slot0 + j); pm[1] = (unsigned)r; pm[2] = seq; pm[3] = (unsigned)(v >> 32); pm[4] = (unsigned)v;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            v = ld_sys(g);
        }
        got = (unsigned)v;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        float t = 0.0f;
        for (int r = 0; r < d.world; ++r) t += __uint_as_float(__shfl(got, j * d.world + r, THALLO_WAVE));
        out[j] = t;
    }
}

}  // namespace thallo
