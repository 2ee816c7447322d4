// dist_device.hpp -- device side of the multi-GPU scalar/ghost-row exchange ("8-rank mailbox all-gather of
// 4-8 B, summed inside the consumer kernel"; reference single-GPU scalars: gauss_newton.t:1994-2000).
//
// One process per GPU.  Every rank owns a MAILBOX in its own HBM: granules of 8 bytes {float value | uint32 seq << 32},
// indexed [slot][source rank].  A producer kernel's last workgroup adds the kernel's per-workgroup partials in the fixed
// single-GPU order and stores ONE granule into every rank's mailbox (peer-to-peer stores over xGMI; write-through,
// system scope).  A consumer kernel polls its own mailbox until the `world` granules of the slot carry the current
// sequence number, then adds them in rank order: every rank holds bit-identical alpha/beta without a host round trip or
// a collective.  seq = Gauss-Newton step counter kept in device memory (so a captured hipGraph can be replayed), a
// granule of an earlier step never matches.  Every spin is bounded: on timeout the error word is set, later waits return
// immediately and the host raises after the step.
#pragma once
#include "device_common.hpp"
#include "thallo_hip.h"

namespace thallo {

typedef unsigned long long u64;
constexpr long long DIST_SPIN_TICKS = 20LL * 100000000LL;     // 20 s of the 100 MHz wall clock: covers host-side skew between ranks (a peer may not have launched yet)

__device__ __forceinline__ u64  ld_sys(const u64* p)   { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(u64* p, u64 v)  { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ctl words
enum { DIST_SEQ = 0, DIST_ERR = 1, DIST_TICKET = 2 };

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
        while ((unsigned)(v >> 32) != seq) {
            if ((it & 1023) == 0) {
                if (ld_agent(d.ctl + DIST_ERR) != 0) break;                          // somebody already timed out: do not pile up
                if (it == 0) t0 = wall_clock64();
            }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > DIST_SPIN_TICKS) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    // first timeout of this rank: leave a post-mortem (slot, source rank, expected seq, granule as found)
                    d.ctl[4] = (unsigned)slot; d.ctl[5] = (unsigned)r; d.ctl[6] = seq; d.ctl[7] = (unsigned)(v >> 32); d.ctl[8] = (unsigned)v;
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

// Workgroup sum of v, then the publish protocol.  Every thread calls it, after its last global store of the kernel.
// `red` >= 17 floats of LDS.  partials = this kernel's per-workgroup partial array (also written, so a single-GPU consumer
// or a test can still read it).
__device__ __forceinline__ void dist_reduce_publish(float v, float* __restrict__ partials, const thallo_dist_t& d, int slot, float* red)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1), wave = threadIdx.x / THALLO_WAVE;
    const int nw = (blockDim.x + THALLO_WAVE - 1) / THALLO_WAVE;
    const float ws = wave_sum_all(v);
    if (lane == 0) red[wave] = ws;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores (incl. peer-to-peer ghost rows) are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < nw; ++w) s += red[w];
        __hip_atomic_store(partials + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(d.ctl + DIST_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[16] = (t == gridDim.x - 1) ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (red[16] != 0.0f && wave == 0) {     // the workgroup that arrived last: every partial of this launch is in memory
        float s = 0.0f;
        for (int i = lane; i < (int)gridDim.x; i += THALLO_WAVE) s += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = wave_sum_all(s);                // same association as sum_partials(): world == 1 reproduces the single-GPU bits
        const unsigned seq = ld_agent(d.ctl + DIST_SEQ);
        if (lane == 0) __hip_atomic_store(d.ctl + DIST_TICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < d.world) st_sys(d.peer_mail[lane] + (long)slot * d.world + d.rank, ((u64)seq << 32) | (u64)__float_as_uint(s));
    }
}

}  // namespace thallo
