// dist_p2p.hip -- host entry points of the multi-GPU device-side exchange (include/thallo_hip.h, dist_device.hpp).
#include "dist_device.hpp"
#include <string.h>
#include <algorithm>
#include <stdlib.h>

using namespace thallo;

namespace {
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

__global__ void k_begin_step(thallo_dist_t d)
{
    if (threadIdx.x == 0) __hip_atomic_fetch_add(d.ctl + DIST_SEQ, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave: local fixed-order sum -> one granule to every rank -> wait for every rank's granule -> rank-ordered sum
__global__ __launch_bounds__(64) void k_exchange(thallo_dist_t d, int slot, thallo_sum_t local, float* __restrict__ out)
{
    __shared__ float vals[8];
    const float s = sum_partials(local.partials, local.count);      // same association as a single-GPU consumer
    const unsigned seq = ld_agent(d.ctl + DIST_SEQ);
    if ((int)threadIdx.x < d.world)
        st_sys(d.peer_mail[threadIdx.x] + (long)slot * d.world + d.rank, ((u64)seq << 32) | (u64)__float_as_uint(s));
    const int slots[1] = { slot };
    float t[1];
    dist_fetch<1>(d, slots, t, vals);
    if (threadIdx.x == 0) out[0] = t[0];
}

__device__ __forceinline__ double wave_sum_all_dd(double v) { return wave_sum_all_f64(v); }

// The exchange of the one-kernel-per-iteration schedule (thallo_hip_iw_pcg_iter): one wave adds this rank's alphaD partials
// (float) and N, S1, S2 partials (double) in the fixed single-GPU order, sends them as 7 granules (the doubles as hi / lo words)
// to every rank's mailbox slots slot0 .. slot0+6, waits for everybody's, adds in rank order and finishes the scalars exactly
// like k_iter_finish: alphaD_k and betaN_k = N - 2 alpha_k S1 + alpha_k^2 S2.
__global__ __launch_bounds__(64) void k_exchange_iter(thallo_dist_t d, int slot0, const float* __restrict__ aD_part, const double* __restrict__ s12, int nb,
                                                      thallo_sum_t aN, float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    const int lane = threadIdx.x;
    const float ad = sum_partials(aD_part, nb);
    double q[3] = { 0.0, 0.0, 0.0 };
    for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s12[3 * i]; q[1] += s12[3 * i + 1]; q[2] += s12[3 * i + 2]; }
    const double q0 = wave_sum_all_dd(q[0]), q1 = wave_sum_all_dd(q[1]), q2 = wave_sum_all_dd(q[2]);
    dist_exchange_iter_wave(d, slot0, ad, q0, q1, q2, aN.count == 1 ? aN.partials[0] : 0.0f, aD_word, bN_word);
}

__global__ __launch_bounds__(64) void k_collect(thallo_dist_t d, int slot0, int nslots, float* __restrict__ out)
{
    __shared__ float vals[8];
    for (int j = 0; j < nslots; ++j) {
        const int slots[1] = { slot0 + j };
        float s[1];
        dist_fetch<1>(d, slots, s, vals);
        if (threadIdx.x == 0) out[j] = s[0];
        __syncthreads();
    }
}

// thallo_hip_dist_xrows (thallo_hip.h): boundary rows of a flat vector into the neighbours' inboxes + the scalars of the exchange to every rank + own inbox -> ghost rows,
// ONE launch.  8-byte system-scope stores / loads for the rows (the inbox is memory another GPU writes while this one reads it).
__device__ __forceinline__ float* inbox_of(unsigned long long* mail, const thallo_xrows_t& x, int parity, int dir)
{
    return (float*)((char*)mail + x.inbox_off) + (long)(parity * 2 + dir) * x.inbox_half;
}
__device__ __forceinline__ void rows_out(const float* __restrict__ vec, const thallo_segs_t& segs, float* dst)
{   // segment lengths and offsets are multiples of 2 floats (checked by the host entry)
    long base = 0;
    for (int k = 0; k < segs.n; ++k) {
        const u64* src = (const u64*)(vec + segs.off[k]);
        u64* out = (u64*)(dst + base);
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < segs.len[k] / 2; i += (long)gridDim.x * blockDim.x) st_sys(out + i, src[i]);
        base += segs.len[k];
    }
}
__device__ __forceinline__ void rows_in(float* __restrict__ vec, const thallo_segs_t& segs, const float* src)
{   // by ONE workgroup
    long base = 0;
    for (int k = 0; k < segs.n; ++k) {
        u64* out = (u64*)(vec + segs.off[k]);
        const u64* in = (const u64*)(src + base);
        for (long i = threadIdx.x; i < segs.len[k] / 2; i += blockDim.x) out[i] = ld_sys(in + i);
        base += segs.len[k];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_xrows(thallo_dist_t d, thallo_xrows_t x, float* __restrict__ vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                                                thallo_sum_t s, const float* __restrict__ aD_part, const double* __restrict__ s3, int nb, int poison,
                                                float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ zstate, int zk, float ztol,
                                                const float* __restrict__ aD2 = nullptr, const double* __restrict__ s3_2 = nullptr, int nb2 = 0,
                                                thallo_units_t us = thallo_units_t{}, thallo_units_t ur = thallo_units_t{}, long unit_slot = 0)
{
    __shared__ unsigned last_wg;
    const unsigned tag = ld_agent(d.ctl + DIST_XSEQ) + 1u;
    const int par = (int)(tag & 1u);
    if (!poison) {
        if (x.above >= 0) rows_out(vec, first, inbox_of(d.peer_mail[x.above], x, par, 1));       // I am BELOW my upper neighbour
        if (x.below >= 0) rows_out(vec, last, inbox_of(d.peer_mail[x.below], x, par, 0));
        if (us.n > 0) {      // partitioned graph: my boundary units' values into EVERY other rank's inbox, area [parity][me] (thallo_units_t; one workgroup)
            int per = 0; for (int k = 0; k < us.nplanes; ++k) per += us.len[k];
            const long total = (long)us.n * per;
            for (long i = threadIdx.x; i < total; i += blockDim.x) {
                const int unit = (int)(i / per); int c = (int)(i - (long)unit * per), k = 0;
                while (c >= us.len[k]) { c -= us.len[k]; ++k; }
                const float v = vec[us.base[k] + (long)us.units[unit] * us.len[k] + c];
                for (int r = 0; r < d.world; ++r)
                    if (r != d.rank) st_sys((float*)((char*)d.peer_mail[r] + x.inbox_off) + ((long)par * d.world + d.rank) * unit_slot + i, v);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the rows went out as write-through system-scope stores: drained = at the peer)
    __syncthreads();
    if (gridDim.x > 1) {                            // (long rows: several workgroups carry them, the last one to arrive goes on)
        if (threadIdx.x == 0) last_wg = __hip_atomic_fetch_add(d.ctl + DIST_XTICKET, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (!last_wg) return;
    }
    // every row of this rank is out and fenced: the granules may go
    const int slot0 = x.ring0 + (int)(tag & 3u) * 16;
    if (threadIdx.x < THALLO_WAVE) {
        const int lane = threadIdx.x;
        if (MODE == 0) {                            // up to two float sums: s -> out0, (aD_part, nb) -> out1
            float w[2], t[2];
            w[0] = s.count > 0 ? sum_partials(s.partials, s.count) : 0.0f;
            w[1] = nb > 0 ? sum_partials(aD_part, nb) : 0.0f;
            if (poison) w[0] = w[1] = __uint_as_float(0x7fc00000u);
            dist_exchange_words_wave_seq<2>(d, tag, slot0, w, t);
            if (lane == 0 && s.count > 0) out0[0] = t[0];
            if (lane == 0 && nb > 0) out1[0] = t[1];
            if (lane == 0 && zstate && reinterpret_cast<unsigned*>(zstate)[1] == 0u) {     // k_lm_zeta's rule on the global q (pcg_kernels.hip)
                const float Q1 = t[0], Q0 = zstate[0];
                const float zt = (float)(zk + 1) * (Q1 - Q0) / Q1;
                const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < ztol;
                if (stop) { reinterpret_cast<unsigned*>(zstate)[1] = 1u; reinterpret_cast<int*>(zstate)[2] = zk + 1; }
                else zstate[0] = Q1;
            }
        } else if (MODE == 2) {                     // an LM iteration's scalars: alphaD, {N, S1, S2} and {U, T1, T2} (s3_2 = the q3 slots) -> alphaD_k, betaN_k, q_{k+1}, the zeta test
            float ad = sum_partials(aD_part, nb);
            double q[3] = { 0.0, 0.0, 0.0 }, u[3] = { 0.0, 0.0, 0.0 }, gu[3];
            for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s3[3 * i]; q[1] += s3[3 * i + 1]; q[2] += s3[3 * i + 2]; u[0] += s3_2[3 * i]; u[1] += s3_2[3 * i + 1]; u[2] += s3_2[3 * i + 2]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = wave_sum_all_f64(q[j]); u[j] = wave_sum_all_f64(u[j]); }
            if (poison) ad = __uint_as_float(0x7fc00000u);
            float gad = 0.0f, al = 0.0f;
            dist_exchange_iter_wave_seq(d, tag, slot0, ad, q[0], q[1], q[2], s.count == 1 ? s.partials[0] : 0.0f, out0, out1, ExtraSums{ false, 0.0f, 0.0, 0.0, 0.0 }, &gad, &al);
            dist_exchange_doubles_wave_seq<3>(d, tag, slot0 + 7, u, gu);
            // (LM divides blindly, gauss_newton.t:226-234: alpha of the exchange above is the guarded quotient -- equal unless alphaD is 0)
            const float Q1 = (float)(0.5 * (gu[0] + (double)al * (gu[1] - gu[2]) - (double)al * (double)al * (double)gad));
            if (lane == 0 && zstate && reinterpret_cast<unsigned*>(zstate)[1] == 0u) {
                const float Q0 = zstate[0];
                const float zt = (float)(zk + 1) * (Q1 - Q0) / Q1;
                const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < ztol;
                if (stop) { reinterpret_cast<unsigned*>(zstate)[1] = 1u; reinterpret_cast<int*>(zstate)[2] = zk + 1; }
                else zstate[0] = Q1;
            }
        } else {
            float ad = sum_partials(aD_part, nb);
            double q[3] = { 0.0, 0.0, 0.0 };
            for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s3[3 * i]; q[1] += s3[3 * i + 1]; q[2] += s3[3 * i + 2]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) q[j] = wave_sum_all_f64(q[j]);
            if (poison) ad = __uint_as_float(0x7fc00000u);
            ExtraSums ex = { false, 0.0f, 0.0, 0.0, 0.0 };
            if (nb2 > 0) {                      // shard form: the replicated block's sums, computed by every rank for itself, join behind the ranks' (k_shard_scalars' order)
                ex.on = true; ex.ad = sum_partials(aD2, nb2);
                double n = 0.0, a = 0.0, b = 0.0;
                for (int i = lane; i < nb2; i += THALLO_WAVE) { n += s3_2[3 * i]; a += s3_2[3 * i + 1]; b += s3_2[3 * i + 2]; }
                ex.q0 = wave_sum_all_f64(n); ex.q1 = wave_sum_all_f64(a); ex.q2 = wave_sum_all_f64(b);
            }
            dist_exchange_iter_wave_seq(d, tag, slot0, ad, q[0], q[1], q[2], s.count == 1 ? s.partials[0] : (s.count > 1 ? sum_partials(s.partials, s.count) : 0.0f), out0, out1, ex);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                                                      // (system scope: the neighbours' rows are behind their granules)
    if (x.above >= 0) rows_in(vec, top, inbox_of(d.mail, x, par, 0));
    if (x.below >= 0) rows_in(vec, bot, inbox_of(d.mail, x, par, 1));
    if (ur.n > 0) {          // my ghost units from my own inbox: ur.src[g] = source rank * unit_slot + position * per (relative to the parity's base)
        int per = 0; for (int k = 0; k < ur.nplanes; ++k) per += ur.len[k];
        const float* base = (const float*)((const char*)d.mail + x.inbox_off) + (long)par * d.world * unit_slot;
        const long total = (long)ur.n * per;
        for (long i = threadIdx.x; i < total; i += blockDim.x) {
            const int g = (int)(i / per); const int within = (int)(i - (long)g * per); int c = within, k = 0;
            while (c >= ur.len[k]) { c -= ur.len[k]; ++k; }
            vec[ur.base[k] + (long)ur.units[g] * ur.len[k] + c] = __hip_atomic_load(base + ur.src[g] + within, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x > 1) __hip_atomic_store(d.ctl + DIST_XTICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d.ctl + DIST_XSEQ, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// thallo_hip_dist_allreduce (thallo_hip.h): reduce-scatter + all-gather by peer stores, one launch, every workgroup resident
__device__ __forceinline__ float* xr_inbox(unsigned long long* mail, const thallo_xreduce_t& x, int world, int which, int par, int src)
{
    return (float*)((char*)mail + x.inbox_off) + ((long)(which * 2 + par) * world + src) * x.chunk;
}
// every thread of the workgroup: wait until `slot` carries `tag` from every rank
__device__ __forceinline__ void xr_wait_all(const thallo_dist_t& d, int slot, unsigned tag)
{
    if ((int)threadIdx.x < d.world) {
        const u64* g = d.mail + (long)slot * d.world + threadIdx.x;
        u64 v = ld_sys(g);
        int it = 0; long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(v >> 32) != tag) {
            if ((it & 1023) == 0) { if (ld_agent(d.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)slot; pm[1] = threadIdx.x; pm[2] = tag; pm[3] = (unsigned)(v >> 32); pm[4] = (unsigned)v;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(4);
            v = ld_sys(g);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
// grid-wide arrival: returns true in every thread of the LAST workgroup to arrive (which then sends the granules of this phase)
__device__ __forceinline__ bool xr_arrive(unsigned* ticket, unsigned* flag_lds)
{
    // every peer store above is a write-through system-scope store: a wave that has drained them (vmcnt(0)) has them at the peer; no L2 write-back needed
    // (a __threadfence_system() here writes back every dirty line of this XCD's L2: three of those per launch cost more than the data movement)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *flag_lds = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    return *flag_lds != 0u;
}
__global__ __launch_bounds__(256) void k_allreduce_p2p(thallo_dist_t d, thallo_xreduce_t x, float* __restrict__ buf, long len, int poison)
{
    __shared__ unsigned flag;
    const unsigned tag = ld_agent(d.ctl + DIST_ASEQ) + 1u;
    const int par = (int)(tag & 1u), world = d.world, me = d.rank;
    const int slot_rs = x.ring0 + (int)(tag & 3u) * 2, slot_ag = slot_rs + 1;
    const long C = x.chunk;
    const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    auto clen = [&](int c) { const long lo = (long)c * C, hi = lo + C < len ? lo + C : len; return hi > lo ? hi - lo : 0L; };
    // 1. my part of every other rank's chunk -> its inbox A
    for (int c = 0; c < world; ++c) {
        if (c == me) continue;
        const u64* src = (const u64*)(buf + (long)c * C);
        u64* dst = (u64*)xr_inbox(d.peer_mail[c], x, world, 0, par, me);
        const long n2 = clen(c) / 2;
        for (long i = gtid; i < n2; i += gsz) st_sys(dst + i, src[i]);
    }
    if (xr_arrive(d.ctl + DIST_ATICKET, &flag) && (int)threadIdx.x < world)
        st_sys(d.peer_mail[threadIdx.x] + (long)slot_rs * world + me, ((u64)tag << 32) | 1ull);
    xr_wait_all(d, slot_rs, tag);
    // 2. my chunk: the contributions in RANK order; the sums into place and into every other rank's inbox B
    {
        const long n2 = clen(me) / 2;
        float* mine = buf + (long)me * C;
        for (long i = gtid; i < n2; i += gsz) {
            float2 acc = make_float2(0.0f, 0.0f);
            for (int r = 0; r < world; ++r) {
                float2 t;
                if (r == me) t = ((const float2*)mine)[i];
                else { const u64 w = ld_sys((const u64*)xr_inbox(d.mail, x, world, 0, par, r) + i); t = make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32))); }
                acc.x = r == 0 ? t.x : acc.x + t.x; acc.y = r == 0 ? t.y : acc.y + t.y;
            }
            if (poison && i == 0) acc.x = __uint_as_float(0x7fc00000u);
            ((float2*)mine)[i] = acc;
            const u64 w = (u64)__float_as_uint(acc.x) | ((u64)__float_as_uint(acc.y) << 32);
            for (int c = 0; c < world; ++c) if (c != me) st_sys((u64*)xr_inbox(d.peer_mail[c], x, world, 1, par, me) + i, w);
        }
    }
    if (xr_arrive(d.ctl + DIST_ATICKET + 1, &flag) && (int)threadIdx.x < world)
        st_sys(d.peer_mail[threadIdx.x] + (long)slot_ag * world + me, ((u64)tag << 32) | 2ull);
    xr_wait_all(d, slot_ag, tag);
    // 3. the other ranks' chunks from my inbox B into place
    for (int c = 0; c < world; ++c) {
        if (c == me) continue;
        const u64* src = (const u64*)xr_inbox(d.mail, x, world, 1, par, c);
        u64* dst = (u64*)(buf + (long)c * C);
        const long n2 = clen(c) / 2;
        for (long i = gtid; i < n2; i += gsz) dst[i] = ld_sys(src + i);
    }
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(d.ctl + DIST_ATICKET + 2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        for (int q = 0; q < 3; ++q) __hip_atomic_store(d.ctl + DIST_ATICKET + q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d.ctl + DIST_ASEQ, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

extern "C" {

// Memory other GPUs write into (mailboxes; the vector block holding the ghost rows) while this GPU's kernels read it: FINE-GRAINED device
// memory (hipDeviceMallocFinegrained), the allocation type HIP defines cross-agent coherence for inside a running kernel.  Plain hipMalloc
// memory is coarse-grained: coherent with other agents at kernel boundaries only, so a poll from inside a running kernel may legally
// keep seeing a stale line in this GPU's L2.  THALLO_DIST_MEM=coarse selects the plain allocation (A/B on one GPU, where every "peer"
// shares the L2 and both behave alike); if the fine-grained allocation or its IPC export is refused the plain one is used and *kind_out
// says so.  kind: 1 fine-grained, 0 coarse.
int thallo_hip_ipc_alloc2(long bytes, void** ptr, void* handle_out64, int* kind_out)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    static_assert(DIST_CTL_WORDS <= THALLO_DIST_CTL_WORDS, "ctl layout");
    if (bytes <= 0 || !ptr || !handle_out64) return -(int)hipErrorInvalidValue;
    const bool want_fine = true;       // fine-grained: the allocation type HIP defines cross-agent coherence for while a kernel runs; plain hipMalloc only if it is refused
    void* p = nullptr;
    hipError_t e = hipErrorUnknown;
    int kind = 0;
    if (want_fine) {
        e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
        if (e == hipSuccess) {
            hipIpcMemHandle_t probe;
            if (hipIpcGetMemHandle(&probe, p) == hipSuccess) kind = 1;
            else { (void)hipFree(p); p = nullptr; e = hipErrorUnknown; (void)hipGetLastError(); }
        } else (void)hipGetLastError();
    }
    if (!p) e = hipMalloc(&p, (size_t)bytes);
    if (e != hipSuccess) return -(int)e;
    if (kind_out) *kind_out = kind;
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return -(int)e; }
    memcpy(handle_out64, &h, 64);
    *ptr = p;
    return 0;
}

int thallo_hip_ipc_alloc(long bytes, void** ptr, void* handle_out64) { return thallo_hip_ipc_alloc2(bytes, ptr, handle_out64, nullptr); }

int thallo_hip_ipc_open(const void* handle64, void** ptr)
{
    if (!handle64 || !ptr) return -(int)hipErrorInvalidValue;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : -(int)e;
}

int thallo_hip_ipc_close(void* ptr) { hipError_t e = hipIpcCloseMemHandle(ptr); return e == hipSuccess ? 0 : -(int)e; }
int thallo_hip_ipc_free(void* ptr)  { hipError_t e = hipFree(ptr); return e == hipSuccess ? 0 : -(int)e; }

static int dist_ok(const thallo_dist_t& d)
{
    return d.world >= 1 && d.world <= THALLO_DIST_MAX_WORLD && d.rank >= 0 && d.rank < d.world && d.mail && d.ctl && d.peer_mail[d.rank] == d.mail;
}

int thallo_hip_dist_begin_step(thallo_dist_t d, thallo_stream_t stream)
{
    if (!dist_ok(d)) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_begin_step, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    return check_launch();
}

int thallo_hip_dist_exchange(thallo_dist_t d, int slot, thallo_sum_t local, float* out, thallo_stream_t stream)
{
    if (!dist_ok(d) || slot < 0 || !out || !local.partials || local.count < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_exchange, dim3(1), dim3(64), 0, (hipStream_t)stream, d, slot, local, out);
    return check_launch();
}

int thallo_hip_dist_exchange_iter(thallo_dist_t d, int slot0, const float* aD_partials, const double* s12_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!dist_ok(d) || slot0 < 0 || !aD_partials || !s12_partials || count < 1 || count > THALLO_MAX_PARTIALS || alphaN.count != 1 || !alphaN.partials ||
        !alphaD_word || !betaN_word || 7 * d.world > 64) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_exchange_iter, dim3(1), dim3(64), 0, (hipStream_t)stream, d, slot0, aD_partials, s12_partials, count, alphaN, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_dist_collect(thallo_dist_t d, int slot0, int nslots, float* out, thallo_stream_t stream)
{
    if (!dist_ok(d) || slot0 < 0 || nslots < 0 || !out) return -(int)hipErrorInvalidValue;
    if (nslots == 0) return 0;
    hipLaunchKernelGGL(k_collect, dim3(1), dim3(64), 0, (hipStream_t)stream, d, slot0, nslots, out);
    return check_launch();
}

static int xrows_impl(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                      int mode, thallo_sum_t s, const float* aD_partials, const double* s3_partials, int count, int poison,
                      float* out0, float* out1, float* zstate, int zk, float ztol, thallo_stream_t stream,
                      const float* aD2 = nullptr, const double* s3_2 = nullptr, int nb2 = 0,
                      thallo_units_t us = thallo_units_t{}, thallo_units_t ur = thallo_units_t{}, long unit_slot = 0)
{
    if (!dist_ok(d) || x.ring0 < 0 || x.inbox_off <= 0 || (x.inbox_off & 7) || x.inbox_half < 0 || (x.inbox_half & 1) || x.above >= d.world || x.below >= d.world ||
        x.above == d.rank || x.below == d.rank || 7 * d.world > 64) return -(int)hipErrorInvalidValue;
    auto total = [](const thallo_segs_t& g, bool& ok) {
        long t = 0;
        if (g.n < 0 || g.n > 8) { ok = false; return 0L; }
        for (int k = 0; k < g.n; ++k) { if (g.len[k] < 0 || (g.len[k] & 1) || (g.off[k] & 1) || g.off[k] < 0) ok = false; t += g.len[k]; }
        return t;
    };
    bool ok = true;
    const long tf = total(first, ok), tl = total(last, ok), tt = total(top, ok), tb = total(bot, ok);
    if (!ok || tf > x.inbox_half || tl > x.inbox_half) return -(int)hipErrorInvalidValue;
    if ((x.above >= 0 && tt != tl && tt != tf) || (x.below >= 0 && tb != tf && tb != tl)) return -(int)hipErrorInvalidValue;   // (slabs are symmetric: what I receive is as long as what I send)
    if ((tf + tl + tt + tb > 0) && !vec) return -(int)hipErrorInvalidValue;
    if (mode == 0) {
        if (s.count < 0 || (s.count > 0 && (!s.partials || !out0)) || s.count > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
        if (count < 0 || (count > 0 && (!aD_partials || !out1)) || count > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;      // the optional second sum
    } else if (mode == 1) { if (!aD_partials || !s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || s.count != 1 || !s.partials || !out0 || !out1) return -(int)hipErrorInvalidValue; }
    else if (mode == 2) { if (!aD_partials || !s3_partials || !s3_2 || s3_2 == s3_partials || count < 1 || count > THALLO_MAX_PARTIALS || s.count != 1 || !s.partials || !out0 || !out1 || !zstate || 13 * 0 + 7 * d.world > 64) return -(int)hipErrorInvalidValue; }
    else return -(int)hipErrorInvalidValue;
    // rows of up to 32 K floats per direction (a 2048-wide image: 2 ghost rows of 4 channels): ONE workgroup, no ticket; longer ones: 8 workgroups
    const int grid = std::max(tf, tl) <= 32768 ? 1 : 8, block = 256;
    if (mode == 2) hipLaunchKernelGGL(k_xrows<2>, dim3(grid), dim3(block), 0, (hipStream_t)stream, d, x, vec, first, last, top, bot, s, aD_partials, s3_partials, count, poison, out0, out1, zstate, zk, ztol, (const float*)nullptr, s3_2, nb2, us, ur, unit_slot);
    else if (mode == 0) hipLaunchKernelGGL(k_xrows<0>, dim3(grid), dim3(block), 0, (hipStream_t)stream, d, x, vec, first, last, top, bot, s, aD_partials, s3_partials, count, poison, out0, out1, zstate, zk, ztol, (const float*)nullptr, (const double*)nullptr, 0, us, ur, unit_slot);
    else           hipLaunchKernelGGL(k_xrows<1>, dim3(grid), dim3(block), 0, (hipStream_t)stream, d, x, vec, first, last, top, bot, s, aD_partials, s3_partials, count, poison, out0, out1, (float*)nullptr, 0, 0.0f, aD2, s3_2, nb2, us, ur, unit_slot);
    return check_launch();
}

int thallo_hip_dist_xrows(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                          int mode, thallo_sum_t s, const float* aD_partials, const double* s3_partials, int count, int poison,
                          float* out0, float* out1, thallo_stream_t stream)
{ return xrows_impl(d, x, vec, first, last, top, bot, mode, s, aD_partials, s3_partials, count, poison, out0, out1, nullptr, 0, 0.0f, stream); }
int thallo_hip_dist_xrows_zeta(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                               thallo_sum_t q_local, const float* second_partials, int second_count, int poison, float* q_out, float* second_out,
                               float* lm_state, int k, float q_tolerance, thallo_stream_t stream)
{
    if (!lm_state || q_local.count < 1 || !q_out) return -(int)hipErrorInvalidValue;
    return xrows_impl(d, x, vec, first, last, top, bot, 0, q_local, second_partials, nullptr, second_count, poison, q_out, second_out, lm_state, k, q_tolerance, stream);
}

int thallo_hip_dist_allreduce(thallo_dist_t d, thallo_xreduce_t x, float* buf, long len, int poison, thallo_stream_t stream)
{
    if (!dist_ok(d) || !buf || len < 4 || (len & 3) || x.chunk < 4 || (x.chunk & 3) || x.chunk * d.world < len || x.inbox_off <= 0 || (x.inbox_off & 15) || x.ring0 < 0) return -(int)hipErrorInvalidValue;
    long per = (x.chunk / 2 + 255) / 256;
    int grid = (int)(per < 1 ? 1 : per > 64 ? 64 : per);         // every workgroup resident at once: the phases wait on each other inside the launch
    hipLaunchKernelGGL(k_allreduce_p2p, dim3(grid), dim3(256), 0, (hipStream_t)stream, d, x, buf, len, poison);
    return check_launch();
}

int thallo_hip_dist_xscalars_shard(thallo_dist_t d, thallo_xrows_t x, thallo_sum_t alphaN, const float* own_alphaD_partials, const double* own_s3_partials, int own_count,
                                   const float* shared_alphaD_partials, const double* shared_s3_partials, int shared_count, int poison,
                                   float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!shared_alphaD_partials || !shared_s3_partials || shared_count < 1 || shared_count > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const thallo_segs_t none = {};
    x.above = -1; x.below = -1;
    return xrows_impl(d, x, nullptr, none, none, none, none, 1, alphaN, own_alphaD_partials, own_s3_partials, own_count, poison, alphaD_word, betaN_word, nullptr, 0, 0.0f, stream,
                      shared_alphaD_partials, shared_s3_partials, shared_count);
}

int thallo_hip_dist_xunits(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_units_t send, thallo_units_t recv, long unit_slot_floats,
                           int mode, thallo_sum_t local_or_alphaN, const float* alphaD_partials, const double* s3_partials, int count, int poison,
                           float* out0, float* out1, thallo_stream_t stream)
{
    long per = 0; for (int k = 0; k < send.nplanes && k < 8; ++k) per += send.len[k];
    if (send.n < 0 || recv.n < 0 || send.nplanes < 1 || send.nplanes > 8 || recv.nplanes != send.nplanes || (send.n && !send.units) || (recv.n && (!recv.units || !recv.src)) ||
        (long)send.n * per > unit_slot_floats || unit_slot_floats > 32768 || ((send.n || recv.n) && !vec)) return -(int)hipErrorInvalidValue;
    const thallo_segs_t none = {};
    x.above = -1; x.below = -1;
    return xrows_impl(d, x, vec, none, none, none, none, mode, local_or_alphaN, alphaD_partials, s3_partials, count, poison, out0, out1, nullptr, 0, 0.0f, stream,
                      nullptr, nullptr, 0, send, recv, unit_slot_floats);
}

int thallo_hip_dist_xrows_lm(thallo_dist_t d, thallo_xrows_t x, float* vec, thallo_segs_t first, thallo_segs_t last, thallo_segs_t top, thallo_segs_t bot,
                             thallo_sum_t alphaN, const float* alphaD_partials, const double* s3_partials, const double* q3_partials, int count, int poison,
                             float* alphaD_word, float* betaN_word, float* lm_state, int k, float q_tolerance, thallo_stream_t stream)
{
    return xrows_impl(d, x, vec, first, last, top, bot, 2, alphaN, alphaD_partials, s3_partials, count, poison, alphaD_word, betaN_word, lm_state, k, q_tolerance, stream,
                      nullptr, q3_partials, count);
}

int thallo_hip_dist_error(thallo_dist_t d, int clear, thallo_stream_t stream)
{
    if (!d.ctl) return -(int)hipErrorInvalidValue;
    unsigned v = 0;
    hipError_t e = hipMemcpyAsync(&v, d.ctl + DIST_ERR, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e == hipSuccess && clear && v) e = hipMemsetAsync(d.ctl + DIST_ERR, 0, 4, (hipStream_t)stream);
    if (e != hipSuccess) return -(int)e;
    return (int)v;
}

}  // extern "C"
