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
                                                float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ zstate, int zk, float ztol)
{
    __shared__ unsigned last_wg;
    const unsigned tag = ld_agent(d.ctl + DIST_XSEQ) + 1u;
    const int par = (int)(tag & 1u);
    if (!poison) {
        if (x.above >= 0) rows_out(vec, first, inbox_of(d.peer_mail[x.above], x, par, 1));       // I am BELOW my upper neighbour
        if (x.below >= 0) rows_out(vec, last, inbox_of(d.peer_mail[x.below], x, par, 0));
    }
    __threadfence_system();
    __syncthreads();
    if (gridDim.x > 1) {                            // (long rows: several workgroups carry them, the last one to arrive goes on)
        if (threadIdx.x == 0) last_wg = __hip_atomic_fetch_add(d.ctl + DIST_XTICKET, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (!last_wg) return;
    }
    // every row of this rank is out and fenced: the granules may go
    const int slot0 = x.ring0 + (int)(tag & 3u) * 8;
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
        } else {
            float ad = sum_partials(aD_part, nb);
            double q[3] = { 0.0, 0.0, 0.0 };
            for (int i = lane; i < nb; i += THALLO_WAVE) { q[0] += s3[3 * i]; q[1] += s3[3 * i + 1]; q[2] += s3[3 * i + 2]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) q[j] = wave_sum_all_f64(q[j]);
            if (poison) ad = __uint_as_float(0x7fc00000u);
            dist_exchange_iter_wave_seq(d, tag, slot0, ad, q[0], q[1], q[2], s.count == 1 ? s.partials[0] : 0.0f, out0, out1);
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                                                      // (system scope: the neighbours' rows are behind their granules)
    if (x.above >= 0) rows_in(vec, top, inbox_of(d.mail, x, par, 0));
    if (x.below >= 0) rows_in(vec, bot, inbox_of(d.mail, x, par, 1));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.x > 1) __hip_atomic_store(d.ctl + DIST_XTICKET, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d.ctl + DIST_XSEQ, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
                      float* out0, float* out1, float* zstate, int zk, float ztol, thallo_stream_t stream)
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
    else return -(int)hipErrorInvalidValue;
    // rows of up to 32 K floats per direction (a 2048-wide image: 2 ghost rows of 4 channels): ONE workgroup, no ticket; longer ones: 8 workgroups
    const int grid = std::max(tf, tl) <= 32768 ? 1 : 8, block = 256;
    if (mode == 0) hipLaunchKernelGGL(k_xrows<0>, dim3(grid), dim3(block), 0, (hipStream_t)stream, d, x, vec, first, last, top, bot, s, aD_partials, s3_partials, count, poison, out0, out1, zstate, zk, ztol);
    else           hipLaunchKernelGGL(k_xrows<1>, dim3(grid), dim3(block), 0, (hipStream_t)stream, d, x, vec, first, last, top, bot, s, aD_partials, s3_partials, count, poison, out0, out1, (float*)nullptr, 0, 0.0f);
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
