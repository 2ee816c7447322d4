// dist_p2p.hip -- host entry points of the multi-GPU device-side exchange (include/thallo_hip.h, dist_device.hpp).
#include "dist_device.hpp"
#include <string.h>
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
