// dist_p2p.hip -- host entry points of the multi-GPU device-side exchange (include/thallo_hip.h, dist_device.hpp).
#include "dist_device.hpp"
#include <string.h>

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

int thallo_hip_ipc_alloc(long bytes, void** ptr, void* handle_out64)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    static_assert(DIST_CTL_WORDS <= THALLO_DIST_CTL_WORDS, "ctl layout");
    if (bytes <= 0 || !ptr || !handle_out64) return -(int)hipErrorInvalidValue;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)bytes);
    if (e != hipSuccess) return -(int)e;
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); return -(int)e; }
    memcpy(handle_out64, &h, 64);
    *ptr = p;
    return 0;
}

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
