// spmv_kernels.hip -- CSR sparse matrix-vector product for the MATERIALIZED schedules of the reference:
//   `[Jt][[J]p]`  J and J^T kept as CSR, two SpMVs per PCG iteration   (gauss_newton.t:1448-1525 cusparseJTJMatVec, csrmv x 2)
//   `[[Jt][J]]p`  J^T J formed once (csrgemm, :1394-1441), one SpMV per iteration (:1462-1481)
// selected in a .t by `r.<residual>.J:set_materialize(true)` / `.JtJ:set_materialize(true)` (thallo.t:5661-5690).
// The Jacobians on this path have short rows (1-2 entries per residual row, <= 5 per unknown), so one thread per row with the
// row's entries read back-to-back beats a wave-per-row layout; x is gathered through L2.  HBM-bound: 8 B per non-zero + 8 B per
// row.  The optional dot product rides along (alphaD = p . Ap), one partial per workgroup.
#include "device_common.hpp"
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {
constexpr int BLOCK = 256;
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

template <bool DOT>
__global__ __launch_bounds__(BLOCK) void k_csr_spmv(int rows, const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
                                                     const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w, float* __restrict__ dot_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += (long)gridDim.x * BLOCK) {
        const int b = rowptr[i], e = rowptr[i + 1];
        float s = 0.0f;
        for (int k = b; k < e; ++k) s += val[k] * x[col[k]];
        y[i] = s;
        if (DOT) acc += w[i] * s;
    }
    if (DOT) block_store_partial(acc, dot_out, red);
}

// ---- materialized J of a generated plugin (dsl_plugin.cpp): every residual row has the same number K of entries (its residual's unknown
// accesses: thallo.t:283-307 `nnz_per_entry`), so J is two dense [rows][K] arrays -- ELL, no row pointers; col = -1 marks an entry without
// an unknown (outside the image, or excluded).  One thread per row; J^T is applied by scattering (float atomics, like the reference's
// residual-wise kernels) -- no sort / transpose per GN iteration (gauss_newton.t:1349-1378 csrsort + csr2csc).
//   MODE 0: Ap += J^T (J p)           [Jt][[J]p] with both products in one pass over J
//   MODE 1: Jp = J p                  first half of Jt[Jp] on a materialized J
//   MODE 2: Ap += J^T Jp              second half
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_ell(long rows, int K, const float* __restrict__ val, const int* __restrict__ col,
                                               const float* __restrict__ p, float* __restrict__ Jp, float* __restrict__ Ap)
{
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += (long)gridDim.x * BLOCK) {
        const float* v = val + i * K; const int* c = col + i * K;
        float s = 0.0f;
        if (MODE != 2) { for (int k = 0; k < K; ++k) if (c[k] >= 0) s = s + v[k] * p[c[k]]; }
        else s = Jp[i];
        if (MODE == 1) { Jp[i] = s; continue; }
        for (int k = 0; k < K; ++k) { const float d = v[k]; if (c[k] >= 0 && d != 0.0f) atomicAdd(Ap + c[k], s * d); }
    }
}
// dense J^T J for small n (gauss_newton.t:560-622, 1216-1241: cuBLAS gemm once per GN iteration, gemv per PCG iteration): accumulated from the
// materialized rows as outer products; then Ap = (J^T J) p, one wave per matrix row
__global__ __launch_bounds__(BLOCK) void k_dense_accumulate(long rows, int K, const float* __restrict__ val, const int* __restrict__ col, long n, float* __restrict__ JtJ)
{
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += (long)gridDim.x * BLOCK) {
        const float* v = val + i * K; const int* c = col + i * K;
        for (int a = 0; a < K; ++a) {
            if (c[a] < 0 || v[a] == 0.0f) continue;
            for (int b = 0; b < K; ++b) if (c[b] >= 0 && v[b] != 0.0f) atomicAdd(JtJ + (long)c[a] * n + c[b], v[a] * v[b]);
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_dense_gemv(long n, const float* __restrict__ M, const float* __restrict__ x, float* __restrict__ y)
{
    const int lane = threadIdx.x & 63;
    for (long r = ((long)blockIdx.x * BLOCK + threadIdx.x) >> 6; r < n; r += ((long)gridDim.x * BLOCK) >> 6) {
        float s = 0.0f;
        for (long k = lane; k < n; k += 64) s += M[r * n + k] * x[k];
        s = wave_sum_all(s);
        if (lane == 0) y[r] = s;
    }
}
}  // namespace

extern "C" {

int thallo_hip_ell_apply(int mode, long rows, int K, const float* val, const int* col, const float* p, float* Jp, float* Ap, thallo_stream_t stream)
{
    if (rows < 0 || K < 1 || !val || !col || mode < 0 || mode > 2) return -(int)hipErrorInvalidValue;
    if ((mode != 2 && !p) || (mode != 0 && !Jp) || (mode != 1 && !Ap)) return -(int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    long want = (rows + BLOCK - 1) / BLOCK; if (want > 4096) want = 4096;
    const dim3 g((unsigned)want), b(BLOCK);
    if (mode == 0) hipLaunchKernelGGL(k_ell<0>, g, b, 0, (hipStream_t)stream, rows, K, val, col, p, Jp, Ap);
    else if (mode == 1) hipLaunchKernelGGL(k_ell<1>, g, b, 0, (hipStream_t)stream, rows, K, val, col, p, Jp, Ap);
    else hipLaunchKernelGGL(k_ell<2>, g, b, 0, (hipStream_t)stream, rows, K, val, col, p, Jp, Ap);
    return check_launch();
}

int thallo_hip_dense_jtj_accumulate(long rows, int K, const float* val, const int* col, long n, float* JtJ, thallo_stream_t stream)
{
    if (rows < 0 || K < 1 || !val || !col || n < 1 || !JtJ) return -(int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    long want = (rows + BLOCK - 1) / BLOCK; if (want > 4096) want = 4096;
    hipLaunchKernelGGL(k_dense_accumulate, dim3((unsigned)want), dim3(BLOCK), 0, (hipStream_t)stream, rows, K, val, col, n, JtJ);
    return check_launch();
}

int thallo_hip_dense_gemv(long n, const float* M, const float* x, float* y, thallo_stream_t stream)
{
    if (n < 1 || !M || !x || !y) return -(int)hipErrorInvalidValue;
    long want = (n * 64 + BLOCK - 1) / BLOCK; if (want > 4096) want = 4096;
    hipLaunchKernelGGL(k_dense_gemv, dim3((unsigned)want), dim3(BLOCK), 0, (hipStream_t)stream, n, M, x, y);
    return check_launch();
}

int thallo_hip_csr_spmv(int rows, const int* rowptr, const int* col, const float* val, const float* x, float* y,
                        const float* dot_with, float* dot_out, thallo_stream_t stream)
{
    if (rows < 0 || !rowptr || !x || !y || ((dot_with == nullptr) != (dot_out == nullptr))) return -(int)hipErrorInvalidValue;
    if (rows == 0) { return 0; }
    long want = ((long)rows + BLOCK - 1) / BLOCK;
    int grid = thallo_hip_device_cu_count() * 4; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS;
    if (want < grid) grid = (int)want;
    if (dot_out) hipLaunchKernelGGL(k_csr_spmv<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, col, val, x, y, dot_with, dot_out);
    else         hipLaunchKernelGGL(k_csr_spmv<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, rows, rowptr, col, val, x, y, dot_with, dot_out);
    int e = check_launch(); return e ? e : grid;
}

}  // extern "C"
