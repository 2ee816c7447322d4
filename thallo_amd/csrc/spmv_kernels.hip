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
}  // namespace

extern "C" {

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
