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
// dense J^T J for small n (gauss_newton.t:560-622, 1216-1241: cuBLAS gemm once per GN iteration, gemv per PCG iteration): formed from the
// materialized rows tile by tile on the matrix cores; then Ap = (J^T J) p, one wave per matrix row
typedef float f32x16 __attribute__((ext_vector_type(16)));
// One WAVE per 32 x 32 tile (unknowns a0.., unknowns b0..) of the lower triangle, on the matrix cores: the rows of J come 64 at a time, each lane densifies ITS row's
// entries that fall into the tile's two column ranges into LDS (a row may name an unknown twice: +=), and the tile gains sum_i J[i][a] J[i][b] as
// 32 x v_mfma_f32_32x32x2_f32 per 64 rows (f32 in, f32 accumulate: an fmaf chain over the rows in their order -- the sum is the same in every run, which the
// atomic scatter this replaces was not).  Operand / result lane maps: cdna_hip_programming.md "fragment layout" (A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31];
// result register v of lane l = element (row (v & 3) + 8 (v >> 2) + 4 (l >> 5), column l & 31)).  The tile is ADDED to J^T J (several residual groups accumulate
// into one matrix, launch after launch on one stream) and mirrored into the upper triangle.
__global__ __launch_bounds__(64) void k_dense_accumulate(long rows, int K, const float* __restrict__ val, const int* __restrict__ col, long n, float* __restrict__ JtJ)
{
    __shared__ float PA[64][33], PB[64][33];
    const long ta = blockIdx.y, tb = blockIdx.x;
    if (tb > ta) return;
    const int a0 = (int)(ta * 32), b0 = (int)(tb * 32);
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    for (long r0 = 0; r0 < rows; r0 += 64) {
        for (int e = 0; e < 33; ++e) { PA[l][e] = 0.0f; PB[l][e] = 0.0f; }
        const long i = r0 + l;
        if (i < rows) {
            const float* v = val + i * K; const int* c = col + i * K;
            for (int k = 0; k < K; ++k) {
                const int ck = c[k]; const float vk = v[k];
                if (ck < 0 || vk == 0.0f) continue;
                if (ck >= a0 && ck < a0 + 32) PA[l][ck - a0] += vk;
                if (ck >= b0 && ck < b0 + 32) PB[l][ck - b0] += vk;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 64; q += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(PA[q + h][m], PB[q + h][m], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const long i = a0 + (v & 3) + 8 * (v >> 2) + 4 * h, j = b0 + m;
        if (i >= n || j >= n) continue;
        if (ta != tb) { JtJ[i * n + j] += acc[v]; JtJ[j * n + i] += acc[v]; }
        else JtJ[i * n + j] += acc[v];            // a diagonal tile holds both triangles itself (P_A = P_B: the product is symmetric bit for bit)
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
// sparse J^T J for a non-constant J ([[Jt][J]]p, gauss_newton.t:1394-1441: csrgemm once per GN iteration): the PATTERN of J^T J follows from the
// rows' unknown indices and is built once per Init on the host (dsl_plugin.cpp); `dest[(i*K + a)*K + b]` = position of the product
// v[i][a] * v[i][b] in the CSR values (-1: no such entry).  The numeric phase is this scatter.
__global__ __launch_bounds__(BLOCK) void k_jtj_scatter(long rows, int K, const float* __restrict__ val, const int* __restrict__ dest, float* __restrict__ out)
{
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += (long)gridDim.x * BLOCK) {
        const float* v = val + i * K; const int* d = dest + i * K * K;
        for (int a = 0; a < K; ++a) {
            const float va = v[a];
            if (va == 0.0f) continue;
            for (int b = 0; b < K; ++b) { const int q = d[a * K + b]; const float w = va * v[b]; if (q >= 0 && w != 0.0f) atomicAdd(out + q, w); }
        }
    }
}

// ---- dense direct solve (gauss_newton.t:1280-1328 cublasDirectSolve: LU, inverse, gemv -- compiled out there by enable_direct_solve = false,
// :22; opt-in here).  J^T J is symmetric positive definite when J has full column rank, so: blocked right-looking Cholesky A = L L^T in
// place (lower triangle, row-major n x n), then L y = b, L^T x = y.  NB = 32: the diagonal block and one panel block live in LDS.
constexpr int NB = 32;
// one workgroup: factor the diagonal block at (k,k), then panel rows below it: L21 = A21 L11^-T.  info[0] = 1 + row of a non-positive pivot.
__global__ __launch_bounds__(BLOCK) void k_potrf_panel(long n, long k, float* __restrict__ A, int* __restrict__ info)
{
    __shared__ float D[NB][NB + 1];
    __shared__ int bad;
    const int t = threadIdx.x, nb = (int)((n - k) < NB ? (n - k) : NB);
    if (t == 0) bad = 0;
    for (int e = t; e < NB * NB; e += BLOCK) { const int r = e / NB, c = e % NB; D[r][c] = (r < nb && c <= r) ? A[(k + r) * n + k + c] : 0.0f; }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {                 // unblocked Cholesky of the nb x nb block, column by column
        if (t == 0) { const float d = D[j][j]; if (!(d > 0.0f)) { if (!bad) bad = j + 1; D[j][j] = 1.0f; } else D[j][j] = sqrtf(d); }
        __syncthreads();
        if (t > j && t < nb) D[t][j] /= D[j][j];
        __syncthreads();
        for (int e = t; e < nb * nb; e += BLOCK) { const int r = e / nb, c = e % nb; if (c > j && r >= c) D[r][c] -= D[r][j] * D[c][j]; }
        __syncthreads();
    }
    if (t == 0 && bad && info[0] == 0) info[0] = (int)k + bad;
    for (int e = t; e < nb * nb; e += BLOCK) { const int r = e / nb, c = e % nb; if (c <= r) A[(k + r) * n + k + c] = D[r][c]; }
    for (long r = k + nb + t; r < n; r += BLOCK) {  // one panel row per thread: x L11^T = a  (forward substitution over the block's columns)
        float x[NB];
        for (int c = 0; c < nb; ++c) {
            float v = A[r * n + k + c];
            for (int q = 0; q < c; ++q) v -= x[q] * D[c][q];
            x[c] = v / D[c][c];
        }
        for (int c = 0; c < nb; ++c) A[r * n + k + c] = x[c];
    }
}
// trailing update A22 -= L21 L21^T (lower triangle only), one NB x NB tile per WAVE on the matrix cores: the tile's product P Q^T (P = the panel rows of the
// tile's rows, Q = of its columns, both NB x nb) is 16 x v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, an fmaf chain over q in natural order, i.e. the same
// arithmetic as a scalar loop.  Operands: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31] of the step's two k; result register v of lane l is
// element (row (v & 3) + 8 (v >> 2) + 4 (l >> 5), column l & 31) (cdna_hip_programming.md, fragment layout).  The panels go through LDS once (coalesced reads).
__global__ __launch_bounds__(64) void k_syrk_tile(long n, long k, int nb, float* __restrict__ A)
{
    static_assert(NB == 32, "one 32x32x2 MFMA tile");
    __shared__ float P[NB][NB + 1], Q[NB][NB + 1];
    const long base = k + nb;
    const long tr = blockIdx.y, tc = blockIdx.x;
    if (tc > tr) return;
    const long r0 = base + tr * NB, c0 = base + tc * NB;
    const int l = threadIdx.x;
    for (int e = l; e < NB * NB; e += 64) {
        const int i = e / NB, j = e % NB;
        P[i][j] = (r0 + i < n && j < nb) ? A[(r0 + i) * n + k + j] : 0.0f;
        Q[i][j] = (c0 + i < n && j < nb) ? A[(c0 + i) * n + k + j] : 0.0f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    const int m = l & 31, h = l >> 5;
#pragma unroll
    for (int q = 0; q < NB; q += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(P[m][q + h], Q[m][q + h], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const long i = r0 + (v & 3) + 8 * (v >> 2) + 4 * h, j = c0 + m;
        if (i < n && j < n && j <= i) A[i * n + j] -= acc[v];
    }
}
// one workgroup: x = L^-T L^-1 b, blockwise (the block's triangular solve by thread 0..nb-1 in LDS, the update of the rest by everyone)
__global__ __launch_bounds__(BLOCK) void k_potrs(long n, const float* __restrict__ L, const float* __restrict__ b, float* __restrict__ x)
{
    __shared__ float xs[NB];
    __shared__ float D[NB][NB + 1];
    const int t = threadIdx.x;
    for (long i = t; i < n; i += BLOCK) x[i] = b[i];
    __syncthreads();
    for (long k = 0; k < n; k += NB) {             // forward: L y = b
        const int nb = (int)((n - k) < NB ? (n - k) : NB);
        for (int e = t; e < nb * nb; e += BLOCK) D[e / nb][e % nb] = L[(k + e / nb) * n + k + e % nb];
        __syncthreads();
        if (t == 0) for (int c = 0; c < nb; ++c) { float v = x[k + c]; for (int q = 0; q < c; ++q) v -= D[c][q] * xs[q]; xs[c] = v / D[c][c]; x[k + c] = xs[c]; }
        __syncthreads();
        for (long r = k + nb + t; r < n; r += BLOCK) { float s = 0.0f; for (int q = 0; q < nb; ++q) s += L[r * n + k + q] * xs[q]; x[r] -= s; }
        __syncthreads();
    }
    for (long k = (n - 1) / NB * NB; k >= 0; k -= NB) {   // backward: L^T x = y
        const int nb = (int)((n - k) < NB ? (n - k) : NB);
        for (int e = t; e < nb * nb; e += BLOCK) D[e / nb][e % nb] = L[(k + e / nb) * n + k + e % nb];
        __syncthreads();
        if (t == 0) for (int c = nb - 1; c >= 0; --c) { float v = x[k + c]; for (int q = c + 1; q < nb; ++q) v -= D[q][c] * xs[q]; xs[c] = v / D[c][c]; x[k + c] = xs[c]; }
        __syncthreads();
        for (long r = t; r < k; r += BLOCK) { float s = 0.0f; for (int q = 0; q < nb; ++q) s += L[(k + q) * n + r] * xs[q]; x[r] -= s; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ per-owner instance lists (generated plugins' gather through index maps, dsl_plugin.cpp build_incidence)
// col[el * K + q]: the flat unknown index that slot q of residual instance el touches (-1: none).  A slot belongs to the owner group when its base >= 0; its owner
// is the pixel (col - base) / ch of the group's index space.  An instance counts ONCE per distinct owner.
struct IncSlots { long base[THALLO_HIP_INC_MAX_SLOTS]; int ch[THALLO_HIP_INC_MAX_SLOTS]; };
__device__ __forceinline__ long inc_owner(const int* __restrict__ col, long el, int K, const IncSlots& S, long npix, int q)
{
    if (S.base[q] < 0) return -1;
    const int u = col[el * K + q]; if (u < 0) return -1;
    const long px = ((long)u - S.base[q]) / S.ch[q];
    return px >= 0 && px < npix ? px : -1;
}
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void k_inc_count_fill(const int* __restrict__ col, long n, int K, IncSlots S, long npix, int* __restrict__ ptr, int* __restrict__ cursor, int* __restrict__ els)
{
    for (long el = (long)blockIdx.x * BLOCK + threadIdx.x; el < n; el += (long)gridDim.x * BLOCK) {
        for (int q = 0; q < K; ++q) {
            const long px = inc_owner(col, el, K, S, npix, q);
            if (px < 0) continue;
            bool dup = false;
            for (int j = 0; j < q; ++j) dup = dup || inc_owner(col, el, K, S, npix, j) == px;
            if (dup) continue;
            if (FILL) els[ptr[px] + atomicAdd(cursor + px, 1)] = (int)el;
            else atomicAdd(ptr + px + 1, 1);
        }
    }
}
// ptr[0] = 0, ptr[1 .. m]: counts -> inclusive prefix sums, in place; one workgroup, 4 elements per thread and pass, the carry in a register.  *total = the sum (as a
// long: more than 2^31 - 1 pairs is the caller's error to raise)
__global__ __launch_bounds__(1024) void k_inc_scan(int* __restrict__ ptr, long m, long* __restrict__ total)
{
    __shared__ long wsum[16];
    __shared__ long carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { carry_s = 0; ptr[0] = 0; }
    __syncthreads();
    for (long base = 1; base <= m; base += 4096) {
        const long i0 = base + 4L * t;
        long v[4], run = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = i0 + j <= m ? (long)ptr[i0 + j] : 0; run += v[j]; v[j] = run; }
        long inc = run;                                     // inclusive scan of the threads' sums over the wave, then over the 16 waves
        for (int d = 1; d < 64; d <<= 1) { const long o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const long excl = before + inc - run;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j <= m) { const long x = excl + v[j]; ptr[i0 + j] = x > 0x7fffffffL ? 0x7fffffff : (int)x; }
        __syncthreads();
        if (t == 1023) carry_s = excl + run;
        __syncthreads();
    }
    if (t == 0) *total = carry_s;
}
// every owner's list in ascending instance order (the atomic cursor filled it in arrival order): what the host-side construction produced, and what makes the sums
// of the gather kernels the same from run to run.  One thread per owner; insertion sort for short lists, heap sort above (in place, O(L log L)).
__global__ __launch_bounds__(BLOCK) void k_inc_sort(const int* __restrict__ ptr, long npix, int* __restrict__ els)
{
    for (long px = (long)blockIdx.x * BLOCK + threadIdx.x; px < npix; px += (long)gridDim.x * BLOCK) {
        int* a = els + ptr[px]; const int L = ptr[px + 1] - ptr[px];
        if (L <= 32) {
            for (int i = 1; i < L; ++i) { const int x = a[i]; int j = i - 1; while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; } a[j + 1] = x; }
            continue;
        }
        auto sift = [&](int root, int end) {
            for (;;) {
                int c = 2 * root + 1; if (c >= end) break;
                if (c + 1 < end && a[c + 1] > a[c]) ++c;
                if (a[root] >= a[c]) break;
                const int tmp = a[root]; a[root] = a[c]; a[c] = tmp; root = c;
            }
        };
        for (int i = L / 2 - 1; i >= 0; --i) sift(i, L);
        for (int e = L - 1; e > 0; --e) { const int tmp = a[0]; a[0] = a[e]; a[e] = tmp; sift(0, e); }
    }
}
}  // namespace

extern "C" {

static inline int inc_grid(long n) { const long g = (n + BLOCK - 1) / BLOCK; return (int)(g < 1 ? 1 : g > 4096 ? 4096 : g); }
int thallo_hip_incidence_count(const int* col, long n, int K, const long* slot_base, const int* slot_ch, long npix, int* ptr, long* total_dev, thallo_stream_t stream)
{
    if (K < 1 || K > THALLO_HIP_INC_MAX_SLOTS || npix < 0 || n < 0) return -(int)hipErrorInvalidValue;
    IncSlots S; for (int q = 0; q < THALLO_HIP_INC_MAX_SLOTS; ++q) { S.base[q] = q < K ? slot_base[q] : -1; S.ch[q] = q < K && slot_ch[q] > 0 ? slot_ch[q] : 1; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ptr, 0, (size_t)(npix + 1) * sizeof(int), s) != hipSuccess) return -(int)hipErrorInvalidValue;
    if (n > 0) hipLaunchKernelGGL(k_inc_count_fill<false>, dim3(inc_grid(n)), dim3(BLOCK), 0, s, col, n, K, S, npix, ptr, (int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_inc_scan, dim3(1), dim3(1024), 0, s, ptr, npix, total_dev);
    return check_launch();
}
int thallo_hip_incidence_fill(const int* col, long n, int K, const long* slot_base, const int* slot_ch, long npix, const int* ptr, int* cursor, int* els, thallo_stream_t stream)
{
    if (K < 1 || K > THALLO_HIP_INC_MAX_SLOTS || npix < 0 || n < 0) return -(int)hipErrorInvalidValue;
    IncSlots S; for (int q = 0; q < THALLO_HIP_INC_MAX_SLOTS; ++q) { S.base[q] = q < K ? slot_base[q] : -1; S.ch[q] = q < K && slot_ch[q] > 0 ? slot_ch[q] : 1; }
    hipStream_t s = (hipStream_t)stream;
    if (npix > 0 && hipMemsetAsync(cursor, 0, (size_t)npix * sizeof(int), s) != hipSuccess) return -(int)hipErrorInvalidValue;
    if (n > 0) hipLaunchKernelGGL(k_inc_count_fill<true>, dim3(inc_grid(n)), dim3(BLOCK), 0, s, col, n, K, S, npix, const_cast<int*>(ptr), cursor, els);
    if (npix > 0) hipLaunchKernelGGL(k_inc_sort, dim3(inc_grid(npix)), dim3(BLOCK), 0, s, ptr, npix, els);
    return check_launch();
}

int thallo_hip_jtj_scatter(long rows, int K, const float* val, const int* dest, float* out, thallo_stream_t stream)
{
    if (rows < 0 || K < 1 || !val || !dest || !out) return -(int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    long want = (rows + BLOCK - 1) / BLOCK; if (want > 4096) want = 4096;
    hipLaunchKernelGGL(k_jtj_scatter, dim3((unsigned)want), dim3(BLOCK), 0, (hipStream_t)stream, rows, K, val, dest, out);
    return check_launch();
}

int thallo_hip_dense_cholesky_solve(long n, float* A, const float* b, float* x, int* info, thallo_stream_t stream)
{
    if (n < 1 || n > 8192 || !A || !b || !x || !info) return -(int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return -(int)hipErrorInvalidValue;
    for (long k = 0; k < n; k += NB) {
        const int nb = (int)((n - k) < NB ? (n - k) : NB);
        hipLaunchKernelGGL(k_potrf_panel, dim3(1), dim3(BLOCK), 0, s, n, k, A, info);
        const long rest = n - k - nb;
        if (rest > 0) { const unsigned tiles = (unsigned)((rest + NB - 1) / NB); hipLaunchKernelGGL(k_syrk_tile, dim3(tiles, tiles), dim3(64), 0, s, n, k, nb, A); }
    }
    hipLaunchKernelGGL(k_potrs, dim3(1), dim3(BLOCK), 0, s, n, A, b, x);
    return check_launch();
}

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
    const unsigned tiles = (unsigned)((n + 31) / 32);
    if (tiles > 65535u) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dense_accumulate, dim3(tiles, tiles), dim3(64), 0, (hipStream_t)stream, rows, K, val, col, n, JtJ);
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
