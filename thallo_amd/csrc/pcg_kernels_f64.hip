// pcg_kernels_f64.hip -- the energy-independent kernels of the PCG loop in double precision (Thallo_InitializationParameters::doublePrecision = 1:
// precision.t:3-6 makes thallo_float a double, so unknowns, the solver's vectors and every reduction are doubles; the reference's own comment there --
// "switch to double to check for precision issues in the solver" -- says what the mode is for).  The reference-shaped unfused schedule
// (gauss_newton.t:712-731 PCGInit1_Finish, :774-787 PCGStep1_Finish, :801-843 PCGStep2, :889-899 PCGStep3, :901-906 PCGLinearUpdate), one launch per step:
// a diagnostic mode, not a hot path -- the float kernels of pcg_kernels.hip and the fused energy kernels are the product's.  Scalars stay on the device:
// every reduction leaves per-workgroup partials, thallo_hip_f64_finish adds them up in index order into one word, and the next kernel reads words.
#include "device_common.hpp"

using namespace thallo;

namespace {

constexpr int B64 = 256;
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }
inline int grid_for(long n) { long g = (n + B64 - 1) / B64; if (g < 1) g = 1; if (g > THALLO_HIP_MAX_PARTIALS) g = THALLO_HIP_MAX_PARTIALS; return (int)g; }

__device__ __forceinline__ double wave_sum_d(double v) { for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64); return v; }
// per-workgroup partial of a per-thread value: wave butterflies, then the four wave sums in wave order
__device__ __forceinline__ void block_partial_d(double v, double* out)
{
    __shared__ double red[B64 / 64];
    const double s = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < B64 / 64; ++w) t += red[w]; out[blockIdx.x] = t; }
}
__device__ __forceinline__ double guarded_invert_d(double d) { const double s = 1.0 + sqrt(d); return 1.0 / (s * s); }      // gauss_newton.t:638-648 as device_common.hpp restates it
__device__ __forceinline__ double safe_div_d(double num, double den) { return den != 0.0 ? num / den : 0.0; }

__global__ __launch_bounds__(B64) void k64_init_finish(const double* __restrict__ r, double* __restrict__ pre, double* __restrict__ z, double* __restrict__ p, long n, int use_precond,
                                                        double* __restrict__ out)
{
    double acc = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) {
        const double m = use_precond ? guarded_invert_d(pre[i]) : 1.0;
        const double ri = r[i], zi = m * ri;
        pre[i] = m; z[i] = zi; p[i] = zi;
        acc += ri * zi;
    }
    block_partial_d(acc, out);
}
__global__ __launch_bounds__(B64) void k64_dot(const double* __restrict__ a, const double* __restrict__ b, long n, double* __restrict__ out)
{
    double acc = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) acc += a[i] * b[i];
    block_partial_d(acc, out);
}
__global__ __launch_bounds__(B64) void k64_step2(double* __restrict__ delta, double* __restrict__ r, double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ Ap,
                                                  const double* __restrict__ pre, long n, const double* __restrict__ aN, const double* __restrict__ aD, double* __restrict__ out)
{
    const double alpha = safe_div_d(aN[0], aD[0]);
    double acc = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) {
        delta[i] += alpha * p[i];
        const double ri = r[i] - alpha * Ap[i], zi = pre[i] * ri;
        r[i] = ri; z[i] = zi;
        acc += zi * ri;
    }
    block_partial_d(acc, out);
}
__global__ __launch_bounds__(B64) void k64_step3(double* __restrict__ p, const double* __restrict__ z, long n, const double* __restrict__ bN, const double* __restrict__ aN)
{
    const double beta = safe_div_d(bN[0], aN[0]);
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) p[i] = z[i] + beta * p[i];
}
__global__ __launch_bounds__(B64) void k64_linear_update(double* __restrict__ X, const double* __restrict__ delta, long n)
{
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) X[i] += delta[i];
}
// one workgroup: the partials in index order (lane-strided, then the wave butterflies, then the wave sums in order) -- the same word in every run
__global__ __launch_bounds__(B64) void k64_finish(const double* __restrict__ partials, int count, double* __restrict__ word)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += B64) acc += partials[i];
    __shared__ double one[1];
    block_partial_d(acc, one);          // (blockIdx.x = 0)
    __syncthreads();
    if (threadIdx.x == 0) word[0] = one[0];
}

// ---- the Levenberg-Marquardt set in double (gauss_newton.t:929-969 PCGSaveSSq / PCGComputeCtC / PCGFinalizeDiagonal, :774-787 PCGStep1_Finish, :801-886 PCGStep2 and its
// two halves around the residual reset); the float forms are pcg_kernels.hip's k_lm_*.  LM divides blindly (gauss_newton.t:226-234).
__global__ __launch_bounds__(B64) void k64_lm_finalize(const double* __restrict__ diag, double* __restrict__ SSq, double* __restrict__ CtC, double* __restrict__ pre,
                                                        const double* __restrict__ r, double* __restrict__ b, double* __restrict__ z, long n, double radius, double min_lm, double max_lm,
                                                        int save_ssq, int use_precond, double* __restrict__ out)
{
    double acc = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) {
        const double d = diag[i], ri = r[i];
        double ss;
        if (save_ssq) { ss = use_precond ? guarded_invert_d(d) : 1.0; SSq[i] = ss; } else ss = SSq[i];
        const double unclamped = d / radius, cm = (1.0 / ss) / radius;
        const double c = fmin(fmax(unclamped, min_lm * cm), max_lm * cm);
        const double m = 1.0 / (c + radius * unclamped), zi = m * ri;
        CtC[i] = c; pre[i] = m; b[i] = ri; z[i] = zi;
        acc += ri * zi;
    }
    block_partial_d(acc, out);
}
__global__ __launch_bounds__(B64) void k64_lm_step1_finish(double* __restrict__ Ap, const double* __restrict__ CtC, const double* __restrict__ p, long n, double* __restrict__ out)
{
    double acc = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) { const double a = Ap[i] + CtC[i] * p[i]; Ap[i] = a; acc += p[i] * a; }
    block_partial_d(acc, out);
}
__global__ __launch_bounds__(B64) void k64_lm_step2(double* __restrict__ delta, double* __restrict__ r, double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ Ap,
                                                     const double* __restrict__ pre, const double* __restrict__ b, long n, const double* __restrict__ aN, const double* __restrict__ aD,
                                                     double* __restrict__ out_bn, double* __restrict__ out_q)
{
    const double alpha = aN[0] / aD[0];
    double acc = 0.0, q = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) {
        const double di = delta[i] + alpha * p[i], ri = r[i] - alpha * Ap[i], zi = pre[i] * ri;
        delta[i] = di; r[i] = ri; z[i] = zi;
        acc += zi * ri; q += 0.5 * di * (ri + b[i]);
    }
    block_partial_d(acc, out_bn);
    __syncthreads();
    block_partial_d(q, out_q);
}
__global__ __launch_bounds__(B64) void k64_lm_axpy(double* __restrict__ delta, const double* __restrict__ p, long n, const double* __restrict__ aN, const double* __restrict__ aD)
{
    const double alpha = aN[0] / aD[0];
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) delta[i] += alpha * p[i];
}
__global__ __launch_bounds__(B64) void k64_lm_reset(double* __restrict__ r, const double* __restrict__ b, const double* __restrict__ Ad, const double* __restrict__ pre, double* __restrict__ z,
                                                     const double* __restrict__ delta, long n, double* __restrict__ out_bn, double* __restrict__ out_q)
{
    double acc = 0.0, q = 0.0;
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) {
        const double ri = b[i] - Ad[i], zi = pre[i] * ri;
        r[i] = ri; z[i] = zi;
        acc += zi * ri; q += 0.5 * delta[i] * (ri + b[i]);
    }
    block_partial_d(acc, out_bn);
    __syncthreads();
    block_partial_d(q, out_q);
}
__global__ __launch_bounds__(B64) void k64_lm_step3(double* __restrict__ p, const double* __restrict__ z, long n, const double* __restrict__ bN, const double* __restrict__ aN)
{
    const double beta = bN[0] / aN[0];
    for (long i = (long)blockIdx.x * B64 + threadIdx.x; i < n; i += (long)gridDim.x * B64) p[i] = z[i] + beta * p[i];
}

}  // namespace

extern "C" {

int thallo_hip_f64_lm_finalize_diagonal(const double* diag, double* SSq, double* CtC, double* pre, const double* r, double* b, double* z, long n, double radius, double min_lm_diagonal,
                                        double max_lm_diagonal, int save_ssq, int use_preconditioner, double* partials_out, thallo_stream_t stream)
{
    if (!diag || !SSq || !CtC || !pre || !r || !b || !z || !partials_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_lm_finalize, dim3(g), dim3(B64), 0, (hipStream_t)stream, diag, SSq, CtC, pre, r, b, z, n, radius, min_lm_diagonal, max_lm_diagonal, save_ssq, use_preconditioner, partials_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_lm_step1_finish(double* Ap, const double* CtC, const double* p, long n, double* partials_out, thallo_stream_t stream)
{
    if (!Ap || !CtC || !p || !partials_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_lm_step1_finish, dim3(g), dim3(B64), 0, (hipStream_t)stream, Ap, CtC, p, n, partials_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_lm_step2(double* delta, double* r, double* z, const double* p, const double* Ap, const double* pre, const double* b, long n, const double* alphaN_word,
                            const double* alphaD_word, double* betaN_out, double* q_out, thallo_stream_t stream)
{
    if (!delta || !r || !z || !p || !Ap || !pre || !b || !alphaN_word || !alphaD_word || !betaN_out || !q_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_lm_step2, dim3(g), dim3(B64), 0, (hipStream_t)stream, delta, r, z, p, Ap, pre, b, n, alphaN_word, alphaD_word, betaN_out, q_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_lm_step2_first_half(double* delta, const double* p, long n, const double* alphaN_word, const double* alphaD_word, thallo_stream_t stream)
{
    if (!delta || !p || !alphaN_word || !alphaD_word || n < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k64_lm_axpy, dim3(grid_for(n)), dim3(B64), 0, (hipStream_t)stream, delta, p, n, alphaN_word, alphaD_word);
    return check_launch();
}
int thallo_hip_f64_lm_step2_second_half(double* r, const double* b, const double* Adelta, const double* pre, double* z, const double* delta, long n, double* betaN_out, double* q_out,
                                        thallo_stream_t stream)
{
    if (!r || !b || !Adelta || !pre || !z || !delta || !betaN_out || !q_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_lm_reset, dim3(g), dim3(B64), 0, (hipStream_t)stream, r, b, Adelta, pre, z, delta, n, betaN_out, q_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_lm_step3(double* p, const double* z, long n, const double* betaN_word, const double* alphaN_word, thallo_stream_t stream)
{
    if (!p || !z || !betaN_word || !alphaN_word || n < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k64_lm_step3, dim3(grid_for(n)), dim3(B64), 0, (hipStream_t)stream, p, z, n, betaN_word, alphaN_word);
    return check_launch();
}

int thallo_hip_f64_init_finish(const double* r, double* pre, double* z, double* p, long n, int use_preconditioner, double* partials_out, thallo_stream_t stream)
{
    if (!r || !pre || !z || !p || !partials_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_init_finish, dim3(g), dim3(B64), 0, (hipStream_t)stream, r, pre, z, p, n, use_preconditioner, partials_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_dot(const double* a, const double* b, long n, double* partials_out, thallo_stream_t stream)
{
    if (!a || !b || !partials_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_dot, dim3(g), dim3(B64), 0, (hipStream_t)stream, a, b, n, partials_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_step2(double* delta, double* r, double* z, const double* p, const double* Ap, const double* pre, long n, const double* alphaN_word, const double* alphaD_word,
                         double* partials_out, thallo_stream_t stream)
{
    if (!delta || !r || !z || !p || !Ap || !pre || !alphaN_word || !alphaD_word || !partials_out || n < 1) return -(int)hipErrorInvalidValue;
    const int g = grid_for(n);
    hipLaunchKernelGGL(k64_step2, dim3(g), dim3(B64), 0, (hipStream_t)stream, delta, r, z, p, Ap, pre, n, alphaN_word, alphaD_word, partials_out);
    const int e = check_launch(); return e ? e : g;
}
int thallo_hip_f64_step3(double* p, const double* z, long n, const double* betaN_word, const double* alphaN_word, thallo_stream_t stream)
{
    if (!p || !z || !betaN_word || !alphaN_word || n < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k64_step3, dim3(grid_for(n)), dim3(B64), 0, (hipStream_t)stream, p, z, n, betaN_word, alphaN_word);
    return check_launch();
}
int thallo_hip_f64_linear_update(double* X, const double* delta, long n, thallo_stream_t stream)
{
    if (!X || !delta || n < 1) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k64_linear_update, dim3(grid_for(n)), dim3(B64), 0, (hipStream_t)stream, X, delta, n);
    return check_launch();
}
int thallo_hip_f64_finish(const double* partials, int count, double* word, thallo_stream_t stream)
{
    if (!partials || !word || count < 1 || count > THALLO_HIP_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k64_finish, dim3(1), dim3(B64), 0, (hipStream_t)stream, partials, count, word);
    return check_launch();
}

}  // extern "C"
