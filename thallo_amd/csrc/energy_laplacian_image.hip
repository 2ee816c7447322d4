// energy_laplacian_image.hip -- plugin for the image-stencil known-answer energy
// (reference: tests/minimal/laplacian.t:1-14, harness tests/minimal/main.cpp:10-71).
//
//   fit(x,y)  = w*(X(x,y) - A(x,y))
//   regx(x,y) = Gx(x,y) ? X(x,y) - X(x+1,y) : 0     Gx = InBounds(x+1,y+1) as shipped (xguard=0)
//                                                        or InBounds(x+1,y)   (xguard=1, gold.png)
//   regy(x,y) = InBounds(x,y+1) ? X(x,y) - X(x,y+1) : 0
//
// Unknown-wise (gather) form of evalJTF / applyJTJ, i.e. what createjtfcentered /
// createjtjcentered (thallo.t:3603-3712) generate, written by hand.  A 1-channel 5-point
// stencil: neighbours come through L1/L2 (each row is re-read by the two adjacent rows of the
// same 64x8 tile), no LDS needed.  Not a perf target -- it exists to pin the solver chain to
// the reference's gold.png.
#include "device_common.hpp"
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {

constexpr int TW = 64, TH = 8, BLOCK = 256;    // each thread: 2 rows of one column

struct Geo { int W, H, tx, ty, ntiles; };

inline Geo make_geo(int W, int H)
{
    Geo g; g.W = W; g.H = H; g.tx = (W + TW - 1) / TW; g.ty = (H + TH - 1) / TH; g.ntiles = g.tx * g.ty;
    return g;
}
inline int grid_for(const Geo& g)
{
    int cap = thallo_hip_device_cu_count() * 4;
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    cap -= cap % 8;
    return g.ntiles < cap ? g.ntiles : cap;
}

__device__ __forceinline__ bool gx(int x, int y, int W, int H, int xguard)
{
    return xguard ? (x + 1 < W) : (x + 1 < W && y + 1 < H);
}

// (J^T J v)(x,y) for the Laplacian energy; also used for J^T F with v = X and the fit term swapped.
__device__ __forceinline__ float lap_apply(const float* __restrict__ v, int x, int y, int W, int H, int xguard, float c)
{
    const long i = (long)y * W + x;
    float s = 0.0f;
    if (gx(x, y, W, H, xguard))              s += c - v[i + 1];
    if (x > 0 && gx(x - 1, y, W, H, xguard)) s -= v[i - 1] - c;
    if (y + 1 < H)                           s += c - v[i + W];
    if (y > 0)                               s -= v[i - W] - c;
    return s;
}

__global__ __launch_bounds__(BLOCK) void k_cost(Geo g, const float* __restrict__ X, const float* __restrict__ A,
                                                float w, int xguard, float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x = (t.cur % g.tx) * TW + (threadIdx.x % TW);
        for (int k = 0; k < TH / 4; ++k) {
            const int y = (t.cur / g.tx) * TH + (threadIdx.x / TW) + 4 * k;
            if (x < g.W && y < g.H) {
                const long i = (long)y * g.W + x;
                const float c = X[i];
                const float f = w * (c - A[i]);
                float s = f * f;
                if (gx(x, y, g.W, g.H, xguard)) { const float d = c - X[i + 1]; s += d * d; }
                if (y + 1 < g.H)                { const float d = c - X[i + g.W]; s += d * d; }
                acc += 0.5f * s;
            }
        }
    }
    block_store_partial(acc, out, red);
}

__global__ __launch_bounds__(BLOCK) void k_init(Geo g, const float* __restrict__ X, const float* __restrict__ A,
                                                float w, int xguard, float* __restrict__ r, float* __restrict__ z,
                                                float* __restrict__ p_prev, float* __restrict__ delta,
                                                float* __restrict__ diag_out, float* __restrict__ aN_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x = (t.cur % g.tx) * TW + (threadIdx.x % TW);
        for (int k = 0; k < TH / 4; ++k) {
            const int y = (t.cur / g.tx) * TH + (threadIdx.x / TW) + 4 * k;
            if (x < g.W && y < g.H) {
                const long i = (long)y * g.W + x;
                const float c = X[i];
                if (diag_out) {     // raw diag(J^T J) = w^2 + number of difference residuals touching the pixel
                    float d = w * w;
                    if (gx(x, y, g.W, g.H, xguard)) d += 1.0f;
                    if (x > 0 && gx(x - 1, y, g.W, g.H, xguard)) d += 1.0f;
                    if (y + 1 < g.H) d += 1.0f;
                    if (y > 0) d += 1.0f;
                    diag_out[i] = d;
                }
                const float jtf = w * (w * (c - A[i])) + lap_apply(X, x, y, g.W, g.H, xguard, c);
                const float res = -jtf;                 // gauss_newton.t:690
                r[i] = res; z[i] = res;                 // identity preconditioner (no UsePreconditioner)
                p_prev[i] = 0.0f; delta[i] = 0.0f;      // :687
                acc += res * res;
            }
        }
    }
    block_store_partial(acc, aN_out, red);
}

// Pass A of the fused step: delta += alpha*p_in ; p_out = z + beta*p_in   (flat, whole image)
// Pass B: Ap = J^T J p_out ; alphaD partials.
// Two kernels here (a 1-channel KAT energy; the fused single-kernel form is image_warping's).
__global__ __launch_bounds__(BLOCK) void k_pupdate(long n, const float* __restrict__ z, const float* __restrict__ p_in,
                                                    float* __restrict__ p_out, float* __restrict__ delta, int first,
                                                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp)
{
    float alpha = 0.0f, beta = 0.0f;
    if (!first) {
        const float an = sum_partials(aNp.partials, aNp.count);
        alpha = safe_div<false>(an, sum_partials(aDp.partials, aDp.count));
        beta  = safe_div<false>(sum_partials(bNp.partials, bNp.count), an);
    }
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
        const float pi = p_in[i];
        if (!first) delta[i] = delta[i] + alpha * pi;
        p_out[i] = z[i] + beta * pi;
    }
}

__global__ __launch_bounds__(BLOCK) void k_apply(Geo g, float w, int xguard, const float* __restrict__ p,
                                                  float* __restrict__ Ap, float* __restrict__ aD_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x = (t.cur % g.tx) * TW + (threadIdx.x % TW);
        for (int k = 0; k < TH / 4; ++k) {
            const int y = (t.cur / g.tx) * TH + (threadIdx.x / TW) + 4 * k;
            if (x < g.W && y < g.H) {
                const long i = (long)y * g.W + x;
                const float c = p[i];
                const float a = w * (w * c) + lap_apply(p, x, y, g.W, g.H, xguard, c);
                Ap[i] = a;
                acc += c * a;
            }
        }
    }
    block_store_partial(acc, aD_out, red);
}

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

}  // namespace

extern "C" {

int thallo_hip_lapimg_cost(int W, int H, const float* X, const float* A, float w_fit, int xguard,
                           float* cost_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, X, A, w_fit, xguard, cost_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_lapimg_pcg_init(int W, int H, const float* X, const float* A, float w_fit, int xguard,
                               float* r, float* z, float* p_prev, float* delta, float* diag_out, float* aN_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, X, A, w_fit, xguard, r, z, p_prev, delta, diag_out, aN_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_lapimg_pcg_step1(int W, int H, float w_fit, int xguard,
                                const float* z, const float* p_in, float* p_out, float* delta, float* Ap,
                                int first, thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp,
                                float* aD_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    const long n = (long)W * H;
    int fg = (int)((n + BLOCK - 1) / BLOCK); if (fg > 1024) fg = 1024;
    hipLaunchKernelGGL(k_pupdate, dim3(fg), dim3(BLOCK), 0, (hipStream_t)stream, n, z, p_in, p_out, delta, first, aNp, aDp, bNp);
    hipLaunchKernelGGL(k_apply, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, w_fit, xguard, (const float*)p_out, Ap, aD_out);
    int e = check_launch(); return e ? e : grid;
}


int thallo_hip_lapimg_apply_jtj(int W, int H, float w_fit, int xguard, const float* p, float* Ap, float* aD_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_apply, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, w_fit, xguard, p, Ap, aD_out);
    int e = check_launch(); return e ? e : grid;
}

}  // extern "C"
