// energy_sfs.hip -- plugin for examples/shape_from_shading/shape_from_shading.t:1-112 (refined depth X from
// an RGB-D frame: depth fit + SH-shading gradient term + Laplacian of back-projected points).
//
// Residuals per pixel q = (x,y) (guarded loads return 0 outside the image, thallo.t:876-883):
//   fit(q)  = [D(q)>0] w_p (X(q) - D(q))
//   sh_h(q) = [1<=x<=W-2, 1<=y<=H-2] w_g edgeMaskR(q) (BI(q) - BI(q+ex))
//   sh_v(q) = [same guard]           w_g edgeMaskC(q) (BI(q) - BI(q+ey))
//   reg(q)  = [valid(q)] w_s (4 P(q) - P(q-ex) - P(q-ey) - P(q+ex) - P(q+ey)),  P(c) = X(c) ((cx-u_x)/f_x, (cy-u_y)/f_y, 1)
//   BI(c)   = [D(c-ex)>0, D(c)>0, D(c-ey)>0] (B(n(c)) - I(c)): SH shading of the normal built from X(c), X(c-ex), X(c-ey)
//   valid(q)= D>0 at q and its 4 neighbours and |X(q)-X(nbr)| < 0.01 (comparisons: zero derivative, ad.t:824-829)
// w_p, w_s, w_g are the square roots of the caller's parameters (shape_from_shading.t:27).
//
// The reference materialises BI as a computed array with gradient images once per GN iteration
// (`B_I_comp:get`, precompute kernels gauss_newton.t:979-986, thallo.t:4046-4094); so does k_precompute:
//   G(c) = (dBI/dX(c), dBI/dX(c-ex), dBI/dX(c-ey), BI(c))   float4 per pixel, 3-wide forward-mode AD
//   Wt(c) = (h, k) = shading row weights incl. guard and edge masks;  fl(c): bit0 D>0, bit1 reg valid
// With those, J is a chain of radius-1 stencils and J^T(Jv) is computed by two gather kernels (no atomics):
//   k_rows : per residual pixel q   U_h = h^2 (dB(q) - dB(q+ex)), U_v = k^2 (dB(q) - dB(q+ey)), R_c = valid w_s (Lap of coef_c v)
//            with dB(c) = G.x v(c) + G.y v(c-ex) + G.z v(c-ey)            [dB = BI itself when evaluating J^T F]
//   k_gather: per unknown pixel i   T(c) = U_h(c) - U_h(c-ex) + U_v(c) - U_v(c-ey)
//            out(i) = w_p^2 [D>0] v(i) + G.x(i) T(i) + G.y(i+ex) T(i+ex) + G.z(i+ey) T(i+ey)
//                     + w_s sum_c coef_c(i) (4 R_c(i) - sum_nbr R_c(nbr))
// 1 float per pixel: the whole working set of a 2048^2 frame is ~250 MB of small planes; neighbours come through
// L1/L2.  Not tuned yet (round 1: correctness + structure).
#include <stdlib.h>
#include <stdint.h>
#include "device_common.hpp"
#include "../../include/thallo_hip.h"
#include "sfs_pair.hpp"
namespace thallo { const char* env_switch(const char* name); }      // solver.cpp: the one table of the library's environment switches

using namespace thallo;

namespace {

constexpr int TW = 64, TH = 4, BLOCK = 256;
// W x H = local image (a row slab may carry 2 ghost rows per side); a kernel visits local rows [ra, rb);
// yoff = global row of local row 0 and Hg = global image height (pixel coordinates and the image-border guard are global)
struct Geo { int W, H, ra, rb, yoff, Hg, tx, ty, ntiles; };
inline Geo make_geo(int W, int H, int ra, int rb, int yoff, int Hg)
{
    Geo g; g.W = W; g.H = H; g.ra = ra; g.rb = rb; g.yoff = yoff; g.Hg = Hg;
    g.tx = (W + TW - 1) / TW; g.ty = (rb - ra + TH - 1) / TH; g.ntiles = g.tx * g.ty; return g;
}
inline int grid_for(const Geo& g)
{
    int cap = thallo_hip_device_cu_count() * 4; if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS; cap -= cap % 8;
    return g.ntiles < cap ? g.ntiles : cap;
}
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

struct Cam { float wp, ws, wg, fx, fy, ux, uy; float L[9]; };

#define FOR_EACH_PIXEL(g) \
    for (TileSweep t_((g).ntiles); t_.valid(); t_.next()) \
        for (int x = (t_.cur % (g).tx) * TW + (threadIdx.x % TW), y = (g).ra + (t_.cur / (g).tx) * TH + (threadIdx.x / TW), once_ = 1; once_; once_ = 0) \
            if (x < (g).W && y < (g).rb)

__device__ __forceinline__ float at(const float* __restrict__ a, int x, int y, int W, int H) { return (x >= 0 && x < W && y >= 0 && y < H) ? a[(long)y * W + x] : 0.0f; }

struct J3 { float v, d0, d1, d2; };
__device__ __forceinline__ J3 k3(float c) { J3 r = { c, 0.f, 0.f, 0.f }; return r; }
__device__ __forceinline__ J3 operator+(J3 a, J3 b) { J3 r = { a.v + b.v, a.d0 + b.d0, a.d1 + b.d1, a.d2 + b.d2 }; return r; }
__device__ __forceinline__ J3 operator-(J3 a, J3 b) { J3 r = { a.v - b.v, a.d0 - b.d0, a.d1 - b.d1, a.d2 - b.d2 }; return r; }
__device__ __forceinline__ J3 operator*(J3 a, J3 b) { J3 r = { a.v * b.v, a.d0 * b.v + a.v * b.d0, a.d1 * b.v + a.v * b.d1, a.d2 * b.v + a.v * b.d2 }; return r; }
__device__ __forceinline__ J3 operator*(J3 a, float c) { J3 r = { a.v * c, a.d0 * c, a.d1 * c, a.d2 * c }; return r; }

// BI and its three partials at pixel (x,y)   (shape_from_shading.t:40-80), from the values at (x,y), (x-1,y), (x,y-1) (0 outside the image)
__device__ __forceinline__ J3 eval_BI_vals(const Cam& cm, float Dl, float Dc, float Du, float Xc, float Xl, float Xu, float Ic, float Il, float Iu, int x, int yg)
{
    if (!(Dl > 0.0f && Dc > 0.0f && Du > 0.0f)) return k3(0.0f);
    const J3 c = { Xc, 1.f, 0.f, 0.f }, l = { Xl, 0.f, 1.f, 0.f }, u = { Xu, 0.f, 0.f, 1.f };
    const float i = (float)x, j = (float)yg;
    const J3 nx = (u * (c - l)) * (1.0f / cm.fy);
    const J3 ny = (l * (c - u)) * (1.0f / cm.fx);
    const J3 nz = (nx * ((cm.ux - i) / cm.fx) + ny * ((cm.uy - j) / cm.fy)) - (l * u) * (1.0f / (cm.fx * cm.fy));
    const J3 sq = nx * nx + ny * ny + nz * nz;
    J3 inv;
    if (sq.v > 0.0f) { inv.v = 1.0f / sqrtf(sq.v); const float k = -0.5f * inv.v / sq.v; inv.d0 = k * sq.d0; inv.d1 = k * sq.d1; inv.d2 = k * sq.d2; }
    else inv = k3(1.0f);
    const J3 n0 = inv * nx, n1 = inv * ny, n2 = inv * nz;
    const float* L = cm.L;
    J3 B = k3(L[0]);
    B = B + n1 * L[1]; B = B + n2 * L[2]; B = B + n0 * L[3];
    B = B + (n0 * n1) * L[4]; B = B + (n1 * n2) * L[5];
    B = B + (((n0 * n0) * -1.0f - n1 * n1) + (n2 * n2) * 2.0f) * L[6];
    B = B + (n2 * n0) * L[7]; B = B + (n0 * n0 - n1 * n1) * L[8];
    const float I = Ic * 0.5f + 0.25f * (Il + Iu);
    return B - k3(I);
}
__device__ __forceinline__ J3 eval_BI(const Cam& cm, const float* __restrict__ X, const float* __restrict__ D, const float* __restrict__ Im,
                                      int x, int y, int W, int H, int yoff)
{
    const float Dl = at(D, x - 1, y, W, H), Dc = at(D, x, y, W, H), Du = at(D, x, y - 1, W, H);
    if (!(Dl > 0.0f && Dc > 0.0f && Du > 0.0f)) return k3(0.0f);
    return eval_BI_vals(cm, Dl, Dc, Du, at(X, x, y, W, H), at(X, x - 1, y, W, H), at(X, x, y - 1, W, H),
                        at(Im, x, y, W, H), at(Im, x - 1, y, W, H), at(Im, x, y - 1, W, H), x, y + yoff);
}

__device__ __forceinline__ float coef(const Cam& cm, int c, int x, int y) { return c == 0 ? ((float)x - cm.ux) / cm.fx : c == 1 ? ((float)y - cm.uy) / cm.fy : 1.0f; }

// precompute: G, Wt, fl
__global__ __launch_bounds__(BLOCK) void k_precompute(Geo g, Cam cm, const float* __restrict__ X, const float* __restrict__ D, const float* __restrict__ Im,
                                                       const unsigned char* __restrict__ mR, const unsigned char* __restrict__ mC,
                                                       float4* __restrict__ G, float2* __restrict__ Wt, unsigned char* __restrict__ fl)
{
    FOR_EACH_PIXEL(g) {
        const long i = (long)y * g.W + x;
        const J3 b = eval_BI(cm, X, D, Im, x, y, g.W, g.H, g.yoff);
        G[i] = make_float4(b.d0, b.d1, b.d2, b.v);
        const int yg = y + g.yoff;
        const bool inner = x >= 1 && x + 1 < g.W && yg >= 1 && yg + 1 < g.Hg;
        Wt[i] = inner ? make_float2(cm.wg * (float)mR[i], cm.wg * (float)mC[i]) : make_float2(0.f, 0.f);
        const float xc = X[i];
        unsigned char f = D[i] > 0.0f ? 1 : 0;
        bool valid = f;
        const int dx[4] = { -1, 0, 1, 0 }, dy[4] = { 0, -1, 0, 1 };
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int xn = x + dx[d], yn = y + dy[d];
            valid = valid && at(D, xn, yn, g.W, g.H) > 0.0f && fabsf(xc - at(X, xn, yn, g.W, g.H)) < 0.01f;
        }
        if (valid) f |= 2;
        fl[i] = f;
    }
}

// cost from the precomputed planes (gauss_newton.t:1067-1079)
__global__ __launch_bounds__(BLOCK) void k_cost(Geo g, Cam cm, const float* __restrict__ X, const float* __restrict__ D,
                                                const float4* __restrict__ G, const float2* __restrict__ Wt, const unsigned char* __restrict__ fl,
                                                float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    FOR_EACH_PIXEL(g) {
        const long i = (long)y * g.W + x;
        const unsigned char f = fl[i];
        const float xc = X[i];
        float s = 0.0f;
        if (f & 1) { const float e = cm.wp * (xc - D[i]); s += e * e; }
        const float2 w = Wt[i];
        if (w.x != 0.0f || w.y != 0.0f) {           // inner pixel: q+ex, q+ey exist
            const float b0 = G[i].w;
            const float eh = w.x * (b0 - G[i + 1].w), ev = w.y * (b0 - G[i + g.W].w);
            s += eh * eh + ev * ev;
        }
        if (f & 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float a = 4.0f * (coef(cm, c, x, y + g.yoff) * xc);
                a -= coef(cm, c, x - 1, y + g.yoff) * at(X, x - 1, y, g.W, g.H); a -= coef(cm, c, x, y - 1 + g.yoff) * at(X, x, y - 1, g.W, g.H);
                a -= coef(cm, c, x + 1, y + g.yoff) * at(X, x + 1, y, g.W, g.H); a -= coef(cm, c, x, y + 1 + g.yoff) * at(X, x, y + 1, g.W, g.H);
                a *= cm.ws; s += a * a;
            }
        }
        acc += 0.5f * s;
    }
    block_store_partial(acc, out, red);
}

// rows: FROM_X = true evaluates residual values (J^T F pass), else J v
template <bool FROM_X>
__global__ __launch_bounds__(BLOCK) void k_rows(Geo g, Cam cm, const float* __restrict__ v, const float4* __restrict__ G,
                                                const float2* __restrict__ Wt, const unsigned char* __restrict__ fl,
                                                float2* __restrict__ U, float* __restrict__ R)
{
    const long N = (long)g.W * g.H;
    FOR_EACH_PIXEL(g) {
        const long i = (long)y * g.W + x;
        const float2 w = Wt[i];
        float uh = 0.0f, uv = 0.0f;
        if (w.x != 0.0f || w.y != 0.0f) {
            float b0, bx, by;
            if (FROM_X) { b0 = G[i].w; bx = G[i + 1].w; by = G[i + g.W].w; }
            else {
                const float vc = v[i], vl = v[i - 1], vu = v[i - g.W];      // inner pixel: all in range
                const float4 g0 = G[i], gx = G[i + 1], gy = G[i + g.W];
                b0 = g0.x * vc + g0.y * vl + g0.z * vu;
                bx = gx.x * v[i + 1] + gx.y * vc + gx.z * v[i + 1 - g.W];
                by = gy.x * v[i + g.W] + gy.y * v[i + g.W - 1] + gy.z * vc;
            }
            uh = w.x * (w.x * (b0 - bx)); uv = w.y * (w.y * (b0 - by));
        }
        U[i] = make_float2(uh, uv);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (fl[i] & 2) {
            const float vc = v[i], vl = at(v, x - 1, y, g.W, g.H), vu = at(v, x, y - 1, g.W, g.H), vr = at(v, x + 1, y, g.W, g.H), vd = at(v, x, y + 1, g.W, g.H);
            float a[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                a[c] = cm.ws * (4.0f * (coef(cm, c, x, y + g.yoff) * vc) - coef(cm, c, x - 1, y + g.yoff) * vl - coef(cm, c, x, y - 1 + g.yoff) * vu
                                - coef(cm, c, x + 1, y + g.yoff) * vr - coef(cm, c, x, y + 1 + g.yoff) * vd);
            r0 = a[0]; r1 = a[1]; r2 = a[2];
        }
        R[i] = r0; R[N + i] = r1; R[2 * N + i] = r2;
    }
}

__device__ __forceinline__ float T_at(const float2* __restrict__ U, int x, int y, int W, int H)
{   // T(c) = U_h(c) - U_h(c-ex) + U_v(c) - U_v(c-ey) ; 0 outside the image
    if (x < 0 || x >= W || y < 0 || y >= H) return 0.0f;
    const long i = (long)y * W + x;
    const float2 u = U[i];
    float t = u.x + u.y;
    if (x > 0) t -= U[i - 1].x;
    if (y > 0) t -= U[i - W].y;
    return t;
}

// gather: MODE 0 = PCGInit1 (r = -J^T F, z = r, p_prev = 0, delta = 0, alphaN), MODE 1 = PCGStep1 (Ap, alphaD)
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gather(Geo g, Cam cm, const float* __restrict__ v, const float* __restrict__ D,
                                                  const float4* __restrict__ G, const float2* __restrict__ U, const float* __restrict__ R,
                                                  const unsigned char* __restrict__ fl, float* __restrict__ out, float* __restrict__ z,
                                                  float* __restrict__ p_prev, float* __restrict__ delta, float* __restrict__ part_out,
                                                  const float* __restrict__ rs = nullptr, double* __restrict__ s3_out = nullptr)
{   // MODE 1 with rs / s3_out: also the Sums3 of the single-reduction PCG form (no preconditioner in this energy: M^-1 = 1)
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    const long N = (long)g.W * g.H;
    float acc = 0.0f; Sums3 sm;
    FOR_EACH_PIXEL(g) {
        const long i = (long)y * g.W + x;
        const float vc = v[i];
        float s = 0.0f;
        if (fl[i] & 1) s += cm.wp * (cm.wp * (MODE == 0 ? vc - D[i] : vc));
        s += G[i].x * T_at(U, x, y, g.W, g.H);
        if (x + 1 < g.W) s += G[i + 1].y * T_at(U, x + 1, y, g.W, g.H);
        if (y + 1 < g.H) s += G[i + g.W].z * T_at(U, x, y + 1, g.W, g.H);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* Rc = R + c * N;
            const float lap = 4.0f * Rc[i] - at(Rc, x - 1, y, g.W, g.H) - at(Rc, x, y - 1, g.W, g.H) - at(Rc, x + 1, y, g.W, g.H) - at(Rc, x, y + 1, g.W, g.H);
            s += cm.ws * (coef(cm, c, x, y + g.yoff) * lap);
        }
        if (MODE == 0) { const float r = -s; out[i] = r; z[i] = r; p_prev[i] = 0.0f; delta[i] = 0.0f; acc += r * r; }
        else { out[i] = s; acc += vc * s; if (s3_out) sm.add(1.0f, rs[i], s); }
    }
    block_store_partial(acc, part_out, red);
    if (MODE == 1 && s3_out) block_store_sums3(sm, s3_out, redd);
}

// ------------------------------------------------------------------------------------------ fused J^T(J v): one LDS-tiled kernel
// k_rows + k_gather above move ~94 B/pixel (v, G, Wt, fl in, U and R out; then v, G, U, R, fl in, out): the intermediate planes U (8 B)
// and R (12 B) make a round trip through HBM.  Fused, they live in LDS: for a 64x16 output tile
//   A  per pixel q of the tile +- 1:  U(q) = (h^2 (dB(q) - dB(q+ex)), k^2 (dB(q) - dB(q+ey))) and R_c(q), computed exactly as k_rows does,
//      from v, G, Wt, fl in global memory (the 3x3 neighbourhoods overlap: L1 / L2 serve the re-reads)                       -> LDS
//   B  per output pixel i:            T at i, i+ex, i+ey from U;  out(i) as k_gather;  alphaD (and the three double sums)
// two workgroup barriers per tile, 24 KB of LDS per workgroup (6 workgroups per CU).
// Bytes per pixel from HBM: read v 4, G 16, Wt 8, fl 1; write out 4 = 33 (the algorithmic bytes of this formulation);
// the tile +- 1 halo (66x18 over 64x16 = 16 %) and the stencil neighbourhoods are re-read through L1 / L2.
// MODE 0 is the J^T F pass of PCGInit1: dB := BI (G.w), v := X, the fit term uses X - D, and the outputs are r = -J^T F, z = r,
// p_prev = 0, delta = 0, alphaN partials.  Same expressions in the same order as k_rows / k_gather: bit-identical outputs.
// `gate` (may be NULL): a device word; non-zero = skip this launch (LM: the PCG loop ended early on the device, solver.cpp).
// MODE 1 with D != NULL: D is the LM diagonal CtC and the output is (J^T J + CtC) v -- PCGStep1_Finish (gauss_newton.t:774-787) folded into the apply.
constexpr int FW = 64, FH = 16;
constexpr int UW = FW + 2, UH = FH + 2;        // U, R: tile +- 1
constexpr int BW = FW + 3, BH = FH + 3;        // b0: tile -1 .. +1 and one more column / row (U(q) needs b0 at q, q+ex, q+ey)
struct FusedTile {
    float b0[BW * BH];                         // the shading row's value at every pixel of that range: dB(q) = G(q) . (v(q), v(q-ex), v(q-ey)), once per pixel
    float uh[UW * UH], uv[UW * UH];
    float r0[UW * UH], r1[UW * UH], r2[UW * UH];
    float cx[UW + 2], cy[UH + 2];              // coef_0 = (x - u_x) / f_x per column, coef_1 = (y - u_y) / f_y per row of the tile +- 2 (coef_2 = 1):
                                               // one division per column / row and tile instead of ten per pixel
};

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_fused(Geo g, Cam cm, const float* __restrict__ v, const float* __restrict__ D,
                                                 const float4* __restrict__ G, const float2* __restrict__ Wt, const unsigned char* __restrict__ fl,
                                                 float* __restrict__ out, float* __restrict__ z, float* __restrict__ p_prev, float* __restrict__ delta,
                                                 float* __restrict__ part_out, const float* __restrict__ rs, double* __restrict__ s3_out,
                                                 const unsigned* __restrict__ gate, FinArgs fin)
{
    __shared__ FusedTile T;
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;
    const int W = g.W, H = g.H;
    const int ftx = (W + FW - 1) / FW, fty = (g.rb - g.ra + FH - 1) / FH, ntiles = ftx * fty;
    float acc = 0.0f; Sums3 sm;
    for (TileSweep t(ntiles); t.valid(); t.next()) {
        const int x0 = (t.cur % ftx) * FW, y0 = g.ra + (t.cur / ftx) * FH;
        if (threadIdx.x < UW + 2) T.cx[threadIdx.x] = coef(cm, 0, x0 + (int)threadIdx.x - 2, 0);
        else if (threadIdx.x >= 128 && threadIdx.x < 128 + UH + 2) T.cy[threadIdx.x - 128] = coef(cm, 1, 0, y0 + (int)threadIdx.x - 128 - 2 + g.yoff);
        // ---- A0: dB on the tile -1 .. +2 (round 2: each pixel's value once, into LDS -- the U of three sites uses it; before, every site re-gathered
        //      three G rows and seven v values: 3 x the G traffic through L1 and twice the arithmetic; same expression, same bits)
        for (int idx = threadIdx.x; idx < BW * BH; idx += BLOCK) {
            const int ly = idx / BW, lx = idx - ly * BW;
            const int qx = x0 + lx - 1, qy = y0 + ly - 1;
            float b = 0.0f;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                const long q = (long)qy * W + qx;
                const float4 g0 = G[q];
                if (MODE == 0) b = g0.w;
                else b = g0.x * v[q] + g0.y * at(v, qx - 1, qy, W, H) + g0.z * at(v, qx, qy - 1, W, H);
            }
            T.b0[idx] = b;
        }
        __syncthreads();
        // ---- A: U and R on the tile +- 1
        for (int idx = threadIdx.x; idx < UW * UH; idx += BLOCK) {
            const int ly = idx / UW, lx = idx - ly * UW;
            const int qx = x0 + lx - 1, qy = y0 + ly - 1;
            float uh = 0.f, uv = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                const long q = (long)qy * W + qx;
                const float2 w = Wt[q];
                if (w.x != 0.0f || w.y != 0.0f) {
                    const int j = ly * BW + lx;
                    const float b0 = T.b0[j], bx = T.b0[j + 1], by = T.b0[j + BW];
                    uh = w.x * (w.x * (b0 - bx)); uv = w.y * (w.y * (b0 - by));
                }
                if (fl[q] & 2) {
                    const float vc = v[q], vl = at(v, qx - 1, qy, W, H), vu = at(v, qx, qy - 1, W, H), vr = at(v, qx + 1, qy, W, H), vd = at(v, qx, qy + 1, W, H);
                    // coef_c at q and its four neighbours: c = 0 varies with x only, c = 1 with y only, c = 2 is 1  (tables start at tile - 2)
                    const float xm = T.cx[lx], xc = T.cx[lx + 1], xp = T.cx[lx + 2], ym = T.cy[ly], yc = T.cy[ly + 1], yp = T.cy[ly + 2];
                    r0 = cm.ws * (4.0f * (xc * vc) - xm * vl - xc * vu - xp * vr - xc * vd);
                    r1 = cm.ws * (4.0f * (yc * vc) - yc * vl - ym * vu - yc * vr - yp * vd);
                    r2 = cm.ws * (4.0f * (1.0f * vc) - 1.0f * vl - 1.0f * vu - 1.0f * vr - 1.0f * vd);
                }
            }
            T.uh[idx] = uh; T.uv[idx] = uv; T.r0[idx] = r0; T.r1[idx] = r1; T.r2[idx] = r2;
        }
        __syncthreads();
        // ---- B: outputs
#pragma unroll
        for (int k = 0; k < FH / (BLOCK / FW); ++k) {
            const int lx = threadIdx.x % FW, ly = threadIdx.x / FW + k * (BLOCK / FW);
            const int x = x0 + lx, y = y0 + ly;
            if (x < W && y < g.rb) {
                const long i = (long)y * W + x;
                const int iu = (ly + 1) * UW + (lx + 1);
                // T(c) = U_h(c) - U_h(c-ex) + U_v(c) - U_v(c-ey), terms outside the image dropped (same order and guards as T_at)
                auto Tat = [&](int j, int cx, int cy) {
                    if (cx < 0 || cx >= W || cy < 0 || cy >= H) return 0.0f;
                    float tt = T.uh[j] + T.uv[j];
                    if (cx > 0) tt -= T.uh[j - 1];
                    if (cy > 0) tt -= T.uv[j - UW];
                    return tt;
                };
                const float vc = v[i];
                float s = 0.0f;
                if (fl[i] & 1) s += cm.wp * (cm.wp * (MODE == 0 ? vc - D[i] : vc));
                s += G[i].x * Tat(iu, x, y);
                if (x + 1 < W) s += G[i + 1].y * Tat(iu + 1, x + 1, y);
                if (y + 1 < H) s += G[i + W].z * Tat(iu + UW, x, y + 1);
                const float* Rp[3] = { T.r0, T.r1, T.r2 };
                const float ci[3] = { T.cx[lx + 2], T.cy[ly + 2], 1.0f };
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* Rc = Rp[c];
                    const float lap = 4.0f * Rc[iu] - Rc[iu - 1] - Rc[iu - UW] - Rc[iu + 1] - Rc[iu + UW];
                    s += cm.ws * (ci[c] * lap);
                }
                if (MODE == 0) { const float r = -s; out[i] = r; z[i] = r; p_prev[i] = 0.0f; delta[i] = 0.0f; acc += r * r; }
                else { if (D) s += D[i] * vc; out[i] = s; acc += vc * s; if (s3_out) sm.add(1.0f, rs[i], s); }      // (MODE 1: D = the LM diagonal CtC or NULL)
            }
        }
        __syncthreads();
    }
    if (MODE == 1 && s3_out) block_finish_sums(acc, sm, part_out, s3_out, fin, red, redd);
    else block_store_partial(acc, part_out, red);
}

// ------------------------------------------------------------------------------------------ marching J^T(J v): no LDS tile, no barrier
// The same J^T(J v) as k_fused<1> (same expressions, same order, same guards), shaped like energy_image_warping_march.hip: a WAVE owns a
// column strip of 64 pixels (lane l: column x0 + l; lanes 2..61 produce output, the outer two on each side carry the radius-2 x halo, so
// strips overlap by 4 columns) and marches down R rows of it.  x neighbours come from the neighbouring lane (DPP wave shifts), y neighbours
// from the lane's own registers: at the step that takes row t the lane forms dB(t), U_h(t), U_v(t-1), R_c(t-1), T(t-1) and the output of
// row t-2.  A segment [ya, yb) therefore takes the rows ya-2 .. yb+1 (4 halo rows, re-read through L2 by the neighbouring segments, which
// run at the same time on the same XCD); rows are prefetched three steps ahead into registers (v, G, Wt, flags of row t; r / CtC of the output
// row t-2).  Every plane is read ONCE per pixel and wave (k_fused: 22 vector loads per pixel through L1), ~1/2 of its VALU work, no LDS traffic.
constexpr int MS_USE = 60, MS_NT = 256;
constexpr int MS_OCC = 2, MS_WG_PER_CU = 2;      // registers for 2 workgroups (8 waves) per CU; grid sized for that many (tools/sfs_probe.py sweeps)
struct MsGeo { int W, H, ra, rb, yoff, R, nstrips, total; };
struct MsRaw { float4 g; float2 w; float v; unsigned f; float rs, ct, pv, av, dl, bb, mi; };      // one row of one lane as loaded (pv: p_{k-1} of the row, PUPD / UPD; av, dl: Ap_{k-1}, delta, UPD; bb: b of the OUTPUT row, mi: M^-1 of the row, LMQ)

// value of lane-1 / lane+1 (wave_shr:1 / wave_shl:1); a lane without a source reads 0 (bound_ctrl) -- lanes 0 / 63 produce no output.
// mov_dpp has no tied "old" operand: one v_mov_b32_dpp per exchange, and the compiler may fold it into the consuming instruction.
__device__ __forceinline__ float ms_left(float v)  { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float ms_right(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ void ms_fence() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void ms_mv(float& d, const float& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
__device__ __forceinline__ void ms_mv(unsigned& d, const unsigned& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
// a prefetch slot moves into fresh registers with real v_mov instructions, so that its refill can be issued into the SAME registers right
// behind (energy_image_warping_march.hip `take`: otherwise the compiler computes in place and the refill turns into a blocking load)
template <bool SUMS, bool CTC, bool INIT, bool PUPD = false, bool UPD = false, bool LMQ = false>
__device__ __forceinline__ void ms_take(MsRaw& d, const MsRaw& s)
{
    d.pv = 0.0f; d.av = 0.0f; d.dl = 0.0f; d.bb = 0.0f; d.mi = 1.0f;
    if (LMQ) { ms_mv(d.bb, s.bb); ms_mv(d.mi, s.mi); }
    if (PUPD || UPD) ms_mv(d.pv, s.pv);
    if (UPD) { ms_mv(d.av, s.av); ms_mv(d.dl, s.dl); }
    ms_mv(d.g.x, s.g.x); ms_mv(d.g.y, s.g.y); ms_mv(d.g.z, s.g.z); ms_mv(d.w.x, s.w.x); ms_mv(d.w.y, s.w.y); ms_mv(d.v, s.v); ms_mv(d.f, s.f);
    d.g.w = 0.0f; d.rs = 0.0f; d.ct = 0.0f;
    if (INIT) ms_mv(d.g.w, s.g.w);
    if (SUMS && !UPD) ms_mv(d.rs, s.rs);
    if (CTC) ms_mv(d.ct, s.ct);
}

// INIT: the J^T F pass of PCGInit1 (k_fused<0>): v := X, dB := BI (G.w), ctc := D (the fit term uses X - D), outputs r = -J^T F, z = r,
// p_prev = 0, delta = 0 and the alphaN partials.
// DIAG (with INIT, LM only): also the raw diag(J^T J) of k_diag -- the same seven shading rows in the same order, from the rows the lane already
// holds (G.x(i), G.y(i+ex), G.z(i+ey) and the row weights of i, i+-ex, i+ey, i+ey-ex, i-ey, i-ey+ex: zero outside the inner image).
// PUPD (LM, one GPU): PCGStep3 rides along -- v is z, the lane forms p_k = z + beta_{k-1} p_{k-1} for every row it takes (halo rows and lanes
// redundantly, like image_warping's marching kernel), stores p_k for its own rows and applies (J^T J + CtC) to it.  p_{k-1} and p_k are different buffers.
struct MsPupd { const float* p_in; float* p_out; thallo_sum_t aN, bN; int first; };
// UPD (GN, one GPU): the whole PCG iteration in this launch -- k_pcg_update rides along.  v is r_{k-1}; per row taken the lane forms
// r_k = r_{k-1} - alpha_{k-1} Ap_{k-1} and p_k = r_k + beta_{k-1} p_{k-1} (no preconditioner in this energy; halo rows and lanes redundantly), stores r_k, p_k
// and delta += alpha_{k-1} p_{k-1} for its own rows, applies J^T J to p_k, and the three sums take r_k from registers.  r, Ap and p ping-pong
// (the neighbours' halo rows re-read the previous iteration's planes while the owner writes this iteration's).
struct MsUpd { float* r_out; const float* A_in; const float* p_in; float* p_out; float* delta; thallo_sum_t aN, aD, bN; int first; int lm; const float* b; const float* pre;
               // the finish of iteration k-1 DEFERRED into this launch (thallo_hip_sfs_pcg_iter_deferred): prev_nb > 0 -- aD.partials are that iteration's alphaD partials,
               // prev_s3 its {N, S1, S2} partials (another buffer than this launch writes); every workgroup adds them up for itself, workgroup 0 leaves the two words
               const double* prev_s3 = nullptr; int prev_nb = 0; float* aD_word = nullptr; float* bN_word = nullptr; };
// LMQ (with UPD, SUMS, CTC; LM on one GPU): the whole LM iteration in this launch -- the vector update of PCGStep2 (iteration k-1's scalars), PCGStep3, (J^T J + CtC) p_k,
// with the LM preconditioner M^-1 (`pre`: z = M^-1 r, one more plane per row taken), and besides alphaD and {N, S1, S2} the three sums {U, T1, T2} of q's expansion in alpha (device_common.hpp SumsQ: delta_k, r_k, p_k, A p_k are in registers, b is
// one more plane read at the output row); the last workgroup finishes alphaD_k, betaN_k, q_{k+1} and the zeta test (block_finish_sums_lm).
template <bool SUMS, bool CTC, bool INIT, bool DIAG, int OCC, bool PUPD = false, bool UPD = false, bool LMQ = false>
__global__ __launch_bounds__(MS_NT, OCC) void k_march(MsGeo g, Cam cm, const float* __restrict__ v, const float* __restrict__ ctc,
                                                      const float4* __restrict__ G, const float2* __restrict__ Wt, const unsigned char* __restrict__ fl,
                                                      float* __restrict__ out, float* __restrict__ part_out, const float* __restrict__ rs,
                                                      double* __restrict__ s3_out, const unsigned* __restrict__ gate, FinArgs fin,
                                                      float* __restrict__ z, float* __restrict__ p_prev, float* __restrict__ delta, float* __restrict__ diag,
                                                      MsPupd pu = MsPupd{}, MsUpd up = MsUpd{}, LmFin lmf = LmFin{})
{
    __shared__ float red[16];
    __shared__ double redd[(LMQ ? 6 : 3) * MS_NT / 64];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;
    float beta = 0.0f;
    if (PUPD && !pu.first) beta = safe_div<true>(sum_partials(pu.bN.partials, pu.bN.count), sum_partials(pu.aN.partials, pu.aN.count));      // as k_pupdate (LM)
    float alpha = 0.0f;
    if (UPD && !LMQ && !up.first && up.prev_nb > 0) {                                  // the deferred finish: last_workgroup_totals' order, block_finish_sums' arithmetic -- the same bits
        float ad, an; double t3[3];
        last_workgroup_totals<3, true>(up.aD.partials, up.prev_s3, nullptr, up.prev_nb, up.aN, red, redd, ad, an, t3);
        alpha = safe_div<false>(an, ad);
        double bnd = t3[0] - 2.0 * (double)alpha * t3[1] + (double)alpha * (double)alpha * t3[2];
        if (!(bnd > 0.0)) bnd = 0.0;
        const float bnf = (float)bnd;
        beta = safe_div<false>(bnf, an);
        if (blockIdx.x == 0 && threadIdx.x == 0) { up.aD_word[0] = ad; up.bN_word[0] = bnf; }
        lds_barrier();                                                                  // (red / redd are the end-of-launch reduction's too)
    } else
    if (UPD && !up.first) {                                                                                                                      // as k_pcg_update
        const float an = sum_partials(up.aN.partials, up.aN.count);
        const float ad = sum_partials(up.aD.partials, up.aD.count), bn = sum_partials(up.bN.partials, up.bN.count);
        alpha = up.lm ? safe_div<true>(an, ad) : safe_div<false>(an, ad);             // (LM divides blindly, gauss_newton.t:226-234)
        beta  = up.lm ? safe_div<true>(bn, an) : safe_div<false>(bn, an);
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));       // (uniform: rows, row addresses and guards stay scalar)
    const int W = g.W, H = g.H;
    // XCD-aware placement as in the image_warping marching kernel: workgroups b and b+8 share an XCD; group b%8 owns a contiguous range of
    // (band of 4 segments, strip) ids, x-adjacent strips first
    int strip = 0, ya = 0, yb = 0;
    {
        const int NG = (gridDim.x % 8) == 0 ? 8 : 1;
        const int grp = blockIdx.x % NG, l = blockIdx.x / NG;
        const long lo = (long)g.total * grp / NG, hi = (long)g.total * (grp + 1) / NG;
        const long id = lo + l;
        if (id < hi) {
            strip = (int)(id % g.nstrips);
            const int seg = (int)(id / g.nstrips) * (MS_NT / 64) + wave;
            ya = g.ra + seg * g.R; yb = ya + g.R;
            if (yb > g.rb) yb = g.rb;
            if (ya > g.rb) ya = g.rb;
        }
    }
    const bool work = ya < yb;
    const int x = strip * MS_USE - 2 + lane;
    const bool xin = x >= 0 && x < W;
    const bool xout = xin && lane >= 2 && lane <= 61;
    const unsigned xc_ = x < 0 ? 0u : x > W - 1 ? (unsigned)(W - 1) : (unsigned)x;
    // coef_0 at x-1, x, x+1 (coef_2 = 1); coef_1 per row, carried
    const float cxm = coef(cm, 0, x - 1, 0), cxc = coef(cm, 0, x, 0), cxp = coef(cm, 0, x + 1, 0);

    float acc = 0.0f; Sums3 sm; SumsQ sq;
    if (work) {
        const int t_first = ya - 2, t_last = yb + 1;
        // loads are unconditional (addresses clamped, validity applied when the row is taken): a load under a branch is waited for at once.
        // Row base pointers are wave-uniform; the lane adds its (unsigned, clamped) column.
        auto issue = [&](MsRaw& s, int t) {
            const int tc = t < 0 ? 0 : t > H - 1 ? H - 1 : t;
            const long rowoff = (long)tc * W;
            s.g = (G + rowoff)[xc_]; s.w = (Wt + rowoff)[xc_]; s.v = (v + rowoff)[xc_];
            if (PUPD) s.pv = (pu.p_in + rowoff)[xc_];
            if (UPD) {
                s.pv = (up.p_in + rowoff)[xc_]; s.av = (up.A_in + rowoff)[xc_];
                const int td = t < ya ? ya : t > yb - 1 ? yb - 1 : t;          // delta: the segment's own rows only
                s.dl = up.delta ? (up.delta + (long)td * W)[xc_] : 0.0f;          // (NULL: the iteration leaves delta alone -- the caller's ring of p planes, thallo_hip_linear_update_n)
                if (LMQ) s.mi = (up.pre + rowoff)[xc_];
            }
            // the aligned dword that holds the pixel's flags byte (shifted when the row is taken): a byte load leaves a zero-extension for the
            // compiler to place, and it places it at the loop latch behind a wait for the fresh load (energy_image_warping_march.hip)
            s.f = *reinterpret_cast<const unsigned*>(fl + ((rowoff + xc_) & ~3L));
            if (SUMS || CTC) {
                const int yo = t - 2 < ya ? ya : t - 2 > yb - 1 ? yb - 1 : t - 2;
                const long ro = (long)yo * W;
                if (SUMS && !UPD) s.rs = (rs + ro)[xc_];
                if (CTC) s.ct = (ctc + ro)[xc_];
                if (LMQ) s.bb = (up.b + ro)[xc_];
            }
        };
        // State carried from row to row: rings of three indexed by the row modulo 3.  Three rows are taken per loop trip, so every index is a
        // compile-time constant and nothing is shifted from register to register (energy_image_warping_march.hip `win`).
        float Vv[3] = { 0.f, 0.f, 0.f }, dB[3] = { 0.f, 0.f, 0.f }, Uh[3] = { 0.f, 0.f, 0.f }, Uv[3] = { 0.f, 0.f, 0.f }, Tt[3] = { 0.f, 0.f, 0.f };
        float Gx[3] = { 0.f, 0.f, 0.f }, Gy[3] = { 0.f, 0.f, 0.f }, Gz[3] = { 0.f, 0.f, 0.f }, Cy[3] = { 0.f, 0.f, 0.f }, Wy[3] = { 0.f, 0.f, 0.f };
        float Wx[3] = { 0.f, 0.f, 0.f };       // (DIAG) the rows' h weights, zero outside the image like Wy
        float Rk[3] = { 0.f, 0.f, 0.f };       // (UPD) r_k of the rows
        float Dk[3] = { 0.f, 0.f, 0.f };       // (LMQ) delta_k of the rows
        float Mk[3] = { 1.f, 1.f, 1.f };       // (LMQ) M^-1 of the rows
        unsigned Fl[3] = { 0u, 0u, 0u };
        bool Wn[3] = { false, false, false };
        float Rr[3][3] = { { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f }, { 0.f, 0.f, 0.f } };
        MsRaw slot[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) slot[j] = MsRaw{};
        // no prologue: the loop starts three rows early with empty slots and its refills are the first loads (one path into the loop header)
        for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = t0 + j;
                const int k0 = j, k1 = (j + 2) % 3, k2 = (j + 1) % 3;          // ring slots of rows t, t-1, t-2 (and t-3 = t)
                MsRaw cur;
                ms_take<SUMS, CTC, INIT, PUPD, UPD, LMQ>(cur, slot[j]);
                ms_fence();
                issue(slot[j], t + 3 > t_last ? t_last : t + 3);
                ms_fence();
                if (t >= t_first && t <= t_last) {                 // (wave-uniform; no load inside)
                    const bool ok = xin && t >= 0 && t < H;
                    float rk = cur.v;
                    if (UPD && !up.first) rk = __builtin_fmaf(-alpha, cur.av, rk);
                    const float v0 = ok ? (PUPD ? cur.v + beta * cur.pv : UPD ? (LMQ ? cur.mi * rk : rk) + beta * cur.pv : cur.v) : 0.0f;      // (LMQ: p_k = M^-1 r_k + beta p_{k-1})
                    if (PUPD && t >= ya && t < yb && xout) (pu.p_out + (long)t * W)[(unsigned)x] = v0;
                    if (UPD) {
                        Rk[k0] = rk;
                        const float dk = up.first ? cur.dl : __builtin_fmaf(alpha, cur.pv, cur.dl);
                        if (LMQ) { Dk[k0] = dk; Mk[k0] = cur.mi; }
                        // own rows -- or a GHOST row of a slab (a row of the local image outside [ra, rb): nobody owns it here, the segment next to it keeps its r and p
                        // current, like image_warping's marching kernel does; its A p comes with the exchange, its delta is never read)
                        const bool mine = t >= ya && t < yb, ghost_row = ok && (t < g.ra || t >= g.rb);
                        if ((mine || ghost_row) && xout) {
                            const long ro = (long)t * W;
                            (up.r_out + ro)[(unsigned)x] = rk; (up.p_out + ro)[(unsigned)x] = v0;
                            if (mine && !up.first && up.delta) (up.delta + ro)[(unsigned)x] = dk;
                        }
                    }
                    const unsigned f0 = ok ? (cur.f >> (8 * (int)(((long)t * W + x) & 3))) & 0xffu : 0u;
                    const float v1 = Vv[k1], v2 = Vv[k2];
                    Cy[k0] = coef(cm, 1, 0, t + g.yoff);
                    // lane exchanges (every lane active here)
                    const float vl0 = ms_left(v0), vl1 = ms_left(v1), vr1 = ms_right(v1);
                    const float dB0 = ok ? (INIT ? cur.g.w : cur.g.x * v0 + cur.g.y * vl0 + cur.g.z * v1) : 0.0f;
                    const float dBr = ms_right(dB0);
                    const bool wn0 = ok && (cur.w.x != 0.0f || cur.w.y != 0.0f);
                    const float Uh0 = wn0 ? cur.w.x * (cur.w.x * (dB0 - dBr)) : 0.0f;
                    const float Uv1 = Wn[k1] ? Wy[k1] * (Wy[k1] * (dB[k1] - dB0)) : 0.0f;
                    float R1[3] = { 0.f, 0.f, 0.f };
                    if (Fl[k1] & 2u) {
                        const float cy0 = Cy[k0], cy1 = Cy[k1], cy2 = Cy[k2];
                        R1[0] = cm.ws * (4.0f * (cxc * v1) - cxm * vl1 - cxc * v2 - cxp * vr1 - cxc * v0);
                        R1[1] = cm.ws * (4.0f * (cy1 * v1) - cy1 * vl1 - cy2 * v2 - cy1 * vr1 - cy0 * v0);
                        R1[2] = cm.ws * (4.0f * (1.0f * v1) - 1.0f * vl1 - 1.0f * v2 - 1.0f * vr1 - 1.0f * v0);
                    }
                    float T1 = Uh[k1] + Uv1;
                    T1 -= ms_left(Uh[k1]);
                    T1 -= Uv[k2];
                    const float T2 = Tt[k2];
                    // G.y(i+ex) T(i+ex): the product as the right neighbour forms it (same operands, same bits), one exchange instead of two
                    const float gT2r = ms_right(Gy[k2] * T2);
                    float Rl[3], Rq[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { Rl[c] = ms_left(Rr[k2][c]); Rq[c] = ms_right(Rr[k2][c]); }
                    const int y = t - 2;
                    float dg_gyR = 0.f, dg_hR = 0.f, dg_kR = 0.f, dg_hL = 0.f, dg_hDL = 0.f, dg_kUR = 0.f; unsigned dg_fL = 0u, dg_fR = 0u;
                    if (DIAG) {     // lane exchanges of the diagonal (every lane active): rows y = t-2 (k2), y+1 = t-1 (k1), y-1 = t-3 (k0, not yet overwritten)
                        dg_gyR = ms_right(Gy[k2]); dg_hR = ms_right(Wx[k2]); dg_kR = ms_right(Wn[k2] ? Wy[k2] : 0.0f); dg_hL = ms_left(Wx[k2]);
                        dg_hDL = ms_left(Wx[k1]); dg_kUR = ms_right(Wn[k0] ? Wy[k0] : 0.0f);
                        dg_fL = (unsigned)__builtin_amdgcn_mov_dpp((int)Fl[k2], 0x138, 0xf, 0xf, true); dg_fR = (unsigned)__builtin_amdgcn_mov_dpp((int)Fl[k2], 0x130, 0xf, 0xf, true);
                    }
                    if (y >= ya && xout) {
                        if (DIAG) {
                            const float gx = Gx[k2], gzD = Gz[k1];
                            const float h0 = Wx[k2], k0w = Wn[k2] ? Wy[k2] : 0.0f, hD = Wx[k1], kD = Wn[k1] ? Wy[k1] : 0.0f, kU = Wn[k0] ? Wy[k0] : 0.0f;
                            float d = (Fl[k2] & 1u) ? cm.wp * cm.wp : 0.0f;
                            float ch, cv;
                            ch = (gx - dg_gyR) * h0;  cv = (gx - gzD) * k0w;  d += ch * ch + cv * cv;       // q = i
                            ch = dg_gyR * dg_hR;      cv = dg_gyR * dg_kR;    d += ch * ch + cv * cv;       // q = i+ex
                            ch = gzD * hD;            cv = gzD * kD;          d += ch * ch + cv * cv;       // q = i+ey
                            ch = -gx * dg_hL;         cv = 0.0f * 0.0f;       d += ch * ch + cv * cv;       // q = i-ex
                            ch = -gzD * dg_hDL;                               d += ch * ch + cv * cv;       // q = i-ex+ey
                            ch = 0.0f; cv = -gx * kU;                         d += ch * ch + cv * cv;       // q = i-ey
                            cv = -dg_gyR * dg_kUR;                            d += ch * ch + cv * cv;       // q = i+ex-ey
                            float cc = 0.0f;
                            { const float k0c = cm.ws * cxc, k1c = cm.ws * Cy[k2], k2c = cm.ws * 1.0f; cc += k0c * k0c; cc += k1c * k1c; cc += k2c * k2c; }
                            float cnt = (Fl[k2] & 2u) ? 16.0f : 0.0f;
                            if (dg_fL & 2u) cnt += 1.0f;
                            if (Fl[k0] & 2u) cnt += 1.0f;
                            if (dg_fR & 2u) cnt += 1.0f;
                            if (Fl[k1] & 2u) cnt += 1.0f;
                            (diag + (long)y * W)[(unsigned)x] = d + cnt * cc;
                        }
                        const float vc = v2;
                        float s = 0.0f;
                        if (Fl[k2] & 1u) s += cm.wp * (cm.wp * (INIT ? vc - cur.ct : vc));
                        s += Gx[k2] * T2;
                        if (x + 1 < W) s += gT2r;
                        if (y + 1 < H) s += Gz[k1] * T1;
                        const float ci[3] = { cxc, Cy[k2], 1.0f };
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float lap = 4.0f * Rr[k2][c] - Rl[c] - Rr[k0][c] - Rq[c] - R1[c];      // (slot k0 still holds row t-3)
                            s += cm.ws * (ci[c] * lap);
                        }
                        if (INIT) {
                            const float r = -s; const long ro = (long)y * W;
                            (out + ro)[(unsigned)x] = r; (z + ro)[(unsigned)x] = r; (p_prev + ro)[(unsigned)x] = 0.0f; (delta + ro)[(unsigned)x] = 0.0f; acc += r * r;
                        } else {
                            if (CTC) s += cur.ct * vc;
                            (out + (long)y * W)[(unsigned)x] = s; acc += vc * s;
                            if (SUMS) sm.add(LMQ ? Mk[k2] : 1.0f, UPD ? Rk[k2] : cur.rs, s);
                            if (LMQ) sq.add(Dk[k2], Rk[k2], cur.bb, vc, s);
                        }
                    }
                    Vv[k0] = v0; Fl[k0] = f0; Wn[k0] = wn0; Wy[k0] = cur.w.y; if (DIAG) Wx[k0] = ok ? cur.w.x : 0.0f; dB[k0] = dB0; Uh[k0] = Uh0; Uv[k1] = Uv1; Tt[k1] = T1;
                    Gx[k0] = cur.g.x; Gy[k0] = cur.g.y; Gz[k0] = cur.g.z;
#pragma unroll
                    for (int c = 0; c < 3; ++c) Rr[k1][c] = R1[c];
                }
            }
        }
    }
    if (LMQ) block_finish_sums_lm(acc, sm, sq, part_out, s3_out, fin, lmf, red, redd);
    else if (SUMS) block_finish_sums(acc, sm, part_out, s3_out, fin, red, redd);
    else block_store_partial(acc, part_out, red);
}

// ------------------------------------------------------------------------------------------ marching precompute
// k_precompute as a marching kernel: a wave owns 64 columns (lanes 1..62 produce output), rows are prefetched three ahead (X, D, Im and the two
// mask dwords: every input once per pixel and wave instead of 13 guarded loads), the left / right neighbours come from the neighbouring lanes,
// the upper / lower ones from the lane's rings.  The step that takes row t writes row t-1.  Same expressions (eval_BI_vals) as k_precompute.
constexpr int MP_USE = 62;
struct MpRaw { float x, d, im; unsigned mr, mc; };
__device__ __forceinline__ void mp_take(MpRaw& d, const MpRaw& s) { ms_mv(d.x, s.x); ms_mv(d.d, s.d); ms_mv(d.im, s.im); ms_mv(d.mr, s.mr); ms_mv(d.mc, s.mc); }

// COST: computeCost (k_cost's terms in k_cost's order) of the rows [c0, c1) rides along, one row behind the planes: the shading rows take BI of the pixel, of its
// right neighbour (lane + 1) and of the row below straight from registers -- for that BI is also evaluated on lane 63 and on the row under the segment.
template <bool COST>
__global__ __launch_bounds__(MS_NT, 2) void k_precompute_march(MsGeo g, int Hg, Cam cm, const float* __restrict__ X, const float* __restrict__ D, const float* __restrict__ Im,
                                                               const unsigned char* __restrict__ mR, const unsigned char* __restrict__ mC,
                                                               float4* __restrict__ G, float2* __restrict__ Wt, unsigned char* __restrict__ fl,
                                                               float* __restrict__ cost_out, int c0, int c1)
{
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = g.W, H = g.H;
    int strip = 0, ya = 0, yb = 0;
    {
        const int NG = (gridDim.x % 8) == 0 ? 8 : 1;
        const int grp = blockIdx.x % NG, l = blockIdx.x / NG;
        const long lo = (long)g.total * grp / NG, hi = (long)g.total * (grp + 1) / NG;
        const long id = lo + l;
        if (id < hi) {
            strip = (int)(id % g.nstrips);
            const int seg = (int)(id / g.nstrips) * (MS_NT / 64) + wave;
            ya = g.ra + seg * g.R; yb = ya + g.R;
            if (yb > g.rb) yb = g.rb;
            if (ya > g.rb) ya = g.rb;
        }
    }
    float acc = 0.0f;
    if (ya < yb) {
    const int x = strip * MP_USE - 1 + lane;
    const bool xin = x >= 0 && x < W;
    const bool xout = xin && lane >= 1 && lane <= 62;
    const unsigned xc_ = x < 0 ? 0u : x > W - 1 ? (unsigned)(W - 1) : (unsigned)x;
    const int t_first = ya - 1, t_last = COST ? yb + 1 : yb;
    const long Nb = (long)W * H - 4;
    auto issue = [&](MpRaw& s, int t) {
        const int tc = t < 0 ? 0 : t > H - 1 ? H - 1 : t;
        const long rowoff = (long)tc * W;
        s.x = (X + rowoff)[xc_]; s.d = (D + rowoff)[xc_]; s.im = (Im + rowoff)[xc_];
        // the dword that holds the pixel's mask byte (a byte load leaves a zero-extension that ends up behind a wait at the loop latch): aligned, except
        // that the last one is pulled back inside a plane whose size is not a multiple of 4 (the host checks W * H >= 4 and 4-byte-aligned planes)
        long b = (rowoff + xc_) & ~3L; if (b > Nb) b = Nb;
        s.mr = *reinterpret_cast<const unsigned*>(mR + b); s.mc = *reinterpret_cast<const unsigned*>(mC + b);
    };
    float Xr[3] = { 0.f, 0.f, 0.f }, Dr[3] = { 0.f, 0.f, 0.f }, Ir[3] = { 0.f, 0.f, 0.f };
    unsigned Mr[3] = { 0u, 0u, 0u }, Mc[3] = { 0u, 0u, 0u };
    float Bv[3] = { 0.f, 0.f, 0.f }, Wxr[3] = { 0.f, 0.f, 0.f }, Wyr[3] = { 0.f, 0.f, 0.f };      // (COST) BI, the row weights and the flags of the rows
    unsigned Fr[3] = { 0u, 0u, 0u };
    MpRaw slot[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) slot[j] = MpRaw{};
    for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int t = t0 + j;
            const int k0 = j, k1 = (j + 2) % 3, k2 = (j + 1) % 3;          // ring slots of rows t, t-1, t-2
            MpRaw cur;
            mp_take(cur, slot[j]);
            ms_fence();
            issue(slot[j], t + 3 > t_last ? t_last : t + 3);
            ms_fence();
            if (t >= t_first && t <= t_last) {
                const bool ok = xin && t >= 0 && t < H;
                const long ipx = (long)t * W + x; long bpx = ipx & ~3L; if (bpx > Nb) bpx = Nb;
                const int sh = ok ? 8 * (int)(ipx - bpx) : 0;
                const float x3 = Xr[k0];                                   // (COST) row t-3, about to be overwritten
                Xr[k0] = ok ? cur.x : 0.0f; Dr[k0] = ok ? cur.d : 0.0f; Ir[k0] = ok ? cur.im : 0.0f;
                Mr[k0] = (cur.mr >> sh) & 0xffu; Mc[k0] = (cur.mc >> sh) & 0xffu;
                // the row being written: y = t-1 (slot k1); its upper row t-2 (k2), its lower row t (k0)
                const float xc = Xr[k1], dc = Dr[k1], ic = Ir[k1];
                const float xl = ms_left(xc), dl = ms_left(dc), il = ms_left(ic), xr = ms_right(xc), dr = ms_right(dc);
                const int y = t - 1;
                const bool own = y >= ya && y < yb && xout;
                if (own || (COST && y >= ya && y <= yb && xin && lane >= 1)) {
                    const J3 b = eval_BI_vals(cm, dl, dc, Dr[k2], xc, xl, Xr[k2], ic, il, Ir[k2], x, y + g.yoff);
                    if (COST) Bv[k1] = b.v;
                    if (own) {
                        const long i = (long)y * W;
                        (G + i)[(unsigned)x] = make_float4(b.d0, b.d1, b.d2, b.v);
                        const int yg = y + g.yoff;
                        const bool inner = x >= 1 && x + 1 < W && yg >= 1 && yg + 1 < Hg;
                        const float2 wv = inner ? make_float2(cm.wg * (float)Mr[k1], cm.wg * (float)Mc[k1]) : make_float2(0.f, 0.f);
                        (Wt + i)[(unsigned)x] = wv;
                        unsigned char f = dc > 0.0f ? 1 : 0;
                        bool valid = f;
                        valid = valid && dl > 0.0f && fabsf(xc - xl) < 0.01f;               // (x-1, y), (x, y-1), (x+1, y), (x, y+1): k_precompute's order
                        valid = valid && Dr[k2] > 0.0f && fabsf(xc - Xr[k2]) < 0.01f;
                        valid = valid && dr > 0.0f && fabsf(xc - xr) < 0.01f;
                        valid = valid && Dr[k0] > 0.0f && fabsf(xc - Xr[k0]) < 0.01f;
                        if (valid) f |= 2;
                        (fl + i)[(unsigned)x] = f;
                        if (COST) { Wxr[k1] = wv.x; Wyr[k1] = wv.y; Fr[k1] = f; }
                    }
                }
                if (COST) {
                    // cost of row t-2 (slot k2): its planes were formed one step ago, BI of the row below just now
                    const float x2 = Xr[k2];
                    const float x2l = ms_left(x2), x2r = ms_right(x2), b2r = ms_right(Bv[k2]);
                    const int yc = t - 2;
                    if (yc >= ya && yc < yb && yc >= c0 && yc < c1 && xout) {
                        float sacc = 0.0f;
                        if (Fr[k2] & 1u) { const float e = cm.wp * (x2 - Dr[k2]); sacc += e * e; }
                        if (Wxr[k2] != 0.0f || Wyr[k2] != 0.0f) {
                            const float b0 = Bv[k2];
                            const float eh = Wxr[k2] * (b0 - b2r), ev = Wyr[k2] * (b0 - Bv[k1]);
                            sacc += eh * eh + ev * ev;
                        }
                        if (Fr[k2] & 2u) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                float a = 4.0f * (coef(cm, c, x, yc + g.yoff) * x2);
                                a -= coef(cm, c, x - 1, yc + g.yoff) * x2l; a -= coef(cm, c, x, yc - 1 + g.yoff) * x3;
                                a -= coef(cm, c, x + 1, yc + g.yoff) * x2r; a -= coef(cm, c, x, yc + 1 + g.yoff) * Xr[k1];
                                a *= cm.ws; sacc += a * a;
                            }
                        }
                        acc += 0.5f * sacc;
                    }
                }
            }
        }
    }
    }
    if (COST) block_store_partial(acc, cost_out, red);
}

// raw diag(J^T J) (LM only): enumerate the rows that contain X(i)
__global__ __launch_bounds__(BLOCK) void k_diag(Geo g, Cam cm, const float4* __restrict__ G, const float2* __restrict__ Wt,
                                                const unsigned char* __restrict__ fl, float* __restrict__ diag)
{
    FOR_EACH_PIXEL(g) {
        const long i = (long)y * g.W + x;
        float d = (fl[i] & 1) ? cm.wp * cm.wp : 0.0f;
        // shading rows at q with X(i) in their support: sh_h: q in {i, i+ex, i+ey, i-ex, i-ex+ey}; sh_v: q in {i, i+ex, i+ey, i-ey, i-ey+ex}
        const int qx[7] = { 0, 1, 0, -1, -1, 0, 1 }, qy[7] = { 0, 0, 1, 0, 1, -1, -1 };
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int X0 = x + qx[k], Y0 = y + qy[k];
            if (X0 < 1 || X0 + 1 >= g.W || Y0 + g.yoff < 1 || Y0 + g.yoff + 1 >= g.Hg || Y0 < 0 || Y0 + 1 >= g.H) continue;      // row guard (global) + local storage
            const long q = (long)Y0 * g.W + X0;
            const float2 w = Wt[q];
            const float4 g0 = G[q], gx = G[q + 1], gy = G[q + g.W];
            // coefficient of X(i) in sh_h(q) and sh_v(q); t = i - q
            const int tx = -qx[k], ty = -qy[k];
            float ch = 0.0f, cv = 0.0f;
            if (tx == 0 && ty == 0) { ch = g0.x - gx.y; cv = g0.x - gy.z; }
            else if (tx == -1 && ty == 0) { ch = g0.y; cv = g0.y; }
            else if (tx == 0 && ty == -1) { ch = g0.z; cv = g0.z; }
            else if (tx == 1 && ty == 0) { ch = -gx.x; }
            else if (tx == 1 && ty == -1) { ch = -gx.z; }
            else if (tx == 0 && ty == 1) { cv = -gy.x; }
            else if (tx == -1 && ty == 1) { cv = -gy.y; }
            ch *= w.x; cv *= w.y;
            d += ch * ch + cv * cv;
        }
        // reg rows: q = i (coefficient 4 w_s coef_c(i)) and the four neighbours (-w_s coef_c(i))
        float cc = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float k = cm.ws * coef(cm, c, x, y + g.yoff); cc += k * k; }
        float cnt = (fl[i] & 2) ? 16.0f : 0.0f;
        if (x > 0 && (fl[i - 1] & 2)) cnt += 1.0f;
        if (y > 0 && (fl[i - g.W] & 2)) cnt += 1.0f;
        if (x + 1 < g.W && (fl[i + 1] & 2)) cnt += 1.0f;
        if (y + 1 < g.H && (fl[i + g.W] & 2)) cnt += 1.0f;
        diag[i] = d + cnt * cc;
    }
}

}  // namespace

extern "C" {

// THALLO_SFS_FUSED=0: the two-pass k_rows + k_gather form (A/B switch)
static bool sfs_fused() { static int v = -1; if (v < 0) { const char* e = thallo::env_switch("THALLO_SFS_FUSED"); v = (e && e[0] == '0') ? 0 : 1; } return v == 1; }
static int fused_grid(int W, int rows)
{
    const int nt = ((W + FW - 1) / FW) * ((rows + FH - 1) / FH);
    int cap = thallo_hip_device_cu_count() * 4; if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS; cap -= cap % 8;      // (1024 partial slots: 4 of the 6 resident workgroups per CU)
    return nt < cap ? nt : cap;
}

// THALLO_SFS_MARCH=0: the LDS-tiled k_fused for J^T(J v) (A/B switch); default: the marching kernel
static int g_ms_rows = 0, g_ms_wgcu = 0, g_ms_force = -1;       // tools / tests: rows per wave segment, workgroups per CU the grid is sized for (0 = automatic), kernel choice (-1 = the environment's)
static bool sfs_march()
{
    static int v = -1; if (v < 0) { const char* e = thallo::env_switch("THALLO_SFS_MARCH"); v = (e && e[0] == '0') ? 0 : 1; }
    return g_ms_force >= 0 ? g_ms_force == 1 : v == 1;
}
static int g_ms_cap = 0;        // tests: workgroup budget the grids are sized for (0 = CUs x workgroups per CU of the device)
static int g_ms_precompute = 1;      // 1: precompute by the marching kernel, 0: k_precompute (tools / tests)
static int g_ms_diag = 1;      // 1: the LM diagonal by the marching J^T F kernel, 0: k_diag (tools / tests)
static bool sfs_march_diag() { return g_ms_diag == 1; }
static int g_ms_pair = -1, g_ms_depth = 0;      // the pixel-pair kernels on the packed planes (energy_sfs_pair.hip): -1 the environment's THALLO_SFS_PAIR (default on), 0 / 1 forced; their rows of prefetch (0 = automatic)
void thallo_hip_sfs_march_debug_set(int what, int value) { if (what == 0) g_ms_rows = value; if (what == 1) g_ms_wgcu = value; if (what == 2) g_ms_force = value; if (what == 3) g_ms_diag = value; if (what == 4) g_ms_precompute = value; if (what == 5) g_ms_cap = value;
                                                           if (what == 6) g_ms_pair = value; if (what == 7) g_ms_depth = value; }
static thallo::SfsTune pair_tune() { thallo::SfsTune t; t.rows = g_ms_rows; t.wgcu = g_ms_wgcu; t.cap = g_ms_cap; t.depth = g_ms_depth; return t; }
static MsGeo make_ms_geo(int W, int H, int ra, int rb, int yoff, int R)
{
    MsGeo g; g.W = W; g.H = H; g.ra = ra; g.rb = rb; g.yoff = yoff; g.R = R;
    g.nstrips = (W + MS_USE - 1) / MS_USE;
    const int nseg = (rb - ra + R - 1) / R;
    g.total = g.nstrips * ((nseg + MS_NT / 64 - 1) / (MS_NT / 64));
    return g;
}
static long ms_cap(int per_cu) { return g_ms_cap > 0 ? g_ms_cap : (long)thallo_hip_device_cu_count() * per_cu; }
// the image has no more column strips than workgroup slots (otherwise the LDS-tiled kernels, which loop over their tiles, run)
static bool sfs_march_fits(int W) { return march_strips_fit((W + MS_USE - 1) / MS_USE, ms_cap(g_ms_wgcu > 0 ? g_ms_wgcu : MS_WG_PER_CU)); }
// Round 6: images of even width run the marching kernels on pixel pairs, on PACKED planes (sfs_pair.hpp) -- every entry point below that takes G / Wt / fl then reads or
// writes that layout (thallo_hip_sfs_planes_layout).  THALLO_SFS_PAIR=0 (THALLO_AB sfs_pair=0): the one-pixel-per-lane kernels on the float4 / float2 / byte planes (A/B).
static bool sfs_pair(int W, int H)
{
    static int v = -1; if (v < 0) { const char* e = thallo::env_switch("THALLO_SFS_PAIR"); v = (e && e[0] == '0') ? 0 : 1; }
    const bool on = g_ms_pair >= 0 ? g_ms_pair == 1 : v == 1;
    return on && sfs_fused() && sfs_march() && sfs_march_fits(W) && thallo::sfs_pair_ok(W, H, pair_tune());
}
int thallo_hip_sfs_planes_layout(int W, int H) { return sfs_pair(W, H) ? 1 : 0; }
static MsGeo pick_ms_geo(int W, int H, int ra, int rb, int yoff)
{
    if (g_ms_rows > 0) return make_ms_geo(W, H, ra, rb, yoff, g_ms_rows);
    const int nstrips = (W + MS_USE - 1) / MS_USE, per_cu = g_ms_wgcu > 0 ? g_ms_wgcu : MS_WG_PER_CU;
    int R = march_rows_per_segment(rb - ra, nstrips, MS_NT / 64, ms_cap(per_cu));
    // wide images (energy_image_warping_march.hip, pick_rows: the same rule): when the strip count leaves more than a quarter of the budget's workgroup slots empty,
    // the budget grows (up to THALLO_MAX_PARTIALS workgroups) until the grid fills its last round of workgroups to 90 %
    if (R > 0 && g_ms_cap <= 0 && g_ms_wgcu <= 0) {
        const long slots = ms_cap(per_cu);
        auto fill_of = [&](int rr) { const long nseg = (rb - ra + rr - 1) / rr, total = (long)nstrips * ((nseg + MS_NT / 64 - 1) / (MS_NT / 64)); return (double)total / (double)(((total + slots - 1) / slots) * slots); };
        double best = fill_of(R);
        for (int m = 2; best < 0.75 && m <= 4; ++m) {
            const int r2 = march_rows_per_segment(rb - ra, nstrips, MS_NT / 64, slots * m);
            if (r2 <= 0) break;
            const double f = fill_of(r2);
            if (f > best + 1e-9) { best = f; R = r2; }
            if (f >= 0.9) break;
        }
    }
    return make_ms_geo(W, H, ra, rb, yoff, R > 0 ? R : rb - ra);      // (R == 0 is excluded by sfs_march_fits() at every call site; one segment per strip otherwise)
}

/* host_params: the 16 scalar parameters of the .t in Inputs{} order: w_p, w_s, w_g (squared weights), f_x, f_y, u_x, u_y, L_1..L_9 */
static Cam cam_of(const float* hp)
{
    Cam c; c.wp = sqrtf(hp[0]); c.ws = sqrtf(hp[1]); c.wg = sqrtf(hp[2]); c.fx = hp[3]; c.fy = hp[4]; c.ux = hp[5]; c.uy = hp[6];
    for (int k = 0; k < 9; ++k) c.L[k] = hp[7 + k];
    return c;
}

static int sfs_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                          const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl, float* cost_out, int c0, int c1, thallo_stream_t stream);
int thallo_hip_sfs_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                              const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl,
                              thallo_stream_t stream)
{ return sfs_precompute(W, H, ra, rb, yoff, Hg, host_params, X, D, Im, edgeMaskR, edgeMaskC, G, Wt, fl, nullptr, 0, 0, stream); }

/* precompute over the rows [ra, rb) AND computeCost over the rows [c0, c1) of them in one launch (marching kernel only: -hipErrorNotSupported otherwise -- the caller
 * then runs thallo_hip_sfs_precompute + thallo_hip_sfs_cost); returns the number of cost partials */
int thallo_hip_sfs_precompute_cost(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                                   const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl,
                                   int c0, int c1, float* cost_out, thallo_stream_t stream)
{
    if (!cost_out || c0 < ra || c1 > rb || c0 >= c1) return -(int)hipErrorInvalidValue;
    return sfs_precompute(W, H, ra, rb, yoff, Hg, host_params, X, D, Im, edgeMaskR, edgeMaskC, G, Wt, fl, cost_out, c0, c1, stream);
}

static int sfs_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* Im,
                          const unsigned char* edgeMaskR, const unsigned char* edgeMaskC, float* G, float* Wt, unsigned char* fl, float* cost_out, int c0, int c1, thallo_stream_t stream)
{
    if (ra < 0 || rb > H || ra >= rb) return -(int)hipErrorInvalidValue;
    if (sfs_pair(W, H)) { (void)fl; return thallo::sfs_pair_precompute(W, H, ra, rb, yoff, Hg, host_params, X, D, Im, edgeMaskR, edgeMaskC, G, Wt, cost_out, c0, c1, pair_tune(), stream); }
    if (sfs_fused() && sfs_march() && sfs_march_fits(W) && g_ms_precompute == 1 && (long)W * H >= 4 && (((uintptr_t)edgeMaskR | (uintptr_t)edgeMaskC) & 3) == 0) {
        const int np = (W + MP_USE - 1) / MP_USE;
        int R = march_rows_per_segment(rb - ra, np, MS_NT / 64, ms_cap(4));
        if (R <= 0) R = rb - ra;                                           // (np <= the k_march strip count, which sfs_march_fits() bounded)
        MsGeo mg = make_ms_geo(W, H, ra, rb, yoff, R); mg.nstrips = np;
        mg.total = mg.nstrips * (((rb - ra + R - 1) / R + MS_NT / 64 - 1) / (MS_NT / 64));
        const int gridp = (mg.total + 7) / 8 * 8;
        if (cost_out) {
            if (gridp > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
            hipLaunchKernelGGL(k_precompute_march<true>, dim3(gridp), dim3(MS_NT), 0, (hipStream_t)stream, mg, Hg, cam_of(host_params), X, D, Im, edgeMaskR, edgeMaskC,
                               (float4*)G, (float2*)Wt, fl, cost_out, c0, c1);
            int e = check_launch(); return e ? e : gridp;
        }
        hipLaunchKernelGGL(k_precompute_march<false>, dim3(gridp), dim3(MS_NT), 0, (hipStream_t)stream, mg, Hg, cam_of(host_params), X, D, Im, edgeMaskR, edgeMaskC,
                           (float4*)G, (float2*)Wt, fl, (float*)nullptr, 0, 0);
        return check_launch();
    }
    if (cost_out) return -(int)hipErrorNotSupported;
    const Geo g = make_geo(W, H, ra, rb, yoff, Hg); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_precompute, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, cam_of(host_params), X, D, Im, edgeMaskR, edgeMaskC,
                       (float4*)G, (float2*)Wt, fl);
    return check_launch();
}

int thallo_hip_sfs_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                        const unsigned char* fl, float* cost_out, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1) return -(int)hipErrorInvalidValue;
    if (sfs_pair(W, H)) return -(int)hipErrorNotSupported;      // (packed planes: the cost comes out of thallo_hip_sfs_precompute_cost)
    const Geo g = make_geo(W, H, row0, row1, yoff, Hg); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g, cam_of(host_params), X, D, (const float4*)G, (const float2*)Wt, fl, cost_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_sfs_pcg_init(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                            const unsigned char* fl, float* U, float* R, float* r, float* z, float* p_prev, float* delta,
                            float* diag_out, float* aN_out, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1) return -(int)hipErrorInvalidValue;
    if (sfs_pair(W, H)) return thallo::sfs_pair_init(W, H, row0, row1, yoff, Hg, host_params, X, D, G, Wt, r, z, p_prev, delta, diag_out, aN_out, thallo::SfsFinDiag{}, pair_tune(), stream);
    // row pass over the owned rows +-1 (they feed the gather of the owned rows), gather over the owned rows
    const Geo g = make_geo(W, H, row0, row1, yoff, Hg); const int grid = grid_for(g);
    const Geo gr = make_geo(W, H, row0 > 0 ? row0 - 1 : 0, row1 < H ? row1 + 1 : H, yoff, Hg); const int gridr = grid_for(gr);
    const Cam cm = cam_of(host_params);
    hipStream_t s = (hipStream_t)stream;
    if (sfs_fused() && sfs_march() && sfs_march_fits(W)) {
        const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
        const int gridm = (mg.total + 7) / 8 * 8;
        if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
        if (diag_out && !sfs_march_diag())
            hipLaunchKernelGGL(k_diag, dim3(grid), dim3(BLOCK), 0, s, g, cm, (const float4*)G, (const float2*)Wt, fl, diag_out);
        if (diag_out && sfs_march_diag())
            hipLaunchKernelGGL((k_march<false, true, true, true, MS_OCC>), dim3(gridm), dim3(MS_NT), 0, s, mg, cm, X, D, (const float4*)G, (const float2*)Wt, fl, r, aN_out,
                               (const float*)nullptr, (double*)nullptr, (const unsigned*)nullptr, FinArgs{}, z, p_prev, delta, diag_out);
        else
            hipLaunchKernelGGL((k_march<false, true, true, false, MS_OCC>), dim3(gridm), dim3(MS_NT), 0, s, mg, cm, X, D, (const float4*)G, (const float2*)Wt, fl, r, aN_out,
                               (const float*)nullptr, (double*)nullptr, (const unsigned*)nullptr, FinArgs{}, z, p_prev, delta, (float*)nullptr);
        int e = check_launch(); return e ? e : gridm;
    }
    if (sfs_fused()) {
        const int gridf = fused_grid(W, row1 - row0);
        hipLaunchKernelGGL(k_fused<0>, dim3(gridf), dim3(BLOCK), 0, s, g, cm, X, D, (const float4*)G, (const float2*)Wt, fl, r, z, p_prev, delta, aN_out,
                           (const float*)nullptr, (double*)nullptr, (const unsigned*)nullptr, FinArgs{});
        if (diag_out) hipLaunchKernelGGL(k_diag, dim3(grid), dim3(BLOCK), 0, s, g, cm, (const float4*)G, (const float2*)Wt, fl, diag_out);
        int e = check_launch(); return e ? e : gridf;
    }
    hipLaunchKernelGGL(k_rows<true>, dim3(gridr), dim3(BLOCK), 0, s, gr, cm, X, (const float4*)G, (const float2*)Wt, fl, (float2*)U, R);
    hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(BLOCK), 0, s, g, cm, X, D, (const float4*)G, (const float2*)U, (const float*)R, fl, r, z, p_prev, delta, aN_out);
    if (diag_out) hipLaunchKernelGGL(k_diag, dim3(grid), dim3(BLOCK), 0, s, g, cm, (const float4*)G, (const float2*)Wt, fl, diag_out);
    int e = check_launch(); return e ? e : grid;
}

static int sfs_apply(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                     float* U, float* R, const float* p, float* Ap, float* aD_out, const float* r, double* s3_out, thallo_stream_t stream, const unsigned* gate = nullptr,
                     thallo_fin_t fin = thallo_fin_t{ { nullptr, 0 }, nullptr, nullptr, nullptr }, const float* ctc = nullptr);

int thallo_hip_sfs_apply_jtj(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                             float* U, float* R, const float* p, float* Ap, float* aD_out, thallo_stream_t stream)
{ return sfs_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, fl, U, R, p, Ap, aD_out, nullptr, nullptr, stream); }

int thallo_hip_sfs_apply_jtj_sums(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                  float* U, float* R, const float* p, float* Ap, float* aD_out, const float* r, double* s3_out, thallo_stream_t stream)
{ if (!r || !s3_out) return -(int)hipErrorInvalidValue; return sfs_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, fl, U, R, p, Ap, aD_out, r, s3_out, stream); }

int thallo_hip_sfs_apply_jtj_sums_fin(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                      float* U, float* R, const float* p, float* Ap, float* aD_out, const float* r, double* s3_out, thallo_fin_t fin, thallo_stream_t stream)
{ if (!r || !s3_out) return -(int)hipErrorInvalidValue; return sfs_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, fl, U, R, p, Ap, aD_out, r, s3_out, stream, nullptr, fin); }

int thallo_hip_sfs_apply_jtj_gated(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                   float* U, float* R, const float* p, float* Ap, float* aD_out, const unsigned* gate, thallo_stream_t stream)
{ return sfs_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, fl, U, R, p, Ap, aD_out, nullptr, nullptr, stream, gate); }

int thallo_hip_sfs_apply_jtj_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                float* U, float* R, const float* p, const float* CtC, float* Ap, float* aD_out, const unsigned* gate, thallo_stream_t stream)
{ return sfs_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, fl, U, R, p, Ap, aD_out, nullptr, nullptr, stream, gate, thallo_fin_t{ { nullptr, 0 }, nullptr, nullptr, nullptr }, CtC); }

int thallo_hip_sfs_lm_pupdate_supported(void) { return sfs_fused() && sfs_march() ? 1 : 0; }
int thallo_hip_sfs_march_fits(int W) { return sfs_fused() && sfs_march() && sfs_march_fits(W) ? 1 : 0; }

/* Round 6 (packed planes only; -hipErrorNotSupported elsewhere: the caller runs the launches they replace).
 * thallo_hip_sfs_pcg_init_lm: PCGInit1's J^T F pass with PCGFinalizeDiagonal riding along (gauss_newton.t:936-969): r = -J^T F, delta = 0, p_prev = 0 and, from the raw
 * diagonal of J^T J formed in the same pass, CtC, M^-1 (pre), b = r, z = M^-1 r, SSq (written when save_ssq, else read), partials of r . z.
 * thallo_hip_sfs_lm_model_cost: delta_out = delta + alpha_kl p_kl (the update the one-launch LM loop owes, thallo_hip_lm_owed_delta's rule) and the partials of
 * delta_out . J^T J delta_out and delta_out . b in ONE launch -- the model cost of an LM step (thallo.t:3845-3865, expanded: solver.cpp step_lm); with X / prevX also
 * savePreviousUnknowns and PCGLinearUpdate (prevX = X, X = X + delta_out; gauss_newton.t:901-906,915-920). */
int thallo_hip_sfs_pcg_init_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* X, const float* D, const float* G, const float* Wt,
                               const unsigned char* fl, float* r, float* z, float* p_prev, float* delta, float* SSq, float* CtC, float* pre, float* b,
                               float radius, float min_lm_diagonal, float max_lm_diagonal, int save_ssq, float* aN_out, thallo_stream_t stream)
{
    (void)fl;
    if (!sfs_pair(W, H)) return -(int)hipErrorNotSupported;
    thallo::SfsFinDiag fd; fd.SSq = SSq; fd.CtC = CtC; fd.pre = pre; fd.b = b; fd.radius = radius; fd.min_lm = min_lm_diagonal; fd.max_lm = max_lm_diagonal; fd.save_ssq = save_ssq;
    if (!SSq || !CtC || !pre || !b) return -(int)hipErrorInvalidValue;
    return thallo::sfs_pair_init(W, H, row0, row1, yoff, Hg, host_params, X, D, G, Wt, r, z, p_prev, delta, nullptr, aN_out, fd, pair_tune(), stream);
}
int thallo_hip_sfs_lm_model_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                 const float* delta, float* delta_out, const float* p_even, const float* p_odd, const float* b, const float* alphaN_words, const float* alphaD_words,
                                 int word_stride, const float* lm_state, int L, float* dJJd_out, float* db_out, float* X, float* prevX, thallo_stream_t stream)
{
    (void)fl;
    if (!sfs_pair(W, H)) return -(int)hipErrorNotSupported;
    return thallo::sfs_pair_model_cost(W, H, row0, row1, yoff, Hg, host_params, G, Wt, delta, delta_out, p_even, p_odd, b, alphaN_words, alphaD_words, word_stride, lm_state, L, dJJd_out, db_out,
                                       X, prevX, pair_tune(), stream);
}

int thallo_hip_sfs_pcg_iter(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                            const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                            thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, float* aD_out, double* s3_out, thallo_fin_t fin, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || !r_in || !r_out || r_in == r_out || !Ap_out || !p_in || !p_out || p_in == p_out || !aD_out || !s3_out) return -(int)hipErrorInvalidValue;
    if (!first && (!Ap_in || Ap_in == Ap_out || !alphaN_prev.partials || !alphaD_prev.partials || !betaN_prev.partials)) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    if (!thallo_hip_sfs_march_fits(W)) return -(int)hipErrorNotSupported;
    if (sfs_pair(W, H)) return thallo::sfs_pair_iter(W, H, row0, row1, yoff, Hg, host_params, G, Wt, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, first, alphaN_prev, alphaD_prev, betaN_prev, nullptr,
                                                     aD_out, s3_out, fin, pair_tune(), stream);
    const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
    const int gridm = (mg.total + 7) / 8 * 8;
    if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const MsUpd up = { r_out, first ? r_in : Ap_in, p_in, p_out, delta, alphaN_prev, alphaD_prev, betaN_prev, first ? 1 : 0, 0, nullptr, nullptr };     // (first: Ap_in is not used; any readable plane)
    const FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, gridm };
    hipLaunchKernelGGL((k_march<true, false, false, false, MS_OCC, false, true>), dim3(gridm), dim3(MS_NT), 0, (hipStream_t)stream, mg, cam_of(host_params), r_in, (const float*)nullptr,
                       (const float4*)G, (const float2*)Wt, fl, Ap_out, aD_out, (const float*)nullptr, s3_out, (const unsigned*)nullptr, fa,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, MsPupd{}, up);
    int e = check_launch(); return e ? e : gridm;
}

int thallo_hip_sfs_pcg_iter_deferred(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                     const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                                     thallo_sum_t alphaN_prev, thallo_prev_t prev, float* aD_out, double* s3_out, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || !r_in || !r_out || r_in == r_out || !Ap_out || !p_in || !p_out || p_in == p_out || !aD_out || !s3_out) return -(int)hipErrorInvalidValue;
    if (!first && (!Ap_in || Ap_in == Ap_out || !alphaN_prev.partials || !prev.alphaD_partials || !prev.s12_partials || prev.s12_partials == s3_out || prev.count < 1 ||
                   prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_word || !prev.betaN_word)) return -(int)hipErrorInvalidValue;
    if (!thallo_hip_sfs_march_fits(W)) return -(int)hipErrorNotSupported;
    if (sfs_pair(W, H)) return thallo::sfs_pair_iter(W, H, row0, row1, yoff, Hg, host_params, G, Wt, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, first, alphaN_prev, alphaN_prev, alphaN_prev, &prev,
                                                     aD_out, s3_out, thallo_fin_t{ { nullptr, 0 }, nullptr, nullptr, nullptr }, pair_tune(), stream);
    const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
    const int gridm = (mg.total + 7) / 8 * 8;
    if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    MsUpd up = { r_out, first ? r_in : Ap_in, p_in, p_out, delta, alphaN_prev, thallo_sum_t{ first ? alphaN_prev.partials : prev.alphaD_partials, first ? 1 : prev.count }, alphaN_prev,
                 first ? 1 : 0, 0, nullptr, nullptr };
    if (!first) { up.prev_s3 = prev.s12_partials; up.prev_nb = prev.count; up.aD_word = prev.alphaD_word; up.bN_word = prev.betaN_word; }
    const FinArgs fa{ alphaN_prev, nullptr, nullptr, nullptr, 0, gridm };          // partials only: the next launch (or thallo_hip_pcg_scalars_finish behind the loop) finishes
    hipLaunchKernelGGL((k_march<true, false, false, false, MS_OCC, false, true>), dim3(gridm), dim3(MS_NT), 0, (hipStream_t)stream, mg, cam_of(host_params), r_in, (const float*)nullptr,
                       (const float4*)G, (const float2*)Wt, fl, Ap_out, aD_out, (const float*)nullptr, s3_out, (const unsigned*)nullptr, fa,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, MsPupd{}, up);
    int e = check_launch(); return e ? e : gridm;
}

int thallo_hip_sfs_pcg_iter_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                               const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, const float* CtC, const float* b,
                               const float* pre, int first, thallo_sum_t alphaN_prev, thallo_sum_t alphaD_prev, thallo_sum_t betaN_prev, float* aD_out, double* s3_out, double* q3_out,
                               thallo_fin_t fin, float* lm_state, int k, float q_tolerance, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || !r_in || !r_out || r_in == r_out || !Ap_out || !p_in || !p_out || p_in == p_out || !delta || !aD_out || !s3_out || !q3_out ||
        q3_out == s3_out || !CtC || !b || !pre || !lm_state) return -(int)hipErrorInvalidValue;
    if (!first && (!Ap_in || Ap_in == Ap_out || !alphaN_prev.partials || !alphaD_prev.partials || !betaN_prev.partials)) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;      // (no tickets: partials only -- a row slab, whose exchange finishes)
    if (!thallo_hip_sfs_march_fits(W)) return -(int)hipErrorNotSupported;
    if (sfs_pair(W, H)) return thallo::sfs_pair_iter_lm(W, H, row0, row1, yoff, Hg, host_params, G, Wt, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, CtC, b, pre, first, alphaN_prev, alphaD_prev, betaN_prev,
                                                        aD_out, s3_out, q3_out, fin, lm_state, k, q_tolerance, pair_tune(), stream);
    const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
    const int gridm = (mg.total + 7) / 8 * 8;
    if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const MsUpd up = { r_out, first ? r_in : Ap_in, p_in, p_out, delta, alphaN_prev, alphaD_prev, betaN_prev, first ? 1 : 0, 1, b, pre };
    const FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, gridm };
    const LmFin lf{ b, q3_out, lm_state, k, q_tolerance };
    hipLaunchKernelGGL((k_march<true, true, false, false, MS_OCC, false, true, true>), dim3(gridm), dim3(MS_NT), 0, (hipStream_t)stream, mg, cam_of(host_params), r_in, CtC,
                       (const float4*)G, (const float2*)Wt, fl, Ap_out, aD_out, (const float*)nullptr, s3_out, reinterpret_cast<const unsigned*>(lm_state) + 1, fa,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, MsPupd{}, up, lf);
    int e = check_launch(); return e ? e : gridm;
}

int thallo_hip_sfs_apply_jtj_lm_pupdate(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                                        const float* z, const float* p_in, float* p_out, const float* CtC, float* Ap, float* aD_out, int first,
                                        thallo_sum_t alphaN_prev, thallo_sum_t betaN_prev, const unsigned* gate, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || !z || !p_in || !p_out || p_in == p_out || !CtC || !Ap || !aD_out) return -(int)hipErrorInvalidValue;
    if (!first && (!alphaN_prev.partials || !betaN_prev.partials || alphaN_prev.count < 1 || betaN_prev.count < 1)) return -(int)hipErrorInvalidValue;
    if (!thallo_hip_sfs_march_fits(W)) return -(int)hipErrorNotSupported;
    if (sfs_pair(W, H)) return thallo::sfs_pair_apply_pupdate(W, H, row0, row1, yoff, Hg, host_params, G, Wt, z, p_in, p_out, CtC, Ap, aD_out, first, alphaN_prev, betaN_prev, gate, pair_tune(), stream);
    const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
    const int gridm = (mg.total + 7) / 8 * 8;
    if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const MsPupd pu = { p_in, p_out, alphaN_prev, betaN_prev, first ? 1 : 0 };
    hipLaunchKernelGGL((k_march<false, true, false, false, MS_OCC, true>), dim3(gridm), dim3(MS_NT), 0, (hipStream_t)stream, mg, cam_of(host_params), z, CtC, (const float4*)G,
                       (const float2*)Wt, fl, Ap, aD_out, (const float*)nullptr, (double*)nullptr, gate, FinArgs{}, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, pu);
    int e = check_launch(); return e ? e : gridm;
}

static int sfs_apply(int W, int H, int row0, int row1, int yoff, int Hg, const float* host_params, const float* G, const float* Wt, const unsigned char* fl,
                     float* U, float* R, const float* p, float* Ap, float* aD_out, const float* r, double* s3_out, thallo_stream_t stream, const unsigned* gate, thallo_fin_t fin, const float* ctc)
{
    if (row0 < 0 || row1 > H || row0 >= row1) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!s3_out || !fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    if (sfs_pair(W, H)) return thallo::sfs_pair_apply(W, H, row0, row1, yoff, Hg, host_params, G, Wt, p, Ap, aD_out, r, s3_out, gate, fin, ctc, pair_tune(), stream);
    const Geo g = make_geo(W, H, row0, row1, yoff, Hg); const int grid = grid_for(g);
    if (sfs_fused() && sfs_march() && sfs_march_fits(W)) {
        const MsGeo mg = pick_ms_geo(W, H, row0, row1, yoff);
        const int gridm = (mg.total + 7) / 8 * 8;
        if (gridm > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;      // (only through the tools' forced rows per segment)
        const FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, gridm };
        const Cam cm = cam_of(host_params);
#define MS_LAUNCH(SUMS, CTC) hipLaunchKernelGGL((k_march<SUMS, CTC, false, false, MS_OCC>), dim3(gridm), dim3(MS_NT), 0, (hipStream_t)stream, mg, cm, p, ctc, (const float4*)G, \
                                                (const float2*)Wt, fl, Ap, aD_out, r, s3_out, gate, fa, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr)
        if (s3_out && ctc) MS_LAUNCH(true, true); else if (s3_out) MS_LAUNCH(true, false); else if (ctc) MS_LAUNCH(false, true); else MS_LAUNCH(false, false);
#undef MS_LAUNCH
        int e = check_launch(); return e ? e : gridm;
    }
    if (sfs_fused()) {
        const int gridf = fused_grid(W, row1 - row0);
        hipLaunchKernelGGL(k_fused<1>, dim3(gridf), dim3(BLOCK), 0, (hipStream_t)stream, g, cam_of(host_params), p, ctc, (const float4*)G, (const float2*)Wt, fl,
                           Ap, (float*)nullptr, (float*)nullptr, (float*)nullptr, aD_out, r, s3_out, gate,
                           FinArgs{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, gridf });
        int e = check_launch(); return e ? e : gridf;
    }
    const Geo gr = make_geo(W, H, row0 > 0 ? row0 - 1 : 0, row1 < H ? row1 + 1 : H, yoff, Hg); const int gridr = grid_for(gr);
    const Cam cm = cam_of(host_params);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_rows<false>, dim3(gridr), dim3(BLOCK), 0, s, gr, cm, p, (const float4*)G, (const float2*)Wt, fl, (float2*)U, R);
    hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(BLOCK), 0, s, g, cm, p, (const float*)nullptr, (const float4*)G, (const float2*)U, (const float*)R, fl,
                       Ap, (float*)nullptr, (float*)nullptr, (float*)nullptr, aD_out, r, s3_out);
    int e = check_launch(); if (e) return e;
    if (ctc) {      // (A/B path: PCGStep1_Finish as a launch of its own over the owned rows; its partials are the ones returned)
        const long o = (long)W * row0, n = (long)W * (row1 - row0);
        return thallo_hip_lm_step1_finish(Ap + o, ctc + o, p + o, n, aD_out, stream);
    }
    if (fin.tickets && (e = thallo_hip_pcg_scalars_finish(aD_out, s3_out, grid, fin.alphaN, fin.alphaD_word, fin.betaN_word, stream)) < 0) return e;      // (A/B path: the words by a launch of their own)
    return grid;
}

}  // extern "C"
