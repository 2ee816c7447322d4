// energy_image_warping.hip -- plugin for examples/image_warping/image_warping.t (the north-star energy).
//
// Energy (image_warping.t:17-31), i = pixel, j = i+d for d in {(1,0),(-1,0),(0,1),(0,-1)}:
//   reg_d(i) = [InBounds(j) & Mask_i==0 & Mask_j==0] * w_reg * ((o_i - o_j) - R(a_i)(u_i - u_j))   in R^2
//   fit(i)   = w_fit * [c_i.x>=0 & c_i.y>=0 & Mask_i==0] * (o_i - c_i)                              in R^2
//   R(a) = [[cos a, -sin a],[sin a, cos a]]  (lib.t:138-142);  unknowns excluded where Mask != 0 (:14-15)
//
// Hand-derived unknown-wise (gather) forms of what createjtfcentered / createjtjcentered
// (thallo.t:3603-3712) generate.  With g_i = R'(a_i)(u_i-u_j), g_j = R'(a_j)(u_j-u_i),
// v = validity of the (i,j) pair, P = (po, pa) the CG direction:
//   (J^T J P)_o(i) = w_reg^2 * sum_d v [ 2(po_i - po_j) - g_i pa_i + g_j pa_j ] + w_fit^2 [fit valid] po_i
//   (J^T J P)_a(i) = -w_reg^2 * sum_d v  g_i . [ (po_i - po_j) - g_i pa_i ]
//
// Kernel design (MI355X): 64x16-pixel tiles + 1-pixel halo staged in LDS (register-staged loads,
// all of a tile's loads issued before the first use -> ~38 KB in flight per workgroup, 4 workgroups
// per CU), persistent XCD-aware tile sweep, one partial per workgroup.  The step kernel also
// applies the previous iteration's PCGStep3 (p = z + beta p) and delta update on the fly, which
// removes one kernel and 31 B/pixel of traffic per PCG iteration versus the reference's
// PCGStep1 + PCGStep2 + PCGStep3 split (gauss_newton.t:1641-1661).
//
// Algorithmic bytes: applyJTJ 48 B/px; this fused kernel = applyJTJ + PCGStep3 +
// the delta part of PCGStep2: reads z 12, p 12, delta 12, cs 8, u 8, flags 1; writes p 12, Ap 12,
// delta 12 = 89 B/px actual.
#include "device_common.hpp"
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {

constexpr int TW = 64, TH = 16, BLOCK = 256;
constexpr int LW = TW + 2, LH = TH + 2, LN = LW * LH;      // 66 x 18 = 1188 halo'd positions
constexpr int PER_THREAD = TH / (BLOCK / TW);              // 4 pixels per thread, one column

struct Geo { int W, H, tx, ty, ntiles; };
inline Geo make_geo(int W, int H)
{
    Geo g; g.W = W; g.H = H; g.tx = (W + TW - 1) / TW; g.ty = (H + TH - 1) / TH; g.ntiles = g.tx * g.ty;
    return g;
}
inline int grid_for(const Geo& g)
{
    int cap = thallo_hip_device_cu_count() * 4;            // LDS: 4 x 38 KB per CU
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    cap -= cap % 8;
    return g.ntiles < cap ? g.ntiles : cap;
}
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

struct Tile {
    float px[LN], py[LN], pa[LN];   // step: CG direction p ; init: offset.x, offset.y, (unused)
    float c[LN], s[LN];             // cos / sin of Angle
    float ux[LN], uy[LN];           // UrShape
    unsigned char f[LN + 4];        // bit0 active, bit1 fit-valid
};

// ------------------------------------------------------------------------------------------ step1
template <bool FUSED>
__global__ __launch_bounds__(BLOCK) void k_step1(Geo g, const float2* __restrict__ cs, const float2* __restrict__ ur,
                                                  const unsigned char* __restrict__ flags, float wf2, float wr2,
                                                  const float* __restrict__ z, const float* __restrict__ p_in,
                                                  float* __restrict__ p_out, float* __restrict__ delta,
                                                  float* __restrict__ Ap, int first,
                                                  thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp,
                                                  float* __restrict__ aD_out)
{
    __shared__ Tile T;
    __shared__ float red[16];
    const long N = (long)g.W * g.H;
    const float2* __restrict__ zo = reinterpret_cast<const float2*>(z);
    const float2* __restrict__ po = reinterpret_cast<const float2*>(p_in);
    float2* __restrict__ qo = reinterpret_cast<float2*>(p_out);
    float2* __restrict__ dlo = reinterpret_cast<float2*>(delta);
    float2* __restrict__ Ao = reinterpret_cast<float2*>(Ap);
    const float* __restrict__ za = z + 2 * N;  const float* __restrict__ pa = p_in + 2 * N;
    float* __restrict__ qa = p_out + 2 * N;    float* __restrict__ dla = delta + 2 * N;
    float* __restrict__ Aa = Ap + 2 * N;

    float alpha = 0.0f, beta = 0.0f;
    if (FUSED && !first) {   // PCGStep3 of iteration k-1 (gauss_newton.t:892-896) and its alpha (:807-812)
        const float an = sum_partials(aNp.partials, aNp.count);
        alpha = safe_div<false>(an, sum_partials(aDp.partials, aDp.count));
        beta  = safe_div<false>(sum_partials(bNp.partials, bNp.count), an);
    }

    float acc = 0.0f;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x0 = (t.cur % g.tx) * TW, y0 = (t.cur / g.tx) * TH;
        // ---- stage A: load halo'd tile, p = z + beta*p_old, delta += alpha*p_old
#pragma unroll
        for (int it = 0; it < (LN + BLOCK - 1) / BLOCK; ++it) {
            const int idx = it * BLOCK + threadIdx.x;
            if (idx < LN) {
                const int ly = idx / LW, lx = idx - ly * LW;
                const int gx = x0 + lx - 1, gy = y0 + ly - 1;
                float npx = 0.f, npy = 0.f, npa = 0.f, cc = 1.f, ss = 0.f, uxx = 0.f, uyy = 0.f;
                unsigned char ff = 0;
                if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H) {
                    const long pix = (long)gy * g.W + gx;
                    const float2 pv = po[pix]; const float pav = pa[pix];
                    const float2 csv = cs[pix]; const float2 uv = ur[pix];
                    ff = flags[pix];
                    if (FUSED) {
                        const float2 zv = zo[pix]; const float zav = za[pix];
                        npx = zv.x + beta * pv.x; npy = zv.y + beta * pv.y; npa = zav + beta * pav;
                    } else { npx = pv.x; npy = pv.y; npa = pav; }
                    cc = csv.x; ss = csv.y; uxx = uv.x; uyy = uv.y;
                    if (FUSED && lx >= 1 && lx <= TW && ly >= 1 && ly <= TH) {      // owned pixel
                        qo[pix] = make_float2(npx, npy); qa[pix] = npa;
                        if (!first) {
                            float2 dv = dlo[pix]; float da = dla[pix];
                            dv.x += alpha * pv.x; dv.y += alpha * pv.y; da += alpha * pav;
                            dlo[pix] = dv; dla[pix] = da;
                        }
                    }
                }
                T.px[idx] = npx; T.py[idx] = npy; T.pa[idx] = npa;
                T.c[idx] = cc; T.s[idx] = ss; T.ux[idx] = uxx; T.uy[idx] = uyy; T.f[idx] = ff;
            }
        }
        __syncthreads();
        // ---- stage B: gather J^T J p
        const int lx = (threadIdx.x % TW) + 1;
        const int gx = x0 + lx - 1;
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int ly = (threadIdx.x / TW) * PER_THREAD + k + 1;
            const int gy = y0 + ly - 1;
            const int i = ly * LW + lx;
            if (gx < g.W && gy < g.H) {
                const long pix = (long)gy * g.W + gx;
                const unsigned char fi = T.f[i];
                float ax = 0.f, ay = 0.f, aa = 0.f;
                const float pxi = T.px[i], pyi = T.py[i], pai = T.pa[i];
                if (fi & 1) {
                    const float ci = T.c[i], si = T.s[i], uxi = T.ux[i], uyi = T.uy[i];
                    const int nb[4] = { i + 1, i - 1, i + LW, i - LW };
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int j = nb[d];
                        if (T.f[j] & 1) {
                            const float dux = uxi - T.ux[j], duy = uyi - T.uy[j];
                            const float gix = -si * dux - ci * duy, giy = ci * dux - si * duy;
                            const float cj = T.c[j], sj = T.s[j], paj = T.pa[j];
                            const float gjx = sj * dux + cj * duy, gjy = -cj * dux + sj * duy;
                            const float dpx = pxi - T.px[j], dpy = pyi - T.py[j];
                            const float ex = dpx - gix * pai, ey = dpy - giy * pai;
                            ax += dpx + ex + gjx * paj;
                            ay += dpy + ey + gjy * paj;
                            aa -= gix * ex + giy * ey;
                        }
                    }
                    ax *= wr2; ay *= wr2; aa *= wr2;
                    if (fi & 2) { ax += wf2 * pxi; ay += wf2 * pyi; }
                }
                Ao[pix] = make_float2(ax, ay); Aa[pix] = aa;
                acc += pxi * ax + pyi * ay + pai * aa;
            }
        }
        __syncthreads();
    }
    block_store_partial(acc, aD_out, red);
}

// ------------------------------------------------------------------------------------------ init
__global__ __launch_bounds__(BLOCK) void k_init(Geo g, const float2* __restrict__ off, const float* __restrict__ ang,
                                                const float2* __restrict__ ur, const float2* __restrict__ cons,
                                                const float* __restrict__ mask, float wf, float wr,
                                                float* __restrict__ r, float* __restrict__ pre, float* __restrict__ z,
                                                float* __restrict__ p_prev, float* __restrict__ delta,
                                                float2* __restrict__ cs, unsigned char* __restrict__ flags,
                                                float* __restrict__ aN_out)
{
    __shared__ Tile T;
    __shared__ float red[16];
    const long N = (long)g.W * g.H;
    const float wr2 = wr * wr, wf2 = wf * wf;
    float acc = 0.0f;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x0 = (t.cur % g.tx) * TW, y0 = (t.cur / g.tx) * TH;
        for (int idx = threadIdx.x; idx < LN; idx += BLOCK) {
            const int ly = idx / LW, lx = idx - ly * LW;
            const int gx = x0 + lx - 1, gy = y0 + ly - 1;
            float ox = 0.f, oy = 0.f, cc = 1.f, ss = 0.f, uxx = 0.f, uyy = 0.f; unsigned char ff = 0;
            if (gx >= 0 && gx < g.W && gy >= 0 && gy < g.H) {
                const long pix = (long)gy * g.W + gx;
                const float2 o = off[pix]; const float2 u = ur[pix];
                ox = o.x; oy = o.y; uxx = u.x; uyy = u.y;
                sincosf(ang[pix], &ss, &cc);
                ff = mask[pix] == 0.0f ? 1 : 0;
            }
            T.px[idx] = ox; T.py[idx] = oy; T.c[idx] = cc; T.s[idx] = ss; T.ux[idx] = uxx; T.uy[idx] = uyy; T.f[idx] = ff;
        }
        __syncthreads();
        const int lx = (threadIdx.x % TW) + 1;
        const int gx = x0 + lx - 1;
        for (int k = 0; k < PER_THREAD; ++k) {
            const int ly = (threadIdx.x / TW) * PER_THREAD + k + 1;
            const int gy = y0 + ly - 1;
            const int i = ly * LW + lx;
            if (gx < g.W && gy < g.H) {
                const long pix = (long)gy * g.W + gx;
                const unsigned char act = T.f[i] & 1;
                float rx = 0.f, ry = 0.f, ra = 0.f, mx = 0.f, my = 0.f, ma = 0.f;
                unsigned char fl = act;
                if (act) {
                    const float oxi = T.px[i], oyi = T.py[i], ci = T.c[i], si = T.s[i], uxi = T.ux[i], uyi = T.uy[i];
                    float jx = 0.f, jy = 0.f, ja = 0.f, dgo = 0.f, dga = 0.f;
                    const int nb[4] = { i + 1, i - 1, i + LW, i - LW };
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int j = nb[d];
                        if (T.f[j] & 1) {
                            const float dux = uxi - T.ux[j], duy = uyi - T.uy[j];
                            const float dox = oxi - T.px[j], doy = oyi - T.py[j];
                            // e_i = (o_i-o_j) - R(a_i)(u_i-u_j) ; e_j = (o_j-o_i) - R(a_j)(u_j-u_i)
                            const float eix = dox - (ci * dux - si * duy), eiy = doy - (si * dux + ci * duy);
                            const float cj = T.c[j], sj = T.s[j];
                            const float ejx = -dox + (cj * dux - sj * duy), ejy = -doy + (sj * dux + cj * duy);
                            const float gix = -si * dux - ci * duy, giy = ci * dux - si * duy;
                            jx += eix - ejx; jy += eiy - ejy;
                            ja -= gix * eix + giy * eiy;
                            dgo += 2.0f; dga += gix * gix + giy * giy;
                        }
                    }
                    jx *= wr2; jy *= wr2; ja *= wr2; dgo *= wr2; dga *= wr2;
                    const float2 cv = cons[pix];
                    if (cv.x >= 0.0f && cv.y >= 0.0f) {            // image_warping.t:27 (Mask==0 already holds)
                        fl |= 2;
                        jx += wf2 * (oxi - cv.x); jy += wf2 * (oyi - cv.y); dgo += wf2;
                    }
                    rx = -jx; ry = -jy; ra = -ja;                   // gauss_newton.t:690
                    mx = guarded_invert(dgo); my = mx; ma = guarded_invert(dga);   // :696, UsePreconditioner(true)
                }
                reinterpret_cast<float2*>(r)[pix] = make_float2(rx, ry);  r[2 * N + pix] = ra;
                reinterpret_cast<float2*>(pre)[pix] = make_float2(mx, my); pre[2 * N + pix] = ma;
                const float zx = mx * rx, zy = my * ry, zaa = ma * ra;
                reinterpret_cast<float2*>(z)[pix] = make_float2(zx, zy);  z[2 * N + pix] = zaa;
                reinterpret_cast<float2*>(p_prev)[pix] = make_float2(0.f, 0.f); p_prev[2 * N + pix] = 0.f;
                reinterpret_cast<float2*>(delta)[pix] = make_float2(0.f, 0.f);  delta[2 * N + pix] = 0.f;
                cs[pix] = make_float2(T.c[i], T.s[i]);
                flags[pix] = fl;
                acc += rx * zx + ry * zy + ra * zaa;                // :701
            }
        }
        __syncthreads();
    }
    block_store_partial(acc, aN_out, red);
}

// ------------------------------------------------------------------------------------------ cost
// computeCost (gauss_newton.t:1067-1079, thallo.t:3939-3949): once per solve (+ per LM step).
constexpr int CW = 64, CH = 4;
__global__ __launch_bounds__(BLOCK) void k_cost(int W, int H, int ctx, int ntiles,
                                                const float2* __restrict__ off, const float* __restrict__ ang,
                                                const float2* __restrict__ ur, const float2* __restrict__ cons,
                                                const float* __restrict__ mask, float wf, float wr, float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (TileSweep t(ntiles); t.valid(); t.next()) {
        const int x = (t.cur % ctx) * CW + (threadIdx.x % CW), y = (t.cur / ctx) * CH + (threadIdx.x / CW);
        if (x < W && y < H) {
            const long i = (long)y * W + x;
            if (mask[i] == 0.0f) {
                const float2 o = off[i]; const float2 u = ur[i];
                float si, ci; sincosf(ang[i], &si, &ci);
                float s2 = 0.f;
                const int dx[4] = { 1, -1, 0, 0 }, dy[4] = { 0, 0, 1, -1 };
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int xn = x + dx[d], yn = y + dy[d];
                    if (xn >= 0 && xn < W && yn >= 0 && yn < H) {
                        const long j = (long)yn * W + xn;
                        if (mask[j] == 0.0f) {
                            const float2 oj = off[j]; const float2 uj = ur[j];
                            const float dux = u.x - uj.x, duy = u.y - uj.y;
                            const float ex = wr * ((o.x - oj.x) - (ci * dux - si * duy));
                            const float ey = wr * ((o.y - oj.y) - (si * dux + ci * duy));
                            s2 += ex * ex + ey * ey;
                        }
                    }
                }
                const float2 cv = cons[i];
                if (cv.x >= 0.0f && cv.y >= 0.0f) {
                    const float fx = wf * (o.x - cv.x), fy = wf * (o.y - cv.y);
                    s2 += fx * fx + fy * fy;
                }
                acc += 0.5f * s2;
            }
        }
    }
    block_store_partial(acc, out, red);
}

}  // namespace

extern "C" {

int thallo_hip_iw_cost(int W, int H, const float* offset, const float* angle, const float* urshape,
                       const float* constraints, const float* mask, float w_fit, float w_reg,
                       float* cost_out, thallo_stream_t stream)
{
    const int ctx = (W + CW - 1) / CW, cty = (H + CH - 1) / CH, nt = ctx * cty;
    int grid = thallo_hip_device_cu_count() * 4; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS;
    grid -= grid % 8; if (nt < grid) grid = nt;
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, W, H, ctx, nt,
                       (const float2*)offset, angle, (const float2*)urshape, (const float2*)constraints, mask, w_fit, w_reg, cost_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_init(int W, int H, const float* offset, const float* angle, const float* urshape,
                           const float* constraints, const float* mask, float w_fit, float w_reg,
                           float* r, float* pre, float* z, float* p_prev, float* delta,
                           float* cs, unsigned char* flags, float* aN_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                       (const float2*)offset, angle, (const float2*)urshape, (const float2*)constraints, mask, w_fit, w_reg,
                       r, pre, z, p_prev, delta, (float2*)cs, flags, aN_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_step1(int W, int H, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg,
                            const float* z, const float* p_in, float* p_out, float* delta, float* Ap,
                            int first, thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp,
                            float* aD_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    hipLaunchKernelGGL(k_step1<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                       (const float2*)cs, (const float2*)urshape, flags, w_fit * w_fit, w_reg * w_reg,
                       z, p_in, p_out, delta, Ap, first, aNp, aDp, bNp, aD_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_apply_jtj(int W, int H, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg, const float* p, float* Ap, float* aD_out, thallo_stream_t stream)
{
    const Geo g = make_geo(W, H); const int grid = grid_for(g);
    thallo_sum_t none; none.partials = nullptr; none.count = 0;
    hipLaunchKernelGGL(k_step1<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                       (const float2*)cs, (const float2*)urshape, flags, w_fit * w_fit, w_reg * w_reg,
                       p, p, Ap, Ap, Ap, 1, none, none, none, aD_out);   /* z/p_out/delta unused when !FUSED */
    int e = check_launch(); return e ? e : grid;
}

}  // extern "C"
