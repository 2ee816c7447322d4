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
// Kernel design (MI355X):
//  * 64x16-pixel tiles (+1-pixel halo) = 1024 pixels per 256-thread workgroup; a thread OWNS one column
//    of 4 pixels from load to store: its centre values stay in registers, only neighbours are read from
//    the LDS image of the tile; the 164-position halo ring is loaded by the first 164 threads.
//  * software pipeline: the loads of tile t+1 are issued right after tile t's LDS image is published and
//    stay in flight during tile t's arithmetic.  The barriers are raw s_barrier + lgkmcnt(0): a
//    __syncthreads() would drain vmcnt(0) and serialise the prefetch.
//  * persistent grid (3 workgroups per CU: 150 VGPRs), XCD-aware contiguous tile ranges so that the halo
//    re-reads of neighbouring tiles hit the same XCD's L2 (measured: HBM fetch = compulsory bytes + 8 %).
//  * PCGStep1 also applies the previous iteration's PCGStep3 (p = z + beta p) and delta update on the
//    fly: one kernel and 31 B/pixel less per PCG iteration than the reference's split
//    (gauss_newton.t:1641-1661).  One partial per workgroup for alphaD, no atomics.
//  * row slabs (multi-GPU): every kernel works on the owned rows [row0,row1) of a local
//    image that may carry ghost rows above/below; ghost rows are read as halo, and the fused step also
//    maintains p on them (p = z + beta p) so only z crosses the wire each iteration.
//
// Bytes per pixel: fused PCGStep1 reads z 12, p 12, delta 12, cs 8, UrShape 8, flags 1; writes p 12,
// Ap 12, delta 12 = 89 actual (algorithmic, reference formulation of the fused ops: 96).
// Plain applyJTJ: reads p 12, cs 8, UrShape 8, flags 1; writes Ap 12 = 41 actual (algorithmic 48).
#include "iw_device.hpp"

using namespace thallo;

namespace {

constexpr int TW = 64, TH = 16, BLOCK = 256;
constexpr int LW = TW + 2, LH = TH + 2, LN = LW * LH;      // 66 x 18 = 1188 halo'd positions
constexpr int PER = TH / (BLOCK / TW);                     // 4 owned pixels per thread (one column)
constexpr int HALO = 2 * LW + 2 * TH;                      // 164 ring positions

inline Geo make_geo(int W, int H, int row0, int row1)
{
    Geo g; g.W = W; g.H = H; g.row0 = row0; g.row1 = row1;
    g.tx = (W + TW - 1) / TW; g.ty = (row1 - row0 + TH - 1) / TH; g.ntiles = g.tx * g.ty;
    return g;
}
inline bool rows_ok(int H, int row0, int row1) { return row0 >= 0 && row1 <= H && row0 < row1; }
inline int grid_for(const Geo& g, int per_cu)
{
    int cap = thallo_hip_device_cu_count() * per_cu;
    if (cap > THALLO_MAX_PARTIALS) cap = THALLO_MAX_PARTIALS;
    cap -= cap % 8;
    return g.ntiles < cap ? g.ntiles : cap;
}
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

// debug hooks of rounds 1-2's kernel experiments (thallo_hip_debug_set; not product API): diagnostic mode 1 = skip arithmetic, 2 = skip loads;
// cache-policy bits for PCGStep1: 1 delta nt, 2 cs/UrShape/flags nt, 4 p_in nt, 8 Ap nt, 32 z nt, 64 p_out nt
int g_iw_debug = 0;
int g_step1_threads = 512; // fused step: 512 threads x 2 px/thread (default) or 256 x 4
int g_step1_per_cu = 3; // microbench: workgroups per CU for the fused step (3 = VGPR limit)
int g_no_grid = 0;      // microbench: 1 = ignore the regular-grid fast path
int g_iter_per_cu = 2;    // microbench: workgroups per CU of the one-kernel iteration (2 = VGPR limit at 512 threads)
int g_iter_nt = 31;      // one-kernel iteration, non-temporal bits: 1 delta, 2 r/Ap loads, 4 r/Ap stores, 8 p loads, 16 p stores, 32 cs/flags (31 measured best)
int g_nt_mask = 1;      // delta non-temporal: measured +1-3 % PCG it/s at 2048^2 (round 1, docs/history)

struct Tile {
    float px[LN], py[LN], pa[LN];   // step: CG direction p ; init: offset.x, offset.y
    float c[LN], s[LN];             // cos / sin of Angle
    float ux[LN], uy[LN];           // UrShape
    unsigned char f[LN + 4];        // bit0 active (Mask==0), bit1 fit-valid
};

struct Owned { float2 zv, pv, dv, csv, uv, ppv; float zav, pav, da, ppa; unsigned char ff; };
struct HaloLd { float2 zv, pv, csv, uv; float zav, pav; unsigned char ff; };

// ------------------------------------------------------------------------------------------ PCGStep1
// GRID = UrShape is the unit pixel grid (what the reference's harness always passes, CombinedSolver.h:158-176):
// u_i - u_j is then exactly -(dx,dy), so the UrShape plane is neither loaded nor staged (-8 B/pixel, -2 LDS planes).
// pcg_init verifies the property bit-exactly on the device every GN step; both paths give identical bits.
template <bool FUSED, bool GRID, int NT>
__device__ __forceinline__ void step1_body(Tile& T, float* red, const Geo& g, const float2* __restrict__ cs, const float2* __restrict__ ur,
                                           const unsigned char* __restrict__ flags, float wf2, float wr2,
                                           const float* __restrict__ z, const float* __restrict__ p_in,
                                           float* __restrict__ p_out, float* __restrict__ delta,
                                           float* __restrict__ Ap, int mode,
                                           thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                           float* __restrict__ aD_out, int dbg)
{   // mode: bit0 = first PCG iteration (p = z, no delta update); bits1-2 = delta update of this launch:
    //   0: delta += alpha_{k-1} p_{k-1}   1: none (deferred)   2: delta += alpha_{k-2} p_{k-2} ; += alpha_{k-1} p_{k-1}
    // (2 reads p_{k-2} from the p_out buffer just before overwriting it; alternating 1 / 2 halves delta's read+write traffic
    //  and gives the same bits as updating every iteration: the adds happen in the same order)
    const int first = mode & 1;
    const int dmode = first ? 1 : (mode >> 1) & 3;
    // GRID: `z` points at r, and z = M^-1 r is formed here from the flags byte (pre_from_flags)
    constexpr int PER = TH / (NT / TW);                    // owned pixels per thread (one column): 4 at 256 threads, 2 at 512
    const int ntm = dbg >> 8; dbg &= 0xff;
    const bool nt_delta = ntm & 1, nt_const = ntm & 2, nt_pin = ntm & 4, nt_ap = ntm & 8, nt_z = ntm & 32, nt_pout = ntm & 64;
    const long N = (long)g.W * g.H;
    const float2* __restrict__ zo = reinterpret_cast<const float2*>(z);
    const float2* __restrict__ po = reinterpret_cast<const float2*>(p_in);
    float2* __restrict__ qo = reinterpret_cast<float2*>(p_out);
    float2* __restrict__ dlo = reinterpret_cast<float2*>(delta);
    float2* __restrict__ Ao = reinterpret_cast<float2*>(Ap);
    const float* __restrict__ za = z + 2 * N;  const float* __restrict__ pa = p_in + 2 * N;
    float* __restrict__ qa = p_out + 2 * N;    float* __restrict__ dla = delta + 2 * N;
    float* __restrict__ Aa = Ap + 2 * N;

    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * PER;      // owned column / first owned row in the tile
    int hlx, hly;                                                        // this thread's halo-ring position
    {
        const int h = threadIdx.x;
        if (h < LW) { hlx = h; hly = 0; }
        else if (h < 2 * LW) { hlx = h - LW; hly = LH - 1; }
        else if (h < 2 * LW + TH) { hlx = 0; hly = h - 2 * LW + 1; }
        else { hlx = LW - 1; hly = h - 2 * LW - TH + 1; }
    }
    const bool has_halo = threadIdx.x < HALO;

    Owned ow[PER]; HaloLd hl;
    bool ow_ld[PER], ow_in[PER]; bool hl_in = false;      // loaded (exists in the local image) / owned (row < row1)

    auto issue_loads = [&](int tile) {
        const int x0 = (tile % g.tx) * TW, y0 = g.row0 + (tile / g.tx) * TH;
        const int gx = x0 + tx;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int gy = y0 + ty + k;
            ow_ld[k] = (dbg != 2) && gx < g.W && gy < g.H;     // a slab's bottom ghost row may fall inside the last tile
            ow_in[k] = ow_ld[k] && gy < g.row1;
            if (ow_ld[k]) {
                const long pix = (long)gy * g.W + gx;
                ow[k].pv = ldf2(po + pix, nt_pin); ow[k].pav = ldf(pa + pix, nt_pin);
                ow[k].csv = ldf2(cs + pix, nt_const); ow[k].ff = ldb(flags + pix, nt_const);
                if (!GRID) ow[k].uv = ldf2(ur + pix, nt_const);
                if (FUSED) {
                    ow[k].zv = ldf2(zo + pix, nt_z); ow[k].zav = ldf(za + pix, nt_z);
                    if (dmode != 1 && ow_in[k]) {
                        ow[k].dv = ldf2(dlo + pix, nt_delta); ow[k].da = ldf(dla + pix, nt_delta);
                        if (dmode == 2) { ow[k].ppv = ldf2(qo + pix, nt_pin); ow[k].ppa = ldf(qa + pix, nt_pin); }
                    }
                }
            }
        }
        if (has_halo) {
            const int hx = x0 + hlx - 1, hy = y0 + hly - 1;
            hl_in = (dbg != 2) && hx >= 0 && hx < g.W && hy >= 0 && hy < g.H;
            if (hl_in) {
                const long pix = (long)hy * g.W + hx;
                hl.pv = po[pix]; hl.pav = pa[pix]; hl.csv = cs[pix]; hl.ff = flags[pix];
                if (!GRID) hl.uv = ur[pix];
                if (FUSED) { hl.zv = zo[pix]; hl.zav = za[pix]; }
            }
        }
    };

    TileSweep t(g.ntiles);
    float alpha = 0.0f, beta = 0.0f, alpha2 = 0.0f;
    if (t.valid()) issue_loads(t.cur);
    if (FUSED && !first) {   // PCGStep3 of iteration k-1 (gauss_newton.t:892-896) and its alpha (:807-812)
        const float an = sum_partials(aNp.partials, aNp.count);
        alpha = safe_div<false>(an, sum_partials(aDp.partials, aDp.count));
        beta  = safe_div<false>(sum_partials(bNp.partials, bNp.count), an);
        if (dmode == 2) alpha2 = safe_div<false>(sum_partials(aNpp.partials, aNpp.count), sum_partials(aDpp.partials, aDpp.count));
    }

    float acc = 0.0f;
    while (t.valid()) {
        const int x0 = (t.cur % g.tx) * TW, y0 = g.row0 + (t.cur / g.tx) * TH;
        const int gx = x0 + tx;
        // ---- publish: p = z + beta*p_old (owned + halo) -> LDS ; owned: p_out, delta += alpha*p_old
        float cpx[PER], cpy[PER], cpa[PER], cc[PER], cs_[PER], cux[PER], cuy[PER];
        unsigned char cf[PER]; bool cin[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = (ty + k + 1) * LW + tx + 1;
            float npx = 0.f, npy = 0.f, npa = 0.f, c1 = 1.f, s1 = 0.f, u1 = 0.f, u2 = 0.f; unsigned char ff = 0;
            cin[k] = ow_in[k];
            if (ow_ld[k]) {
                const long pix = (long)(y0 + ty + k) * g.W + gx;
                if (FUSED) {
                    float zx = ow[k].zv.x, zy = ow[k].zv.y, zq = ow[k].zav;
                    if (GRID) { float mo, ma; pre_from_flags(ow[k].ff, wf2, wr2, mo, ma); zx *= mo; zy *= mo; zq *= ma; }
                    npx = zx + beta * ow[k].pv.x; npy = zy + beta * ow[k].pv.y; npa = zq + beta * ow[k].pav;
                    stf2(qo + pix, make_float2(npx, npy), nt_pout); stf(qa + pix, npa, nt_pout);      // owned, or ghost row kept current
                    if (dmode != 1 && ow_in[k]) {
                        float dx = ow[k].dv.x, dy = ow[k].dv.y, dq = ow[k].da;
                        // explicit fma everywhere a delta update is applied (here, linear_update, linear_update2): deferring an
                        // update must not change a single bit
                        if (dmode == 2) { dx = __builtin_fmaf(alpha2, ow[k].ppv.x, dx); dy = __builtin_fmaf(alpha2, ow[k].ppv.y, dy); dq = __builtin_fmaf(alpha2, ow[k].ppa, dq); }
                        stf2(dlo + pix, make_float2(__builtin_fmaf(alpha, ow[k].pv.x, dx), __builtin_fmaf(alpha, ow[k].pv.y, dy)), nt_delta);
                        stf(dla + pix, __builtin_fmaf(alpha, ow[k].pav, dq), nt_delta);
                    }
                } else { npx = ow[k].pv.x; npy = ow[k].pv.y; npa = ow[k].pav; }
                c1 = ow[k].csv.x; s1 = ow[k].csv.y; ff = ow[k].ff;
                if (!GRID) { u1 = ow[k].uv.x; u2 = ow[k].uv.y; }
            }
            T.px[i] = npx; T.py[i] = npy; T.pa[i] = npa; T.c[i] = c1; T.s[i] = s1; T.f[i] = ff;
            if (!GRID) { T.ux[i] = u1; T.uy[i] = u2; }
            cpx[k] = npx; cpy[k] = npy; cpa[k] = npa; cc[k] = c1; cs_[k] = s1; cux[k] = u1; cuy[k] = u2; cf[k] = ff;
        }
        if (has_halo) {
            const int i = hly * LW + hlx;
            float npx = 0.f, npy = 0.f, npa = 0.f, c1 = 1.f, s1 = 0.f, u1 = 0.f, u2 = 0.f; unsigned char ff = 0;
            if (hl_in) {
                if (FUSED) {
                    float zx = hl.zv.x, zy = hl.zv.y, zq = hl.zav;
                    if (GRID) { float mo, ma; pre_from_flags(hl.ff, wf2, wr2, mo, ma); zx *= mo; zy *= mo; zq *= ma; }
                    npx = zx + beta * hl.pv.x; npy = zy + beta * hl.pv.y; npa = zq + beta * hl.pav;
                    // ghost row of a slab (not owned by any tile of this rank): keep its p current
                    const int hy = y0 + hly - 1;
                    if ((hy < g.row0 || hy >= g.row1) && hlx >= 1 && hlx <= TW) {
                        const long pix = (long)hy * g.W + (x0 + hlx - 1);
                        qo[pix] = make_float2(npx, npy); qa[pix] = npa;
                    }
                } else { npx = hl.pv.x; npy = hl.pv.y; npa = hl.pav; }
                c1 = hl.csv.x; s1 = hl.csv.y; ff = hl.ff;
                if (!GRID) { u1 = hl.uv.x; u2 = hl.uv.y; }
            }
            T.px[i] = npx; T.py[i] = npy; T.pa[i] = npa; T.c[i] = c1; T.s[i] = s1; T.f[i] = ff;
            if (!GRID) { T.ux[i] = u1; T.uy[i] = u2; }
        }
        lds_barrier();
        // ---- prefetch the next tile while this one is computed
        const int cur_y0 = y0;
        t.next();
        if (t.valid()) issue_loads(t.cur);
        // ---- gather J^T J p for the owned pixels
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (cin[k] || (dbg == 2 && gx < g.W && cur_y0 + ty + k < g.row1)) {
                const int i = (ty + k + 1) * LW + tx + 1;
                const long pix = (long)(cur_y0 + ty + k) * g.W + gx;
                float ax = 0.f, ay = 0.f, aa = 0.f;
                const float pxi = cpx[k], pyi = cpy[k], pai = cpa[k];
                if ((cf[k] & 1) && dbg != 1) {
                    const float ci = cc[k], si = cs_[k], uxi = cux[k], uyi = cuy[k];
                    const int nb[4] = { i + 1, i - 1, i + LW, i - LW };
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int j = nb[d];
                        if (T.f[j] & 1) {
                            float dux, duy;
                            if (GRID) { dux = d == 0 ? -1.0f : d == 1 ? 1.0f : 0.0f; duy = d == 2 ? -1.0f : d == 3 ? 1.0f : 0.0f; }
                            else { dux = uxi - T.ux[j]; duy = uyi - T.uy[j]; }
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
                    if (cf[k] & 2) { ax += wf2 * pxi; ay += wf2 * pyi; }
                }
                stf2(Ao + pix, make_float2(ax, ay), nt_ap); stf(Aa + pix, aa, nt_ap);
                acc += pxi * ax + pyi * ay + pai * aa;
            }
        }
        lds_barrier();
    }
    block_store_partial(acc, aD_out, red);
}

template <bool FUSED, int MINW, int NT>
__global__ __launch_bounds__(NT, MINW) void k_step1(Geo g, const float2* __restrict__ cs, const float2* __restrict__ ur,
                                                        const unsigned char* __restrict__ flags, float wf2, float wr2,
                                                        const float* __restrict__ z, const float* __restrict__ p_in,
                                                        float* __restrict__ p_out, float* __restrict__ delta,
                                                        float* __restrict__ Ap, int mode,
                                                        thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                                        float* __restrict__ aD_out, int dbg, const int* __restrict__ irregular,
                                                        const float* __restrict__ r)
{
    __shared__ Tile T;
    __shared__ float red[16];
    // `irregular` = number of pixels whose UrShape neighbours are not at unit offsets (written by pcg_init); wave-uniform
    const bool grid = irregular != nullptr && __builtin_amdgcn_readfirstlane(irregular[0]) == 0;
    if (grid) step1_body<FUSED, true, NT>(T, red, g, cs, ur, flags, wf2, wr2, FUSED ? r : z, p_in, p_out, delta, Ap, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, dbg);
    else      step1_body<FUSED, false, NT>(T, red, g, cs, ur, flags, wf2, wr2, z, p_in, p_out, delta, Ap, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, dbg);
}

// ------------------------------------------------------------------------------------------ PCGStep2
// gauss_newton.t:801-843 without the delta half (fused into the next PCGStep1).  On the GRID path M^-1 comes from the
// flags byte and z is not written: reads r 12, Ap 12, flags 1; writes r 12 = 37 B/pixel instead of 60.
// Otherwise the generic form (reads pre, writes z).  Two float4 ranges: Offset plane rows, Angle plane rows.
template <bool DIST>
__global__ __launch_bounds__(BLOCK) void k_step2_iw(float4* __restrict__ r, const float4* __restrict__ Ap, const float4* __restrict__ pre,
                                                     float4* __restrict__ z, const unsigned char* __restrict__ flags, long N, float wf2, float wr2,
                                                     long off0, long len0, long off1, long len1,
                                                     thallo_sum_t aN, thallo_sum_t aD, const int* __restrict__ irregular, float* __restrict__ bN_out,
                                                     thallo_dist_t dd, long row_o4, long row_a4)
{   // DIST (multi-GPU row slabs): the first / last owned row of r is also stored into the neighbour's ghost row, peer-to-peer.
    // row_o4 / row_a4 = float4s per image row in the Offset / Angle plane
    __shared__ float red[16];
    const bool grid = DIST || (irregular != nullptr && __builtin_amdgcn_readfirstlane(irregular[0]) == 0);
    const float alpha = safe_div<false>(sum_partials(aN.partials, aN.count), sum_partials(aD.partials, aD.count));
    float acc = 0.0f;
    const long n4 = len0 + len1;
    for (long j = (long)blockIdx.x * BLOCK + threadIdx.x; j < n4; j += (long)gridDim.x * BLOCK) {
        const bool in_off = j < len0;
        const long i = in_off ? off0 + j : off1 + (j - len0);
        float4 rv = ldf4(r + i, true);
        const float4 av = Ap[i];
        rv.x -= alpha * av.x; rv.y -= alpha * av.y; rv.z -= alpha * av.z; rv.w -= alpha * av.w;
        float4 m;
        if (grid) {
            if (in_off) {               // elements 4i..4i+3 of the Offset plane = pixels 2i, 2i+1 (x,y each)
                const unsigned f2 = reinterpret_cast<const unsigned short*>(flags)[i];
                float m0, m1, dummy;
                pre_from_flags((unsigned char)(f2 & 255), wf2, wr2, m0, dummy); pre_from_flags((unsigned char)(f2 >> 8), wf2, wr2, m1, dummy);
                m = make_float4(m0, m0, m1, m1);
            } else {                    // Angle plane: element e -> pixel e - 2N (a multiple of 4: 2N % 4 == 0)
                const unsigned f4 = reinterpret_cast<const unsigned*>(flags)[i - N / 2];
                float d, a0, a1, a2, a3;
                pre_from_flags((unsigned char)(f4 & 255), wf2, wr2, d, a0); pre_from_flags((unsigned char)((f4 >> 8) & 255), wf2, wr2, d, a1);
                pre_from_flags((unsigned char)((f4 >> 16) & 255), wf2, wr2, d, a2); pre_from_flags((unsigned char)(f4 >> 24), wf2, wr2, d, a3);
                m = make_float4(a0, a1, a2, a3);
            }
        } else m = pre[i];
        const float4 zv = make_float4(m.x * rv.x, m.y * rv.y, m.z * rv.z, m.w * rv.w);
        stf4(r + i, rv, true);
        if (!grid) z[i] = zv;
        if (DIST) {     // write-through system-scope stores; complete (acknowledged) when this kernel ends, i.e. before the
                        // exchange kernel behind it on the stream sends this rank's betaN granule
            const long jj = in_off ? j : j - len0, rowlen = in_off ? row_o4 : row_a4, total = in_off ? len0 : len1;
            if (jj < rowlen && dd.peer_r[0]) {
                float* dst = dd.peer_r[0] + (in_off ? dd.peer_off_o[0] : dd.peer_off_a[0]) + 4 * jj;
                st_sys(dst, rv.x); st_sys(dst + 1, rv.y); st_sys(dst + 2, rv.z); st_sys(dst + 3, rv.w);
            }
            if (jj >= total - rowlen && dd.peer_r[1]) {
                float* dst = dd.peer_r[1] + (in_off ? dd.peer_off_o[1] : dd.peer_off_a[1]) + 4 * (jj - (total - rowlen));
                st_sys(dst, rv.x); st_sys(dst + 1, rv.y); st_sys(dst + 2, rv.z); st_sys(dst + 3, rv.w);
            }
        }
        acc += zv.x * rv.x + zv.y * rv.y + zv.z * rv.z + zv.w * rv.w;
    }
    block_store_partial(acc, bN_out, red);
}

// ------------------------------------------------------------------------------------------ PCGInit1 (+_Finish)
// Once per GN iteration: evalJTF in gather form, guardedInvert, z = M^-1 r, p_prev = 0, delta = 0, the
// (cos,sin) and validity planes, alphaN partials.  Not pipelined (1 % of a GN iteration).
__global__ __launch_bounds__(BLOCK) void k_init(Geo g, const float2* __restrict__ off, const float* __restrict__ ang,
                                                const float2* __restrict__ ur, const float2* __restrict__ cons,
                                                const float* __restrict__ mask, float wf, float wr,
                                                float* __restrict__ r, float* __restrict__ pre, float* __restrict__ z,
                                                float* __restrict__ p_prev, float* __restrict__ delta,
                                                float2* __restrict__ cs, unsigned char* __restrict__ flags,
                                                float* __restrict__ diag_out, int* __restrict__ irregular, float* __restrict__ aN_out)
{
    __shared__ Tile T;
    __shared__ float red[16];
    const long N = (long)g.W * g.H;
    const float wr2 = wr * wr, wf2 = wf * wf;
    float acc = 0.0f;
    int bad = 0;
    for (TileSweep t(g.ntiles); t.valid(); t.next()) {
        const int x0 = (t.cur % g.tx) * TW, y0 = g.row0 + (t.cur / g.tx) * TH;
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
                if ((gy < g.row0 || gy >= g.row1) && lx >= 1 && lx <= TW) {
                    // ghost row of a slab: the step kernel reads these planes as halo
                    cs[pix] = make_float2(cc, ss); flags[pix] = ff;
                    reinterpret_cast<float2*>(p_prev)[pix] = make_float2(0.f, 0.f); p_prev[2 * N + pix] = 0.f;
                }
            }
            T.px[idx] = ox; T.py[idx] = oy; T.c[idx] = cc; T.s[idx] = ss; T.ux[idx] = uxx; T.uy[idx] = uyy; T.f[idx] = ff;
        }
        __syncthreads();
        const int lx = (threadIdx.x % TW) + 1;
        const int gx = x0 + lx - 1;
        for (int k = 0; k < PER; ++k) {
            const int ly = (threadIdx.x / TW) * PER + k + 1;
            const int gy = y0 + ly - 1;
            const int i = ly * LW + lx;
            if (gx < g.W && gy < g.row1) {
                const long pix = (long)gy * g.W + gx;
                const unsigned char act = T.f[i] & 1;
                // is UrShape the unit pixel grid here?  (right and down neighbour inside the local image)
                if (gx + 1 < g.W && (T.ux[i + 1] - T.ux[i] != 1.0f || T.uy[i + 1] - T.uy[i] != 0.0f)) bad = 1;
                if (gy + 1 < g.H && (T.ux[i + LW] - T.ux[i] != 0.0f || T.uy[i + LW] - T.uy[i] != 1.0f)) bad = 1;
                float rx = 0.f, ry = 0.f, ra = 0.f, mx = 0.f, my = 0.f, ma = 0.f, dgo_raw = 0.f, dga_raw = 0.f;
                unsigned char fl = act;
                if (act) {
                    const float oxi = T.px[i], oyi = T.py[i], ci = T.c[i], si = T.s[i], uxi = T.ux[i], uyi = T.uy[i];
                    float jx = 0.f, jy = 0.f, ja = 0.f, dgo = 0.f, dga = 0.f;
                    int cnt = 0; bool unit = true;
                    const int nb[4] = { i + 1, i - 1, i + LW, i - LW };
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int j = nb[d];
                        if (T.f[j] & 1) {
                            const float dux = uxi - T.ux[j], duy = uyi - T.uy[j];
                            unit = unit && dux == (d == 0 ? -1.0f : d == 1 ? 1.0f : 0.0f) && duy == (d == 2 ? -1.0f : d == 3 ? 1.0f : 0.0f);
                            ++cnt;
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
                    if (unit) dga = (float)cnt * wr2;      // |R'(a) du|^2 = 1 exactly on the unit grid (same value the GRID step path recomputes)
                    fl |= (unsigned char)(cnt << 2);
                    const float2 cv = cons[pix];
                    if (cv.x >= 0.0f && cv.y >= 0.0f) {            // image_warping.t:27 (Mask==0 already holds)
                        fl |= 2;
                        jx += wf2 * (oxi - cv.x); jy += wf2 * (oyi - cv.y); dgo += wf2;
                    }
                    rx = -jx; ry = -jy; ra = -ja;                   // gauss_newton.t:690
                    mx = guarded_invert(dgo); my = mx; ma = guarded_invert(dga);   // :696, UsePreconditioner(true)
                    dgo_raw = dgo; dga_raw = dga;
                }
                if (diag_out) {                                     // raw diag(J^T J): LM's computeCtC input (thallo.t:3929-3933)
                    reinterpret_cast<float2*>(diag_out)[pix] = make_float2(dgo_raw, dgo_raw); diag_out[2 * N + pix] = dga_raw;
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
    if (irregular && __any(bad) && (threadIdx.x & 63) == 0) atomicAdd(irregular, 1);     // int atomic: exact, order-free
    block_store_partial(acc, aN_out, red);
}

// ------------------------------------------------------------------------------------------ computeCost
// gauss_newton.t:1067-1079, thallo.t:3939-3949: once per solve (+ once per LM step); neighbours via L1/L2.
constexpr int CW = 64, CH = 4;
__global__ __launch_bounds__(BLOCK) void k_cost(int W, int H, int row0, int row1, int ctx, int ntiles,
                                                const float2* __restrict__ off, const float* __restrict__ ang,
                                                const float2* __restrict__ ur, const float2* __restrict__ cons,
                                                const float* __restrict__ mask, float wf, float wr, float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (TileSweep t(ntiles); t.valid(); t.next()) {
        const int x = (t.cur % ctx) * CW + (threadIdx.x % CW), y = row0 + (t.cur / ctx) * CH + (threadIdx.x / CW);
        if (x < W && y < row1) {
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

// ------------------------------------------------------------------------------------------ one kernel per PCG iteration
// The whole PCG iteration k in ONE pass (replaces PCGStep1 + PCGStep2 + PCGStep3, gauss_newton.t:734-752,801-843,889-899):
//     r_k   = r_{k-1} - alpha_{k-1} A p_{k-1}        (owned + halo; fma, correctly rounded)          [first: r_0 as is]
//     z_k   = M^-1 r_k ;  p_k = z_k + beta_{k-1} p_{k-1} ;  delta += ... (deferred as in PCGStep1)
//     A p_k = J^T J p_k ;  alphaD_k = sum p_k . A p_k
// and, so that beta_k is known without a second pass over r,
//     N_k = sum r_k . M^-1 r_k ,  S1_k = sum r_k . M^-1 A p_k ,  S2_k = sum (A p_k) . M^-1 (A p_k)      (double precision)
//     betaN_k = r_{k+1} . M^-1 r_{k+1} = N_k - 2 alpha_k S1_k + alpha_k^2 S2_k        (k_iter_finish, in double)
// which is the same number the reference adds up from the rounded r_{k+1} to ~1e-7 relative (the products are exact in
// double, r_{k+1} itself still follows the true recurrence).  One reduction point per iteration instead of two.
// r, Ap and p ping-pong (a neighbouring tile reads the old value of a pixel this tile overwrites).
// Bytes per pixel (pixel grid): read r 12, Ap 12, p 12, cs 8, flags 1; write r 12, p 12, Ap 12 = 81 (+ 18 deferred delta on
// average) against 75 + 37 for the two-kernel form.
#ifdef THALLO_MARCH_SWEEP
// tools/march_probe.py MB_MODE=stamps: where a launch of the tile kernel spends its time (100 MHz wall clock, thread 0 of every workgroup)
__device__ unsigned long long* g_stamps = nullptr;
#define IW_STAMP(k) do { if (threadIdx.x == 0 && g_stamps) g_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define IW_STAMP(k) do { } while (0)
#endif
struct TileI : Tile { float rx[LN], ry[LN], ra[LN]; };      // + r_k of the owned pixels (read back in the gather phase)
struct OwnedI { float2 rv, av, pv, csv, uv, dv, ppv, mv; float ra, aa, pav, da, ppa, ma; unsigned char ff; };
struct HaloI { float2 rv, av, pv, csv, uv, mv; float ra, aa, pav, ma; unsigned char ff; };

template <bool GRID, int NT, bool DIST = false>
__device__ __forceinline__ void iter_body(TileI& T, float* red, double* redd, const Geo& g, const float2* __restrict__ cs, const float2* __restrict__ ur,
                                          const unsigned char* __restrict__ flags, const float* __restrict__ pre, float wf2, float wr2,
                                          const float* __restrict__ r_in, float* __restrict__ r_out, const float* __restrict__ A_in, float* __restrict__ A_out,
                                          const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta, int mode,
                                          thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                          float* __restrict__ aD_out, double* __restrict__ s12_out, int ntm, const thallo_dist_t* dd = nullptr,
                                          unsigned* __restrict__ fin_tickets = nullptr, float* __restrict__ aD_word = nullptr, float* __restrict__ bN_word = nullptr,
                                          int xslot = 0, PrevSums prev = PrevSums{ nullptr, nullptr, 0, nullptr, nullptr })
{
    constexpr int PER = TH / (NT / TW);
    const int first = mode & 1;
    const int dmode = first ? 1 : (mode >> 1) & 3;                   // as in PCGStep1: 0 single, 1 none, 2 double delta update
    const bool nt_delta = ntm & 1, nt_ra = ntm & 2, nt_out = ntm & 4, nt_pin = ntm & 8, nt_pout = ntm & 16, nt_const = ntm & 32;
    const long N = (long)g.W * g.H;
    const float2* __restrict__ ro = reinterpret_cast<const float2*>(r_in);   const float* __restrict__ ra = r_in + 2 * N;
    const float2* __restrict__ ao = reinterpret_cast<const float2*>(A_in);   const float* __restrict__ aa = A_in + 2 * N;
    const float2* __restrict__ po = reinterpret_cast<const float2*>(p_in);   const float* __restrict__ pa = p_in + 2 * N;
    const float2* __restrict__ mo_ = reinterpret_cast<const float2*>(pre);   const float* __restrict__ ma_ = pre + 2 * N;
    float2* __restrict__ Ro = reinterpret_cast<float2*>(r_out);  float* __restrict__ Ra = r_out + 2 * N;
    float2* __restrict__ Ao = reinterpret_cast<float2*>(A_out);  float* __restrict__ Aa = A_out + 2 * N;
    float2* __restrict__ qo = reinterpret_cast<float2*>(p_out);  float* __restrict__ qa = p_out + 2 * N;
    float2* __restrict__ dlo = reinterpret_cast<float2*>(delta); float* __restrict__ dla = delta + 2 * N;

    const int tx = threadIdx.x % TW, ty = (threadIdx.x / TW) * PER;
    int hlx, hly;
    {
        const int h = threadIdx.x;
        if (h < LW) { hlx = h; hly = 0; }
        else if (h < 2 * LW) { hlx = h - LW; hly = LH - 1; }
        else if (h < 2 * LW + TH) { hlx = 0; hly = h - 2 * LW + 1; }
        else { hlx = LW - 1; hly = h - 2 * LW - TH + 1; }
    }
    const bool has_halo = threadIdx.x < HALO;
    OwnedI ow[PER]; HaloI hl;
    bool ow_ld[PER], ow_in[PER]; bool hl_in = false;

    auto issue_loads = [&](int tile) {
        const int x0 = (tile % g.tx) * TW, y0 = g.row0 + (tile / g.tx) * TH;
        const int gx = x0 + tx;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int gy = y0 + ty + k;
            ow_ld[k] = gx < g.W && gy < g.H;
            ow_in[k] = ow_ld[k] && gy < g.row1;
            if (ow_ld[k]) {
                const long pix = (long)gy * g.W + gx;
                ow[k].rv = ldf2(ro + pix, nt_ra); ow[k].ra = ldf(ra + pix, nt_ra);
                if (!first) { ow[k].av = ldf2(ao + pix, nt_ra); ow[k].aa = ldf(aa + pix, nt_ra); }
                ow[k].pv = ldf2(po + pix, nt_pin); ow[k].pav = ldf(pa + pix, nt_pin);
                ow[k].csv = ldf2(cs + pix, nt_const); ow[k].ff = ldb(flags + pix, nt_const);
                if (!GRID) { ow[k].uv = ur[pix]; ow[k].mv = mo_[pix]; ow[k].ma = ma_[pix]; }
                if (dmode != 1 && ow_in[k]) {
                    ow[k].dv = ldf2(dlo + pix, nt_delta); ow[k].da = ldf(dla + pix, nt_delta);
                    if (dmode == 2) { ow[k].ppv = qo[pix]; ow[k].ppa = qa[pix]; }
                }
            }
        }
        if (has_halo) {
            const int hx = x0 + hlx - 1, hy = y0 + hly - 1;
            hl_in = hx >= 0 && hx < g.W && hy >= 0 && hy < g.H;
            if (hl_in) {
                const long pix = (long)hy * g.W + hx;
                hl.rv = ro[pix]; hl.ra = ra[pix];
                if (!first) { hl.av = ao[pix]; hl.aa = aa[pix]; }
                hl.pv = po[pix]; hl.pav = pa[pix]; hl.csv = cs[pix]; hl.ff = flags[pix];
                if (!GRID) { hl.uv = ur[pix]; hl.mv = mo_[pix]; hl.ma = ma_[pix]; }
            }
        }
    };

    TileSweep t(g.ntiles);
    IW_STAMP(0);
#ifdef THALLO_MARCH_SWEEP
    const long long clk0 = clock64();
#endif
    if (t.valid()) issue_loads(t.cur);
    IW_STAMP(1);
    float alpha = 0.0f, beta = 0.0f, alpha2 = 0.0f;
    if (!first) {
        // ONE wave adds the previous iteration's partials up (the same additions in the same order as before, so the same bits) and hands
        // alpha / beta to the others through LDS: the other seven waves of the workgroup would only repeat its 1,000+ loads and ~250 VALU
        // instructions, and on a one-tile launch (512^2: one workgroup per CU) instruction issue is what the workgroup waits for.  The barrier
        // costs nothing: the tile's loads, issued above, are still on their way.
        if (threadIdx.x < THALLO_WAVE) {
            iteration_scalars(aNp, aDp, bNp, prev, alpha, beta, blockIdx.x == 0 && threadIdx.x == 0);
            if (dmode == 2) alpha2 = safe_div<false>(sum_partials(aNpp.partials, aNpp.count), sum_partials(aDpp.partials, aDpp.count));
            if (threadIdx.x == 0) { red[12] = alpha; red[13] = beta; red[14] = alpha2; }
        }
        lds_barrier();
        alpha = red[12]; beta = red[13]; alpha2 = red[14];
    }
    IW_STAMP(2);

#ifdef THALLO_AD_DOUBLE
    double acc = 0.0;
#else
    float acc = 0.0f;
#endif
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    while (t.valid()) {
        const int x0 = (t.cur % g.tx) * TW, y0 = g.row0 + (t.cur / g.tx) * TH;
        const int gx = x0 + tx;
        float cmo[PER], cmy[PER], cma[PER];           // non-GRID only: M^-1 of the owned pixels (GRID recomputes it from the flags byte)
        bool cin[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = (ty + k + 1) * LW + tx + 1;
            float npx = 0.f, npy = 0.f, npa = 0.f, c1 = 1.f, s1_ = 0.f, u1 = 0.f, u2 = 0.f, zx = 0.f, zy = 0.f, zq = 0.f, mo = 0.f, my = 0.f, ma = 0.f; unsigned char ff = 0;
            float rnx = 0.f, rny = 0.f, rnq = 0.f;
            cin[k] = ow_in[k];
            if (ow_ld[k]) {
                const long pix = (long)(y0 + ty + k) * g.W + gx;
                float rx = ow[k].rv.x, ry = ow[k].rv.y, rq = ow[k].ra;
                if (!first) { rx = __builtin_fmaf(-alpha, ow[k].av.x, rx); ry = __builtin_fmaf(-alpha, ow[k].av.y, ry); rq = __builtin_fmaf(-alpha, ow[k].aa, rq); }
                rnx = rx; rny = ry; rnq = rq;
                stf2(Ro + pix, make_float2(rx, ry), nt_out); stf(Ra + pix, rq, nt_out);     // owned, or a slab's ghost row kept current
                if (GRID) { pre_from_flags(ow[k].ff, wf2, wr2, mo, ma); my = mo; }
                else { mo = ow[k].mv.x; my = ow[k].mv.y; ma = ow[k].ma; }
                zx = mo * rx; zy = my * ry; zq = ma * rq;
                npx = zx + beta * ow[k].pv.x; npy = zy + beta * ow[k].pv.y; npa = zq + beta * ow[k].pav;
                stf2(qo + pix, make_float2(npx, npy), nt_pout); stf(qa + pix, npa, nt_pout);
                if (dmode != 1 && ow_in[k]) {
                    float dx = ow[k].dv.x, dy = ow[k].dv.y, dq = ow[k].da;
                    if (dmode == 2) { dx = __builtin_fmaf(alpha2, ow[k].ppv.x, dx); dy = __builtin_fmaf(alpha2, ow[k].ppv.y, dy); dq = __builtin_fmaf(alpha2, ow[k].ppa, dq); }
                    stf2(dlo + pix, make_float2(__builtin_fmaf(alpha, ow[k].pv.x, dx), __builtin_fmaf(alpha, ow[k].pv.y, dy)), nt_delta);
                    stf(dla + pix, __builtin_fmaf(alpha, ow[k].pav, dq), nt_delta);
                }
                c1 = ow[k].csv.x; s1_ = ow[k].csv.y; ff = ow[k].ff;
                if (!GRID) { u1 = ow[k].uv.x; u2 = ow[k].uv.y; }
            }
            T.px[i] = npx; T.py[i] = npy; T.pa[i] = npa; T.c[i] = c1; T.s[i] = s1_; T.f[i] = ff;
            T.rx[i] = rnx; T.ry[i] = rny; T.ra[i] = rnq;
            if (!GRID) { T.ux[i] = u1; T.uy[i] = u2; cmo[k] = mo; cmy[k] = my; cma[k] = ma; }
        }
        if (has_halo) {
            const int i = hly * LW + hlx;
            float npx = 0.f, npy = 0.f, npa = 0.f, c1 = 1.f, s1_ = 0.f, u1 = 0.f, u2 = 0.f; unsigned char ff = 0;
            if (hl_in) {
                float rx = hl.rv.x, ry = hl.rv.y, rq = hl.ra;
                if (!first) { rx = __builtin_fmaf(-alpha, hl.av.x, rx); ry = __builtin_fmaf(-alpha, hl.av.y, ry); rq = __builtin_fmaf(-alpha, hl.aa, rq); }
                float mo, my, ma;
                if (GRID) { pre_from_flags(hl.ff, wf2, wr2, mo, ma); my = mo; } else { mo = hl.mv.x; my = hl.mv.y; ma = hl.ma; }
                npx = mo * rx + beta * hl.pv.x; npy = my * ry + beta * hl.pv.y; npa = ma * rq + beta * hl.pav;
                const int hy = y0 + hly - 1;
                if ((hy < g.row0 || hy >= g.row1) && hlx >= 1 && hlx <= TW) {      // ghost row of a slab: keep its r and p current
                    const long pix = (long)hy * g.W + (x0 + hlx - 1);
                    Ro[pix] = make_float2(rx, ry); Ra[pix] = rq;
                    qo[pix] = make_float2(npx, npy); qa[pix] = npa;
                }
                c1 = hl.csv.x; s1_ = hl.csv.y; ff = hl.ff;
                if (!GRID) { u1 = hl.uv.x; u2 = hl.uv.y; }
            }
            T.px[i] = npx; T.py[i] = npy; T.pa[i] = npa; T.c[i] = c1; T.s[i] = s1_; T.f[i] = ff;
            if (!GRID) { T.ux[i] = u1; T.uy[i] = u2; }
        }
        IW_STAMP(3);
        lds_barrier();
        IW_STAMP(4);
        const int cur_y0 = y0;
        t.next();
        if (t.valid()) issue_loads(t.cur);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (cin[k]) {
                const int i = (ty + k + 1) * LW + tx + 1;
                const long pix = (long)(cur_y0 + ty + k) * g.W + gx;
                float ax = 0.f, ay = 0.f, av = 0.f;
                const float pxi = T.px[i], pyi = T.py[i], pai = T.pa[i];
                const unsigned char cfk = T.f[i];
                if (cfk & 1) {
                    const float ci = T.c[i], si = T.s[i];
                    float uxi = 0.f, uyi = 0.f;
                    if (!GRID) { uxi = T.ux[i]; uyi = T.uy[i]; }
                    const int nb[4] = { i + 1, i - 1, i + LW, i - LW };
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int j = nb[d];
                        if (T.f[j] & 1) {
                            float dux, duy;
                            if (GRID) { dux = d == 0 ? -1.0f : d == 1 ? 1.0f : 0.0f; duy = d == 2 ? -1.0f : d == 3 ? 1.0f : 0.0f; }
                            else { dux = uxi - T.ux[j]; duy = uyi - T.uy[j]; }
                            const float gix = -si * dux - ci * duy, giy = ci * dux - si * duy;
                            const float cj = T.c[j], sj = T.s[j], paj = T.pa[j];
                            const float gjx = sj * dux + cj * duy, gjy = -cj * dux + sj * duy;
                            const float dpx = pxi - T.px[j], dpy = pyi - T.py[j];
                            const float ex = dpx - gix * pai, ey = dpy - giy * pai;
                            ax += dpx + ex + gjx * paj;
                            ay += dpy + ey + gjy * paj;
                            av -= gix * ex + giy * ey;
                        }
                    }
                    ax *= wr2; ay *= wr2; av *= wr2;
                    if (cfk & 2) { ax += wf2 * pxi; ay += wf2 * pyi; }
                }
                float mo, my, ma;
                if (GRID) { pre_from_flags(cfk, wf2, wr2, mo, ma); my = mo; } else { mo = cmo[k]; my = cmy[k]; ma = cma[k]; }
                stf2(Ao + pix, make_float2(ax, ay), nt_out); stf(Aa + pix, av, nt_out);
                if (DIST) {     // multi-GPU row slabs: the first / last owned row of A p_k also goes into the neighbour's ghost row of its
                                // Ap_out buffer (peer-to-peer, write-through; complete when this kernel ends, i.e. before the exchange
                                // kernel behind it sends this rank's granules)
                    const int gyk = cur_y0 + ty + k;
                    if (gyk == g.row0 && dd->peer_r[0]) {
                        float* d2 = dd->peer_r[0] + dd->peer_off_o[0] + 2 * gx; st_sys(d2, ax); st_sys(d2 + 1, ay);
                        st_sys(dd->peer_r[0] + dd->peer_off_a[0] + gx, av);
                    }
                    if (gyk == g.row1 - 1 && dd->peer_r[1]) {
                        float* d2 = dd->peer_r[1] + dd->peer_off_o[1] + 2 * gx; st_sys(d2, ax); st_sys(d2 + 1, ay);
                        st_sys(dd->peer_r[1] + dd->peer_off_a[1] + gx, av);
                    }
                }
#ifdef THALLO_AD_DOUBLE
                acc += (double)pxi * (double)ax + (double)pyi * (double)ay + (double)pai * (double)av;
#else
                acc += pxi * ax + pyi * ay + pai * av;
#endif
                // exact products of the float data, accumulated in double: N = sum r.M^-1.r, S1 = sum r.M^-1.Ap, S2 = sum Ap.M^-1.Ap
                const double dmo = mo, dmy = my, dma = ma, drx = T.rx[i], dry = T.ry[i], dra = T.ra[i], dax = ax, day = ay, daa = av;
                s0 += dmo * (drx * drx) + dmy * (dry * dry) + dma * (dra * dra);
                s1 += dmo * (drx * dax) + dmy * (dry * day) + dma * (dra * daa);
                s2 += dmo * (dax * dax) + dmy * (day * day) + dma * (daa * daa);
            }
        }
        IW_STAMP(5);
        lds_barrier();
    }
    iter_tail<NT, DIST>(acc, s0, s1, s2, red, redd, aD_out, s12_out, bNp, dd, fin_tickets, aD_word, bN_word, xslot);
    IW_STAMP(6);
#ifdef THALLO_MARCH_SWEEP
    if (threadIdx.x == 0 && g_stamps) g_stamps[blockIdx.x * 8 + 7] = (unsigned long long)(clock64() - clk0);
#endif
}

template <int MINW, int NT, bool DIST>
__global__ __launch_bounds__(NT, MINW) void k_iter(Geo g, const float2* __restrict__ cs, const float2* __restrict__ ur, const unsigned char* __restrict__ flags,
                                                       const float* __restrict__ pre, float wf2, float wr2,
                                                       const float* __restrict__ r_in, float* __restrict__ r_out, const float* __restrict__ A_in, float* __restrict__ A_out,
                                                       const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta, int mode,
                                                       thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                                       float* __restrict__ aD_out, double* __restrict__ s12_out, int ntm, const int* __restrict__ irregular,
                                                       thallo_dist_t dd, unsigned* __restrict__ fin_tickets, float* __restrict__ aD_word, float* __restrict__ bN_word, int xslot,
                                                       PrevSums prev)
{
    __shared__ TileI T;
    __shared__ float red[16];
    __shared__ double redd[48];
    const bool grid = irregular != nullptr && __builtin_amdgcn_readfirstlane(irregular[0]) == 0;
    if (grid) iter_body<true, NT, DIST>(T, red, redd, g, cs, ur, flags, pre, wf2, wr2, r_in, r_out, A_in, A_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, ntm, &dd, fin_tickets, aD_word, bN_word, xslot, prev);
    else      iter_body<false, NT, DIST>(T, red, redd, g, cs, ur, flags, pre, wf2, wr2, r_in, r_out, A_in, A_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, ntm, &dd, fin_tickets, aD_word, bN_word, xslot, prev);
}

// one wave: alphaD_k (float partials, the usual order), S1_k, S2_k (double partials, same lane-strided order), then
// betaN_k = alphaN_k - 2 alpha_k S1_k + alpha_k^2 S2_k  with alpha_k = alphaN_k / alphaD_k exactly as every consumer forms it
__global__ __launch_bounds__(64) void k_iter_finish(const float* __restrict__ aD_part, const double* __restrict__ s12, int nb, thallo_sum_t aN,
                                                    float* __restrict__ aD_word, float* __restrict__ bN_word)
{
    const int lane = threadIdx.x;
    const float ad = sum_partials(aD_part, nb);
    double n = 0.0, a = 0.0, b = 0.0;
    for (int i = lane; i < nb; i += THALLO_WAVE) { n += s12[3 * i]; a += s12[3 * i + 1]; b += s12[3 * i + 2]; }
    n = wave_sum_all_d(n); a = wave_sum_all_d(a); b = wave_sum_all_d(b);
    const float an = sum_partials(aN.partials, aN.count);          // what every consumer divides by (the float alphaN_k)
    const float alpha = safe_div<false>(an, ad);
    // the expansion starts from the double N = r_k.M^-1.r_k of the very same float data as S1, S2: the cancellation (betaN can be
    // 1e-3 of alphaN) then costs nothing, which it would with the rounded float alphaN_k
    double bn = n - 2.0 * (double)alpha * a + (double)alpha * (double)alpha * b;
    if (!(bn > 0.0)) bn = 0.0;                       // r . M^-1 r is a sum of squares; guards the last bits at convergence
    if (lane == 0) { aD_word[0] = ad; bN_word[0] = (float)bn; }
}

}  // namespace

extern "C" {

#ifdef THALLO_MARCH_SWEEP
int thallo_hip_debug_stamps(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif
void thallo_hip_debug_set(int what, int value) { if (what == 0) g_iw_debug = value; if (what == 3) g_nt_mask = value; if (what == 4) g_no_grid = value; if (what == 5) g_step1_per_cu = value; if (what == 6) g_step1_threads = value; if (what == 7) g_iter_nt = value; if (what == 8) g_iter_per_cu = value < 1 ? 1 : value > 2 ? 2 : value; }

int thallo_hip_iw_cost(int W, int H, int row0, int row1, const float* offset, const float* angle, const float* urshape,
                       const float* constraints, const float* mask, float w_fit, float w_reg,
                       float* cost_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1)) return -(int)hipErrorInvalidValue;
    const int ctx = (W + CW - 1) / CW, cty = (row1 - row0 + CH - 1) / CH, nt = ctx * cty;
    int grid = thallo_hip_device_cu_count() * 4; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS;
    grid -= grid % 8; if (nt < grid) grid = nt;
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, W, H, row0, row1, ctx, nt,
                       (const float2*)offset, angle, (const float2*)urshape, (const float2*)constraints, mask, w_fit, w_reg, cost_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_init(int W, int H, int row0, int row1, const float* offset, const float* angle, const float* urshape,
                           const float* constraints, const float* mask, float w_fit, float w_reg,
                           float* r, float* pre, float* z, float* p_prev, float* delta,
                           float* cs, unsigned char* flags, float* diag_out, int* irregular_out, float* aN_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1)) return -(int)hipErrorInvalidValue;
    const Geo g = make_geo(W, H, row0, row1); const int grid = grid_for(g, 4);
    if (irregular_out && hipMemsetAsync(irregular_out, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_init, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                       (const float2*)offset, angle, (const float2*)urshape, (const float2*)constraints, mask, w_fit, w_reg,
                       r, pre, z, p_prev, delta, (float2*)cs, flags, diag_out, irregular_out, aN_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_step1(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg,
                            const float* z, const float* p_in, float* p_out, float* delta, float* Ap,
                            int mode, thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                            const int* irregular, const float* r, float* aD_out, thallo_stream_t stream)
{
    if (!r || ((2L * W * H) & 3)) irregular = nullptr;      // the z-free path needs r and 16-byte planes
    if (!rows_ok(H, row0, row1)) return -(int)hipErrorInvalidValue;
    const Geo g = make_geo(W, H, row0, row1);
    int grid;
    if (g_step1_threads == 512) {       // 2 workgroups of 8 waves per CU (128 VGPRs): 16 waves/CU and 4096 tiles / 512 = 8 tiles each at 2048^2
        grid = grid_for(g, g_step1_per_cu > 2 ? 2 : g_step1_per_cu);
        hipLaunchKernelGGL((k_step1<true, 4, 512>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g,
                           (const float2*)cs, (const float2*)urshape, flags, w_fit * w_fit, w_reg * w_reg,
                           z, p_in, p_out, delta, Ap, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, g_iw_debug | (g_nt_mask << 8), g_no_grid ? nullptr : irregular, r);
    } else {                            // 3 workgroups of 4 waves per CU (165 VGPRs)
        grid = grid_for(g, g_step1_per_cu);
        hipLaunchKernelGGL((k_step1<true, 3, 256>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                           (const float2*)cs, (const float2*)urshape, flags, w_fit * w_fit, w_reg * w_reg,
                           z, p_in, p_out, delta, Ap, mode, aNp, aDp, bNp, aNpp, aDpp, aD_out, g_iw_debug | (g_nt_mask << 8), g_no_grid ? nullptr : irregular, r);
    }
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_step2(int W, int H, int row0, int row1, const unsigned char* flags, float w_fit, float w_reg,
                            float* r, const float* Ap, const float* pre, float* z,
                            thallo_sum_t aN, thallo_sum_t aD, const int* irregular, float* bN_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1)) return -(int)hipErrorInvalidValue;
    const long N = (long)W * H, rows = row1 - row0;
    const long off0 = 2L * W * row0, len0 = 2L * W * rows, off1 = 2 * N + (long)W * row0, len1 = (long)W * rows;
    if (g_no_grid || ((off0 | len0 | off1 | len1 | (2 * N)) & 3)) {     // not 16-byte granular: the generic kernels (read pre, write z)
        if (row0 == 0 && row1 == H) return thallo_hip_pcg_step2(r, Ap, pre, z, 3 * N, aN, aD, bN_out, stream);
        return thallo_hip_pcg_step2_ranges(r, Ap, pre, z, off0, len0, off1, len1, aN, aD, bN_out, stream);
    }
    long want = ((len0 + len1) / 4 + BLOCK - 1) / BLOCK;
    int grid = thallo_hip_device_cu_count() * 4; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS; grid -= grid % 8;
    if (want < grid) grid = (int)(want < 1 ? 1 : want);
    hipLaunchKernelGGL(k_step2_iw<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)r, (const float4*)Ap, (const float4*)pre, (float4*)z,
                       flags, N, w_fit * w_fit, w_reg * w_reg, off0 / 4, len0 / 4, off1 / 4, len1 / 4, aN, aD, irregular, bN_out, thallo_dist_t{}, 0L, 0L);
    int e = check_launch(); return e ? e : grid;
}

static bool dist_args_ok(const thallo_dist_t& d)
{
    return d.world >= 1 && d.world <= THALLO_DIST_MAX_WORLD && d.rank >= 0 && d.rank < d.world && d.mail && d.ctl && d.peer_mail[d.rank] == d.mail;
}

int thallo_hip_iw_pcg_step2_dist(int W, int H, int row0, int row1, const unsigned char* flags, float w_fit, float w_reg,
                                 float* r, const float* Ap, thallo_sum_t aN, thallo_sum_t aD, thallo_dist_t d,
                                 float* bN_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1) || !dist_args_ok(d) || (W & 3)) return -(int)hipErrorInvalidValue;
    const long N = (long)W * H, rows = row1 - row0;
    const long off0 = 2L * W * row0, len0 = 2L * W * rows, off1 = 2 * N + (long)W * row0, len1 = (long)W * rows;
    if ((off0 | len0 | off1 | len1 | (2 * N)) & 3) return -(int)hipErrorInvalidValue;
    for (int k = 0; k < 2; ++k) if (d.peer_r[k] && ((d.peer_off_o[k] | d.peer_off_a[k]) & 3)) return -(int)hipErrorInvalidValue;
    long want = ((len0 + len1) / 4 + BLOCK - 1) / BLOCK;
    int grid = thallo_hip_device_cu_count() * 4; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS; grid -= grid % 8;
    if (want < grid) grid = (int)(want < 1 ? 1 : want);
    hipLaunchKernelGGL(k_step2_iw<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, (float4*)r, (const float4*)Ap, (const float4*)nullptr, (float4*)nullptr,
                       flags, N, w_fit * w_fit, w_reg * w_reg, off0 / 4, len0 / 4, off1 / 4, len1 / 4, aN, aD, (const int*)nullptr, bN_out, d,
                       2L * W / 4, (long)W / 4);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_iter(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                           float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                           const float* p_in, float* p_out, float* delta, int mode,
                           thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                           const int* irregular, float* aD_out, double* s12_out,
                           unsigned* fin_tickets, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1) || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (!(mode & 1) && !Ap_in) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    const Geo g = make_geo(W, H, row0, row1);
    const int grid = grid_for(g, g_iter_per_cu);
    // two register budgets of the same kernel: up to one workgroup per CU (every image the plugin sends here by default: < 0.4 Mpixel) the
    // 139 registers it wants, no scratch; beyond that the 128-register build -- two workgroups per CU, 28 B/lane of scratch -- is 30 % faster
    // (2048^2 with an irregular UrShape: 9.0 vs 11.8 ms per GN step)
    if (grid <= thallo_hip_device_cu_count())
        hipLaunchKernelGGL((k_iter<3, 512, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, thallo_dist_t{}, fin_tickets, aD_word, bN_word, 0, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
    else
        hipLaunchKernelGGL((k_iter<4, 512, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, thallo_dist_t{}, fin_tickets, aD_word, bN_word, 0, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_iter_deferred(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                                    float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                    const float* p_in, float* p_out, float* delta, int mode,
                                    thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, thallo_prev_t prev,
                                    const int* irregular, float* aD_out, double* s12_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1) || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (!(mode & 1) && (!Ap_in || prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word || !prev.betaN_word ||
                        prev.s12_partials == s12_out)) return -(int)hipErrorInvalidValue;
    const Geo g = make_geo(W, H, row0, row1);
    const int grid = grid_for(g, g_iter_per_cu);
    const thallo_sum_t none = { nullptr, 0 };
    const PrevSums ps = (mode & 1) ? PrevSums{ nullptr, nullptr, 0, nullptr, nullptr } : PrevSums{ prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word };
    // two register budgets of the same kernel: up to one workgroup per CU (every image the plugin sends here by default: < 0.4 Mpixel) the
    // 139 registers it wants, no scratch; beyond that the 128-register build -- two workgroups per CU, 28 B/lane of scratch -- is 30 % faster
    // (2048^2 with an irregular UrShape: 9.0 vs 11.8 ms per GN step)
    if (grid <= thallo_hip_device_cu_count())
        hipLaunchKernelGGL((k_iter<3, 512, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, thallo_dist_t{}, nullptr, nullptr, nullptr, 0, ps);
    else
        hipLaunchKernelGGL((k_iter<4, 512, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, thallo_dist_t{}, nullptr, nullptr, nullptr, 0, ps);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_iter_dist(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags, const float* pre,
                                float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                const float* p_in, float* p_out, float* delta, int mode,
                                thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                const int* irregular, thallo_dist_t d, float* aD_out, double* s12_out,
                                unsigned* fin_tickets, int slot0, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    if (fin_tickets && (slot0 < 0 || !d.mail || !d.ctl || 7 * d.world > 64 || bNp.count != 1)) return -(int)hipErrorInvalidValue;
    if (!rows_ok(H, row0, row1) || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if ((!(mode & 1) && !Ap_in) || d.world < 1 || d.world > THALLO_DIST_MAX_WORLD) return -(int)hipErrorInvalidValue;
    const Geo g = make_geo(W, H, row0, row1);
    const int grid = grid_for(g, g_iter_per_cu);
    // two register budgets of the same kernel: up to one workgroup per CU (every image the plugin sends here by default: < 0.4 Mpixel) the
    // 139 registers it wants, no scratch; beyond that the 128-register build -- two workgroups per CU, 28 B/lane of scratch -- is 30 % faster
    // (2048^2 with an irregular UrShape: 9.0 vs 11.8 ms per GN step)
    if (grid <= thallo_hip_device_cu_count())
        hipLaunchKernelGGL((k_iter<3, 512, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, d, fin_tickets, aD_word, bN_word, slot0, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
    else
        hipLaunchKernelGGL((k_iter<4, 512, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, g, (const float2*)cs, (const float2*)urshape, flags, pre,
                       w_fit * w_fit, w_reg * w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp,
                       aD_out, s12_out, g_iter_nt, g_no_grid ? nullptr : irregular, d, fin_tickets, aD_word, bN_word, slot0, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_iw_pcg_iter_finish(const float* aD_partials, const double* s12_partials, int count, thallo_sum_t alphaN,
                                  float* alphaD_word, float* betaN_word, thallo_stream_t stream)
{
    if (!aD_partials || !s12_partials || count < 1 || count > THALLO_MAX_PARTIALS || !alphaD_word || !betaN_word) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_iter_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, aD_partials, s12_partials, count, alphaN, alphaD_word, betaN_word);
    return check_launch();
}

int thallo_hip_iw_apply_jtj(int W, int H, int row0, int row1, const float* cs, const float* urshape, const unsigned char* flags,
                            float w_fit, float w_reg, const float* p, float* Ap, const int* irregular, float* aD_out, thallo_stream_t stream)
{
    if (!rows_ok(H, row0, row1)) return -(int)hipErrorInvalidValue;
    const Geo g = make_geo(W, H, row0, row1); const int grid = grid_for(g, 4);
    thallo_sum_t none; none.partials = nullptr; none.count = 0;
    /* z / p_out / delta are unused when !FUSED: pass valid dummies */
    hipLaunchKernelGGL((k_step1<false, 4, 256>), dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, g,
                       (const float2*)cs, (const float2*)urshape, flags, w_fit * w_fit, w_reg * w_reg,
                       p, p, Ap, Ap, Ap, 1, none, none, none, none, none, aD_out, g_iw_debug, g_no_grid ? nullptr : irregular, p);
    int e = check_launch(); return e ? e : grid;
}

}  // extern "C"
