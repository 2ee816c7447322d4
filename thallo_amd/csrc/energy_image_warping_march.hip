// energy_image_warping_march.hip -- the one-kernel PCG iteration of image_warping as a barrier-free "marching" stencil, A p kept as a plane.
//
// Same mathematics, arguments and outputs as k_iter (energy_image_warping.hip; replaces PCGStep1 + PCGStep2 + PCGStep3 of
// gauss_newton.t:734-752,801-843,889-899 in ONE pass), for the unit-pixel-grid UrShape (the GRID path).  What differs is the
// shape of the computation on the chip:
//
//   * a WAVE (not a workgroup) owns a column strip of 128 pixels (lane l: the two x-adjacent pixels x0+2l, x0+2l+1) and marches down
//     R rows of it.  Lanes 1..62 produce output (124 pixels), lanes 0 and 63 carry the x halo (strips overlap by 4 pixels);
//   * x neighbours come from the neighbouring lane through DPP wave shifts (v_mov_b32_dpp wave_shr / wave_shl), y neighbours from the
//     lane's own registers: it keeps a window of three published rows (p_k, cos, sin, flags of rows y-1, y, y+1).  No LDS tile, no
//     barrier in the loop: every wave is its own software pipeline and the memory system sees ~1,000 desynchronised streams;
//   * the raw rows (r, Ap, p, cs, flags [, delta, p_{k-2}]) are prefetched three rows ahead into registers; each plane moves in 16-byte
//     (Offset part: 2 pixels x float2) or 8-byte (Angle part) accesses;
//   * M^-1 = guardedInvert(diag) is a function of the 5-bit flags value only (iw_device.hpp): a 32-entry table in LDS built once per
//     workgroup replaces two divisions + two square roots per pixel;
//   * the y halo (rows ya-1 and yb of a segment) is published redundantly by the wave.
//
// Bytes per pixel as for k_iter: read r 12, Ap 12, p 12, cs 8, flags 1; write r 12, p 12, Ap 12 = 81 (+ 18 deferred delta on average).
//
// Round 4: on one GPU the iterations k >= 1 of a whole image run energy_image_warping_march_rc.hip instead (A p_{k-1} recomputed from the p_{k-1} rows:
// 57 + 18 B/pixel, same bits).  This file stays as the FIRST iteration of every GN step, the multi-GPU slab form (the A p ghost rows ARE the exchange),
// the in-kernel-finish form and the A/B (THALLO_MARCH=3).  The round-2/3 research variants of this kernel (DBG modes, workgroup maps, phase stamps, the
// streaming reference) live in probe/march_probe_kernels.hip, which only `make VARIANT=sweep` builds -- in place of this file.
#include "iw_march.hpp"

namespace thallo {
int g_march_rows = 0;
int g_march_cap = 0;
}

using namespace thallo;

namespace {

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

template <bool FIRST>
struct Raw {                                // one row of one lane (2 pixels) as loaded
    float4 ro, po, cs, ao;                  // r / p / (c0,s0,c1,s1) / Ap: Offset part (x0,y0,x1,y1)
    float2 ra, pa, aa;                      // Angle part (a0, a1)
    unsigned f;                             // the dword holding the flags bytes of the 2 pixels
};
struct RawD { float4 dlo, ppo; float2 dla, ppa; };     // delta and p_{k-2} of the row (deferred delta update)
template <bool FIRST>
__device__ __forceinline__ void take(Raw<FIRST>& d, const Raw<FIRST>& s)      // 23 moves per row (iw_march.hpp: take1)
{
    take4(d.ro, s.ro); take2(d.ra, s.ra); take4(d.po, s.po); take2(d.pa, s.pa); take4(d.cs, s.cs); take1(d.f, s.f);
    if (!FIRST) { take4(d.ao, s.ao); take2(d.aa, s.aa); }
}
template <int DMODE>
__device__ __forceinline__ void take(RawD& d, const RawD& s)
{
    take4(d.dlo, s.dlo); take2(d.dla, s.dla);
    if (DMODE == 2) { take4(d.ppo, s.ppo); take2(d.ppa, s.ppa); }
}

struct Row { float px[2], py[2], pa[2], c[2], s[2], rx[2], ry[2], ra[2], mo[2], ma[2]; unsigned f; };     // a published row: p_k, cos, sin, r_k, M^-1, flags

template <bool FIRST, int DMODE, int NTM, bool DIST, int OCC>
__global__ __launch_bounds__(MARCH_NT, OCC) void k_iter_march(MarchGeo g, const float* __restrict__ cs, const unsigned char* __restrict__ flags, float wf2, float wr2,
                                                         const float* __restrict__ r_in, float* __restrict__ r_out, const float* __restrict__ A_in, float* __restrict__ A_out,
                                                         const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta,
                                                         thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                                         float* __restrict__ aD_out, double* __restrict__ s12_out, const int* __restrict__ irregular,
                                                         thallo_dist_t dd, unsigned* __restrict__ fin_tickets, float* __restrict__ aD_word, float* __restrict__ bN_word, int xslot,
                                                         PrevSums prev)
{
    __shared__ float2 lut[32];
    __shared__ float red[16];
    __shared__ double redd[48];
    constexpr bool nt_delta = NTM & 1, nt_ra = NTM & 2, nt_out = NTM & 4, nt_pin = NTM & 8, nt_pout = NTM & 16, nt_const = NTM & 32;
    // this kernel is the unit-pixel-grid form only; the caller checked that at Init.  Should the word pcg_init wrote this GN step say
    // otherwise, poison the scalars (NaN cost downstream) instead of computing with the wrong Jacobian.
    if (irregular != nullptr && __builtin_amdgcn_readfirstlane(irregular[0]) != 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && aD_word) { aD_word[0] = __builtin_nanf(""); bN_word[0] = __builtin_nanf(""); }
        if (blockIdx.x == 0 && threadIdx.x == 0 && prev.count > 0) { prev.aD_word[0] = __builtin_nanf(""); prev.bN_word[0] = __builtin_nanf(""); }
        if (threadIdx.x == 0) { aD_out[blockIdx.x] = __builtin_nanf(""); }
        return;
    }
    if (threadIdx.x < 32) { float mo, ma; pre_from_flags((unsigned char)threadIdx.x, wf2, wr2, mo, ma); lut[threadIdx.x] = make_float2(mo, ma); }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long N = (long)g.W * g.H;
    const int W2 = g.W >> 1;                                  // pixel pairs per row
    int strip, ya, yb;
    march_place(g, wave, strip, ya, yb);
    const bool work = ya < yb;
    const int x0 = strip * MARCH_USE - 2 + 2 * lane;          // first of this lane's two pixels
    const bool xin = x0 >= 0 && x0 < g.W;                     // W even: both pixels exist or neither
    const bool xout = xin && lane >= 1 && lane <= 62;         // this lane's pixels are outputs of this wave

    const float4* __restrict__ ro4 = reinterpret_cast<const float4*>(r_in);  const float2* __restrict__ ra2 = reinterpret_cast<const float2*>(r_in + 2 * N);
    const float4* __restrict__ ao4 = reinterpret_cast<const float4*>(A_in);  const float2* __restrict__ aa2 = reinterpret_cast<const float2*>(A_in + 2 * N);
    const float4* __restrict__ po4 = reinterpret_cast<const float4*>(p_in);  const float2* __restrict__ pa2 = reinterpret_cast<const float2*>(p_in + 2 * N);
    const float4* __restrict__ cs4 = reinterpret_cast<const float4*>(cs);
    const unsigned* __restrict__ f4 = reinterpret_cast<const unsigned*>(flags);
    float4* __restrict__ Ro4 = reinterpret_cast<float4*>(r_out);  float2* __restrict__ Ra2 = reinterpret_cast<float2*>(r_out + 2 * N);
    float4* __restrict__ Ao4 = reinterpret_cast<float4*>(A_out);  float2* __restrict__ Aa2 = reinterpret_cast<float2*>(A_out + 2 * N);
    float4* __restrict__ qo4 = reinterpret_cast<float4*>(p_out);  float2* __restrict__ qa2 = reinterpret_cast<float2*>(p_out + 2 * N);
    float4* __restrict__ dl4 = reinterpret_cast<float4*>(delta);  float2* __restrict__ dl2 = reinterpret_cast<float2*>(delta + 2 * N);

    float alpha = 0.0f, beta = 0.0f, alpha2 = 0.0f;
    // alpha, beta are added up inside the row loop, behind the first three rows' loads (see there) -- except in the "apply two delta updates" variant,
    // which sits at the 256-register limit and would spill around that block: it adds them up here, in front of the loop
    constexpr bool SCALARS_IN_LOOP = DMODE != 2;
    // the words of iteration k-1 are left behind by the one wave that owns the first segment of strip 0 (small grids leave whole workgroups --
    // workgroup 0 included -- without rows, and a wave without rows never adds the scalars up)
    const bool scal_writer = work && strip == 0 && ya == g.row0 && lane == 0;
    if (!FIRST && !SCALARS_IN_LOOP) {
        iteration_scalars<1>(aNp, aDp, bNp, prev, alpha, beta, scal_writer);
        alpha2 = safe_div<false>(sum_partials(aNpp.partials, aNpp.count), sum_partials(aDpp.partials, aDpp.count));
    }

    typedef Raw<FIRST> RawT;
    RawT slot[3]; RawD dsl[3];
    auto row_exists = [&](int t) { return t >= 0 && t < g.H; };
    auto row_owned = [&](int t) { return t >= g.row0 && t < g.row1; };
    // Loads are UNCONDITIONAL (addresses clamped into the image / the segment, validity applied at publish): a load under a divergent
    // or even a uniform branch makes the compiler merge its result with the slot's old value right behind the branch, i.e. wait for
    // it at once -- which silently turns the prefetch into a blocking load (seen in the ISA as s_waitcnt vmcnt(0) in the loop).
    const int xc = x0 < 0 ? 0 : x0 > g.W - 2 ? g.W - 2 : x0;
    auto issue = [&](RawT& s, int t) {
        const int tc = t < 0 ? 0 : t > g.H - 1 ? g.H - 1 : t;
        const long i2 = (long)tc * W2 + (xc >> 1);
        s.ro = ldf4(ro4 + i2, nt_ra); s.ra = ldf2(ra2 + i2, nt_ra);
        if (!FIRST) { s.ao = ldf4(ao4 + i2, nt_ra); s.aa = ldf2(aa2 + i2, nt_ra); }
        s.po = ldf4(po4 + i2, nt_pin); s.pa = ldf2(pa2 + i2, nt_pin);
        s.cs = ldf4(cs4 + i2, nt_const);
        // the aligned dword that holds the pair's two flag bytes (publish shifts): a 16-bit load would leave a zero-extension of the
        // raw value for the compiler to place -- it places it at the loop latch, behind a wait for the fresh load
        s.f = nt_const ? __builtin_nontemporal_load(f4 + (i2 >> 1)) : f4[i2 >> 1];
    };
    // delta (and p_{k-2}) of the segment's own rows, prefetched two rows ahead (the halo rows re-read a row of the segment, unused)
    auto issue_d = [&](RawD& s, int t) {
        const int td = t < ya ? ya : t > yb - 1 ? yb - 1 : t;
        const long j2 = (long)td * W2 + (xc >> 1);
        s.dlo = ldf4(dl4 + j2, nt_delta); s.dla = ldf2(dl2 + j2, nt_delta);
        if (DMODE == 2) { s.ppo = qo4[j2]; s.ppa = qa2[j2]; }
    };

    // Window of three published rows.  Three rows are processed per loop iteration, so the roles (y-1, y, y+1) rotate through
    // win[0..2] with compile-time indices: no register-to-register window shift, no loop-carried copies.
    Row win[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { win[i].px[q] = 0.f; win[i].py[q] = 0.f; win[i].pa[q] = 0.f; win[i].c[q] = 1.f; win[i].s[q] = 0.f;
                                      win[i].rx[q] = 0.f; win[i].ry[q] = 0.f; win[i].ra[q] = 0.f; win[i].mo[q] = 0.f; win[i].ma[q] = 0.f; }
        win[i].f = 0u;
    }

    // publish row t from its raw slot into wn: r_k = r - alpha Ap, p_k = M^-1 r_k + beta p ; stores for owned rows (segment rows) and a slab's ghost rows
    auto publish = [&](const RawT& s, const RawD& sd, int t, Row& wn) {     // s, sd: copies made by take()
        const bool ok = xin && row_exists(t);
        float rx[2] = { s.ro.x, s.ro.z }, ry[2] = { s.ro.y, s.ro.w }, rq[2] = { s.ra.x, s.ra.y };
        if (!FIRST) {
            rx[0] = __builtin_fmaf(-alpha, s.ao.x, rx[0]); ry[0] = __builtin_fmaf(-alpha, s.ao.y, ry[0]);
            rx[1] = __builtin_fmaf(-alpha, s.ao.z, rx[1]); ry[1] = __builtin_fmaf(-alpha, s.ao.w, ry[1]);
            rq[0] = __builtin_fmaf(-alpha, s.aa.x, rq[0]); rq[1] = __builtin_fmaf(-alpha, s.aa.y, rq[1]);
        }
        const float ppx[2] = { s.po.x, s.po.z }, ppy[2] = { s.po.y, s.po.w }, ppq[2] = { s.pa.x, s.pa.y };
        // outside the image / beyond the segment: inactive pixels (M^-1 = 0, p = 0)
        const unsigned fl = ok ? (s.f >> (((((long)t * W2 + (x0 >> 1)) & 1) != 0) ? 16 : 0)) & 0xffffu : 0u;
        const float2 m0 = lut[fl & 31u], m1 = lut[(fl >> 8) & 31u];
        const float mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wn.px[q] = ok ? mo[q] * rx[q] + beta * ppx[q] : 0.f; wn.py[q] = ok ? mo[q] * ry[q] + beta * ppy[q] : 0.f; wn.pa[q] = ok ? ma[q] * rq[q] + beta * ppq[q] : 0.f;
            wn.rx[q] = rx[q]; wn.ry[q] = ry[q]; wn.ra[q] = rq[q]; wn.mo[q] = mo[q]; wn.ma[q] = ma[q];
        }
        wn.c[0] = s.cs.x; wn.s[0] = s.cs.y; wn.c[1] = s.cs.z; wn.s[1] = s.cs.w; wn.f = fl;
        const bool mine = t >= ya && t < yb;
        if (ok && xout && (mine || !row_owned(t))) {       // this wave's own rows, or a ghost row of the slab (kept current here)
            const long i2 = (long)t * W2 + (x0 >> 1);
            stf4(Ro4 + i2, make_float4(rx[0], ry[0], rx[1], ry[1]), nt_out); stf2(Ra2 + i2, make_float2(rq[0], rq[1]), nt_out);
            stf4(qo4 + i2, make_float4(wn.px[0], wn.py[0], wn.px[1], wn.py[1]), nt_pout); stf2(qa2 + i2, make_float2(wn.pa[0], wn.pa[1]), nt_pout);
            if (DMODE != 1 && mine) {
                float4 d = sd.dlo; float2 da = sd.dla;
                if (DMODE == 2) {
                    d.x = __builtin_fmaf(alpha2, sd.ppo.x, d.x); d.y = __builtin_fmaf(alpha2, sd.ppo.y, d.y);
                    d.z = __builtin_fmaf(alpha2, sd.ppo.z, d.z); d.w = __builtin_fmaf(alpha2, sd.ppo.w, d.w);
                    da.x = __builtin_fmaf(alpha2, sd.ppa.x, da.x); da.y = __builtin_fmaf(alpha2, sd.ppa.y, da.y);
                }
                d.x = __builtin_fmaf(alpha, ppx[0], d.x); d.y = __builtin_fmaf(alpha, ppy[0], d.y);
                d.z = __builtin_fmaf(alpha, ppx[1], d.z); d.w = __builtin_fmaf(alpha, ppy[1], d.w);
                da.x = __builtin_fmaf(alpha, ppq[0], da.x); da.y = __builtin_fmaf(alpha, ppq[1], da.y);
                stf4(dl4 + i2, d, nt_delta); stf2(dl2 + i2, da, nt_delta);
            }
        }
    };

    float acc = 0.0f; double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    // gather J^T J p_k for the centre row y of the window (wm = y-1, wc = y, wn = y+1)
    auto stencil = [&](int y, const Row& wm, const Row& wc, const Row& wn) {
        float ax[2], ay[2], av[2], am[2], ac[2], an[2], wfm[2], wfc[2], wfn[2];
        flag_masks(wm.f, wf2, am, wfm); flag_masks(wc.f, wf2, ac, wfc); flag_masks(wn.f, wf2, an, wfn);
        jtjp_pair(wm, wc, wn, wm, wc, wn, am, ac, an, wfc, wr2, ax, ay, av);      // (every lane: the DPP shifts read the neighbouring lanes' registers)
        if (xout) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                iter_sums_pixel(wc.px[q], wc.py[q], wc.pa[q], ax[q], ay[q], av[q], wc.rx[q], wc.ry[q], wc.ra[q], wc.mo[q], wc.ma[q], acc, s0, s1, s2);
            const long i2 = (long)y * W2 + (x0 >> 1);
            stf4(Ao4 + i2, make_float4(ax[0], ay[0], ax[1], ay[1]), nt_out); stf2(Aa2 + i2, make_float2(av[0], av[1]), nt_out);
            if (DIST) {     // multi-GPU row slabs: the first / last owned row of A p_k also goes into the neighbour's ghost row of its Ap_out
                            // buffer (peer-to-peer, write-through); drained by every wave before the arrival ticket (iter_tail)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (y == (k == 0 ? g.row0 : g.row1 - 1) && dd.peer_r[k]) {
                        float* d2 = dd.peer_r[k] + dd.peer_off_o[k] + 2 * x0;
                        st_sys(d2, ax[0]); st_sys(d2 + 1, ay[0]); st_sys(d2 + 2, ax[1]); st_sys(d2 + 3, ay[1]);
                        float* d1 = dd.peer_r[k] + dd.peer_off_a[k] + x0;
                        st_sys(d1, av[0]); st_sys(d1 + 1, av[1]);
                    }
                }
            }
        }
    };

    if (work) {
        const int t_first = ya - 1, t_last = yb;            // rows to publish: the segment and its two halo rows
        // No branch around a load anywhere in this loop (see `issue`): the step count is rounded up to a multiple of 3, rows beyond
        // t_last are clamped re-reads of the last row (cache hits) whose publish / stencil are predicated off.
        // There is no prologue either: the loop starts three rows early with empty slots (publish predicated off) and its refills are
        // the first loads.  A separate prologue is a second path into the loop header with its own (compiler-scheduled) issue order,
        // and the wait at the header is the conservative merge of both paths -- in practice vmcnt(0).
        // Entering an iteration at row t0: win[0] = row t0-2, win[1] = row t0-1.
#pragma unroll
        for (int j = 0; j < 3; ++j) { slot[j] = RawT{}; dsl[j] = RawD{}; }
        for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) {
            // The iteration's scalars, at the start of the SECOND trip: the first trip only issued the loads of rows t_first .. t_first + 2, nothing
            // needed alpha / beta yet; now the partial (or word) loads queue up behind those row loads and the additions run while the rows arrive.
            if (!FIRST && SCALARS_IN_LOOP && t0 == t_first) iteration_scalars<1>(aNp, aDp, bNp, prev, alpha, beta, scal_writer);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = t0 + j;
                Row& wn = win[(j + 2) % 3]; Row& wc = win[(j + 1) % 3]; Row& wm = win[j % 3];
                RawT cur; RawD curd = RawD{};
                take(cur, slot[j]);                          // (the only place that waits for memory: s_waitcnt vmcnt(N) with the other two slots still in flight)
                if (DMODE != 1) take<DMODE>(curd, dsl[j]);
                fence_order();                               // the refill stays behind the moves ...
                issue(slot[j], t + 3 > t_last ? t_last : t + 3);
                if (DMODE != 1) issue_d(dsl[(j + 2) % 3], t + 2);      // (the slot taken one step ago)
                fence_order();                               // ... and in front of the arithmetic
                // (wave-uniform branch, no load inside: the three lead-in rows and the rounding-up rows skip the arithmetic; the window keeps its zeros)
                if (t >= t_first && t <= t_last) publish(cur, curd, t, wn);
                if (t - 1 >= ya && t <= t_last) stencil(t - 1, wm, wc, wn);
            }
        }
    }
    iter_tail<MARCH_NT, DIST>(acc, s0, s1, s2, red, redd, aD_out, s12_out, bNp, &dd, fin_tickets, aD_word, bN_word, xslot);
}

// Is UrShape the unit pixel grid?  Counts the pixels whose right / down neighbour is not at the exact unit offset (the property
// pcg_init re-verifies on the device every GN step); run once per Init so that the HOST can pick this kernel.
__global__ __launch_bounds__(256) void k_urshape_check(int W, int H, const float2* __restrict__ ur, int* __restrict__ bad_out)
{
    int bad = 0;
    const long N = (long)W * H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)(i / W);
        const float2 u = ur[i];
        if (x + 1 < W) { const float2 v = ur[i + 1]; if (v.x - u.x != 1.0f || v.y - u.y != 0.0f) bad = 1; }
        if (y + 1 < H) { const float2 v = ur[i + W]; if (v.x - u.x != 0.0f || v.y - u.y != 1.0f) bad = 1; }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(bad_out, 1);
}

constexpr int MARCH_OCC = 2;     // register budget: two workgroups of 4 waves per CU (<= 256 VGPRs), the grid is sized for one

template <bool DIST>
int launch_march(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                 const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                 thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp, const int* irregular, thallo_dist_t d,
                 float* aD_out, double* s12_out, unsigned* fin_tickets, float* aD_word, float* bN_word, int xslot, hipStream_t stream, PrevSums prev = PrevSums{ nullptr, nullptr, 0, nullptr, nullptr })
{
    const int R = march_pick_rows(W, row1 - row0);
    if (R <= 0) return -(int)hipErrorNotSupported;                          // wider than the workgroup budget: thallo_hip_iw_march_rows() said so
    const MarchGeo g = make_march_geo(W, H, row0, row1, R);
    const int grid = (g.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;      // (only reachable through the tools' forced rows-per-segment)
    const bool first = mode & 1;
    const int dmode = first ? 1 : (mode >> 1) & 3;
    const float wf2 = w_fit * w_fit, wr2 = w_reg * w_reg;
#define MARCH_LAUNCH(F, DM) hipLaunchKernelGGL((k_iter_march<F, DM, MARCH_NTM, DIST, MARCH_OCC>), dim3(grid), dim3(MARCH_NT), 0, stream, g, cs, flags, wf2, wr2, \
        r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, irregular, d, fin_tickets, aD_word, bN_word, xslot, prev)
    // (the multi-GPU variant has no "apply two delta updates" form: peer stores on top of that variant's 256 registers spill, and at slab sizes the
    //  6 B/pixel it saves do not matter -- solver_dist.cpp updates delta every iteration on the device-side transport)
    if (DIST && dmode == 2) return -(int)hipErrorInvalidValue;
    if (first) MARCH_LAUNCH(true, 1);
    else if (dmode == 1) MARCH_LAUNCH(false, 1);
    else if (dmode == 2) { if constexpr (!DIST) MARCH_LAUNCH(false, 2); }
    else MARCH_LAUNCH(false, 0);
#undef MARCH_LAUNCH
    int e = check_launch(); return e ? e : grid;
}

}  // namespace

extern "C" {
int thallo_hip_iw_pcg_iter_march(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                 float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                 const float* p_in, float* p_out, float* delta, int mode,
                                 thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                 const int* irregular, float* aD_out, double* s12_out,
                                 unsigned* fin_tickets, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || (W & 1) || W < 2) return -(int)hipErrorInvalidValue;
    if (!cs || !flags || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (!(mode & 1) && (!Ap_in || !delta)) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    return launch_march<false>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode,
                               aNp, aDp, bNp, aNpp, aDpp, irregular, thallo_dist_t{}, aD_out, s12_out, fin_tickets, aD_word, bN_word, 0, (hipStream_t)stream);
}

int thallo_hip_iw_pcg_iter_march_deferred(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                          float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                          const float* p_in, float* p_out, float* delta, int mode,
                                          thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, thallo_prev_t prev,
                                          const int* irregular, float* aD_out, double* s12_out, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || (W & 1) || W < 2) return -(int)hipErrorInvalidValue;
    if (!cs || !flags || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (!(mode & 1) && (!Ap_in || !delta || prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word ||
                        !prev.betaN_word || prev.s12_partials == s12_out)) return -(int)hipErrorInvalidValue;
    const thallo_sum_t none = { nullptr, 0 };
    return launch_march<false>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode,
                               aNp, none, none, aNpp, aDpp, irregular, thallo_dist_t{}, aD_out, s12_out, nullptr, nullptr, nullptr, 0, (hipStream_t)stream,
                               (mode & 1) ? PrevSums{ nullptr, nullptr, 0, nullptr, nullptr } : PrevSums{ prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word });
}

int thallo_hip_iw_pcg_iter_march_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags,
                                      float w_fit, float w_reg, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                                      const float* p_in, float* p_out, float* delta, int mode,
                                      thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                      const int* irregular, thallo_dist_t d, float* aD_out, double* s12_out,
                                      unsigned* fin_tickets, int slot0, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || (W & 1) || W < 2) return -(int)hipErrorInvalidValue;
    if (!cs || !flags || !r_in || !r_out || !Ap_out || !p_in || !p_out || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if ((!(mode & 1) && (!Ap_in || !delta)) || d.world < 1 || d.world > THALLO_DIST_MAX_WORLD) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    if (fin_tickets && (slot0 < 0 || !d.mail || !d.ctl || 7 * d.world > 64 || bNp.count != 1)) return -(int)hipErrorInvalidValue;
    for (int k = 0; k < 2; ++k) if (d.peer_r[k] && ((d.peer_off_o[k] | d.peer_off_a[k]) & 1)) return -(int)hipErrorInvalidValue;
    return launch_march<true>(W, H, row0, row1, cs, flags, w_fit, w_reg, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, mode,
                              aNp, aDp, bNp, aNpp, aDpp, irregular, d, aD_out, s12_out, fin_tickets, aD_word, bN_word, slot0, (hipStream_t)stream);
}

int thallo_hip_iw_urshape_irregular(int W, int H, const float* urshape, int* count_out, thallo_stream_t stream)
{
    if (W < 1 || H < 1 || !urshape || !count_out) return -(int)hipErrorInvalidValue;
    if (hipMemsetAsync(count_out, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return -(int)hipErrorInvalidValue;
    int grid = thallo_hip_device_cu_count() * 8;
    const long want = ((long)W * H + 255) / 256;
    if (want < grid) grid = (int)want;
    hipLaunchKernelGGL(k_urshape_check, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, (const float2*)urshape, count_out);
    return check_launch();
}


void thallo_hip_march_debug_set(int what, int value)
{
    if (what == 0) g_march_rows = value;
    if (what == 6) g_march_cap = value;
}

/* rows per wave segment the marching kernels would use on `rows` owned rows of a W-wide image; 0 = the image has more column strips than the device
 * has workgroup slots, the marching kernels return -hipErrorNotSupported and the caller stays on the tile kernel */
int thallo_hip_iw_march_rows(int W, int rows)
{
    if (W < 2 || (W & 1) || rows < 1) return 0;
    return march_pick_rows(W, rows);
}

}  // extern "C"
