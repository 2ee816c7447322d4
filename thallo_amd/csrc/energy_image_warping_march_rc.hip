// energy_image_warping_march_rc.hip -- the marching one-kernel PCG iteration of image_warping WITHOUT an A p plane (round 4).
//
// energy_image_warping_march.hip moves 81 B/pixel per iteration (+ 18 of deferred delta): launch k-1 writes A p_{k-1} (12 B/pixel) only so that launch k
// can read it (12 B/pixel) to form r_k = r_{k-1} - alpha_{k-1} A p_{k-1}.  Here launch k RECOMPUTES A p_{k-1} from the rows of p_{k-1} it loads anyway
// (it needs p_{k-1} for p_k = M^-1 r_k + beta_{k-1} p_{k-1}): the stencil runs twice per pixel and launch -- once on p_{k-1} for the residual update, once
// on p_k for the iteration's sums -- and the plane never exists.  Per pixel: read r 12, p 12, cs 8, flags 1; write r 12, p 12 = 57 B (+ 18 of deferred
// delta on average = 75 B against 99).  The kernel is bandwidth- and latency-bound at one wave per SIMD; the second stencil fills issue slots the wave
// spent waiting.  Replaces PCGStep1 + PCGStep2 + PCGStep3 of gauss_newton.t:734-752,801-843,889-899.
//
// Same expressions on the same inputs as the stored-plane kernel (jtjp_pair / iter_sums_pixel of iw_march.hpp, -ffp-contract=on in both files), same
// strips, segments and summation order: r, p, delta and every alpha_k / beta_k are BIT-identical to it (tests/test_gpu_parity.py).
//
// Shape: as the stored-plane kernel -- a wave owns a column strip of 128 pixels (lane l: pixels x0+2l, x0+2l+1; lanes 1..62 are outputs) and marches down
// its R rows; x neighbours through DPP wave shifts, y neighbours from the lane's own registers.  What changes is the depth of the pipeline: at the step
// that takes row t (p_{k-1}, cos / sin, flags of row t; r_{k-1}, delta of row t-1) the lane forms
//     A p_{k-1}(t-1)  from p_{k-1}(t-2 .. t)            ->  r_k(t-1), p_k(t-1)   (stored for the segment's own rows)
//     A p_k(t-2)      from p_k(t-3 .. t-1)              ->  the sums of row t-2
// so a segment [ya, yb) takes rows ya-2 .. yb+1 of p / cs / flags (4 halo rows instead of 2) and rows ya-1 .. yb of r.  Row state lives in rings of four
// indexed by the row modulo 4, four rows per loop trip (compile-time indices, no register shifts); rows are prefetched DEPTH steps ahead into registers.
#include "iw_march.hpp"

using namespace thallo;

namespace {

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

template <int DMODE>
struct RawRc {                              // what one step takes, for one lane (2 pixels)
    float4 po, cs; float2 pa; unsigned f;   // row t:   p_{k-1} (Offset part x0,y0,x1,y1 | Angle part a0,a1), (c0,s0,c1,s1), the dword holding the pair's flags bytes
    float4 ro; float2 ra;                   // row t-1: r_{k-1}
    float4 dlo, ppo; float2 dla, ppa;       // row t-1: delta (DMODE 0, 2) and p_{k-2} (DMODE 2)
};
template <int DMODE>
__device__ __forceinline__ void take(RawRc<DMODE>& d, const RawRc<DMODE>& s)
{
    take4(d.po, s.po); take2(d.pa, s.pa); take4(d.cs, s.cs); take1(d.f, s.f); take4(d.ro, s.ro); take2(d.ra, s.ra);
    if (DMODE != 1) { take4(d.dlo, s.dlo); take2(d.dla, s.dla); }
    if (DMODE == 2) { take4(d.ppo, s.ppo); take2(d.ppa, s.ppa); }
}

struct PRow { float px[2], py[2], pa[2]; };                      // p of a lane's pixel pair in one row
struct GRow { float c[2], s[2]; unsigned f; };                   // cos / sin of Angle and the two flags bytes
struct RRow { float rx[2], ry[2], ra[2], mo[2], ma[2]; };        // r_k and M^-1

// DMODE: the delta update this launch carries (THALLO_IW_STEP1_MODE): 0 delta += alpha p_{k-1}; 1 none; 2 delta += alpha_{k-2} p_{k-2} + alpha_{k-1} p_{k-1}
template <int DMODE, int DEPTH, int NTM, int OCC>
__global__ __launch_bounds__(MARCH_NT, OCC) void k_iter_march_rc(MarchGeo g, const float* __restrict__ cs, const unsigned char* __restrict__ flags, float wf2, float wr2,
                                                            const float* __restrict__ r_in, float* __restrict__ r_out,
                                                            const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta,
                                                            thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                                            float* __restrict__ aD_out, double* __restrict__ s12_out, const int* __restrict__ irregular,
                                                            unsigned* __restrict__ fin_tickets, float* __restrict__ aD_word, float* __restrict__ bN_word, PrevSums prev)
{
    static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "the prefetch slots rotate inside a trip of four rows");
    __shared__ float2 lut[32];
    __shared__ float red[16];
    __shared__ double redd[48];
    constexpr bool nt_delta = NTM & 1, nt_ra = NTM & 2, nt_out = NTM & 4, nt_pin = NTM & 8, nt_pout = NTM & 16, nt_const = NTM & 32;
    // unit-pixel-grid form only; should the word pcg_init wrote this GN step say otherwise, poison the scalars (NaN cost downstream)
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
    const float4* __restrict__ po4 = reinterpret_cast<const float4*>(p_in);  const float2* __restrict__ pa2 = reinterpret_cast<const float2*>(p_in + 2 * N);
    const float4* __restrict__ cs4 = reinterpret_cast<const float4*>(cs);
    const unsigned* __restrict__ f4 = reinterpret_cast<const unsigned*>(flags);
    float4* __restrict__ Ro4 = reinterpret_cast<float4*>(r_out);  float2* __restrict__ Ra2 = reinterpret_cast<float2*>(r_out + 2 * N);
    float4* __restrict__ qo4 = reinterpret_cast<float4*>(p_out);  float2* __restrict__ qa2 = reinterpret_cast<float2*>(p_out + 2 * N);
    float4* __restrict__ dl4 = reinterpret_cast<float4*>(delta);  float2* __restrict__ dl2 = reinterpret_cast<float2*>(delta + 2 * N);

    float alpha = 0.0f, beta = 0.0f, alpha2 = 0.0f;
    // the words of iteration k-1 are left behind by the one wave that owns the first segment of strip 0
    const bool scal_writer = work && strip == 0 && ya == g.row0 && lane == 0;

    typedef RawRc<DMODE> RawT;
    RawT slot[DEPTH];
    // Loads are UNCONDITIONAL (addresses clamped into the image / the segment, validity applied when the row enters the rings): a load under a branch is
    // merged with the slot's old value right behind the branch, i.e. waited for at once (energy_image_warping_march.hip)
    const int xc = x0 < 0 ? 0 : x0 > g.W - 2 ? g.W - 2 : x0;
    auto issue = [&](RawT& s, int t) {
        const int tc = t < 0 ? 0 : t > g.H - 1 ? g.H - 1 : t;
        const long i2 = (long)tc * W2 + (xc >> 1);
        s.po = ldf4(po4 + i2, nt_pin); s.pa = ldf2(pa2 + i2, nt_pin);
        s.cs = ldf4(cs4 + i2, nt_const);
        s.f = nt_const ? __builtin_nontemporal_load(f4 + (i2 >> 1)) : f4[i2 >> 1];
        const int tr = t - 1 < 0 ? 0 : t - 1 > g.H - 1 ? g.H - 1 : t - 1;
        const long j2 = (long)tr * W2 + (xc >> 1);
        s.ro = ldf4(ro4 + j2, nt_ra); s.ra = ldf2(ra2 + j2, nt_ra);
        if (DMODE != 1) {       // delta (and p_{k-2}) of the segment's own rows only (the halo rows re-read a row of the segment, unused)
            const int td = t - 1 < ya ? ya : t - 1 > yb - 1 ? yb - 1 : t - 1;
            const long d2 = (long)td * W2 + (xc >> 1);
            s.dlo = ldf4(dl4 + d2, nt_delta); s.dla = ldf2(dl2 + d2, nt_delta);
            if (DMODE == 2) { s.ppo = qo4[d2]; s.ppa = qa2[d2]; }
        }
    };

    // rings of four rows, indexed by the row modulo 4 (compile-time inside a trip of four rows)
    PRow pp[4], pk[4]; GRow gg[4]; RRow rr[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) { pp[i].px[q] = 0.f; pp[i].py[q] = 0.f; pp[i].pa[q] = 0.f; pk[i].px[q] = 0.f; pk[i].py[q] = 0.f; pk[i].pa[q] = 0.f; gg[i].c[q] = 1.f; gg[i].s[q] = 0.f; }
        gg[i].f = 0u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) { rr[i].rx[q] = 0.f; rr[i].ry[q] = 0.f; rr[i].ra[q] = 0.f; rr[i].mo[q] = 0.f; rr[i].ma[q] = 0.f; }

    float acc = 0.0f; double s0 = 0.0, s1 = 0.0, s2 = 0.0;

    if (work) {
        const int t_first = ya - 2, t_last = yb + 1;          // rows of p_{k-1} / cs / flags to take
        // No prologue (a second path into the loop header makes its wait the conservative merge of both): the loop starts DEPTH rows early with empty slots.
        // The rings are indexed by j, the position inside the trip, so any starting row works.
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) slot[j] = RawT{};
        const int t_begin = t_first - DEPTH;
        for (int t0 = t_begin; t0 <= t_last; t0 += 4) {
            // The iteration's scalars, at the start of the SECOND trip: the first trip issued the loads of the first rows and entered (at most) rows ya-2, ya-1,
            // which need neither alpha nor beta; the partial loads queue up behind those row loads and the additions run while the rows arrive.
            // (DEPTH 1 reaches row ya in its first trip: in front of the loop.)
            if (DEPTH == 1 ? t0 == t_begin : t0 == t_begin + 4) {
                iteration_scalars<1>(aNp, aDp, bNp, prev, alpha, beta, scal_writer);
                if (DMODE == 2) alpha2 = safe_div<false>(sum_partials(aNpp.partials, aNpp.count), sum_partials(aDpp.partials, aDpp.count));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + j;
                // ring roles at this step: row t -> index j, t-1 -> j+3, t-2 -> j+2, t-3 -> j+1 (mod 4)
                PRow& p0 = pp[j % 4]; const PRow& p1 = pp[(j + 3) % 4]; const PRow& p2 = pp[(j + 2) % 4];
                GRow& g0 = gg[j % 4]; const GRow& g1 = gg[(j + 3) % 4]; const GRow& g2 = gg[(j + 2) % 4]; const GRow& g3 = gg[(j + 1) % 4];
                PRow& k1 = pk[(j + 3) % 4]; const PRow& k2 = pk[(j + 2) % 4]; const PRow& k3 = pk[(j + 1) % 4];
                RRow& r1 = rr[(j + 1) % 2]; const RRow& r2 = rr[j % 2];
                RawT cur;
                take(cur, slot[j % DEPTH]);                  // (the only place that waits for memory)
                fence_order();                               // the refill stays behind the moves ...
                issue(slot[j % DEPTH], t + DEPTH > t_last ? t_last : t + DEPTH);
                fence_order();                               // ... and in front of the arithmetic
                if (t < t_first || t > t_last) continue;     // (wave-uniform, no load inside: lead-in and rounding-up rows; the rings keep their zeros)
                // ---- row t enters the p_{k-1} / geometry rings
                {
                    const bool ok = xin && t >= 0 && t < g.H;
                    const unsigned fl = ok ? (cur.f >> (((((long)t * W2 + (x0 >> 1)) & 1) != 0) ? 16 : 0)) & 0xffffu : 0u;
                    p0.px[0] = ok ? cur.po.x : 0.f; p0.py[0] = ok ? cur.po.y : 0.f; p0.px[1] = ok ? cur.po.z : 0.f; p0.py[1] = ok ? cur.po.w : 0.f;
                    p0.pa[0] = ok ? cur.pa.x : 0.f; p0.pa[1] = ok ? cur.pa.y : 0.f;
                    g0.c[0] = cur.cs.x; g0.s[0] = cur.cs.y; g0.c[1] = cur.cs.z; g0.s[1] = cur.cs.w; g0.f = fl;
                }
                // ---- row u = t-1: A p_{k-1}(u) -> r_k(u), p_k(u)
                const int u = t - 1;
                if (u >= ya - 1 && u <= yb) {
                    float ax[2], ay[2], av[2];
                    jtjp_pair(p2, p1, p0, g2, g1, g0, wf2, wr2, ax, ay, av);
                    const bool ok = xin && u >= 0 && u < g.H;
                    float rx[2] = { cur.ro.x, cur.ro.z }, ry[2] = { cur.ro.y, cur.ro.w }, rq[2] = { cur.ra.x, cur.ra.y };
                    rx[0] = __builtin_fmaf(-alpha, ax[0], rx[0]); ry[0] = __builtin_fmaf(-alpha, ay[0], ry[0]);
                    rx[1] = __builtin_fmaf(-alpha, ax[1], rx[1]); ry[1] = __builtin_fmaf(-alpha, ay[1], ry[1]);
                    rq[0] = __builtin_fmaf(-alpha, av[0], rq[0]); rq[1] = __builtin_fmaf(-alpha, av[1], rq[1]);
                    const float2 m0 = lut[g1.f & 31u], m1 = lut[(g1.f >> 8) & 31u];
                    const float mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        k1.px[q] = ok ? mo[q] * rx[q] + beta * p1.px[q] : 0.f; k1.py[q] = ok ? mo[q] * ry[q] + beta * p1.py[q] : 0.f; k1.pa[q] = ok ? ma[q] * rq[q] + beta * p1.pa[q] : 0.f;
                        r1.rx[q] = rx[q]; r1.ry[q] = ry[q]; r1.ra[q] = rq[q]; r1.mo[q] = mo[q]; r1.ma[q] = ma[q];
                    }
                    if (ok && xout && u >= ya && u < yb) {
                        const long i2 = (long)u * W2 + (x0 >> 1);
                        stf4(Ro4 + i2, make_float4(rx[0], ry[0], rx[1], ry[1]), nt_out); stf2(Ra2 + i2, make_float2(rq[0], rq[1]), nt_out);
                        stf4(qo4 + i2, make_float4(k1.px[0], k1.py[0], k1.px[1], k1.py[1]), nt_pout); stf2(qa2 + i2, make_float2(k1.pa[0], k1.pa[1]), nt_pout);
                        if (DMODE != 1) {
                            float4 d = cur.dlo; float2 da = cur.dla;
                            if (DMODE == 2) {
                                d.x = __builtin_fmaf(alpha2, cur.ppo.x, d.x); d.y = __builtin_fmaf(alpha2, cur.ppo.y, d.y);
                                d.z = __builtin_fmaf(alpha2, cur.ppo.z, d.z); d.w = __builtin_fmaf(alpha2, cur.ppo.w, d.w);
                                da.x = __builtin_fmaf(alpha2, cur.ppa.x, da.x); da.y = __builtin_fmaf(alpha2, cur.ppa.y, da.y);
                            }
                            d.x = __builtin_fmaf(alpha, p1.px[0], d.x); d.y = __builtin_fmaf(alpha, p1.py[0], d.y);
                            d.z = __builtin_fmaf(alpha, p1.px[1], d.z); d.w = __builtin_fmaf(alpha, p1.py[1], d.w);
                            da.x = __builtin_fmaf(alpha, p1.pa[0], da.x); da.y = __builtin_fmaf(alpha, p1.pa[1], da.y);
                            stf4(dl4 + i2, d, nt_delta); stf2(dl2 + i2, da, nt_delta);
                        }
                    }
                }
                // ---- row v = t-2: A p_k(v) and the iteration's sums
                const int v = t - 2;
                if (v >= ya && v < yb) {
                    float ax[2], ay[2], av[2];
                    jtjp_pair(k3, k2, k1, g3, g2, g1, wf2, wr2, ax, ay, av);
                    if (xout) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            iter_sums_pixel(k2.px[q], k2.py[q], k2.pa[q], ax[q], ay[q], av[q], r2.rx[q], r2.ry[q], r2.ra[q], r2.mo[q], r2.ma[q], acc, s0, s1, s2);
                    }
                }
            }
        }
    }
    iter_tail<MARCH_NT, false>(acc, s0, s1, s2, red, redd, aD_out, s12_out, bNp, nullptr, fin_tickets, aD_word, bN_word, 0);
}

}  // namespace

namespace thallo {
int g_march_rc_depth = 2;      // tools: rows of prefetch (1, 2, 4)
int g_march_rc_occ = 2;        // tools: register budget -- workgroups of 4 waves per CU the kernel is compiled for (2: <= 256 registers, 1: <= 512; the grid is sized for one)
}

namespace {
int launch_march_rc(int W, int H, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                    const float* r_in, float* r_out, const float* p_in, float* p_out, float* delta, int mode,
                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp, const int* irregular,
                    float* aD_out, double* s12_out, unsigned* fin_tickets, float* aD_word, float* bN_word, hipStream_t stream, PrevSums prev)
{
    const int R = march_pick_rows(W, H);
    if (R <= 0) return -(int)hipErrorNotSupported;
    const MarchGeo g = make_march_geo(W, H, 0, H, R);
    const int grid = (g.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const int dmode = (mode >> 1) & 3;
    const float wf2 = w_fit * w_fit, wr2 = w_reg * w_reg;
#define RC_LAUNCH(DM, DP, OCC) hipLaunchKernelGGL((k_iter_march_rc<DM, DP, MARCH_NTM, OCC>), dim3(grid), dim3(MARCH_NT), 0, stream, g, cs, flags, wf2, wr2, \
        r_in, r_out, p_in, p_out, delta, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, irregular, fin_tickets, aD_word, bN_word, prev)
#define RC_BY_DEPTH(DM) do { if (g_march_rc_occ == 1) { if (g_march_rc_depth == 4) RC_LAUNCH(DM, 4, 1); else RC_LAUNCH(DM, 2, 1); } \
                             else if (g_march_rc_depth == 1) RC_LAUNCH(DM, 1, 2); else RC_LAUNCH(DM, 2, 2); } while (0)
    if (dmode == 1) RC_BY_DEPTH(1); else if (dmode == 2) RC_BY_DEPTH(2); else RC_BY_DEPTH(0);
#undef RC_BY_DEPTH
#undef RC_LAUNCH
    int e = check_launch(); return e ? e : grid;
}
}  // namespace

extern "C" {

int thallo_hip_iw_pcg_iter_march_rc(int W, int H, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    const float* r_in, float* r_out, const float* p_in, float* p_out, float* delta, int mode,
                                    thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp,
                                    const int* irregular, float* aD_out, double* s12_out,
                                    unsigned* fin_tickets, float* aD_word, float* bN_word, thallo_stream_t stream)
{
    if ((W & 1) || W < 2 || H < 1 || (mode & 1)) return -(int)hipErrorInvalidValue;          // (the first iteration of a GN step has no A p_{k-1}: the stored-plane kernel runs it)
    if (!cs || !flags || !r_in || !r_out || !p_in || !p_out || !delta || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (!fin_tickets || !aD_word || !bN_word) { fin_tickets = nullptr; aD_word = nullptr; bN_word = nullptr; }
    return launch_march_rc(W, H, cs, flags, w_fit, w_reg, r_in, r_out, p_in, p_out, delta, mode, aNp, aDp, bNp, aNpp, aDpp, irregular,
                           aD_out, s12_out, fin_tickets, aD_word, bN_word, (hipStream_t)stream, PrevSums{ nullptr, nullptr, 0, nullptr, nullptr });
}

int thallo_hip_iw_pcg_iter_march_rc_deferred(int W, int H, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                             const float* r_in, float* r_out, const float* p_in, float* p_out, float* delta, int mode,
                                             thallo_sum_t aNp, thallo_sum_t aNpp, thallo_sum_t aDpp, thallo_prev_t prev,
                                             const int* irregular, float* aD_out, double* s12_out, thallo_stream_t stream)
{
    if ((W & 1) || W < 2 || H < 1 || (mode & 1)) return -(int)hipErrorInvalidValue;
    if (!cs || !flags || !r_in || !r_out || !p_in || !p_out || !delta || !aD_out || !s12_out) return -(int)hipErrorInvalidValue;
    if (prev.count < 1 || prev.count > THALLO_MAX_PARTIALS || !prev.alphaD_partials || !prev.s12_partials || !prev.alphaD_word || !prev.betaN_word ||
        prev.s12_partials == s12_out) return -(int)hipErrorInvalidValue;
    const thallo_sum_t none = { nullptr, 0 };
    return launch_march_rc(W, H, cs, flags, w_fit, w_reg, r_in, r_out, p_in, p_out, delta, mode, aNp, none, none, aNpp, aDpp, irregular,
                           aD_out, s12_out, nullptr, nullptr, nullptr, (hipStream_t)stream,
                           PrevSums{ prev.alphaD_partials, prev.s12_partials, prev.count, prev.alphaD_word, prev.betaN_word });
}

void thallo_hip_march_rc_debug_set(int what, int value) { if (what == 0) g_march_rc_depth = value; if (what == 1) g_march_rc_occ = value; }

}  // extern "C"
