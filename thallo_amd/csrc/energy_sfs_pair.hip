// energy_sfs_pair.hip -- shape_from_shading's marching kernels on PIXEL PAIRS (round 6).
//
// energy_sfs.hip's k_march gives a lane ONE pixel: 60 useful pixels per wave row, 4-byte loads, one scalar instruction per pixel and operation -- 0.47-0.59 of the HBM
// peak where image_warping's marching kernel reaches 0.62 on two pixels per lane.  Here
//   * a lane owns two horizontally adjacent pixels (lane l: columns x0 + 2l, x0 + 2l + 1; lanes 1..62 produce output = 124 pixels per wave row, the outer lanes carry the
//     radius-2 halo), every per-pixel quantity is a register PAIR (px0, px1) and every operation of the chain one packed instruction (v_pk_fma_f32 / v_pk_mul_f32 /
//     v_pk_add_f32) on both pixels -- the same operations, in the same order, with the same roundings per pixel as k_march's;
//   * the x neighbours of a pair: left = (left lane's pixel 1, own pixel 0), right = (own pixel 1, right lane's pixel 0) -- one DPP wave shift and one move per exchange
//     of TWO pixels (k_march: one DPP per pixel);
//   * every plane is read with 8-byte raw-buffer loads through a descriptor with the row offset in an SGPR (no vector address arithmetic in the row step);
//   * the planes the iteration reads are PACKED (sfs_pair.hpp): Gx, Gy, Gz planar (12 B/pixel: BI, which only PCGInit1 and the cost read, is a plane of its own), the
//     flags byte and the two edge-mask bytes in one dword per pixel (4 B/pixel instead of 8 + 1 for float row weights): 40 bytes per pixel and GN iteration instead of 49;
//   * precompute (a-11; gauss_newton.t:979-986, thallo.t:4046-4094) writes those planes from CLOSED-FORM partials of the shading term (the normal, the SH polynomial and
//     its gradient written out: dBI/dX_q = h . dn/dX_q with h = |n|^-1 (I - nn^T) grad_n B) instead of 3-wide forward-mode duals (~1/3 of the arithmetic);
//   * the LM step loses three launches: PCGFinalizeDiagonal (gauss_newton.t:936-969) rides in the J^T F pass (FIN), and the model cost's three launches (owed delta
//     update, applyJTJ, dot) are one (MODEL).
// Structure (segments, rings of three rows, prefetch slots, deferred finish, slab ghost rows) is k_march's; results agree with it to rounding (tests/test_gpu_parity.py,
// tools/sfs_probe.py: planes to 1e-6 of their largest entry, sums to the float / double rounding of a different summation order).  Built with -ffp-contract=on: equal
// source expressions give equal bits in every instantiation.
#include <stdlib.h>
#include <stdint.h>
#include "iw_march.hpp"
#include "sfs_pair.hpp"
#include "sfs_pair_device.hpp"

using namespace thallo;

namespace {

constexpr int PM_WG_PER_CU = 1;           // grid sizing: one workgroup of 4 waves per CU (tools/sfs_pair_time.py at 2048^2: 32.6 us against 33.9 for two, 36.0 for three); registers for 2
struct PmGeo { int W, H, ra, rb, yoff, R, nstrips, total; };
// which strip and rows a wave works on: energy_sfs.hip's placement (workgroups b and b + 8 share an XCD; group b % 8 owns a contiguous range of (band of 4 segments,
// strip) ids, x-adjacent strips first)
__device__ __forceinline__ void pm_place(const PmGeo& g, int wave, int& strip, int& ya, int& yb)
{
    strip = 0; ya = 0; yb = 0;
    const int NG = (gridDim.x % 8) == 0 ? 8 : 1;
    const int grp = blockIdx.x % NG, l = blockIdx.x / NG;
    const long lo = (long)g.total * grp / NG, hi = (long)g.total * (grp + 1) / NG;
    const long id = lo + l;
    if (id < hi) {
        strip = (int)(id % g.nstrips);
        const int seg = (int)(id / g.nstrips) * (PM_NT / 64) + wave;
        ya = g.ra + seg * g.R; yb = ya + g.R;
        if (yb > g.rb) yb = g.rb;
        if (ya > g.rb) ya = g.rb;
    }
}

struct PmPupd { const float* p_in; float* p_out; thallo_sum_t aN, bN; int first; };
struct PmUpd { float* r_out; const float* A_in; const float* p_in; float* p_out; float* delta; thallo_sum_t aN, aD, bN; int first; int lm; const float* b; const float* pre;
               const double* prev_s3; int prev_nb; float* aD_word; float* bN_word; };
struct PmFin { float* SSq; float* CtC; float* pre; float* b; float radius, min_lm, max_lm; int save_ssq; };
struct PmModel { const float* p_even; const float* p_odd; const float* b; const float* aN_words; const float* aD_words; int stride; const float* state; int L; float* db_out;
                 float* X; float* prevX; };      // X != NULL: savePreviousUnknowns + PCGLinearUpdate ride along (prevX = X, X += delta) on the segment's own rows

// one row of one lane (two pixels) as loaded
struct PmRaw { u32x2 gx, gy, gz, bi, fw, v, rs, ct, pv, av, dl, bb, mi; };

// MODE bits (template): what rides along -- as k_march's SUMS / CTC / INIT / DIAG / PUPD / UPD / LMQ; FIN (with INIT, DIAG): PCGFinalizeDiagonal folded in;
// MODEL: the LM model cost (delta += alpha p, J^T J delta, the two dot products)
template <bool SUMS, bool CTC, bool INIT, bool DIAG, int OCC, bool PUPD, bool UPD, bool LMQ, bool FIN, bool MODEL, int DEPTH>
__global__ __launch_bounds__(PM_NT, OCC) void k_pmarch(PmGeo g, PCam cm, const float* __restrict__ Gp, const unsigned* __restrict__ Fw, const float* __restrict__ v, const float* __restrict__ ctc,
                                                       float* __restrict__ out, float* __restrict__ part_out, const float* __restrict__ rs,
                                                       double* __restrict__ s3_out, const unsigned* __restrict__ gate, FinArgs fin,
                                                       float* __restrict__ z, float* __restrict__ p_prev, float* __restrict__ delta, float* __restrict__ diag,
                                                       PmPupd pu, PmUpd up, LmFin lmf, PmFin fd, PmModel md)
{
    static_assert(DEPTH == 3 || DEPTH == 6, "the prefetch slots rotate inside a trip of DEPTH rows, the rings of three inside it");
    __shared__ float red[16];
    __shared__ double redd[(LMQ ? 6 : 3) * PM_NT / 64];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;
    float beta = 0.0f, alpha = 0.0f;
    if (PUPD && !pu.first) beta = safe_div<true>(sum_partials(pu.bN.partials, pu.bN.count), sum_partials(pu.aN.partials, pu.aN.count));
    if (UPD && !LMQ && !up.first && up.prev_nb > 0) {                                  // the deferred finish of iteration k-1 (k_march's: the same order, the same arithmetic)
        float ad, an; double t3[3];
        last_workgroup_totals<3, true>(up.aD.partials, up.prev_s3, nullptr, up.prev_nb, up.aN, red, redd, ad, an, t3);
        alpha = safe_div<false>(an, ad);
        double bnd = t3[0] - 2.0 * (double)alpha * t3[1] + (double)alpha * (double)alpha * t3[2];
        if (!(bnd > 0.0)) bnd = 0.0;
        const float bnf = (float)bnd;
        beta = safe_div<false>(bnf, an);
        if (blockIdx.x == 0 && threadIdx.x == 0) { up.aD_word[0] = ad; up.bN_word[0] = bnf; }
        lds_barrier();
    } else if (UPD && !up.first) {
        const float an = sum_partials(up.aN.partials, up.aN.count);
        const float ad = sum_partials(up.aD.partials, up.aD.count), bn = sum_partials(up.bN.partials, up.bN.count);
        alpha = up.lm ? safe_div<true>(an, ad) : safe_div<false>(an, ad);
        beta  = up.lm ? safe_div<true>(bn, an) : safe_div<false>(bn, an);
    }
    const float* __restrict__ model_p = nullptr;
    if (MODEL) {                                                                        // thallo_hip_lm_owed_delta's rule: the iteration the loop ended on
        const unsigned gt = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(md.state)[1]);
        const int done = gt ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(md.state)[2]) : md.L;
        const int kl = done - 1;
        model_p = (kl & 1) ? md.p_odd : md.p_even;
        if (kl >= 0) alpha = safe_div<true>(md.aN_words[(long)kl * md.stride], md.aD_words[(long)kl * md.stride]);
        else { alpha = 0.0f; model_p = md.p_even; }
    }
    alpha = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(alpha))); beta = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(beta)));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = g.W, H = g.H;
    int strip, ya, yb;
    pm_place(g, wave, strip, ya, yb);
    const bool work = ya < yb;
    const int x0 = strip * PM_USE - 2 + 2 * lane;             // first of the lane's two pixels (W even: both inside the image or neither)
    const bool xin = x0 >= 0 && x0 < W;
    const bool xout = xin && lane >= 1 && lane <= 62;
    const int xcl = x0 < 0 ? 0 : x0 > W - 2 ? W - 2 : x0;
    const unsigned vo = (unsigned)xcl * 4u;                   // the pair's byte offset inside a row of floats / dwords
    const unsigned rowb = (unsigned)W * 4u, planeb = (unsigned)W * (unsigned)H * 4u;
    const M2 xp1 = { true, x0 + 2 < W };                      // x + 1 < W per pixel
    // coef_0 at x-1, x, x+1 of both pixels (coef_2 = 1); coef_1 per row, carried
    const v2f cxc = { coef0(cm, x0), coef0(cm, x0 + 1) }, cxm = { coef0(cm, x0 - 1), cxc.x }, cxp = { cxc.y, coef0(cm, x0 + 2) };

    const rsrc_t RS_G = make_rsrc(Gp), RS_F = make_rsrc(Fw), RS_V = make_rsrc(v);
    const rsrc_t RS_RS = make_rsrc(rs), RS_CT = make_rsrc(ctc);
    const rsrc_t RS_PI = make_rsrc(PUPD ? pu.p_in : UPD ? up.p_in : MODEL ? model_p : (const float*)v);
    const rsrc_t RS_AI = make_rsrc(UPD ? up.A_in : (const float*)v), RS_DL = make_rsrc(UPD ? (const float*)up.delta : (const float*)v);
    const rsrc_t RS_MI = make_rsrc(LMQ ? up.pre : (const float*)v), RS_BB = make_rsrc(LMQ ? up.b : MODEL ? md.b : (const float*)v);
    const rsrc_t RS_OUT = make_rsrc(out), RS_PO = make_rsrc(PUPD ? pu.p_out : UPD ? up.p_out : out), RS_RO = make_rsrc(UPD ? up.r_out : out);

    v2f acc = { 0.f, 0.f }, acc2 = { 0.f, 0.f }; Sums3 sm; SumsQ sq;
    if (work) {
        const int t_first = ya - 2, t_last = yb + 1;
        // loads are unconditional (rows clamped into the image, columns into the row; validity applied when the row is taken)
        auto issue = [&](PmRaw& s, int t) {
            const unsigned tc = (unsigned)(t < 0 ? 0 : t > H - 1 ? H - 1 : t), row = tc * rowb;
            s.gx = bld2(RS_G, vo, row); s.gy = bld2(RS_G, vo, planeb + row); s.gz = bld2(RS_G, vo, 2u * planeb + row);
            if (INIT) s.bi = bld2(RS_G, vo, 3u * planeb + row);
            s.fw = bld2(RS_F, vo, row);
            s.v = bld2(RS_V, vo, row);
            if (PUPD || UPD || MODEL) s.pv = bld2(RS_PI, vo, row);
            if (UPD) {
                s.av = bld2(RS_AI, vo, row);
                const unsigned td = (unsigned)(t < ya ? ya : t > yb - 1 ? yb - 1 : t);          // delta: the segment's own rows only
                if (up.delta) s.dl = bld2(RS_DL, vo, td * rowb); else s.dl = u32x2{ 0u, 0u };
                if (LMQ) s.mi = bld2(RS_MI, vo, row);
            }
            if (SUMS || CTC || MODEL) {
                const unsigned yo = (unsigned)(t - 2 < ya ? ya : t - 2 > yb - 1 ? yb - 1 : t - 2), ro = yo * rowb;
                if (SUMS && !UPD) s.rs = bld2(RS_RS, vo, ro);
                if (CTC) s.ct = bld2(RS_CT, vo, ro);
                if (LMQ || MODEL) s.bb = bld2(RS_BB, vo, ro);
            }
        };
        auto take = [&](PmRaw& d, const PmRaw& s) {
            d = PmRaw{};
            take2u(d.gx, s.gx); take2u(d.gy, s.gy); take2u(d.gz, s.gz); take2u(d.fw, s.fw); take2u(d.v, s.v);
            if (INIT) take2u(d.bi, s.bi);
            if (PUPD || UPD || MODEL) take2u(d.pv, s.pv);
            if (UPD) { take2u(d.av, s.av); take2u(d.dl, s.dl); if (LMQ) take2u(d.mi, s.mi); }
            if (SUMS && !UPD) take2u(d.rs, s.rs);
            if (CTC) take2u(d.ct, s.ct);
            if (LMQ || MODEL) take2u(d.bb, s.bb);
        };
        const v2f Z2 = { 0.f, 0.f };
        // state carried from row to row: rings of three indexed by the row modulo 3 (DEPTH rows per loop trip: every index is a compile-time constant)
        v2f Vv[3] = { Z2, Z2, Z2 }, dB[3] = { Z2, Z2, Z2 }, Uh[3] = { Z2, Z2, Z2 }, Uv[3] = { Z2, Z2, Z2 }, Tt[3] = { Z2, Z2, Z2 };
        v2f Gx[3] = { Z2, Z2, Z2 }, Gy[3] = { Z2, Z2, Z2 }, Gz[3] = { Z2, Z2, Z2 }, Wy[3] = { Z2, Z2, Z2 }, Wx[3] = { Z2, Z2, Z2 };
        v2f Rk[3] = { Z2, Z2, Z2 }, Dk[3] = { Z2, Z2, Z2 }, Mk[3] = { splat(1.f), splat(1.f), splat(1.f) };
        float Cy[3] = { 0.f, 0.f, 0.f };
        unsigned Fl[3] = { 0u, 0u, 0u };
        M2 Wn[3] = { { false, false }, { false, false }, { false, false } };
        v2f Rr[3][3] = { { Z2, Z2, Z2 }, { Z2, Z2, Z2 }, { Z2, Z2, Z2 } };
        PmRaw slot[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) slot[j] = PmRaw{};
        // no prologue: the loop starts DEPTH rows early with empty slots and its refills are the first loads (one path into the loop header)
        for (int t0 = t_first - DEPTH; t0 <= t_last; t0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int t = t0 + j;
                const int k0 = j % 3, k1 = (j + 2) % 3, k2 = (j + 1) % 3;          // ring slots of rows t, t-1, t-2 (and t-3 = t)
                PmRaw cur;
                take(cur, slot[j]);
                fence_order();
                issue(slot[j], t + DEPTH > t_last ? t_last : t + DEPTH);
                fence_order();
                if (t >= t_first && t <= t_last) {                 // (wave-uniform; no load inside)
                    const bool rowok = t >= 0 && t < H, ok = xin && rowok;
                    const v2f vin = f2(cur.v), pv = f2(cur.pv), mi = LMQ ? f2(cur.mi) : splat(1.f);
                    v2f rk = vin;
                    if (UPD && !up.first) rk = fma2(-alpha, f2(cur.av), rk);
                    v2f v0;
                    if (PUPD) v0 = vin + beta * pv;
                    else if (UPD) { const v2f zk = LMQ ? mi * rk : rk; v0 = zk + beta * pv; }      // (LMQ: p_k = M^-1 r_k + beta p_{k-1})
                    else if (MODEL) v0 = fma2(alpha, pv, vin);                                      // delta + alpha p
                    else v0 = vin;
                    v0 = sel(ok, v0, Z2);
                    const bool mine = t >= ya && t < yb;
                    if (PUPD && mine && xout) bst2(RS_PO, vo, (unsigned)t * rowb, v0);
                    if (MODEL && mine && xout) {
                        bst2(RS_OUT, vo, (unsigned)t * rowb, v0);                                  // (out = the updated delta: ANOTHER plane than the one the neighbouring segments' halo rows still read)
                        if (md.X) {      // gauss_newton.t:915-920 (savePreviousUnknowns) + :901-906 (PCGLinearUpdate: X = X + delta, k_linear_update's expression); nobody else reads X here
                            const rsrc_t RS_X = make_rsrc(md.X), RS_PX = make_rsrc(md.prevX);
                            const v2f xo = f2(bld2(RS_X, vo, (unsigned)t * rowb));
                            bst2(RS_PX, vo, (unsigned)t * rowb, xo); bst2(RS_X, vo, (unsigned)t * rowb, xo + v0);
                        }
                    }
                    if (UPD) {
                        Rk[k0] = rk;
                        const v2f dl = f2(cur.dl);
                        const v2f dk = up.first ? dl : fma2(alpha, pv, dl);
                        if (LMQ) { Dk[k0] = dk; Mk[k0] = mi; }
                        // own rows -- or a GHOST row of a slab (a row of the local image outside [ra, rb): the segment next to it keeps its r and p current; its A p comes
                        // with the exchange, its delta is never read)
                        const bool ghost_row = rowok && (t < g.ra || t >= g.rb);
                        if ((mine || ghost_row) && xout) {
                            const unsigned ro = (unsigned)t * rowb;
                            bst2(RS_RO, vo, ro, rk); bst2(RS_PO, vo, ro, v0);
                            if (mine && !up.first && up.delta) bst2(RS_DL, vo, ro, dk);
                        }
                    }
                    const unsigned f0 = ok ? ((cur.fw.x & 0xffu) | ((cur.fw.y & 0xffu) << 8)) : 0u;
                    const v2f wx = cm.wg * v2f{ (float)((cur.fw.x >> 8) & 0xffu), (float)((cur.fw.y >> 8) & 0xffu) };
                    const v2f wy = cm.wg * v2f{ (float)((cur.fw.x >> 16) & 0xffu), (float)((cur.fw.y >> 16) & 0xffu) };
                    const v2f gx = f2(cur.gx), gy = f2(cur.gy), gz = f2(cur.gz);
                    const v2f v1 = Vv[k1], v2 = Vv[k2];
                    Cy[k0] = coef1(cm, t + g.yoff);
                    // lane exchanges (every lane active here)
                    const v2f vl0 = nbL(v0), vl1 = nbL(v1), vr1 = nbR(v1);
                    const v2f dB0 = sel(ok, INIT ? f2(cur.bi) : gx * v0 + gy * vl0 + gz * v1, Z2);
                    const v2f dBr = nbR(dB0);
                    const M2 wn0 = { ok && (wx.x != 0.0f || wy.x != 0.0f), ok && (wx.y != 0.0f || wy.y != 0.0f) };
                    const v2f Uh0 = sel(wn0, wx * (wx * (dB0 - dBr)), Z2);
                    const v2f Uv1 = sel(Wn[k1], Wy[k1] * (Wy[k1] * (dB[k1] - dB0)), Z2);
                    v2f R1[3];
                    {
                        const float cy0 = Cy[k0], cy1 = Cy[k1], cy2 = Cy[k2];
                        const M2 f2b = bit(Fl[k1], 2u);
                        R1[0] = sel(f2b, cm.ws * (4.0f * (cxc * v1) - cxm * vl1 - cxc * v2 - cxp * vr1 - cxc * v0), Z2);
                        R1[1] = sel(f2b, cm.ws * (4.0f * (cy1 * v1) - cy1 * vl1 - cy2 * v2 - cy1 * vr1 - cy0 * v0), Z2);
                        R1[2] = sel(f2b, cm.ws * (4.0f * v1 - vl1 - v2 - vr1 - v0), Z2);
                    }
                    v2f T1 = Uh[k1] + Uv1;
                    T1 -= nbL(Uh[k1]);
                    T1 -= Uv[k2];
                    const v2f T2 = Tt[k2];
                    // G.y(i+ex) T(i+ex): the product as the right neighbour forms it (same operands, same bits), one exchange instead of two
                    const v2f gT2r = nbR(Gy[k2] * T2);
                    v2f Rl[3], Rq[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { Rl[c] = nbL(Rr[k2][c]); Rq[c] = nbR(Rr[k2][c]); }
                    const int y = t - 2;
                    v2f dg_gyR = Z2, dg_hR = Z2, dg_kR = Z2, dg_hL = Z2, dg_hDL = Z2, dg_kUR = Z2; unsigned dg_fL = 0u, dg_fR = 0u;
                    if (DIAG) {     // lane exchanges of the diagonal (every lane active): rows y = t-2 (k2), y+1 = t-1 (k1), y-1 = t-3 (k0, not yet overwritten)
                        dg_gyR = nbR(Gy[k2]); dg_hR = nbR(Wx[k2]); dg_kR = nbR(sel(Wn[k2], Wy[k2], Z2)); dg_hL = nbL(Wx[k2]);
                        dg_hDL = nbL(Wx[k1]); dg_kUR = nbR(sel(Wn[k0], Wy[k0], Z2));
                        dg_fL = fl_left(Fl[k2]); dg_fR = fl_right(Fl[k2]);
                    }
                    if (y >= ya && xout) {
                        const unsigned ro = (unsigned)y * rowb;
                        v2f dgv = Z2;
                        if (DIAG) {
                            const v2f gxx = Gx[k2], gzD = Gz[k1];
                            const v2f h0 = Wx[k2], k0w = sel(Wn[k2], Wy[k2], Z2), hD = Wx[k1], kD = sel(Wn[k1], Wy[k1], Z2), kU = sel(Wn[k0], Wy[k0], Z2);
                            v2f d = sel(bit(Fl[k2], 1u), splat(cm.wp * cm.wp), Z2);
                            v2f ch, cv;
                            ch = (gxx - dg_gyR) * h0; cv = (gxx - gzD) * k0w; d += ch * ch + cv * cv;       // q = i
                            ch = dg_gyR * dg_hR;      cv = dg_gyR * dg_kR;   d += ch * ch + cv * cv;       // q = i+ex
                            ch = gzD * hD;            cv = gzD * kD;         d += ch * ch + cv * cv;       // q = i+ey
                            ch = -gxx * dg_hL;        cv = Z2;               d += ch * ch + cv * cv;       // q = i-ex
                            ch = -gzD * dg_hDL;                              d += ch * ch + cv * cv;       // q = i-ex+ey
                            ch = Z2; cv = -gxx * kU;                         d += ch * ch + cv * cv;       // q = i-ey
                            cv = -dg_gyR * dg_kUR;                           d += ch * ch + cv * cv;       // q = i+ex-ey
                            v2f cc = Z2;
                            { const v2f k0c = cm.ws * cxc; const float k1c = cm.ws * Cy[k2], k2c = cm.ws * 1.0f; cc += k0c * k0c; cc += splat(k1c * k1c); cc += splat(k2c * k2c); }
                            v2f cnt = sel(bit(Fl[k2], 2u), splat(16.0f), Z2);
                            cnt = sel(bit(dg_fL, 2u), cnt + 1.0f, cnt);
                            cnt = sel(bit(Fl[k0], 2u), cnt + 1.0f, cnt);
                            cnt = sel(bit(dg_fR, 2u), cnt + 1.0f, cnt);
                            cnt = sel(bit(Fl[k1], 2u), cnt + 1.0f, cnt);
                            dgv = d + cnt * cc;
                            if (diag) { const rsrc_t RS_DG = make_rsrc(diag); bst2(RS_DG, vo, ro, dgv); }
                        }
                        const v2f vc = v2, ct = f2(cur.ct);
                        v2f s = Z2;
                        s = sel(bit(Fl[k2], 1u), s + cm.wp * (cm.wp * (INIT ? vc - ct : vc)), s);
                        s += Gx[k2] * T2;
                        s = sel(xp1, s + gT2r, s);
                        if (y + 1 < H) s += Gz[k1] * T1;
                        {
                            v2f lap;
                            lap = 4.0f * Rr[k2][0] - Rl[0] - Rr[k0][0] - Rq[0] - R1[0]; s += cm.ws * (cxc * lap);      // (slot k0 still holds row t-3)
                            lap = 4.0f * Rr[k2][1] - Rl[1] - Rr[k0][1] - Rq[1] - R1[1]; s += cm.ws * (Cy[k2] * lap);
                            lap = 4.0f * Rr[k2][2] - Rl[2] - Rr[k0][2] - Rq[2] - R1[2]; s += cm.ws * (1.0f * lap);
                        }
                        if (INIT) {
                            const v2f r = -s;
                            bst2(RS_OUT, vo, ro, r);
                            { const rsrc_t RS_PP = make_rsrc(p_prev), RS_D0 = make_rsrc(delta); bst2(RS_PP, vo, ro, Z2); bst2(RS_D0, vo, ro, Z2); }
                            if (FIN) {            // PCGFinalizeDiagonal (k_lm_finalize's expressions per element; no preconditioner in this energy: SSq = 1 at the first step)
                                const rsrc_t RS_SS = make_rsrc(fd.SSq), RS_CC = make_rsrc(fd.CtC), RS_PR = make_rsrc(fd.pre), RS_B = make_rsrc(fd.b), RS_Z = make_rsrc(z);
                                // SSq: this energy has no preconditioner, so PCGSaveSSq stores 1 at the first step and every later step reads 1 (k_lm_finalize: `else s4 = 1`):
                                // the plane is written once for whoever else reads it and never loaded; (1 / SSq) / radius = 1 / radius exactly, one division per launch instead
                                // of two per pixel
                                if (fd.save_ssq) bst2(RS_SS, vo, ro, splat(1.0f));
                                const float inv_radius = 1.0f / fd.radius;
                                v2f cc2, mm, zz;
                                {
                                    const float dd[2] = { dgv.x, dgv.y }, rr[2] = { r.x, r.y };
                                    float c_[2], m_[2], z_[2];
#pragma unroll
                                    for (int q = 0; q < 2; ++q) {
                                        const float unclamped = dd[q] * inv_radius;
                                        const float cmq = (1.0f / 1.0f) / fd.radius;
                                        const float c = fminf(fmaxf(unclamped, fd.min_lm * cmq), fd.max_lm * cmq);
                                        c_[q] = c; m_[q] = 1.0f / (c + fd.radius * unclamped); z_[q] = m_[q] * rr[q];
                                    }
                                    cc2 = v2f{ c_[0], c_[1] }; mm = v2f{ m_[0], m_[1] }; zz = v2f{ z_[0], z_[1] };
                                }
                                bst2(RS_CC, vo, ro, cc2); bst2(RS_PR, vo, ro, mm); bst2(RS_B, vo, ro, r); bst2(RS_Z, vo, ro, zz);
                                acc += r * zz;
                            } else {
                                const rsrc_t RS_Z = make_rsrc(z); bst2(RS_Z, vo, ro, r);
                                acc += r * r;
                            }
                        } else if (MODEL) {
                            acc += vc * s; acc2 += vc * f2(cur.bb);
                        } else {
                            if (CTC) s += ct * vc;
                            bst2(RS_OUT, vo, ro, s); acc += vc * s;
                            if (SUMS) {
                                const v2f rr = UPD ? Rk[k2] : f2(cur.rs), mk = LMQ ? Mk[k2] : splat(1.0f);
                                sm.add(mk.x, rr.x, s.x); sm.add(mk.y, rr.y, s.y);
                                if (LMQ) { const v2f bb = f2(cur.bb); sq.add(Dk[k2].x, rr.x, bb.x, vc.x, s.x); sq.add(Dk[k2].y, rr.y, bb.y, vc.y, s.y); }
                            }
                        }
                    }
                    Vv[k0] = v0; Fl[k0] = f0; Wn[k0] = wn0; Wy[k0] = wy; if (DIAG) Wx[k0] = sel(ok, wx, Z2); dB[k0] = dB0; Uh[k0] = Uh0; Uv[k1] = Uv1; Tt[k1] = T1;
                    Gx[k0] = gx; Gy[k0] = gy; Gz[k0] = gz;
#pragma unroll
                    for (int c = 0; c < 3; ++c) Rr[k1][c] = R1[c];
                }
            }
        }
    }
    const float accf = acc.x + acc.y;
    if (MODEL) { float vv[2] = { accf, acc2.x + acc2.y }; float* __restrict__ const oo[2] = { part_out, md.db_out }; block_store_partials<2>(vv, oo, red); }
    else if (LMQ) block_finish_sums_lm(accf, sm, sq, part_out, s3_out, fin, lmf, red, redd);
    else if (SUMS) block_finish_sums(accf, sm, part_out, s3_out, fin, red, redd);
    else block_store_partial(accf, part_out, red);
}

// ------------------------------------------------------------------------------------------ precompute on pixel pairs, closed-form partials
// BI(c) = B(n^(X(c), X(c-ex), X(c-ey))) - I(c) and its three partials (shape_from_shading.t:40-80).  With c, l, u the three depths,
//   n = ( u (c-l) / f_y,  l (c-u) / f_x,  n_x a_x + n_y a_y - l u / (f_x f_y) ),   a_x = (u_x - x) / f_x, a_y = (u_y - y) / f_y,   n^ = n / |n|,
//   B = L1 + L2 n^_y + L3 n^_z + L4 n^_x + L5 n^_x n^_y + L6 n^_y n^_z + L7 (-n^_x^2 - n^_y^2 + 2 n^_z^2) + L8 n^_z n^_x + L9 (n^_x^2 - n^_y^2)
// the chain rule collapses to ONE 3-vector:  h = |n|^-1 (g - n^ (g . n^)),  g = grad_n^ B;  dB/dq = h . dn/dq  for q in {c, l, u}, and dn/dq are the products above with
// one factor removed.  Values go through eval_BI_vals' operations in its order (energy_sfs.hip); the partials agree with its forward-mode duals to rounding.
struct BIv { v2f b, dc, dl, du; };
__device__ __forceinline__ BIv eval_BI_pair(const PCam& cm, v2f Dl, v2f Dc, v2f Du, v2f c, v2f l, v2f u, v2f Ic, v2f Il, v2f Iu, v2f ax, float ay)
{
    const v2f Z2 = { 0.f, 0.f };
    const M2 on = { Dl.x > 0.0f && Dc.x > 0.0f && Du.x > 0.0f, Dl.y > 0.0f && Dc.y > 0.0f && Du.y > 0.0f };
    const float ify = 1.0f / cm.fy, ifx = 1.0f / cm.fx, kxy = 1.0f / (cm.fx * cm.fy);
    const v2f cl = c - l, cu = c - u;
    const v2f nx = (u * cl) * ify;
    const v2f ny = (l * cu) * ifx;
    const v2f nz = (nx * ax + ny * ay) - (l * u) * kxy;
    const v2f sq = nx * nx + ny * ny + nz * nz;
    const M2 pos = { sq.x > 0.0f, sq.y > 0.0f };
    const v2f inv = sel(pos, v2f{ 1.0f / sqrtf(sq.x), 1.0f / sqrtf(sq.y) }, splat(1.0f));
    const v2f n0 = inv * nx, n1 = inv * ny, n2 = inv * nz;
    const float* L = cm.L;
    v2f B = splat(L[0]);
    B = B + n1 * L[1]; B = B + n2 * L[2]; B = B + n0 * L[3];
    B = B + (n0 * n1) * L[4]; B = B + (n1 * n2) * L[5];
    B = B + (((n0 * n0) * -1.0f - n1 * n1) + (n2 * n2) * 2.0f) * L[6];
    B = B + (n2 * n0) * L[7]; B = B + (n0 * n0 - n1 * n1) * L[8];
    const v2f I = Ic * 0.5f + 0.25f * (Il + Iu);
    // gradient of the SH polynomial in n^
    const v2f g0 = L[3] + n1 * L[4] - (2.0f * L[6]) * n0 + n2 * L[7] + (2.0f * L[8]) * n0;
    const v2f g1 = L[1] + n0 * L[4] + n2 * L[5] - (2.0f * L[6]) * n1 - (2.0f * L[8]) * n1;
    const v2f g2 = L[2] + n1 * L[5] + (4.0f * L[6]) * n2 + n0 * L[7];
    const v2f gn = g0 * n0 + g1 * n1 + g2 * n2;
    const v2f h0 = sel(pos, inv * (g0 - n0 * gn), g0), h1 = sel(pos, inv * (g1 - n1 * gn), g1), h2 = sel(pos, inv * (g2 - n2 * gn), g2);
    const v2f A = h0 + h2 * ax, Bq = h1 + h2 * ay, Cq = h2 * kxy;
    const v2f uf = u * ify, lf = l * ifx;
    BIv r;
    r.b  = sel(on, B - I, Z2);
    r.dc = sel(on, A * uf + Bq * lf, Z2);
    r.dl = sel(on, Bq * (cu * ifx) - A * uf - Cq * u, Z2);
    r.du = sel(on, A * (cl * ify) - Bq * lf - Cq * l, Z2);
    return r;
}

struct PpRaw { u32x2 x, d, im; unsigned mr, mc; };

// COST: computeCost (k_cost's terms in k_cost's order) of the rows [c0, c1) rides along, one row behind the planes
template <bool COST>
__global__ __launch_bounds__(PM_NT, 2) void k_pprecompute(PmGeo g, int Hg, PCam cm, const float* __restrict__ X, const float* __restrict__ D, const float* __restrict__ Im,
                                                          const unsigned char* __restrict__ mR, const unsigned char* __restrict__ mC,
                                                          float* __restrict__ Gp, unsigned* __restrict__ Fw, float* __restrict__ cost_out, int c0, int c1)
{
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = g.W, H = g.H;
    int strip, ya, yb;
    pm_place(g, wave, strip, ya, yb);
    v2f acc = { 0.f, 0.f };
    if (ya < yb) {
        const v2f Z2 = { 0.f, 0.f };
        const int x0 = strip * PM_USE - 2 + 2 * lane;
        const bool xin = x0 >= 0 && x0 < W;
        const bool xout = xin && lane >= 1 && lane <= 62;
        const int xcl = x0 < 0 ? 0 : x0 > W - 2 ? W - 2 : x0;
        const unsigned vo = (unsigned)xcl * 4u, vob = (unsigned)xcl;
        const unsigned rowb = (unsigned)W * 4u, planeb = (unsigned)W * (unsigned)H * 4u;
        const rsrc_t RS_X = make_rsrc(X), RS_D = make_rsrc(D), RS_I = make_rsrc(Im), RS_MR = make_rsrc(mR), RS_MC = make_rsrc(mC), RS_G = make_rsrc(Gp), RS_F = make_rsrc(Fw);
        const v2f ax = { (cm.ux - (float)x0) / cm.fx, (cm.ux - (float)(x0 + 1)) / cm.fx };
        const v2f cxc = { coef0(cm, x0), coef0(cm, x0 + 1) }, cxm = { coef0(cm, x0 - 1), cxc.x }, cxp = { cxc.y, coef0(cm, x0 + 2) };
        const M2 xinner = { x0 >= 1 && x0 + 1 < W, x0 + 2 < W };      // x >= 1 && x + 1 < W per pixel (x0 + 1 >= 1 always)
        const int t_first = ya - 1, t_last = COST ? yb + 1 : yb;
        auto issue = [&](PpRaw& s, int t) {
            const unsigned tc = (unsigned)(t < 0 ? 0 : t > H - 1 ? H - 1 : t), row = tc * rowb;
            s.x = bld2(RS_X, vo, row); s.d = bld2(RS_D, vo, row); s.im = bld2(RS_I, vo, row);
            // the pair's two mask bytes: one aligned 16-bit load each (W and x0 are even)
            s.mr = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(RS_MR, vob, tc * (unsigned)W, 0);
            s.mc = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(RS_MC, vob, tc * (unsigned)W, 0);
        };
        v2f Xr[3] = { Z2, Z2, Z2 }, Dr[3] = { Z2, Z2, Z2 }, Ir[3] = { Z2, Z2, Z2 };
        unsigned Mr[3] = { 0u, 0u, 0u }, Mc[3] = { 0u, 0u, 0u };
        v2f Bv[3] = { Z2, Z2, Z2 }, Wxr[3] = { Z2, Z2, Z2 }, Wyr[3] = { Z2, Z2, Z2 };      // (COST) BI, the row weights and the flags of the rows
        unsigned Fr[3] = { 0u, 0u, 0u };
        PpRaw slot[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) slot[j] = PpRaw{};
        for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = t0 + j;
                const int k0 = j, k1 = (j + 2) % 3, k2 = (j + 1) % 3;          // ring slots of rows t, t-1, t-2
                PpRaw cur = PpRaw{};
                take2u(cur.x, slot[j].x); take2u(cur.d, slot[j].d); take2u(cur.im, slot[j].im); take1(cur.mr, slot[j].mr); take1(cur.mc, slot[j].mc);
                fence_order();
                issue(slot[j], t + 3 > t_last ? t_last : t + 3);
                fence_order();
                if (t >= t_first && t <= t_last) {
                    const bool ok = xin && t >= 0 && t < H;
                    const v2f x3 = Xr[k0];                                   // (COST) row t-3, about to be overwritten
                    Xr[k0] = sel(ok, f2(cur.x), Z2); Dr[k0] = sel(ok, f2(cur.d), Z2); Ir[k0] = sel(ok, f2(cur.im), Z2);
                    Mr[k0] = ok ? cur.mr : 0u; Mc[k0] = ok ? cur.mc : 0u;
                    // the row being written: y = t-1 (slot k1); its upper row t-2 (k2), its lower row t (k0)
                    const v2f xc = Xr[k1], dc = Dr[k1], ic = Ir[k1];
                    const v2f xl = nbL(xc), dl = nbL(dc), il = nbL(ic), xr = nbR(xc), dr = nbR(dc);
                    const int y = t - 1;
                    const bool own = y >= ya && y < yb && xout;
                    const BIv b = eval_BI_pair(cm, dl, dc, Dr[k2], xc, xl, Xr[k2], ic, il, Ir[k2], ax, (cm.uy - (float)(y + g.yoff)) / cm.fy);
                    if (COST) Bv[k1] = b.b;
                    if (own) {
                        const unsigned ro = (unsigned)y * rowb;
                        bst2(RS_G, vo, ro, b.dc); bst2(RS_G, vo, planeb + ro, b.dl); bst2(RS_G, vo, 2u * planeb + ro, b.du); bst2(RS_G, vo, 3u * planeb + ro, b.b);
                        const int yg = y + g.yoff;
                        const bool yinner = yg >= 1 && yg + 1 < Hg;
                        const unsigned mr0 = (yinner && xinner.x) ? Mr[k1] & 0xffu : 0u, mr1 = (yinner && xinner.y) ? (Mr[k1] >> 8) & 0xffu : 0u;
                        const unsigned mc0 = (yinner && xinner.x) ? Mc[k1] & 0xffu : 0u, mc1 = (yinner && xinner.y) ? (Mc[k1] >> 8) & 0xffu : 0u;
                        unsigned fa = dc.x > 0.0f ? 1u : 0u, fb = dc.y > 0.0f ? 1u : 0u;
                        bool va = fa != 0u, vb = fb != 0u;
                        va = va && dl.x > 0.0f && fabsf(xc.x - xl.x) < 0.01f;               // (x-1, y), (x, y-1), (x+1, y), (x, y+1): k_precompute's order
                        vb = vb && dl.y > 0.0f && fabsf(xc.y - xl.y) < 0.01f;
                        va = va && Dr[k2].x > 0.0f && fabsf(xc.x - Xr[k2].x) < 0.01f;
                        vb = vb && Dr[k2].y > 0.0f && fabsf(xc.y - Xr[k2].y) < 0.01f;
                        va = va && dr.x > 0.0f && fabsf(xc.x - xr.x) < 0.01f;
                        vb = vb && dr.y > 0.0f && fabsf(xc.y - xr.y) < 0.01f;
                        va = va && Dr[k0].x > 0.0f && fabsf(xc.x - Xr[k0].x) < 0.01f;
                        vb = vb && Dr[k0].y > 0.0f && fabsf(xc.y - Xr[k0].y) < 0.01f;
                        if (va) fa |= 2u;
                        if (vb) fb |= 2u;
                        bst2u(RS_F, vo, ro, fa | (mr0 << 8) | (mc0 << 16), fb | (mr1 << 8) | (mc1 << 16));
                        if (COST) { Wxr[k1] = cm.wg * v2f{ (float)mr0, (float)mr1 }; Wyr[k1] = cm.wg * v2f{ (float)mc0, (float)mc1 }; Fr[k1] = fa | (fb << 8); }
                    }
                    if (COST) {
                        // cost of row t-2 (slot k2): its planes were formed one step ago, BI of the row below just now
                        const v2f x2 = Xr[k2];
                        const v2f x2l = nbL(x2), x2r = nbR(x2), b2r = nbR(Bv[k2]);
                        const int yc = t - 2;
                        if (yc >= ya && yc < yb && yc >= c0 && yc < c1 && xout) {
                            v2f sacc = Z2;
                            { const v2f e = cm.wp * (x2 - Dr[k2]); sacc = sel(bit(Fr[k2], 1u), sacc + e * e, sacc); }
                            {
                                const M2 wn = { Wxr[k2].x != 0.0f || Wyr[k2].x != 0.0f, Wxr[k2].y != 0.0f || Wyr[k2].y != 0.0f };
                                const v2f b0 = Bv[k2];
                                const v2f eh = Wxr[k2] * (b0 - b2r), ev = Wyr[k2] * (b0 - Bv[k1]);
                                sacc = sel(wn, sacc + (eh * eh + ev * ev), sacc);
                            }
                            {
                                const float cym = coef1(cm, yc - 1 + g.yoff), cyc = coef1(cm, yc + g.yoff), cyp = coef1(cm, yc + 1 + g.yoff);
                                v2f sr = sacc, a;
                                a = 4.0f * (cxc * x2); a -= cxm * x2l; a -= cxc * x3; a -= cxp * x2r; a -= cxc * Xr[k1]; a *= cm.ws; sr += a * a;
                                a = 4.0f * (cyc * x2); a -= cyc * x2l; a -= cym * x3; a -= cyc * x2r; a -= cyp * Xr[k1]; a *= cm.ws; sr += a * a;
                                a = 4.0f * (1.0f * x2); a -= 1.0f * x2l; a -= 1.0f * x3; a -= 1.0f * x2r; a -= 1.0f * Xr[k1]; a *= cm.ws; sr += a * a;
                                sacc = sel(bit(Fr[k2], 2u), sr, sacc);
                            }
                            acc += 0.5f * sacc;
                        }
                    }
                }
            }
        }
    }
    if (COST) block_store_partial(acc.x + acc.y, cost_out, red);
}

PmGeo make_geo(int W, int H, int ra, int rb, int yoff, int R)
{
    PmGeo g; g.W = W; g.H = H; g.ra = ra; g.rb = rb; g.yoff = yoff; g.R = R;
    g.nstrips = (W + PM_USE - 1) / PM_USE;
    const int nseg = (rb - ra + R - 1) / R;
    g.total = g.nstrips * ((nseg + PM_NT / 64 - 1) / (PM_NT / 64));
    return g;
}
long pm_cap(const SfsTune& t, int per_cu) { return t.cap > 0 ? t.cap : (long)thallo_hip_device_cu_count() * per_cu; }
PmGeo pick_geo(int W, int H, int ra, int rb, int yoff, const SfsTune& t, int per_cu_default = PM_WG_PER_CU)
{
    if (t.rows > 0) return make_geo(W, H, ra, rb, yoff, t.rows);
    const int nstrips = (W + PM_USE - 1) / PM_USE, per_cu = t.wgcu > 0 ? t.wgcu : per_cu_default;
    int R = march_rows_per_segment(rb - ra, nstrips, PM_NT / 64, pm_cap(t, per_cu));
    // wide images: when the strip count leaves more than a quarter of the budget's workgroup slots empty the budget grows (energy_sfs.hip pick_ms_geo: the same rule)
    if (R > 0 && t.cap <= 0 && t.wgcu <= 0) {
        const long slots = pm_cap(t, per_cu);
        auto fill_of = [&](int rr) { const long nseg = (rb - ra + rr - 1) / rr, total = (long)nstrips * ((nseg + PM_NT / 64 - 1) / (PM_NT / 64)); return (double)total / (double)(((total + slots - 1) / slots) * slots); };
        double best = fill_of(R);
        for (int m = 2; best < 0.75 && m <= 4; ++m) {
            const int r2 = march_rows_per_segment(rb - ra, nstrips, PM_NT / 64, slots * m);
            if (r2 <= 0) break;
            const double f = fill_of(r2);
            if (f > best + 1e-9) { best = f; R = r2; }
            if (f >= 0.9) break;
        }
    }
    return make_geo(W, H, ra, rb, yoff, R > 0 ? R : rb - ra);
}
inline bool rows_ok(int H, int row0, int row1) { return !(row0 < 0 || row1 > H || row0 >= row1); }

}  // namespace

namespace thallo {

bool sfs_pair_ok(int W, int H, const SfsTune& t)
{
    if (W < 2 || (W & 1) || H < 1) return false;
    if (16.0 * (double)W * (double)H >= 4294967296.0) return false;              // the G buffer (four planes) behind one descriptor
    return march_strips_fit((W + PM_USE - 1) / PM_USE, pm_cap(t, t.wgcu > 0 ? t.wgcu : PM_WG_PER_CU));
}

int sfs_pair_precompute(int W, int H, int ra, int rb, int yoff, int Hg, const float* hp, const float* X, const float* D, const float* Im, const unsigned char* mR, const unsigned char* mC,
                        float* G, float* Fw, float* cost_out, int c0, int c1, const SfsTune& t, thallo_stream_t stream)
{
    if (!rows_ok(H, ra, rb) || !X || !D || !Im || !mR || !mC || !G || !Fw) return -(int)hipErrorInvalidValue;
    if (((uintptr_t)mR | (uintptr_t)mC) & 1) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, ra, rb, yoff, t, 2);      // (two workgroups per CU: tools/sfs_pair_time.py, cold launches at 2048^2: 39.5 us against 43.3 for one, 42.9 for four)
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    if (cost_out) {
        hipLaunchKernelGGL(k_pprecompute<true>, dim3(grid), dim3(PM_NT), 0, (hipStream_t)stream, mg, Hg, cam_of(hp), X, D, Im, mR, mC, G, (unsigned*)Fw, cost_out, c0, c1);
        int e = check_launch(); return e ? e : grid;
    }
    hipLaunchKernelGGL(k_pprecompute<false>, dim3(grid), dim3(PM_NT), 0, (hipStream_t)stream, mg, Hg, cam_of(hp), X, D, Im, mR, mC, G, (unsigned*)Fw, (float*)nullptr, 0, 0);
    return check_launch();
}

#define PM_ARGS_NONE PmPupd{}, PmUpd{}, LmFin{}, PmFin{}, PmModel{}
// Rows of prefetch: three.  Six (two trips ahead) measured 4-6 % slower on every form at 2048^2 and 640 x 480 (tools/sfs_pair_time.py, round 6) and pushes the LM form into
// scratch; the product instantiates depth 3 only (-DTHALLO_SFS_PAIR_DEPTH6: both, for the tool's sweep).
#ifdef THALLO_SFS_PAIR_DEPTH6
#define PM_LAUNCH(SUMS, CTC, INIT, DIAG, PUPD, UPD, LMQ, FIN, MODEL, ...) do { \
    if (t.depth == 6) hipLaunchKernelGGL((k_pmarch<SUMS, CTC, INIT, DIAG, 2, PUPD, UPD, LMQ, FIN, MODEL, 6>), dim3(grid), dim3(PM_NT), 0, (hipStream_t)stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((k_pmarch<SUMS, CTC, INIT, DIAG, 2, PUPD, UPD, LMQ, FIN, MODEL, 3>), dim3(grid), dim3(PM_NT), 0, (hipStream_t)stream, __VA_ARGS__); } while (0)
#else
#define PM_LAUNCH(SUMS, CTC, INIT, DIAG, PUPD, UPD, LMQ, FIN, MODEL, ...) do { (void)t.depth; \
    hipLaunchKernelGGL((k_pmarch<SUMS, CTC, INIT, DIAG, 2, PUPD, UPD, LMQ, FIN, MODEL, 3>), dim3(grid), dim3(PM_NT), 0, (hipStream_t)stream, __VA_ARGS__); } while (0)
#endif

int sfs_pair_init(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* X, const float* D, const float* G, const float* Fw,
                  float* r, float* z, float* p_prev, float* delta, float* diag_out, float* aN_out, const SfsFinDiag& fdh, const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if (!rows_ok(H, row0, row1) || !X || !D || !G || !Fw || !r || !z || !p_prev || !delta || !aN_out) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const PCam cm = cam_of(hp);
    const bool fin = fdh.CtC != nullptr;
    if (fin && (!fdh.SSq || !fdh.pre || !fdh.b || !(fdh.radius > 0.0f))) return -(int)hipErrorInvalidValue;
    const PmFin fd = { fdh.SSq, fdh.CtC, fdh.pre, fdh.b, fdh.radius, fdh.min_lm, fdh.max_lm, fdh.save_ssq };
#define PM_INIT_ARGS mg, cm, G, (const unsigned*)Fw, X, D, r, aN_out, (const float*)nullptr, (double*)nullptr, (const unsigned*)nullptr, FinArgs{}, z, p_prev, delta, diag_out, PmPupd{}, PmUpd{}, LmFin{}, fd, PmModel{}
    if (fin) PM_LAUNCH(false, true, true, true, false, false, false, true, false, PM_INIT_ARGS);
    else if (diag_out) PM_LAUNCH(false, true, true, true, false, false, false, false, false, PM_INIT_ARGS);
    else PM_LAUNCH(false, true, true, false, false, false, false, false, false, PM_INIT_ARGS);
#undef PM_INIT_ARGS
    int e = check_launch(); return e ? e : grid;
}

int sfs_pair_apply(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* p, float* Ap, float* aD_out,
                   const float* r, double* s3_out, const unsigned* gate, thallo_fin_t fin, const float* ctc, const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if (!rows_ok(H, row0, row1) || !G || !Fw || !p || !Ap || !aD_out) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!s3_out || !fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    if (s3_out && !r) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, grid };
    const PCam cm = cam_of(hp);
#define PM_APPLY_ARGS mg, cm, G, (const unsigned*)Fw, p, ctc, Ap, aD_out, r, s3_out, gate, fa, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, PM_ARGS_NONE
    if (s3_out && ctc) PM_LAUNCH(true, true, false, false, false, false, false, false, false, PM_APPLY_ARGS);
    else if (s3_out)   PM_LAUNCH(true, false, false, false, false, false, false, false, false, PM_APPLY_ARGS);
    else if (ctc)      PM_LAUNCH(false, true, false, false, false, false, false, false, false, PM_APPLY_ARGS);
    else               PM_LAUNCH(false, false, false, false, false, false, false, false, false, PM_APPLY_ARGS);
#undef PM_APPLY_ARGS
    int e = check_launch(); return e ? e : grid;
}

int sfs_pair_apply_pupdate(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* z, const float* p_in, float* p_out,
                           const float* ctc, float* Ap, float* aD_out, int first, thallo_sum_t aN_prev, thallo_sum_t bN_prev, const unsigned* gate, const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if (!rows_ok(H, row0, row1) || !G || !Fw || !z || !p_in || !p_out || p_in == p_out || !ctc || !Ap || !aD_out) return -(int)hipErrorInvalidValue;
    if (!first && (!aN_prev.partials || !bN_prev.partials || aN_prev.count < 1 || bN_prev.count < 1)) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const PmPupd pu = { p_in, p_out, aN_prev, bN_prev, first ? 1 : 0 };
    PM_LAUNCH(false, true, false, false, true, false, false, false, false, mg, cam_of(hp), G, (const unsigned*)Fw, z, ctc, Ap, aD_out, (const float*)nullptr, (double*)nullptr, gate, FinArgs{},
              (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, pu, PmUpd{}, LmFin{}, PmFin{}, PmModel{});
    int e = check_launch(); return e ? e : grid;
}

int sfs_pair_iter(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw,
                  const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int first,
                  thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev, const thallo_prev_t* prev, float* aD_out, double* s3_out, thallo_fin_t fin,
                  const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if (!rows_ok(H, row0, row1) || !G || !Fw || !r_in || !r_out || r_in == r_out || !Ap_out || !p_in || !p_out || p_in == p_out || !aD_out || !s3_out) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    PmUpd up = { r_out, first ? r_in : Ap_in, p_in, p_out, delta, aN_prev, aD_prev, bN_prev, first ? 1 : 0, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr };
    FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, grid };
    if (prev) {
        if (!first && (!prev->alphaD_partials || !prev->s12_partials || prev->s12_partials == s3_out || prev->count < 1 || prev->count > THALLO_MAX_PARTIALS || !prev->alphaD_word || !prev->betaN_word))
            return -(int)hipErrorInvalidValue;
        up.aD = thallo_sum_t{ first ? aN_prev.partials : prev->alphaD_partials, first ? 1 : prev->count }; up.bN = aN_prev;
        if (!first) { up.prev_s3 = prev->s12_partials; up.prev_nb = prev->count; up.aD_word = prev->alphaD_word; up.bN_word = prev->betaN_word; }
        fa = FinArgs{ aN_prev, nullptr, nullptr, nullptr, 0, grid };      // partials only: the next launch (or thallo_hip_pcg_scalars_finish behind the loop) finishes
    }
    PM_LAUNCH(true, false, false, false, false, true, false, false, false, mg, cam_of(hp), G, (const unsigned*)Fw, r_in, (const float*)nullptr, Ap_out, aD_out, (const float*)nullptr, s3_out,
              (const unsigned*)nullptr, fa, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, PmPupd{}, up, LmFin{}, PmFin{}, PmModel{});
    int e = check_launch(); return e ? e : grid;
}

int sfs_pair_iter_lm(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw,
                     const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, const float* CtC, const float* b,
                     const float* pre, int first, thallo_sum_t aN_prev, thallo_sum_t aD_prev, thallo_sum_t bN_prev, float* aD_out, double* s3_out, double* q3_out,
                     thallo_fin_t fin, float* lm_state, int k, float q_tol, const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if (!rows_ok(H, row0, row1) || !G || !Fw) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const PmUpd up = { r_out, first ? r_in : Ap_in, p_in, p_out, delta, aN_prev, aD_prev, bN_prev, first ? 1 : 0, 1, b, pre, nullptr, 0, nullptr, nullptr };
    const FinArgs fa{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, grid };
    const LmFin lf{ b, q3_out, lm_state, k, q_tol };
    PM_LAUNCH(true, true, false, false, false, true, true, false, false, mg, cam_of(hp), G, (const unsigned*)Fw, r_in, CtC, Ap_out, aD_out, (const float*)nullptr, s3_out,
              reinterpret_cast<const unsigned*>(lm_state) + 1, fa, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, PmPupd{}, up, lf, PmFin{}, PmModel{});
    int e = check_launch(); return e ? e : grid;
}

int sfs_pair_model_cost(int W, int H, int row0, int row1, int yoff, int Hg, const float* hp, const float* G, const float* Fw, const float* delta, float* delta_out, const float* p_even, const float* p_odd,
                        const float* b, const float* alphaN_words, const float* alphaD_words, int word_stride, const float* lm_state, int L, float* dJJd_out, float* db_out,
                        float* X, float* prevX, const SfsTune& t, thallo_stream_t stream)
{
    (void)Hg;
    if ((X != nullptr) != (prevX != nullptr) || (X && X == prevX)) return -(int)hipErrorInvalidValue;
    if (!rows_ok(H, row0, row1) || row0 != 0 || row1 != H || !G || !Fw || !delta || !delta_out || delta == delta_out || !p_even || !p_odd || !b || !alphaN_words || !alphaD_words || word_stride < 1 || !lm_state || L < 0 ||
        !dJJd_out || !db_out) return -(int)hipErrorInvalidValue;
    const PmGeo mg = pick_geo(W, H, row0, row1, yoff, t);
    const int grid = (mg.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;
    const PmModel md = { p_even, p_odd, b, alphaN_words, alphaD_words, word_stride, lm_state, L, db_out, X, prevX };
    PM_LAUNCH(false, false, false, false, false, false, false, false, true, mg, cam_of(hp), G, (const unsigned*)Fw, delta, (const float*)nullptr, delta_out, dJJd_out, (const float*)nullptr,
              (double*)nullptr, (const unsigned*)nullptr, FinArgs{}, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, PmPupd{}, PmUpd{}, LmFin{}, PmFin{}, md);
    int e = check_launch(); return e ? e : grid;
}

}  // namespace thallo
