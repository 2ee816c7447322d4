// iw_march.hpp -- pieces shared by the wave-marching image_warping kernels: energy_image_warping_march.hip (A p kept as a plane between
// launches; also the multi-GPU slab form) and energy_image_warping_march_rc.hip (A p_{k-1} recomputed from the p_{k-1} rows, no A p plane).
// Both state J^T J p through jtjp_pair() below and both files are built with -ffp-contract=on, so equal inputs give equal bits in either kernel.
#pragma once
#include "iw_device.hpp"

namespace thallo {

extern int g_march_rows;     // tests / tools: rows per wave segment (0 = automatic) -- the resident kernel's bitwise test forces its own R on the marching kernels
extern int g_march_cap;      // tests: workgroup budget the grid is sized for (0 = CUs x workgroups per CU of the device)

constexpr int MARCH_USE = 124;            // output pixels per wave row (lanes 1..62 x 2)
constexpr int MARCH_NT = 256;             // threads per workgroup: 4 waves = 4 vertically adjacent segments of one strip
constexpr int MARCH_WG_PER_CU = 1;        // grid sizing: ~1 workgroup (4 waves) per CU measured best at 2048^2 (36 rows per wave; 18: +5 %, 48: +12 %)

// value of the lane to the left (lane-1) / right (lane+1); lanes without a source (0 / 63) read 0.  mov_dpp has no tied "old" operand: one v_mov_b32_dpp per exchange
__device__ __forceinline__ float from_left(float v)  { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }   // wave_shr:1
__device__ __forceinline__ float from_right(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }   // wave_shl:1

// Move a prefetch slot into fresh registers with REAL v_mov instructions (early-clobber outputs: never the slot's own registers).
// All arithmetic then works on the copy, the slot's registers die here and its refill -- issued right behind -- loads into the SAME
// registers again: the loop-carried slot needs no copy at the loop latch.  Without this the compiler computes in place in the
// slot registers (v_fmac), keeps the results there as window state, refills into other registers and copies -- after an
// s_waitcnt for the fresh load -- at the latch, which silently turns the prefetch into a blocking load.
// fence_order: nothing moves across, neither in the optimiser (memory clobber) nor in the machine scheduler
__device__ __forceinline__ void fence_order() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void take1(float& d, const float& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
__device__ __forceinline__ void take1(unsigned& d, const unsigned& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
__device__ __forceinline__ void take4(float4& d, const float4& s) { take1(d.x, s.x); take1(d.y, s.y); take1(d.z, s.z); take1(d.w, s.w); }
__device__ __forceinline__ void take2(float2& d, const float2& s) { take1(d.x, s.x); take1(d.y, s.y); }
// ... a register PAIR with one v_mov_b64 (the pair stays a pair: what the packed arithmetic below wants)
__device__ __forceinline__ void take_pair(unsigned long long& d, const unsigned long long& s) { asm volatile("v_mov_b64 %0, %1" : "=&v"(d) : "v"(s)); }

// one neighbour's contribution to (J^T J p)_i on the unit grid; D = 0: x+1, 1: x-1, 2: y+1, 3: y-1.  k_iter's expressions with
// u_i - u_j = -(dx,dy) folded in by hand (the compiler may not drop the products with 0.0f): with g_i = R'(a_i)(u_i-u_j),
// g_j = R'(a_j)(u_j-u_i):  D=0: g_i = (s_i,-c_i), g_j = (-s_j,c_j);  D=1: negated;  D=2: g_i = (c_i,s_i), g_j = (-c_j,-s_j);  D=3: negated.
// BRANCH-FREE: m = 1.0f for a valid neighbour pair, 0.0f otherwise, enters as the multiplicand of the accumulating fma -- fma(1, t, a) rounds like a + t, so
// the sums carry the bits of k_iter's `if (valid) a += t` (round 4: the predicated form cost an exec-mask branch per term, 16 per row, in an issue-bound loop).
template <int D>
__device__ __forceinline__ void nb_term(float m, float ci, float si, float pxi, float pyi, float pai,
                                        float pxj, float pyj, float paj, float cj, float sj, float& ax, float& ay, float& av)
{
    const float gix = D == 0 ? si : D == 1 ? -si : D == 2 ? ci : -ci;
    const float giy = D == 0 ? -ci : D == 1 ? ci : D == 2 ? si : -si;
    const float gjx = D == 0 ? -sj : D == 1 ? sj : D == 2 ? -cj : cj;
    const float gjy = D == 0 ? cj : D == 1 ? -cj : D == 2 ? -sj : sj;
    const float dpx = pxi - pxj, dpy = pyi - pyj;
    const float ex = dpx - gix * pai, ey = dpy - giy * pai;
    const float tx = dpx + ex + gjx * paj;
    const float ty = dpy + ey + gjy * paj;
    const float ta = gix * ex + giy * ey;
    ax = __builtin_fmaf(m, tx, ax); ay = __builtin_fmaf(m, ty, ay); av = __builtin_fmaf(-m, ta, av);
}

// (J^T J p) at the lane's two pixels of a centre row.  P rows carry p as px[2], py[2], pa[2]; G rows the geometry: c[2], s[2] (cos / sin of Angle).  am / ac / an:
// the pixels' ACTIVE bits (flags bit 0) of rows y-1 / y / y+1 as floats 0 / 1; wfit[q] = w_fit^2 where the pixel's fit residual is valid (flags bit 1), else 0.
// The x neighbours across the lane boundary (the left lane's pixel 1, the right lane's pixel 0) come through DPP wave shifts.  Every lane calls it (lanes 0 / 63
// get garbage for their outer pixel); an inactive centre gives exactly 0 (its scale is w_reg^2 * 0, and a fit-valid pixel is active: k_init).
template <class P, class G>
__device__ __forceinline__ void jtjp_pair(const P& pm, const P& pc, const P& pn, const G& gm, const G& gc, const G& gn,
                                          const float (&am)[2], const float (&ac)[2], const float (&an)[2], const float (&wfit)[2], float wr2,
                                          float (&ax)[2], float (&ay)[2], float (&av)[2])
{
    const float Lpx = from_left(pc.px[1]), Lpy = from_left(pc.py[1]), Lpa = from_left(pc.pa[1]), Lc = from_left(gc.c[1]), Ls = from_left(gc.s[1]), La = from_left(ac[1]);
    const float Rpx = from_right(pc.px[0]), Rpy = from_right(pc.py[0]), Rpa = from_right(pc.pa[0]), Rc = from_right(gc.c[0]), Rs = from_right(gc.s[0]), Ra = from_right(ac[0]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        ax[q] = 0.f; ay[q] = 0.f; av[q] = 0.f;
        const float pxi = pc.px[q], pyi = pc.py[q], pai = pc.pa[q], ci = gc.c[q], si = gc.s[q];
        if (q == 0) {
            nb_term<0>(ac[1], ci, si, pxi, pyi, pai, pc.px[1], pc.py[1], pc.pa[1], gc.c[1], gc.s[1], ax[q], ay[q], av[q]);
            nb_term<1>(La, ci, si, pxi, pyi, pai, Lpx, Lpy, Lpa, Lc, Ls, ax[q], ay[q], av[q]);
        } else {
            nb_term<0>(Ra, ci, si, pxi, pyi, pai, Rpx, Rpy, Rpa, Rc, Rs, ax[q], ay[q], av[q]);
            nb_term<1>(ac[0], ci, si, pxi, pyi, pai, pc.px[0], pc.py[0], pc.pa[0], gc.c[0], gc.s[0], ax[q], ay[q], av[q]);
        }
        nb_term<2>(an[q], ci, si, pxi, pyi, pai, pn.px[q], pn.py[q], pn.pa[q], gn.c[q], gn.s[q], ax[q], ay[q], av[q]);
        nb_term<3>(am[q], ci, si, pxi, pyi, pai, pm.px[q], pm.py[q], pm.pa[q], gm.c[q], gm.s[q], ax[q], ay[q], av[q]);
        const float w = wr2 * ac[q];
        ax[q] *= w; ay[q] *= w; av[q] *= w;
        ax[q] = __builtin_fmaf(wfit[q], pxi, ax[q]); ay[q] = __builtin_fmaf(wfit[q], pyi, ay[q]);
    }
}
// ---- the same J^T J p on (x, y) PAIRS (round 5; energy_image_warping_march_rc.hip).  Offset's two channels sit next to each other in every plane (r, p: {x0, y0, x1, y1};
// cs: {c, s} per pixel) and the marching kernels are issue-bound -- a lone wave issues an instruction every ~5 cycles whatever it is, v_pk_fma_f32 included
// (profiles/r04/issue_rates_pk_rate.txt).  Per neighbour d = p_i - p_j, e = d - g_i a_i, t = d + e + g_j a_j and a_xy = fma(m, t, a_xy) are ONE packed instruction
// each; t_a = g_i . e and its accumulation stay scalar.  The x-direction terms need (s, -c): kept per pixel next to (c, s) (`gx`, formed once when a row enters
// the rings) -- D = 0: g_i = gx_i, g_j = -gx_j; D = 1: negated; D = 2: g_i = cs_i, g_j = -cs_j; D = 3: negated.  Every component goes through the operations and
// roundings of nb_term above (the same expression shapes under -ffp-contract=on, the same explicit fmas): BIT-identical results, 8 instead of 13 instructions.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(float m, v2f t, v2f a) { return __builtin_elementwise_fma(v2f{ m, m }, t, a); }
template <int D>
__device__ __forceinline__ void nb_term_xy(float m, v2f csi, v2f gxi, v2f pi, float pai, v2f pj, float paj, v2f csj, v2f gxj, v2f& axy, float& av)
{
    const v2f gi = D == 0 ? gxi : D == 1 ? -gxi : D == 2 ? csi : -csi;
    const v2f gj = D == 0 ? -gxj : D == 1 ? gxj : D == 2 ? -csj : csj;
    const v2f dp = pi - pj;
    const v2f e = dp - gi * pai;
    const v2f t = dp + e + gj * paj;
    const float ta = gi.x * e.x + gi.y * e.y;
    axy = fma2(m, t, axy); av = __builtin_fmaf(-m, ta, av);
}
// rows: P { v2f xy[2]; float pa[2]; }  G { v2f cs[2], gx[2]; float a[2]; }
template <class P, class G>
__device__ __forceinline__ void jtjp_pair_xy(const P& pm, const P& pc, const P& pn, const G& gm, const G& gc, const G& gn,
                                             const float (&am)[2], const float (&ac)[2], const float (&an)[2], const float (&wfit)[2], float wr2,
                                             v2f (&axy)[2], float (&av)[2])
{
    const v2f Lp = { from_left(pc.xy[1].x), from_left(pc.xy[1].y) }, Lcs = { from_left(gc.cs[1].x), from_left(gc.cs[1].y) }, Lgx = { Lcs.y, -Lcs.x };
    const float Lpa = from_left(pc.pa[1]), La = from_left(ac[1]);
    const v2f Rp = { from_right(pc.xy[0].x), from_right(pc.xy[0].y) }, Rcs = { from_right(gc.cs[0].x), from_right(gc.cs[0].y) }, Rgx = { Rcs.y, -Rcs.x };
    const float Rpa = from_right(pc.pa[0]), Ra = from_right(ac[0]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        v2f a2 = { 0.f, 0.f }; float a_v = 0.f;
        const v2f pi = pc.xy[q], csi = gc.cs[q], gxi = gc.gx[q]; const float pai = pc.pa[q];
        if (q == 0) {
            nb_term_xy<0>(ac[1], csi, gxi, pi, pai, pc.xy[1], pc.pa[1], gc.cs[1], gc.gx[1], a2, a_v);
            nb_term_xy<1>(La, csi, gxi, pi, pai, Lp, Lpa, Lcs, Lgx, a2, a_v);
        } else {
            nb_term_xy<0>(Ra, csi, gxi, pi, pai, Rp, Rpa, Rcs, Rgx, a2, a_v);
            nb_term_xy<1>(ac[0], csi, gxi, pi, pai, pc.xy[0], pc.pa[0], gc.cs[0], gc.gx[0], a2, a_v);
        }
        nb_term_xy<2>(an[q], csi, gxi, pi, pai, pn.xy[q], pn.pa[q], gn.cs[q], gn.gx[q], a2, a_v);
        nb_term_xy<3>(am[q], csi, gxi, pi, pai, pm.xy[q], pm.pa[q], gm.cs[q], gm.gx[q], a2, a_v);
        const float w = wr2 * ac[q];
        a2 *= w; a_v *= w;
        a2 = fma2(wfit[q], pi, a2);
        axy[q] = a2; av[q] = a_v;
    }
}

// the masks of one row's flags pair (bits 0-7 pixel 0, 8-15 pixel 1)
__device__ __forceinline__ void flag_masks(unsigned f, float wf2, float (&a)[2], float (&wfit)[2])
{
    a[0] = (float)(f & 1u); a[1] = (float)((f >> 8) & 1u);
    wfit[0] = (f & 2u) ? wf2 : 0.f; wfit[1] = (f & 0x200u) ? wf2 : 0.f;
}

// the iteration's sums over one pixel: float alphaD term; N = sum r.M^-1.r, S1 = sum r.M^-1.Ap, S2 = sum Ap.M^-1.Ap as exact products of the float
// data accumulated in double (both Offset channels share M^-1: 5 double operations per sum)
__device__ __forceinline__ void iter_sums_pixel(float pxi, float pyi, float pai, float ax, float ay, float av, float rx, float ry, float ra, float mo, float ma,
                                                float& acc, double& s0, double& s1, double& s2)
{
    acc += pxi * ax + pyi * ay + pai * av;
    const double dmo = mo, dma = ma, drx = rx, dry = ry, dra = ra, dax = ax, day = ay, daa = av;
    s0 = __builtin_fma(dmo, __builtin_fma(dry, dry, drx * drx), __builtin_fma(dma, dra * dra, s0));
    s1 = __builtin_fma(dmo, __builtin_fma(dry, day, drx * dax), __builtin_fma(dma, dra * daa, s1));
    s2 = __builtin_fma(dmo, __builtin_fma(day, day, dax * dax), __builtin_fma(dma, daa * daa, s2));
}

// ... the same with a 0 / 1 multiplicand instead of a branch (msum = 1: fma(1, t, acc) rounds like acc + t; the caller passes M^-1 = 0 where msum = 0, so the
// double sums add exact zeros there)
__device__ __forceinline__ void iter_sums_pixel_masked(float msum, float pxi, float pyi, float pai, float ax, float ay, float av, float rx, float ry, float ra, float mo, float ma,
                                                       float& acc, double& s0, double& s1, double& s2)
{
    const float t = pxi * ax + pyi * ay + pai * av;
    acc = __builtin_fmaf(msum, t, acc);
    const double dmo = mo, dma = ma, drx = rx, dry = ry, dra = ra, dax = ax, day = ay, daa = av;
    s0 = __builtin_fma(dmo, __builtin_fma(dry, dry, drx * drx), __builtin_fma(dma, dra * dra, s0));
    s1 = __builtin_fma(dmo, __builtin_fma(dry, day, drx * dax), __builtin_fma(dma, dra * daa, s1));
    s2 = __builtin_fma(dmo, __builtin_fma(day, day, dax * dax), __builtin_fma(dma, daa * daa, s2));
}

struct MarchGeo { int W, H, row0, row1, R, nstrips, nwgrow, total; };

// XCD-aware placement: workgroups b and b+8 share an XCD (MI355X_MICROARCH.md), group b%8 owns a contiguous range of (band of 4 segments, strip) ids,
// x-adjacent strips first: x-halo columns and y-halo rows are re-read from the same L2.  Returns the wave's strip and its rows [ya, yb) (empty: no work).
__device__ __forceinline__ void march_place(const MarchGeo& g, int wave, int& strip, int& ya, int& yb)
{
    strip = 0; ya = 0; yb = 0;
    const int G = (gridDim.x % 8) == 0 ? 8 : 1;
    const int grp = blockIdx.x % G, l = blockIdx.x / G;
    const long lo = (long)g.total * grp / G, hi = (long)g.total * (grp + 1) / G;
    const long id = lo + l;
    if (id < hi) {
        strip = (int)(id % g.nstrips);
        const int seg = (int)(id / g.nstrips) * (MARCH_NT / 64) + wave;
        ya = g.row0 + seg * g.R; yb = ya + g.R;
        if (yb > g.row1) yb = g.row1;
        if (ya > g.row1) ya = g.row1;
    }
}

inline MarchGeo make_march_geo(int W, int H, int row0, int row1, int R)
{
    MarchGeo g; g.W = W; g.H = H; g.row0 = row0; g.row1 = row1; g.R = R;
    g.nstrips = (W + MARCH_USE - 1) / MARCH_USE;
    const int nseg = (row1 - row0 + R - 1) / R;
    g.nwgrow = (nseg + MARCH_NT / 64 - 1) / (MARCH_NT / 64);
    g.total = g.nstrips * g.nwgrow;
    return g;
}

// rows per wave segment: every workgroup resident at once, about one workgroup of 4 waves per CU, and at most 1024 workgroups (partial slots).  0: the
// image is wider than that many strips (W > ~31.7k pixels on a 256-CU device) -- the caller runs the tile kernel instead.
// Wide images: the grid is (strips) x (bands of 4 segments), so with ONE workgroup per CU as the budget a width whose strip count does not divide the CU
// count leaves CUs without work -- 16384 pixels: 133 strips x 1 band = 133 of 256 CUs.  Measured (round 3): 16384 x 2048 920 -> 697 us per launch with the
// budget grown until the grid fills its last round of workgroups; 8192 x 2048 (67 x 3 = 201 of 256) gains nothing from 469 workgroups, so a fill of 75 %
// counts as full.  The budget grows to 2, 3, 4 workgroups per CU (at most THALLO_MAX_PARTIALS): once it grows, up to the smallest multiple with a fill of
// 90 %, else the best.  Images for which one workgroup per CU already fills the chip -- every size the kernels are compared at bit for bit -- keep their rows.
inline long march_cap(int occ) { return g_march_cap > 0 ? g_march_cap : (long)thallo_hip_device_cu_count() * occ; }
inline int march_pick_rows(int W, int rows, int occ = MARCH_WG_PER_CU)
{
    if (g_march_rows > 0) return g_march_rows;
    const int nstrips = (W + MARCH_USE - 1) / MARCH_USE, wpw = MARCH_NT / 64;
    if (g_march_cap > 0) return march_rows_per_segment(rows, nstrips, wpw, march_cap(occ));       // (a forced budget -- tests, tools -- is taken as it is)
    const long cus = march_cap(1);
    int best_R = 0; double best_fill = -1.0;
    for (int m = occ; m <= 4; ++m) {
        const int R = march_rows_per_segment(rows, nstrips, wpw, cus * m);
        if (R <= 0) break;                      // more strips than one workgroup per CU has slots: thallo_hip_iw_march_rows() says 0, the tile kernel runs
        const long nseg = (rows + R - 1) / R, total = (long)nstrips * ((nseg + wpw - 1) / wpw);
        const double fill = (double)total / (double)(((total + cus - 1) / cus) * cus);
        if (best_R == 0 || fill > best_fill + 1e-9) { best_R = R; best_fill = fill; }
        if (fill >= (m == occ ? 0.75 : 0.9)) break;
    }
    return best_R;
}

constexpr int MARCH_NTM = 5;     // product cache policy (tools/march_probe.py sweeps, profiles/r02): delta and the r (/ A p) stores non-temporal
                                 // bits: 1 delta, 2 r/Ap loads, 4 r/Ap stores, 8 p loads, 16 p stores, 32 cs/flags

}  // namespace thallo
