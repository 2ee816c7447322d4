// energy_image_warping_march.hip -- the one-kernel PCG iteration of image_warping as a barrier-free "marching" stencil.
//
// Same mathematics, arguments and outputs as k_iter (energy_image_warping.hip; replaces PCGStep1 + PCGStep2 + PCGStep3 of
// gauss_newton.t:734-752,801-843,889-899 in ONE pass), for the unit-pixel-grid UrShape (the GRID path).  What differs is the
// shape of the computation on the chip:
//
//   * a WAVE (not a workgroup) owns a column strip of 128 pixels (lane l: the two x-adjacent pixels x0+2l, x0+2l+1) and marches down
//     R rows of it.  Lanes 1..62 produce output (124 pixels), lanes 0 and 63 carry the x halo (strips overlap by 4 pixels);
//   * x neighbours come from the neighbouring lane through DPP wave shifts (v_mov_b32_dpp wave_shr / wave_shl), y neighbours from the
//     lane's own registers: it keeps a window of three published rows (p_k, cos, sin, flags of rows y-1, y, y+1).  No LDS tile, no
//     barrier in the loop: every wave is its own software pipeline and the memory system sees ~2,000 desynchronised streams;
//   * the raw rows (r, Ap, p, cs, flags [, delta, p_{k-2}]) are prefetched D rows ahead into registers; each plane moves in 16-byte
//     (Offset part: 2 pixels x float2) or 8-byte (Angle part) accesses;
//   * M^-1 = guardedInvert(diag) is a function of the 5-bit flags value only (iw_device.hpp): a 32-entry table in LDS built once per
//     workgroup replaces two divisions + two square roots per pixel.
//   * the y halo (rows ya-1 and yb of a segment) is published redundantly by the wave; with R = 16 that is 12.5 % more L2 reads
//     (the tile kernel's 64x16 tiles: 16 %), no extra HBM traffic (neighbouring segments run at the same time on the same XCD).
//
// Bytes per pixel as for k_iter: read r 12, Ap 12, p 12, cs 8, flags 1; write r 12, p 12, Ap 12 = 81 (+ 18 deferred delta on average).
#include "iw_device.hpp"

namespace thallo {
int g_march_rows = 0;      // tests / tools: rows per wave segment (0 = automatic) -- the resident kernel's bitwise test forces its own R on this kernel
int g_march_cap = 0;       // tests: workgroup budget the grid is sized for (0 = CUs x workgroups per CU of the device)
#ifdef THALLO_MARCH_SWEEP  // tools/march_probe.py only (make VARIANT=sweep): the product has neither the knobs nor the extra instantiations
int g_march_depth = 2;     // prefetch depth in rows (1, 2 or 3)
int g_march_nt = 5;        // non-temporal bits as for k_iter: 1 delta, 2 r/Ap loads, 4 r/Ap stores, 8 p loads, 16 p stores, 32 cs/flags
int g_march_occ = 2;       // workgroups per CU the kernel is compiled for (register budget) and sized for (rows per segment)
int g_march_dbg = 0;       // 1 = no stencil arithmetic (Ap := p), 2 = no double sums, 3 = 1 + no halo rows / lanes, 4 = segments marched bottom-up
int g_march_map = 0;       // 1 = the 4 waves of a workgroup side by side (x-adjacent strips) instead of stacked segments
#else
constexpr int g_march_dbg = 0, g_march_map = 0;
#endif
#ifdef THALLO_MARCH_SWEEP
constexpr bool MARCH_MAPS = true;
#else
constexpr bool MARCH_MAPS = false;      // the product has no workgroup-shape experiments (and no s_barrier in its row loop)
#endif
}

using namespace thallo;

namespace {

constexpr int MARCH_USE = 124;            // output pixels per wave row (lanes 1..62 x 2)
constexpr int MARCH_NT = 256;             // threads per workgroup: 4 waves = 4 vertically adjacent segments of one strip

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

// value of the lane to the left (lane-1) / right (lane+1); lanes without a source keep `self` (never used: lanes 0 / 63 produce no output)
__device__ __forceinline__ float from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ float from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));   // wave_shl:1
}
__device__ __forceinline__ unsigned from_left(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned from_right(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x130, 0xf, 0xf, false); }

template <bool FIRST>
struct Raw {                                // one row of one lane (2 pixels) as loaded
    float4 ro, po, cs, ao;                  // r / p / (c0,s0,c1,s1) / Ap: Offset part (x0,y0,x1,y1)
    float2 ra, pa, aa;                      // Angle part (a0, a1)
    unsigned f;                             // the dword holding the flags bytes of the 2 pixels
};
struct RawD { float4 dlo, ppo; float2 dla, ppa; };     // delta and p_{k-2} of the row (deferred delta update)
// Move a prefetch slot into fresh registers with REAL v_mov instructions (early-clobber outputs: never the slot's own registers).
// All arithmetic then works on the copy, the slot's registers die here and its refill -- issued right behind -- loads into the SAME
// registers again: the loop-carried slot needs no copy at the loop latch.  Without this the compiler computes in place in the
// slot registers (v_fmac), keeps the results there as window state, refills into other registers and copies -- after an
// s_waitcnt for the fresh load -- at the latch, which silently turns the prefetch into a blocking load.  23 moves per row.
// nothing moves across: neither in the optimiser (memory clobber) nor in the machine scheduler
__device__ __forceinline__ void fence_order() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void take1(float& d, const float& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
__device__ __forceinline__ void take1(unsigned& d, const unsigned& s) { asm volatile("v_mov_b32 %0, %1" : "=&v"(d) : "v"(s)); }
__device__ __forceinline__ void take4(float4& d, const float4& s) { take1(d.x, s.x); take1(d.y, s.y); take1(d.z, s.z); take1(d.w, s.w); }
__device__ __forceinline__ void take2(float2& d, const float2& s) { take1(d.x, s.x); take1(d.y, s.y); }
template <bool FIRST>
__device__ __forceinline__ void take(Raw<FIRST>& d, const Raw<FIRST>& s)
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

struct MarchGeo { int W, H, row0, row1, R, nstrips, nwgrow, total, map, use; };   // map / use: tools only (workgroup shape, pixels per wave row)

// one neighbour's contribution to (J^T J p)_i on the unit grid; D = 0: x+1, 1: x-1, 2: y+1, 3: y-1.  k_iter's expressions with
// u_i - u_j = -(dx,dy) folded in by hand (the compiler may not drop the products with 0.0f): with g_i = R'(a_i)(u_i-u_j),
// g_j = R'(a_j)(u_j-u_i):  D=0: g_i = (s_i,-c_i), g_j = (-s_j,c_j);  D=1: negated;  D=2: g_i = (c_i,s_i), g_j = (-c_j,-s_j);  D=3: negated
template <int D>
__device__ __forceinline__ void nb_term(bool valid, float ci, float si, float pxi, float pyi, float pai,
                                        float pxj, float pyj, float paj, float cj, float sj, float& ax, float& ay, float& av)
{
    if (valid) {
        const float gix = D == 0 ? si : D == 1 ? -si : D == 2 ? ci : -ci;
        const float giy = D == 0 ? -ci : D == 1 ? ci : D == 2 ? si : -si;
        const float gjx = D == 0 ? -sj : D == 1 ? sj : D == 2 ? -cj : cj;
        const float gjy = D == 0 ? cj : D == 1 ? -cj : D == 2 ? -sj : sj;
        const float dpx = pxi - pxj, dpy = pyi - pyj;
        const float ex = dpx - gix * pai, ey = dpy - giy * pai;
        ax += dpx + ex + gjx * paj;
        ay += dpy + ey + gjy * paj;
        av -= gix * ex + giy * ey;
    }
}

#ifdef THALLO_MARCH_SWEEP
// tools/march_probe.py MB_MODE=stamps: where a launch spends its time (100 MHz wall clock, lane 0 of every wave)
__device__ unsigned long long* g_stamps_m = nullptr;
// (the pointer is read ONCE per kernel, MARCH_STAMP_INIT: a read of the __device__ word inside the row loop is a load the compiler waits for with vmcnt(0) -- every row's
//  prefetch drained; the sweep build ran like that until round 3 and timed the same as the product)
#ifdef THALLO_MARCH_STAMPS      // (make VARIANT=stamps EXTRA="-DTHALLO_MARCH_SWEEP -DTHALLO_MARCH_STAMPS": the stamp stores are FLAT stores, and one FLAT instruction in the row loop makes
                                //  the compiler wait with vmcnt(0) at the top of every trip -- the plain sweep build has none, so that its loop is the product's)
#define MARCH_STAMP_INIT unsigned long long* const stamps_l = g_stamps_m
#define MARCH_STAMP(k) do { if ((threadIdx.x & 63) == 0 && stamps_l) stamps_l[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MARCH_STAMP_INIT do { } while (0)
#define MARCH_STAMP(k) do { } while (0)
#endif
#else
#define MARCH_STAMP_INIT do { } while (0)
#define MARCH_STAMP(k) do { } while (0)
#endif

template <bool FIRST, int DMODE, int DEPTH, int NTM, bool DIST, int OCC, int DBG>
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
    MARCH_STAMP_INIT;
    MARCH_STAMP(0);
    if (threadIdx.x < 32) { float mo, ma; pre_from_flags((unsigned char)threadIdx.x, wf2, wr2, mo, ma); lut[threadIdx.x] = make_float2(mo, ma); }
    __syncthreads();
    MARCH_STAMP(1);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long N = (long)g.W * g.H;
    const int W2 = g.W >> 1;                                  // pixel pairs per row
    // XCD-aware placement: workgroups b and b+8 share an XCD (MI355X_MICROARCH.md), group b%8 owns a contiguous range of
    // (band of 4 segments, strip) ids, x-adjacent strips first: x-halo columns and y-halo rows are re-read from the same L2
    int strip = 0, ya = 0, yb = 0, sya = 0, syb = 0;      // (sya, syb: the workgroup's segment -- map 2: waves without a strip still take part in its barriers)
    {
        const int G = (MARCH_MAPS && g.map == 3) ? 1 : (gridDim.x % 8) == 0 ? 8 : 1;      // (map 3, microbench: no XCD-aware placement -- workgroup b takes id b)
        const int grp = blockIdx.x % G, l = blockIdx.x / G;
        const long lo = (long)g.total * grp / G, hi = (long)g.total * (grp + 1) / G;
        const long id = lo + l;
        if (MARCH_MAPS && id < hi && (g.map == 1 || g.map == 2)) {        // (microbench) the 4 waves side by side: one segment of 4 x-adjacent strips; map 2: + a workgroup barrier per loop trip (three rows),
            const int nsb = (g.nstrips + 3) / 4;                             // so that the four waves touch the same image rows -- the same DRAM pages -- at the same time
            strip = (int)(id % nsb) * 4 + wave;
            const int seg = (int)(id / nsb);
            sya = g.row0 + seg * g.R; syb = sya + g.R; if (syb > g.row1) syb = g.row1; if (sya > g.row1) sya = g.row1;
            if (strip < g.nstrips) { ya = sya; yb = syb; }
        } else if (id < hi) {
            strip = (int)(id % g.nstrips);
            const int seg = (int)(id / g.nstrips) * (MARCH_NT / 64) + wave;
            ya = g.row0 + seg * g.R; yb = ya + g.R;
            if (yb > g.row1) yb = g.row1;
            if (ya > g.row1) ya = g.row1;
        }
    }
    const bool work = ya < yb;
    const int x0 = (DBG == 3 || DBG == 6 || DBG == 8) ? strip * 128 + 2 * lane : strip * MARCH_USE - 2 + 2 * lane;          // first of this lane's two pixels
    const bool xin = x0 >= 0 && x0 < g.W;                     // W even: both pixels exist or neither
    const bool xout = DBG == 6 ? true : xin && (DBG == 3 || DBG == 8 || (lane >= 1 && lane <= 62));      // (DBG 6, timing only, W a multiple of 128: DBG 3 with every store unconditional)         // this lane's pixels are outputs of this wave

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
    auto publish = [&](const RawT& s, const RawD& sd, int t, bool live, Row& wn) {     // s, sd: copies made by take()
        const bool ok = DBG == 6 ? true : live && xin && row_exists(t);
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
        if (DBG == 6 || (ok && xout && (mine || !row_owned(t)))) {       // this wave's own rows, or a ghost row of the slab (kept current here)
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
        // x neighbours across the lane boundary: the left lane's pixel 1, the right lane's pixel 0
        const float Lpx = from_left(wc.px[1]), Lpy = from_left(wc.py[1]), Lpa = from_left(wc.pa[1]), Lc = from_left(wc.c[1]), Ls = from_left(wc.s[1]);
        const float Rpx = from_right(wc.px[0]), Rpy = from_right(wc.py[0]), Rpa = from_right(wc.pa[0]), Rc = from_right(wc.c[0]), Rs = from_right(wc.s[0]);
        const unsigned Lf = from_left(wc.f) >> 8, Rf = from_right(wc.f);
        if (DBG == 1 || DBG == 3 || DBG == 6 || DBG == 8) {
            if (xout) {
                const long i2 = (long)y * W2 + (x0 >> 1);
                stf4(Ao4 + i2, make_float4(wc.px[0] + Lpx, wc.py[0], wc.px[1] + Rpx, wc.py[1]), nt_out); stf2(Aa2 + i2, make_float2(wc.pa[0] + wn.pa[0], wc.pa[1] + wm.pa[1]), nt_out);
                acc += wc.px[0];
            }
            return;
        }
        if (xout) {
            float ax[2], ay[2], av[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                ax[q] = 0.f; ay[q] = 0.f; av[q] = 0.f;
                const unsigned fq = (wc.f >> (8 * q)) & 255u;
                const float pxi = wc.px[q], pyi = wc.py[q], pai = wc.pa[q];
                if (fq & 1u) {
                    const float ci = wc.c[q], si = wc.s[q];
                    if (q == 0) {
                        nb_term<0>((wc.f >> 8) & 1u, ci, si, pxi, pyi, pai, wc.px[1], wc.py[1], wc.pa[1], wc.c[1], wc.s[1], ax[q], ay[q], av[q]);
                        nb_term<1>(Lf & 1u, ci, si, pxi, pyi, pai, Lpx, Lpy, Lpa, Lc, Ls, ax[q], ay[q], av[q]);
                    } else {
                        nb_term<0>(Rf & 1u, ci, si, pxi, pyi, pai, Rpx, Rpy, Rpa, Rc, Rs, ax[q], ay[q], av[q]);
                        nb_term<1>(wc.f & 1u, ci, si, pxi, pyi, pai, wc.px[0], wc.py[0], wc.pa[0], wc.c[0], wc.s[0], ax[q], ay[q], av[q]);
                    }
                    nb_term<2>((wn.f >> (8 * q)) & 1u, ci, si, pxi, pyi, pai, wn.px[q], wn.py[q], wn.pa[q], wn.c[q], wn.s[q], ax[q], ay[q], av[q]);
                    nb_term<3>((wm.f >> (8 * q)) & 1u, ci, si, pxi, pyi, pai, wm.px[q], wm.py[q], wm.pa[q], wm.c[q], wm.s[q], ax[q], ay[q], av[q]);
                    ax[q] *= wr2; ay[q] *= wr2; av[q] *= wr2;
                    if (fq & 2u) { ax[q] += wf2 * pxi; ay[q] += wf2 * pyi; }
                }
                acc += pxi * ax[q] + pyi * ay[q] + pai * av[q];
                // exact products of the float data, accumulated in double: N = sum r.M^-1.r, S1 = sum r.M^-1.Ap, S2 = sum Ap.M^-1.Ap
                // (both Offset channels share M^-1: 5 double operations per sum)
                if (DBG == 2) continue;
                const double dmo = wc.mo[q], dma = wc.ma[q], drx = wc.rx[q], dry = wc.ry[q], dra = wc.ra[q], dax = ax[q], day = ay[q], daa = av[q];
                s0 = __builtin_fma(dmo, __builtin_fma(dry, dry, drx * drx), __builtin_fma(dma, dra * dra, s0));
                s1 = __builtin_fma(dmo, __builtin_fma(dry, day, drx * dax), __builtin_fma(dma, dra * daa, s1));
                s2 = __builtin_fma(dmo, __builtin_fma(day, day, dax * dax), __builtin_fma(dma, daa * daa, s2));
            }
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
        const int t_first = (DBG == 3 || DBG == 6 || DBG == 8) ? ya : ya - 1, t_last = (DBG == 3 || DBG == 6 || DBG == 8) ? yb - 1 : yb;            // rows to publish: the segment and its two halo rows
        // No branch around a load anywhere in this loop (see `issue`): the step count is rounded up to a multiple of 3, rows beyond
        // t_last are clamped re-reads of the last row (cache hits) whose publish / stencil are predicated off.
        // There is no prologue either: the loop starts three rows early with empty slots (publish predicated off) and its refills are
        // the first loads.  A separate prologue is a second path into the loop header with its own (compiler-scheduled) issue order,
        // and the wait at the header is the conservative merge of both paths -- in practice vmcnt(0).
        // Entering an iteration at row t0: win[0] = row t0-2, win[1] = row t0-1.
#pragma unroll
        for (int j = 0; j < 3; ++j) { slot[j] = RawT{}; dsl[j] = RawD{}; }
        for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) {
            if (MARCH_MAPS && g.map == 2) __builtin_amdgcn_s_barrier();
            // The iteration's scalars, at the start of the SECOND trip: the first trip only issued the loads of rows t_first .. t_first + 2, nothing
            // needed alpha / beta yet; now the partial (or word) loads queue up behind those row loads and the additions run while the rows arrive.
            if (t0 == t_first) MARCH_STAMP(2);
            if (DBG == 8) { alpha = 0.5f; beta = 0.25f; }      // (tools: DBG 3 without the iteration's scalars and without the reduction tail -- what a launch's fixed parts cost)
            else
            if (!FIRST && SCALARS_IN_LOOP && t0 == t_first) iteration_scalars<1>(aNp, aDp, bNp, prev, alpha, beta, scal_writer);
            if (t0 == t_first) MARCH_STAMP(3);
            if (t0 == t_first + 3) MARCH_STAMP(4);
            if (DBG == 7) {     // (tools: BATCHED trips, no delta variants -- the three rows' loads taken together, their 24 refills issued back to back, then the three rows'
                                // arithmetic and their 18 stores: fewer, longer bursts per wave, like the streaming reference)
                RawT cur3[3]; const RawD nod = RawD{};
#pragma unroll
                for (int j = 0; j < 3; ++j) take(cur3[j], slot[j]);
                fence_order();
#pragma unroll
                for (int j = 0; j < 3; ++j) issue(slot[j], t0 + j + 3 > t_last ? t_last : t0 + j + 3);
                fence_order();
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int t = t0 + j;
                    Row& wn = win[(j + 2) % 3]; Row& wc = win[(j + 1) % 3]; Row& wm = win[j % 3];
                    if (t >= t_first && t <= t_last) publish(cur3[j], nod, t, true, wn);
                    if (t - 1 >= ya && t <= t_last) stencil(t - 1, wm, wc, wn);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = t0 + j;
                Row& wn = win[(j + 2) % 3]; Row& wc = win[(j + 1) % 3]; Row& wm = win[j % 3];
                RawT cur; RawD curd = RawD{};
                take(cur, slot[j]);                          // (the only place that waits for memory: s_waitcnt vmcnt(N) with the other two slots still in flight)
                if (DMODE != 1) take<DMODE>(curd, dsl[j]);
                fence_order();                               // the refill stays behind the moves ...
                // DBG == 4 (sweep build): the segment marched bottom-up -- step t works on row ya + yb - 1 - t, the window roles y-1 / y+1 swap
                const auto phys = [&](int tt) { return DBG == 4 ? ya + yb - 1 - tt : tt; };
                issue(slot[j], phys(t + 3 > t_last ? t_last : t + 3));
                if (DMODE != 1) issue_d(dsl[(j + 2) % 3], phys(t + 2));      // (the slot taken one step ago)
                fence_order();                               // ... and in front of the arithmetic
                // (wave-uniform branch, no load inside: the three lead-in rows and the rounding-up rows skip the arithmetic; the window keeps its zeros)
                if (DBG == 6) {     // no branch around a store: lead-in / rounding-up trips store (garbage, later overwritten / identical) into the clamped row
                    const int tq = t < t_first ? t_first : t > t_last ? t_last : t;
                    publish(cur, curd, tq, true, wn); stencil(tq, wc, wn, wm);
                    continue;
                }
                if (t >= t_first && t <= t_last) publish(cur, curd, phys(t), true, wn);
                if (DBG == 3 || DBG == 8) { if (t >= t_first && t <= t_last) stencil(t, wc, wn, wm); }      // (timing only)
                else if (DBG == 4) { if (t - 1 >= ya && t <= t_last) stencil(phys(t - 1), wn, wc, wm); }
                else if (t - 1 >= ya && t <= t_last) stencil(t - 1, wm, wc, wn);
            }
        }
    }
    else if (MARCH_MAPS && g.map == 2 && sya < syb) {      // a wave without a strip: the same number of barriers as its three siblings
        const int t_first = DBG == 3 ? sya : sya - 1, t_last = DBG == 3 ? syb - 1 : syb;
        for (int t0 = t_first - 3; t0 <= t_last; t0 += 3) __builtin_amdgcn_s_barrier();
    }
    if (DBG == 8) { if ((threadIdx.x & 63) == 0) aD_out[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 1023] = acc; return; }
    MARCH_STAMP(5);
    iter_tail<MARCH_NT, DIST>(acc, s0, s1, s2, red, redd, aD_out, s12_out, bNp, &dd, fin_tickets, aD_word, bN_word, xslot);
    MARCH_STAMP(6);
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

#ifdef THALLO_MARCH_SWEEP
// tools/march_probe.py only: the same planes moved with the same access widths by a flat grid-stride loop and trivial arithmetic --
// the streaming ceiling of this byte mix (81 B/pixel, or 117 with the delta / p_{k-2} planes) on this box.
template <int NTM, bool DELTA, bool FUSED = false>
__global__ __launch_bounds__(256) void k_stream_ref(long n2, long N, int span, int rev, const float* __restrict__ cs, const unsigned char* __restrict__ flags,
                                                    const float* __restrict__ r_in, float* __restrict__ r_out, const float* __restrict__ A_in, float* __restrict__ A_out,
                                                    const float* __restrict__ p_in, float* __restrict__ p_out, float* __restrict__ delta, float alpha)
{
    constexpr bool nt_delta = NTM & 1, nt_ra = NTM & 2, nt_out = NTM & 4, nt_pin = NTM & 8, nt_pout = NTM & 16, nt_const = NTM & 32;
    const float4* ro4 = (const float4*)r_in; const float2* ra2 = (const float2*)(r_in + 2 * N);
    const float4* ao4 = (const float4*)A_in; const float2* aa2 = (const float2*)(A_in + 2 * N);
    const float4* po4 = (const float4*)p_in; const float2* pa2 = (const float2*)(p_in + 2 * N);
    const float4* cs4 = (const float4*)cs; const unsigned short* f2 = (const unsigned short*)flags;
    float4* Ro4 = (float4*)r_out; float2* Ra2 = (float2*)(r_out + 2 * N);
    float4* Ao4 = (float4*)A_out; float2* Aa2 = (float2*)(A_out + 2 * N);
    float4* qo4 = (float4*)p_out; float2* qa2 = (float2*)(p_out + 2 * N);
    float4* dl4 = (float4*)delta; float2* dl2 = (float2*)(delta + 2 * N);
    // traversal order: span = 1 is the flat grid-stride sweep (the whole chip inside a narrow moving window); span = K lets every
    // workgroup walk K consecutive 256-thread chunks before it jumps ahead by gridDim.x * K chunks -- K = chunks / gridDim.x is the
    // order of the marching kernel (every workgroup owns one contiguous region, the chip touches the whole plane all the time)
    const long chunks = (n2 + 255) / 256;
    // span < 0 (tools: the marching kernel's traversal without its halo, window or arithmetic): a WAVE owns a column strip of 128 pixels and walks down -span... rows of it;
    // the image width comes in through `rev` (pixel pairs per row)
    const int W2s = span < 0 ? rev : 0, strips = span < 0 ? W2s / 64 : 0;
    const long rows_total = span < 0 ? n2 / W2s : 0, waves = (long)gridDim.x * 4, segs_per_strip = span < 0 ? waves / strips : 0;
    const long Rw = span < 0 ? (rows_total + segs_per_strip - 1) / segs_per_strip : 0;
    for (long it = 0;; ++it) {
        long i;
        if (span < 0) {
            const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
            const long strip = wid % strips, seg = wid / strips;
            const long row = seg * Rw + it;
            if (it >= Rw || seg >= segs_per_strip) break;
            if (row >= rows_total) break;
            i = row * W2s + strip * 64 + (threadIdx.x & 63);
        } else {
        const long band = it / span, within = it % span;
        const long chunk = (band * gridDim.x + blockIdx.x) * span + within;
        if (band * (long)gridDim.x * span >= chunks) break;
        // rev: the same chunks in the opposite order -- what the previous launch touched LAST is read FIRST (the part of it the 256 MB Infinity Cache still holds)
        i = (rev ? chunks - 1 - chunk : chunk) * 256 + threadIdx.x;
        }
        if (i >= n2 || i < 0) continue;
        // FUSED (layout experiment): every solver vector as ONE stream of 6 floats per pixel pair (x0 y0 a0 x1 y1 a1) instead of an Offset plane and an Angle plane:
        // 9 address streams per workgroup instead of 15
        float4 r, a, pp; float2 ra, aa, pa;
        if (FUSED) {
            const float2* R6 = (const float2*)r_in + 3 * i; const float2* A6 = (const float2*)A_in + 3 * i; const float2* P6 = (const float2*)p_in + 3 * i;
            const float2 r0 = ldf2(R6, nt_ra), r1 = ldf2(R6 + 1, nt_ra), r2 = ldf2(R6 + 2, nt_ra); r = make_float4(r0.x, r0.y, r1.x, r1.y); ra = r2;
            const float2 a0 = ldf2(A6, nt_ra), a1 = ldf2(A6 + 1, nt_ra), a2 = ldf2(A6 + 2, nt_ra); a = make_float4(a0.x, a0.y, a1.x, a1.y); aa = a2;
            const float2 p0 = ldf2(P6, nt_pin), p1 = ldf2(P6 + 1, nt_pin), p2 = ldf2(P6 + 2, nt_pin); pp = make_float4(p0.x, p0.y, p1.x, p1.y); pa = p2;
        } else {
            r = ldf4(ro4 + i, nt_ra); ra = ldf2(ra2 + i, nt_ra);
            a = ldf4(ao4 + i, nt_ra); aa = ldf2(aa2 + i, nt_ra);
            pp = ldf4(po4 + i, nt_pin); pa = ldf2(pa2 + i, nt_pin);
        }
        const float4 c = ldf4(cs4 + i, nt_const); const unsigned f = nt_const ? (unsigned)__builtin_nontemporal_load(f2 + i) : (unsigned)f2[i];
        r.x -= alpha * a.x; r.y -= alpha * a.y; r.z -= alpha * a.z; r.w -= alpha * a.w; ra.x -= alpha * aa.x; ra.y -= alpha * aa.y;
        const float m = (f & 1u) ? 0.5f : 0.25f;
        const float4 q = make_float4(m * r.x + c.x * pp.x, m * r.y + c.y * pp.y, m * r.z + c.z * pp.z, m * r.w + c.w * pp.w);
        const float2 qa = make_float2(m * ra.x + pa.x, m * ra.y + pa.y);
        if (FUSED) {
            float2* R6 = (float2*)r_out + 3 * i; float2* Q6 = (float2*)p_out + 3 * i; float2* A6 = (float2*)A_out + 3 * i;
            stf2(R6, make_float2(r.x, r.y), nt_out); stf2(R6 + 1, make_float2(r.z, r.w), nt_out); stf2(R6 + 2, ra, nt_out);
            stf2(Q6, make_float2(q.x, q.y), nt_pout); stf2(Q6 + 1, make_float2(q.z, q.w), nt_pout); stf2(Q6 + 2, qa, nt_pout);
            stf2(A6, make_float2(q.x + r.x, q.y + r.y), nt_out); stf2(A6 + 1, make_float2(q.z + r.z, q.w + r.w), nt_out); stf2(A6 + 2, make_float2(qa.x + ra.x, qa.y + ra.y), nt_out);
        } else {
        stf4(Ro4 + i, r, nt_out); stf2(Ra2 + i, ra, nt_out);
        stf4(qo4 + i, q, nt_pout); stf2(qa2 + i, qa, nt_pout);
        stf4(Ao4 + i, make_float4(q.x + r.x, q.y + r.y, q.z + r.z, q.w + r.w), nt_out); stf2(Aa2 + i, make_float2(qa.x + ra.x, qa.y + ra.y), nt_out);
        }
        if (DELTA) {
            float4 d; float2 da; float4 o; float2 oa;
            if (FUSED) {
                const float2* D6 = (const float2*)delta + 3 * i; const float2* O6 = (const float2*)p_out + 3 * i;
                const float2 d0 = ldf2(D6, nt_delta), d1 = ldf2(D6 + 1, nt_delta), d2 = ldf2(D6 + 2, nt_delta); d = make_float4(d0.x, d0.y, d1.x, d1.y); da = d2;
                const float2 o0 = O6[0], o1 = O6[1], o2 = O6[2]; o = make_float4(o0.x, o0.y, o1.x, o1.y); oa = o2;
            } else {
            d = ldf4(dl4 + i, nt_delta); da = ldf2(dl2 + i, nt_delta);
            o = qo4[i]; oa = qa2[i];     // (stand-in for p_{k-2}: the plane about to be overwritten)
            }
            d.x += alpha * (pp.x + o.x); d.y += alpha * (pp.y + o.y); d.z += alpha * (pp.z + o.z); d.w += alpha * (pp.w + o.w);
            da.x += alpha * (pa.x + oa.x); da.y += alpha * (pa.y + oa.y);
            if (FUSED) { float2* D6 = (float2*)delta + 3 * i; stf2(D6, make_float2(d.x, d.y), nt_delta); stf2(D6 + 1, make_float2(d.z, d.w), nt_delta); stf2(D6 + 2, da, nt_delta); }
            else { stf4(dl4 + i, d, nt_delta); stf2(dl2 + i, da, nt_delta); }
        }
    }
}
#endif

inline MarchGeo make_march_geo(int W, int H, int row0, int row1, int R)
{
    MarchGeo g; g.W = W; g.H = H; g.row0 = row0; g.row1 = row1; g.R = R;
    g.map = g_march_map; g.use = (g_march_dbg == 3 || g_march_dbg == 6 || g_march_dbg == 8) ? 128 : MARCH_USE;
    g.nstrips = (W + g.use - 1) / g.use;
    const int nseg = (row1 - row0 + R - 1) / R;
    g.nwgrow = (nseg + MARCH_NT / 64 - 1) / (MARCH_NT / 64);
    g.total = g.nstrips * g.nwgrow;
    if (g.map == 1 || g.map == 2) g.total = ((g.nstrips + 3) / 4) * nseg;
    return g;
}

// rows per wave segment: every workgroup resident at once (the 190-VGPR variant fits 2 workgroups of 4 waves per CU), i.e. about
// 2 waves per SIMD, and at most 1024 workgroups (partial slots).  0: the image is wider than that many strips (W > ~31.7k pixels on a 256-CU
// device, ~3.9k on a 32-CU partition) -- the caller runs the tile kernel instead (thallo_hip_iw_march_fits).
inline long march_cap(int occ) { return g_march_cap > 0 ? g_march_cap : (long)thallo_hip_device_cu_count() * occ; }
// Wide images: the grid is (strips) x (bands of 4 segments), so with ONE workgroup per CU as the budget a width whose strip count does not divide the CU count leaves
// CUs without work -- 16384 pixels: 133 strips x 1 band = 133 of 256 CUs.  Measured (round 3, tools/march_probe.py MB_MODE=ab with and without MB_CAP=256): 16384 x 2048
// 920 -> 697 us per launch with the budget grown until the grid fills its last round of workgroups; 8192 x 2048 (67 x 3 = 201 of 256) gains nothing from 469
// workgroups, so a fill of 75 % counts as full.  The budget grows to 2, 3, 4 workgroups per CU (two are resident at once, the rest follow as CUs free up; at most
// THALLO_MAX_PARTIALS): once it grows, up to the smallest multiple with a fill of 90 % (16384 wide: 931 workgroups; 737 us with 399), else the best.  Images for which one workgroup per CU already fills the chip -- every size the other
// kernels are compared with bit for bit -- keep their rows per segment.
inline int pick_rows(int W, int rows, int occ)
{
    if (g_march_rows > 0) return g_march_rows;
    const int use = (g_march_dbg == 3 || g_march_dbg == 6 || g_march_dbg == 8) ? 128 : MARCH_USE;
    const int nstrips = (W + use - 1) / use, wpw = MARCH_NT / 64;
    if (g_march_cap > 0) return march_rows_per_segment(rows, nstrips, wpw, march_cap(occ));       // (a forced budget -- tests, tools -- is taken as it is)
    const long cus = march_cap(1);
    int best_R = 0; double best_fill = -1.0;
    for (int m = occ; m <= 4; ++m) {
        const int R = march_rows_per_segment(rows, nstrips, wpw, cus * m);
        if (R <= 0) break;                      // more strips than one workgroup per CU has slots: thallo_hip_iw_march_fits() says no, the tile kernel runs
        const long nseg = (rows + R - 1) / R, total = (long)nstrips * ((nseg + wpw - 1) / wpw);
        const double fill = (double)total / (double)(((total + cus - 1) / cus) * cus);
        if (best_R == 0 || fill > best_fill + 1e-9) { best_R = R; best_fill = fill; }
        if (fill >= (m == occ ? 0.75 : 0.9)) break;
    }
    return best_R;
}

constexpr int MARCH_DEPTH = 2, MARCH_NTM = 5, MARCH_OCC = 2;     // product configuration (tools/march_probe.py sweeps, profiles/r02): delta and the r / Ap stores non-temporal
constexpr int MARCH_WG_PER_CU = 1;                                // grid sizing: ~1 workgroup (4 waves) per CU measured best at 2048^2 (36 rows per wave; 18: +5 %, 48: +12 %)

// every configuration tools/march_probe.py may select: (prefetch depth, non-temporal mask, workgroups per CU, debug)
#define MARCH_SWEEP_CFGS(X) \
    X(2, 0, 2, 0) X(2, 1, 2, 0) X(2, 3, 2, 0) X(2, 9, 2, 0) X(2, 11, 2, 0) X(2, 33, 2, 0) X(2, 35, 2, 0) X(2, 41, 2, 0) X(2, 43, 2, 0) \
    X(2, 5, 2, 0) X(2, 17, 2, 0) X(2, 21, 2, 0) X(2, 31, 2, 0) X(2, 63, 2, 0) \
    X(2, 1, 3, 0) X(2, 11, 3, 0) X(2, 43, 3, 0) X(1, 1, 3, 0) X(1, 11, 3, 0) X(1, 43, 3, 0) X(1, 1, 4, 0) X(1, 11, 4, 0) \
    X(2, 5, 2, 6) X(2, 11, 2, 6) X(2, 5, 2, 7) X(2, 0, 2, 7) X(2, 5, 2, 8) X(2, 5, 2, 4) X(2, 0, 2, 4) X(2, 1, 2, 4) X(2, 4, 2, 0) X(2, 4, 2, 4) X(2, 16, 2, 0) X(2, 16, 2, 4) \
    X(2, 1, 2, 1) X(2, 1, 2, 2) X(1, 1, 2, 0) X(3, 1, 2, 0) X(2, 5, 2, 1) X(2, 11, 2, 1) X(2, 0, 2, 1) X(2, 1, 2, 3) X(2, 5, 2, 3) X(2, 11, 2, 3) X(2, 0, 2, 3)

template <bool DIST>
int launch_march(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                 const float* r_in, float* r_out, const float* Ap_in, float* Ap_out, const float* p_in, float* p_out, float* delta, int mode,
                 thallo_sum_t aNp, thallo_sum_t aDp, thallo_sum_t bNp, thallo_sum_t aNpp, thallo_sum_t aDpp, const int* irregular, thallo_dist_t d,
                 float* aD_out, double* s12_out, unsigned* fin_tickets, float* aD_word, float* bN_word, int xslot, hipStream_t stream, PrevSums prev = PrevSums{ nullptr, nullptr, 0, nullptr, nullptr })
{
    const int R = pick_rows(W, row1 - row0, MARCH_WG_PER_CU);
    if (R <= 0) return -(int)hipErrorNotSupported;                          // wider than the workgroup budget: thallo_hip_iw_march_fits() said so
    const MarchGeo g = make_march_geo(W, H, row0, row1, R);
    const int grid = (g.total + 7) / 8 * 8;
    if (grid > THALLO_MAX_PARTIALS) return -(int)hipErrorInvalidValue;      // (only reachable through the tools' forced rows-per-segment)
    const bool first = mode & 1;
    const int dmode = first ? 1 : (mode >> 1) & 3;
    const float wf2 = w_fit * w_fit, wr2 = w_reg * w_reg;
#define MARCH_LAUNCH(F, DM, DP, NTM, OCC, DBG) hipLaunchKernelGGL((k_iter_march<F, DM, DP, NTM, DIST, OCC, DBG>), dim3(grid), dim3(MARCH_NT), 0, stream, g, cs, flags, wf2, wr2, \
        r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, aNp, aDp, bNp, aNpp, aDpp, aD_out, s12_out, irregular, d, fin_tickets, aD_word, bN_word, xslot, prev)
    // (the multi-GPU variant has no "apply two delta updates" form: peer stores on top of that variant's 256 registers spill, and at slab sizes the
    //  6 B/pixel it saves do not matter -- solver_dist.cpp updates delta every iteration on the device-side transport)
    if (DIST && dmode == 2) return -(int)hipErrorInvalidValue;
#define MARCH_BY_MODE(DP, NTM, OCC, DBG) do { if (first) MARCH_LAUNCH(true, 1, DP, NTM, OCC, DBG); else if (dmode == 1) MARCH_LAUNCH(false, 1, DP, NTM, OCC, DBG); \
        else if (dmode == 2) { if constexpr (!DIST) MARCH_LAUNCH(false, 2, DP, NTM, OCC, DBG); } else MARCH_LAUNCH(false, 0, DP, NTM, OCC, DBG); } while (0)
    bool launched = false;
#ifdef THALLO_MARCH_SWEEP
    if constexpr (!DIST) {
#define X(DP, NTM, OCC, DBG) if (!launched && g_march_depth == DP && g_march_nt == NTM && g_march_occ == OCC && g_march_dbg == DBG) { MARCH_BY_MODE(DP, NTM, OCC, DBG); launched = true; }
        MARCH_SWEEP_CFGS(X)
#undef X
        if (!launched) return -(int)hipErrorInvalidValue;       // not an instantiated configuration
    }
#endif
    if (!launched) MARCH_BY_MODE(MARCH_DEPTH, MARCH_NTM, MARCH_OCC, 0);
#undef MARCH_BY_MODE
#undef MARCH_LAUNCH
    int e = check_launch(); return e ? e : grid;
}

}  // namespace

extern "C" {
#ifdef THALLO_MARCH_SWEEP
int thallo_hip_debug_stamps_march(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_m), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif

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

#ifdef THALLO_MARCH_SWEEP
int thallo_hip_iw_stream_ref(int W, int H, const float* cs, const unsigned char* flags, const float* r_in, float* r_out, const float* Ap_in, float* Ap_out,
                             const float* p_in, float* p_out, float* delta, int with_delta, int ntm, int blocks_per_cu, int span, thallo_stream_t stream)
{
    const long N = (long)W * H, n2 = N / 2;
    const int grid = thallo_hip_device_cu_count() * blocks_per_cu;
    int rev = span >= 1000 ? 1 : 0; if (rev) span -= 1000;              // (tools: span + 1000 = the reversed traversal)
    if (span < 0) { if ((W / 2) % 64) return -(int)hipErrorInvalidValue; rev = W / 2; }      // (tools: span < 0 = column strips walked down by waves, like the marching kernel)
    else if (span == 0) span = (int)(((n2 + 255) / 256 + grid - 1) / grid);      // 0: one contiguous region per workgroup
#define SR(NTM) do { if (with_delta) hipLaunchKernelGGL((k_stream_ref<NTM, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n2, N, span, rev, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, 0.5f); \
                     else hipLaunchKernelGGL((k_stream_ref<NTM, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n2, N, span, rev, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, 0.5f); } while (0)
#define SRF(NTM) do { if (with_delta) hipLaunchKernelGGL((k_stream_ref<NTM, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n2, N, span, rev, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, 0.5f); \
                     else hipLaunchKernelGGL((k_stream_ref<NTM, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n2, N, span, rev, cs, flags, r_in, r_out, Ap_in, Ap_out, p_in, p_out, delta, 0.5f); } while (0)
    if (ntm == 100) { SRF(0); return check_launch(); } if (ntm == 105) { SRF(5); return check_launch(); } if (ntm == 111) { SRF(11); return check_launch(); }
    if (ntm == 5) { SR(5); return check_launch(); }
    if (ntm == 0) SR(0); else if (ntm == 1) SR(1); else if (ntm == 11) SR(11); else if (ntm == 43) SR(43); else if (ntm == 31) SR(31); else if (ntm == 63) SR(63); else return -(int)hipErrorInvalidValue;
#undef SR
    return check_launch();
}
#endif

void thallo_hip_march_debug_set(int what, int value)
{
    if (what == 0) g_march_rows = value;
    if (what == 6) g_march_cap = value;
#ifdef THALLO_MARCH_SWEEP
    if (what == 1) g_march_depth = value;
    if (what == 2) g_march_nt = value;
    if (what == 3) g_march_occ = value;
    if (what == 4) g_march_dbg = value;
    if (what == 5) g_march_map = value;
#endif
}

/* rows per wave segment the marching kernels would use on `rows` owned rows of a W-wide image; 0 = the image has more column strips than the device
 * has workgroup slots, the marching kernels return -hipErrorNotSupported and the caller stays on the tile kernel */
int thallo_hip_iw_march_rows(int W, int rows)
{
    if (W < 2 || (W & 1) || rows < 1) return 0;
    return pick_rows(W, rows, MARCH_WG_PER_CU);
}

}  // extern "C"
