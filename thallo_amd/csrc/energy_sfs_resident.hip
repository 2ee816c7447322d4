// energy_sfs_resident.hip -- the whole PCG loop of one Gauss-Newton step of shape_from_shading in ONE launch, for images whose solver state fits the register
// files of the chip (the reference's own data set, 640 x 480; round 6).
//
// Why: at 0.3 Mpixel a launch per PCG iteration (energy_sfs_pair.hip) is all launch boundary, lead-in rows and reduction tail -- 9.7 us per launch back to back, 13 us
// through the solver, for 12 MB of traffic (1.5 us at the HBM peak) and ~2 us of arithmetic.  Here a wave keeps r, p, A p, delta and the per-GN-step planes (Gx, Gy,
// Gz, flags | edge masks) of its R rows x 124 pixels -- plus the radius-2 halo: two rows above, two below, one lane (two pixels) left and right -- in REGISTERS for all L
// iterations; what moves per iteration is what another wave needs, through global memory as 8-byte {value | tag} granules (write-through, agent scope; the data IS
// the flag: energy_image_warping_resident.hip, MI355X_MICROARCH.md "handoff-1to1"):
//   * the first two and the last two rows of A p_k of a segment to the waves above / below (lanes 1..62);
//   * lane 1's / lane 62's pixels of A p_k of every row to the strips left / right -- whose lanes 63 / 0 also take the halo ROWS' corner pixels from the records of
//     the diagonal neighbours (the chain dB -> U -> T -> J^T reaches (x +- 1, y -+ 1), (x +- 2, y -+ 1), (x +- 1, y -+ 2));
//   * the workgroup's four sums {alphaD | N, S1, S2} to every workgroup.
// All of it is published at the end of an iteration's arithmetic and consumed in front of the next iteration's vector update: ONE synchronisation point per iteration,
// neighbours and scalars only, no grid barrier.  A p_{k-1} is exchanged rather than p_k (p_k on the halo needs alpha / beta, i.e. the global sums, first); r and p on the
// halo are recomputed locally, as the slabs of a multi-GPU run do for their ghost rows.
//
// Geometry, per-pixel arithmetic and summation order are energy_sfs_pair.hip's k_pmarch<SUMS, UPD> (a wave = a 124-pixel strip x R rows, workgroup = 4 stacked segments,
// its XCD-aware placement; the row step's expressions verbatim; per-lane accumulation by rows, wave butterfly, workgroup sum, lane-strided sum of the workgroups'
// partials), both files built with -ffp-contract=on: r, p, delta, A p and every alpha / beta are bit-identical to L launches of the marching kernel run with the same R
// (tests/test_gpu_parity.py).  Gauss-Newton, whole image on one GPU.  Replaces gauss_newton.t:1615-1687 (the PCG loop) for these shapes.
#include "sfs_pair.hpp"
#include "sfs_pair_device.hpp"
#include <cstring>

using namespace thallo;

namespace {

constexpr int SR_NT = 256;                // 4 waves = 4 vertically adjacent segments of one strip (one workgroup per CU)
constexpr int SR_MIN_R = 2, SR_MAX_R = 8; // rows per segment the kernel is instantiated for (two halo rows come from ONE neighbouring segment: R >= 2; 14 registers per held row and lane)

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// exchange buffers as raw buffers; aux 16 = sc1: write-through stores / L1-bypassing loads (agent scope).  A 16-byte access moves TWO granules, each 8-byte half with its own tag.
__device__ __forceinline__ rsrc_t make_xrsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ u32x4 ld2g(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ u32x2 ld1g(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16); }
__device__ __forceinline__ void st2g(rsrc_t r, unsigned off, unsigned tag, float v0, float v1)
{ u32x4 d; d.x = __float_as_uint(v0); d.y = tag; d.z = __float_as_uint(v1); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 16); }
__device__ __forceinline__ void st1g(rsrc_t r, unsigned off, unsigned tag, unsigned v) { u32x2 d; d.x = v; d.y = tag; __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 16); }

// control words of a resident launch (device memory)
enum { SR_SEQ = 0, SR_ERR = 1, SR_SPIN_MS = 2, SR_PM = 4, SR_CTL_WORDS = 16 };

struct SrGeo { int W, H, yoff, R, nstrips, nseg, nwgrow, total; };

// exchange buffers of one plan (thallo_hip_sfs_resident_bytes); parity = iteration & 1
struct SrBufs {
    u64* rowh;        // [2 parity][waves][2 sides: 0 = the wave's FIRST two rows (for the wave above), 1 = its LAST two (for the wave below)][64 lanes][4: row a px 0, px 1, row b px 0, px 1]
    u64* colh;        // [2 parity][waves][2 sides: 0 = lane 1's pixels (for the strip to the left), 1 = lane 62's (for the strip to the right)][64: word 2 * row + pixel]
    u64* sums;        // [2 parity][1024 workgroups][8: alphaD, N hi, N lo, S1 hi, S1 lo, S2 hi, S2 lo, -]
    unsigned* ctl;    // SR_CTL_WORDS
};

struct SrArgs {
    SrGeo g; SrBufs b; PCam cm;
    const float* G; const unsigned* Fw;           // packed planes (sfs_pair.hpp)
    const float* r_in; const float* p_in;         // r_0 and p_{-1} (zeros): what PCGInit1 wrote
    float* r_out; float* A_out; float* p_out;     // r_{L-1}, A p_{L-1}, p_{L-1}: what L launches of the marching kernel leave behind
    float* delta;                                 // in: 0; out: sum_{k < L-1} alpha_k p_k (PCGLinearUpdate adds the last term, like behind the launches)
    thallo_sum_t aN0;                             // alphaN_0
    float* words;                                 // words[2k] = alphaD_k, words[2k + 1] = betaN_k
    float* X;                                     // the unknowns, or NULL: PCGLinearUpdate stays a launch of its own
    int L;
};

#ifdef THALLO_MARCH_SWEEP
// tools/sfs_resident_probe.py stamps: where an iteration spends its time (100 MHz wall clock, lane 0 of every wave, iterations 4..7)
__device__ unsigned long long* g_stamps_sr = nullptr;
#define SR_STAMP(k, i) do { if ((threadIdx.x & 63) == 0 && g_stamps_sr && (k) >= 4 && (k) < 8) g_stamps_sr[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + ((k) - 4)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SR_STAMP(k, i) do { } while (0)
#endif
struct Spin { unsigned n; long long t0; };
// bounded wait bookkeeping: true = give up (this wave or somebody else timed out; every later wait of the wave falls through at once)
__device__ __forceinline__ bool spin_fail(Spin& sp, unsigned* ctl, unsigned what, unsigned idx, unsigned tag)
{
    __builtin_amdgcn_s_sleep(1);
    if (((++sp.n) & 127u) != 0u) return false;
    if (__hip_atomic_load(ctl + SR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    const long long now = wall_clock64();
    if (sp.t0 == 0) { sp.t0 = now; return false; }
    const unsigned ms = __hip_atomic_load(ctl + SR_SPIN_MS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long bound = ms ? (long long)ms * 100000LL : 2LL * 100000000LL;        // default: 2 s of the 100 MHz wall clock
    if (now - sp.t0 <= bound) return false;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_exchange(ctl + SR_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        unsigned* pm = ctl + SR_PM;        // first timeout of the launch: what was waited for
        pm[0] = what; pm[1] = blockIdx.x; pm[2] = threadIdx.x >> 6; pm[3] = idx; pm[4] = tag;
    }
    return true;
}

// LDS words shared by the four waves of a workgroup
struct SrLds {
    unsigned qtag[4];                 // quarter-sweep exchange: wave w's column is complete for tag ...
    unsigned wtag[4];                 // wave sums of an iteration are in place
    unsigned q[4][7][64];             // per wave: the 7 words of the 64 slots it swept
    float wa[4]; double wd[4][3];     // per wave: alphaD part, {N, S1, S2} parts
    float cst[4][2][64];              // per wave: lane 1's / lane 62's A p of its rows (word 2 * row + pixel), so that ONE store instruction publishes a column
    float crx[4][2][64];              // per wave: the received columns (word 2 * held row + pixel), for lanes 0 / 63 to pick up
};

template <int R>
__global__ __launch_bounds__(SR_NT, 1) void k_sfs_resident(SrArgs a)
{
    constexpr int NR = R + 4;             // held rows: jj = 0, 1 the rows above, 2 .. R + 1 my own, R + 2, R + 3 the rows below (row t = ya - 2 + jj)
    __shared__ SrLds S;
    const SrGeo g = a.g;
    const PCam cm = a.cm;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned* const ctl = a.b.ctl;
    const int W = g.W, H = g.H;

    // workgroup -> (strip, first segment): k_pmarch's placement (pm_place), so that a workgroup's partials sit in the same slot
    const int grid = (int)gridDim.x;
    const int NG = (grid % 8) == 0 ? 8 : 1;
    auto wg_id = [&](int b, long& id) { const int grp = b % NG, l = b / NG; const long lo = (long)g.total * grp / NG, hi = (long)g.total * (grp + 1) / NG; id = lo + l; return id < hi; };
    long id;
    if (!wg_id(blockIdx.x, id)) return;                                  // (a slot without rows: its sums are zeros, the sweeps know)
    const bool writer = id == 0 && threadIdx.x == 0;
    if (threadIdx.x < 4) { S.qtag[threadIdx.x] = 0u; S.wtag[threadIdx.x] = 0u; }
    for (int i = threadIdx.x; i < 4 * 2 * 64; i += SR_NT) { (&S.cst[0][0][0])[i] = 0.0f; (&S.crx[0][0][0])[i] = 0.0f; }
    __syncthreads();                                                      // (the only barrier of the launch)

    const int strip = (int)(id % g.nstrips), seg = (int)(id / g.nstrips) * (SR_NT / 64) + wave;
    int ya = seg * g.R, yb = ya + g.R;
    if (yb > H) yb = H;
    if (ya > H) ya = H;
    const int nr = yb - ya;                                               // rows of this wave (0: a wave of the last workgroup row without a segment)
    const int x0 = strip * PM_USE - 2 + 2 * lane;
    const bool xin = x0 >= 0 && x0 < W;
    const bool xout = xin && lane >= 1 && lane <= 62;
    const int xcl = x0 < 0 ? 0 : x0 > W - 2 ? W - 2 : x0;
    const long N = (long)W * H;
    const M2 xp1 = { true, x0 + 2 < W };
    const v2f cxc = { coef0(cm, x0), coef0(cm, x0 + 1) }, cxm = { coef0(cm, x0 - 1), cxc.x }, cxp = { cxc.y, coef0(cm, x0 + 2) };
    const unsigned seq = __hip_atomic_load(ctl + SR_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // tag of iteration k: seq + k + 1

    // ---- who my neighbours are (a neighbour exists = somebody publishes the granules I would wait for).  A segment with fewer than R rows is the last of its strip.
    const int wid = strip * g.nseg + seg;
    const bool has_up = nr > 0 && seg > 0, has_dn = nr == R && yb < H;
    const bool has_lf = nr > 0 && strip > 0, has_rt = nr > 0 && strip + 1 < g.nstrips;
    const long waves = (long)g.nstrips * g.nseg;
    const rsrc_t RS_ROW = make_xrsrc(a.b.rowh), RS_COL = make_xrsrc(a.b.colh), RS_SUM = make_xrsrc(a.b.sums);
    auto rowh = [&](int par, int w, int side) { return (unsigned)(((((long)par * waves + w) * 2 + side) * 64 + lane) * 32); };      // this lane's 4 granules
    auto colh = [&](int par, int w, int side, int i) { return (unsigned)(((((long)par * waves + w) * 2 + side) * 64 + i) * 8); };
    auto sumw = [&](int par, int slot) { return (unsigned)((((long)par * THALLO_MAX_PARTIALS + slot) * 8) * 8); };
    // which column word this lane fetches at the synchronisation point, and where it belongs (S.crx word 2 * jj + pixel):
    //   lanes 0 .. 2R-1: my own rows, from the strip beside me; 2R .. 2R+3: the two rows above, from the strip beside the wave above (its last two rows);
    //   2R+4 .. 2R+7: the two rows below, from the strip beside the wave below (its first two rows)
    int c_dw = 0, c_word = 0, c_dst = 0; bool c_any = false;
    if (lane < 2 * R) { c_dw = 0; c_word = lane; c_dst = lane + 4; c_any = nr > 0; }
    else if (lane < 2 * R + 4) { c_dw = -1; c_word = 2 * (R - 2) + (lane - 2 * R); c_dst = lane - 2 * R; c_any = has_up; }
    else if (lane < 2 * R + 8) { c_dw = 1; c_word = lane - 2 * R - 4; c_dst = lane; c_any = has_dn; }
    const bool need_cl = c_any && has_lf, need_cr = c_any && has_rt;
    const bool col_pub = lane < 2 * R && nr > 0;

    // ---- state
    v2f rr[NR], pp[NR], Ap[NR], gx[NR], gy[NR], gz[NR], dl[R];
    unsigned fwx[NR], fwy[NR];
    {
        const float* Gp = a.G;
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            const int t = ya - 2 + jj;
            const int tc = t < 0 ? 0 : t > H - 1 ? H - 1 : t;
            const long i = (long)tc * W + xcl;
            const float2 r2 = *reinterpret_cast<const float2*>(a.r_in + i), p2 = *reinterpret_cast<const float2*>(a.p_in + i);
            const float2 g0 = *reinterpret_cast<const float2*>(Gp + i), g1 = *reinterpret_cast<const float2*>(Gp + N + i), g2 = *reinterpret_cast<const float2*>(Gp + 2 * N + i);
            const uint2 fw = *reinterpret_cast<const uint2*>(a.Fw + i);
            rr[jj] = v2f{ r2.x, r2.y }; pp[jj] = v2f{ p2.x, p2.y }; Ap[jj] = v2f{ 0.f, 0.f };
            gx[jj] = v2f{ g0.x, g0.y }; gy[jj] = v2f{ g1.x, g1.y }; gz[jj] = v2f{ g2.x, g2.y };
            fwx[jj] = fw.x; fwy[jj] = fw.y;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool mine = j < nr && xout;
            float2 d2 = make_float2(0.f, 0.f);
            if (mine) d2 = *reinterpret_cast<const float2*>(a.delta + (long)(ya + j) * W + x0);
            dl[j] = v2f{ d2.x, d2.y };
        }
    }

    const float aN0 = sum_partials(a.aN0.partials, a.aN0.count);
    float aN_prev = aN0;                  // alphaN_{k-1}
    float alpha = 0.0f, beta = 0.0f;
    Spin sp; sp.n = 0; sp.t0 = 0;
    bool dead = false;                    // a bounded wait ran out (here or elsewhere): no more waiting, the host raises
    const int slot = 64 * wave + lane;    // the sums slot this lane sweeps
    long sid;
    const bool slot_live = slot < grid && wg_id(slot, sid);
    const v2f Z2 = { 0.f, 0.f };

    // Publish my quarter of the sums of an iteration (the 7 words of slot 64 w + lane) to the other waves of the workgroup, wait for theirs, and add all slots up in
    // the order the launch-per-iteration path uses (last_workgroup_totals: lane-strided over the slots, then the wave butterfly) -- same bits everywhere.
    auto exchange_scalars = [&](unsigned T, const unsigned (&w7)[7], float& ad_o, double& n_o, double& s1_o, double& s2_o) {
#pragma unroll
        for (int c = 0; c < 7; ++c) S.q[wave][c][lane] = w7[c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&S.qtag[wave], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sp.n = 0; sp.t0 = 0;
        while (!dead) {
            const unsigned t0 = __hip_atomic_load(&S.qtag[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), t1 = __hip_atomic_load(&S.qtag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned t2 = __hip_atomic_load(&S.qtag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), t3 = __hip_atomic_load(&S.qtag[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (t0 == T && t1 == T && t2 == T && t3 == T) break;
            if (spin_fail(sp, ctl, 2u, 0u, T)) dead = true;
        }
        dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");            // (no instruction: keeps the column reads below the tag polls)
        float t = 0.0f; double n = 0.0, a1 = 0.0, b1 = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned* qq = &S.q[w][0][lane];
            t += __uint_as_float(qq[0]);
            n += __hiloint2double((int)qq[64], (int)qq[128]);
            a1 += __hiloint2double((int)qq[192], (int)qq[256]);
            b1 += __hiloint2double((int)qq[320], (int)qq[384]);
        }
        ad_o = wave_sum_all(t);
        n_o = wave_sum_all_f64(n); s1_o = wave_sum_all_f64(a1); s2_o = wave_sum_all_f64(b1);
    };
    // alphaD_k, betaN_k from the sums: alpha_k = alphaN_k / alphaD_k, betaN_k = N - 2 alpha S1 + alpha^2 S2 (k_pmarch's deferred finish, its expressions)
    auto scalars_from_sums = [&](float aN, float ad, double n, double a1, double b1, float& aD_o, float& bN_o) {
        const float al = safe_div<false>(aN, ad);
        double bnd = n - 2.0 * (double)al * a1 + (double)al * (double)al * b1;
        if (!(bnd > 0.0)) bnd = 0.0;
        aD_o = ad; bN_o = (float)bnd;
    };

    for (int k = 0; k < a.L; ++k) {
        const unsigned T = seq + (unsigned)k + 1u, Tp = T - 1u;
        const int par = k & 1, parp = par ^ 1;
        SR_STAMP(k, 0);
        if (k > 0) {
            // ---- the one synchronisation point: my quarter of the sums of iteration k-1, the two rows of A p_{k-1} from the wave above and from the wave below, one
            // word per lane of the columns from the strips to the left / right.  Everything is polled in ONE loop, all loads of a pass in flight together.
            unsigned w7[7]; float cvl = 0.f, cvr = 0.f;
            v2f ru0 = Z2, ru1 = Z2, rd0 = Z2, rd1 = Z2;
#pragma unroll
            for (int c = 0; c < 7; ++c) w7[c] = 0u;
            {
                const bool need_u = xout && has_up, need_d = xout && has_dn, need_s = slot_live;
                const unsigned usrc = rowh(parp, has_up ? wid - 1 : wid, 1), dsrc = rowh(parp, has_dn ? wid + 1 : wid, 0);
                const unsigned ssrc = sumw(parp, slot);
                const unsigned clsrc = colh(parp, need_cl ? wid - g.nseg + c_dw : wid, 1, c_word), crsrc = colh(parp, need_cr ? wid + g.nseg + c_dw : wid, 0, c_word);
                bool ok_s = !need_s, ok_u = !need_u, ok_d = !need_d, ok_cl = !need_cl, ok_cr = !need_cr;
                sp.n = 0; sp.t0 = 0;
                while (!(ok_s && ok_u && ok_d && ok_cl && ok_cr) && !dead) {
                    asm volatile("" ::: "memory");                     // (every pass re-reads: nothing may be hoisted out of the loop)
                    u32x4 vs[4], vu[2], vd[2]; u32x2 vcl, vcr;
                    if (!ok_s) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) vs[c] = ld2g(RS_SUM, ssrc + 16 * c);
                    }
                    if (!ok_u) { vu[0] = ld2g(RS_ROW, usrc); vu[1] = ld2g(RS_ROW, usrc + 16); }
                    if (!ok_d) { vd[0] = ld2g(RS_ROW, dsrc); vd[1] = ld2g(RS_ROW, dsrc + 16); }
                    if (!ok_cl) vcl = ld1g(RS_COL, clsrc);
                    if (!ok_cr) vcr = ld1g(RS_COL, crsrc);
                    if (!ok_s) {
                        w7[0] = vs[0].x; w7[1] = vs[0].z; w7[2] = vs[1].x; w7[3] = vs[1].z; w7[4] = vs[2].x; w7[5] = vs[2].z; w7[6] = vs[3].x;
                        ok_s = vs[0].y == Tp && vs[0].w == Tp && vs[1].y == Tp && vs[1].w == Tp && vs[2].y == Tp && vs[2].w == Tp && vs[3].y == Tp;
                    }
                    if (!ok_u) {
                        ru0 = v2f{ __uint_as_float(vu[0].x), __uint_as_float(vu[0].z) }; ru1 = v2f{ __uint_as_float(vu[1].x), __uint_as_float(vu[1].z) };
                        ok_u = vu[0].y == Tp && vu[0].w == Tp && vu[1].y == Tp && vu[1].w == Tp;
                    }
                    if (!ok_d) {
                        rd0 = v2f{ __uint_as_float(vd[0].x), __uint_as_float(vd[0].z) }; rd1 = v2f{ __uint_as_float(vd[1].x), __uint_as_float(vd[1].z) };
                        ok_d = vd[0].y == Tp && vd[0].w == Tp && vd[1].y == Tp && vd[1].w == Tp;
                    }
                    if (!ok_cl) { cvl = __uint_as_float(vcl.x); ok_cl = vcl.y == Tp; }
                    if (!ok_cr) { cvr = __uint_as_float(vcr.x); ok_cr = vcr.y == Tp; }
                    if (!(ok_s && ok_u && ok_d && ok_cl && ok_cr) && spin_fail(sp, ctl, !ok_s ? 1u : !(ok_u && ok_d) ? 3u : 4u, (unsigned)wid, Tp)) dead = true;
                }
                if (need_u) { Ap[0] = ru0; Ap[1] = ru1; }
                if (need_d) { Ap[R + 2] = rd0; Ap[R + 3] = rd1; }
            }
            dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
            SR_STAMP(k, 1);
            // the columns go through LDS to the two lanes that hold them (lane 0 / 63); same wave: program order + lgkmcnt(0)
            if (need_cl) S.crx[wave][0][c_dst] = cvl;
            if (need_cr) S.crx[wave][1][c_dst] = cvr;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((lane == 0 && has_lf) || (lane == 63 && has_rt)) {
                const float* cx = &S.crx[wave][lane == 0 ? 0 : 1][0];
#pragma unroll
                for (int jj = 0; jj < NR; ++jj) Ap[jj] = v2f{ cx[2 * jj], cx[2 * jj + 1] };
            }
            float ad; double n, a1, b1;
            exchange_scalars(Tp, w7, ad, n, a1, b1);
            float aD, bN;
            scalars_from_sums(aN_prev, ad, n, a1, b1, aD, bN);
            alpha = safe_div<false>(aN_prev, aD);
            beta = safe_div<false>(bN, aN_prev);
            if (writer) { a.words[2 * (k - 1)] = aD; a.words[2 * (k - 1) + 1] = bN; }
            aN_prev = bN;
        }
        SR_STAMP(k, 2);
        // ---- r_k = r_{k-1} - alpha A p_{k-1} ; delta += alpha p_{k-1} ; p_k = r_k + beta p_{k-1}     (every row I hold, halo included; k_pmarch<UPD>'s expressions)
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            const int t = ya - 2 + jj;
            const bool ok = nr > 0 && xin && t >= 0 && t < H;
            v2f rk = rr[jj];
            if (k > 0) rk = fma2(-alpha, Ap[jj], rk);
            const v2f pv = pp[jj];
            if (k > 0 && jj >= 2 && jj < R + 2) dl[jj - 2] = fma2(alpha, pv, dl[jj - 2]);
            const v2f zk = rk;
            v2f v0 = zk + beta * pv;
            v0 = sel(ok, v0, Z2);
            rr[jj] = rk; pp[jj] = v0;
        }
        SR_STAMP(k, 3);
        // ---- A p_k for my rows (the row step of k_pmarch: dB -> U_h, U_v -> T -> J^T, the Laplacian rows), the four sums
        v2f acc = Z2; Sums3 sm;
        {
            v2f dB[NR], Uh[NR], Uv[NR], Tt[NR], Rr[NR][3];
            unsigned Fl[NR]; float Cy[NR];
#pragma unroll
            for (int jj = 0; jj < NR; ++jj) {
                const int t = ya - 2 + jj;
                const bool ok = nr > 0 && xin && t >= 0 && t < H;
                Fl[jj] = ok ? ((fwx[jj] & 0xffu) | ((fwy[jj] & 0xffu) << 8)) : 0u;
                Cy[jj] = coef1(cm, t + g.yoff);
                dB[jj] = Z2; Uh[jj] = Z2; Uv[jj] = Z2; Tt[jj] = Z2;
#pragma unroll
                for (int c = 0; c < 3; ++c) Rr[jj][c] = Z2;
            }
#pragma unroll
            for (int jj = 1; jj < NR; ++jj) {                       // dB(row jj) from p(jj), p(jj - 1); U_h(jj)
                const int t = ya - 2 + jj;
                const bool ok = nr > 0 && xin && t >= 0 && t < H;
                const v2f v0 = pp[jj], v1 = pp[jj - 1];
                const v2f vl0 = nbL(v0);
                const v2f dB0 = sel(ok, gx[jj] * v0 + gy[jj] * vl0 + gz[jj] * v1, Z2);
                const v2f dBr = nbR(dB0);
                const v2f wx = cm.wg * v2f{ (float)((fwx[jj] >> 8) & 0xffu), (float)((fwy[jj] >> 8) & 0xffu) };
                const v2f wy = cm.wg * v2f{ (float)((fwx[jj] >> 16) & 0xffu), (float)((fwy[jj] >> 16) & 0xffu) };
                const M2 wn0 = { ok && (wx.x != 0.0f || wy.x != 0.0f), ok && (wx.y != 0.0f || wy.y != 0.0f) };
                dB[jj] = dB0;
                Uh[jj] = sel(wn0, wx * (wx * (dB0 - dBr)), Z2);
            }
#pragma unroll
            for (int jj = 1; jj < NR - 1; ++jj) {                   // U_v(jj) from dB(jj), dB(jj + 1); the Laplacian rows R(jj) from p(jj - 1), p(jj), p(jj + 1)
                const int t = ya - 2 + jj;
                const bool ok = nr > 0 && xin && t >= 0 && t < H;
                const v2f wx = cm.wg * v2f{ (float)((fwx[jj] >> 8) & 0xffu), (float)((fwy[jj] >> 8) & 0xffu) };
                const v2f wy = cm.wg * v2f{ (float)((fwx[jj] >> 16) & 0xffu), (float)((fwy[jj] >> 16) & 0xffu) };
                const M2 wn = { ok && (wx.x != 0.0f || wy.x != 0.0f), ok && (wx.y != 0.0f || wy.y != 0.0f) };
                Uv[jj] = sel(wn, wy * (wy * (dB[jj] - dB[jj + 1])), Z2);
                const v2f v0 = pp[jj + 1], v1 = pp[jj], v2 = pp[jj - 1];
                const v2f vl1 = nbL(v1), vr1 = nbR(v1);
                const float cy0 = Cy[jj + 1], cy1 = Cy[jj], cy2 = Cy[jj - 1];
                const M2 f2b = bit(Fl[jj], 2u);
                Rr[jj][0] = sel(f2b, cm.ws * (4.0f * (cxc * v1) - cxm * vl1 - cxc * v2 - cxp * vr1 - cxc * v0), Z2);
                Rr[jj][1] = sel(f2b, cm.ws * (4.0f * (cy1 * v1) - cy1 * vl1 - cy2 * v2 - cy1 * vr1 - cy0 * v0), Z2);
                Rr[jj][2] = sel(f2b, cm.ws * (4.0f * v1 - vl1 - v2 - vr1 - v0), Z2);
            }
#pragma unroll
            for (int jj = 2; jj < NR - 1; ++jj) {                   // T(jj)
                v2f T1 = Uh[jj] + Uv[jj];
                T1 -= nbL(Uh[jj]);
                T1 -= Uv[jj - 1];
                Tt[jj] = T1;
            }
#pragma unroll
            for (int jj = 2; jj < R + 2; ++jj) {                    // output row y = ya + jj - 2
                const int y = ya - 2 + jj;
                const v2f T2 = Tt[jj], T1 = Tt[jj + 1];
                const v2f gT2r = nbR(gy[jj] * T2);
                v2f Rl[3], Rq[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { Rl[c] = nbL(Rr[jj][c]); Rq[c] = nbR(Rr[jj][c]); }
                if (jj - 2 < nr && xout) {
                    const v2f vc = pp[jj];
                    v2f s = Z2;
                    s = sel(bit(Fl[jj], 1u), s + cm.wp * (cm.wp * vc), s);
                    s += gx[jj] * T2;
                    s = sel(xp1, s + gT2r, s);
                    if (y + 1 < H) s += gz[jj + 1] * T1;
                    {
                        v2f lap;
                        lap = 4.0f * Rr[jj][0] - Rl[0] - Rr[jj - 1][0] - Rq[0] - Rr[jj + 1][0]; s += cm.ws * (cxc * lap);
                        lap = 4.0f * Rr[jj][1] - Rl[1] - Rr[jj - 1][1] - Rq[1] - Rr[jj + 1][1]; s += cm.ws * (Cy[jj] * lap);
                        lap = 4.0f * Rr[jj][2] - Rl[2] - Rr[jj - 1][2] - Rq[2] - Rr[jj + 1][2]; s += cm.ws * (1.0f * lap);
                    }
                    acc += vc * s;
                    const v2f rk = rr[jj];
                    sm.add(1.0f, rk.x, s.x); sm.add(1.0f, rk.y, s.y);
                    Ap[jj] = s;
                    if (lane == 1 || lane == 62) { float* d = &S.cst[wave][lane == 1 ? 0 : 1][2 * (jj - 2)]; d[0] = s.x; d[1] = s.y; }
                }
            }
        }
        SR_STAMP(k, 4);
        // ---- publish: the boundary rows, my two columns (one store instruction each), then the workgroup's sums (wave butterflies -> LDS -> wave 0 adds the four
        // waves up in order and publishes 7 granules)
        {
            if (xout && has_up) { const unsigned d = rowh(par, wid, 0); st2g(RS_ROW, d, T, Ap[2].x, Ap[2].y); st2g(RS_ROW, d + 16, T, Ap[3].x, Ap[3].y); }
            if (xout && has_dn) { const unsigned d = rowh(par, wid, 1); st2g(RS_ROW, d, T, Ap[R].x, Ap[R].y); st2g(RS_ROW, d + 16, T, Ap[R + 1].x, Ap[R + 1].y); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the columns written to LDS above)
            if (col_pub && has_lf) st1g(RS_COL, colh(par, wid, 0, lane), T, __float_as_uint(S.cst[wave][0][lane]));
            if (col_pub && has_rt) st1g(RS_COL, colh(par, wid, 1, lane), T, __float_as_uint(S.cst[wave][1][lane]));
            const float accf = acc.x + acc.y;
            const float wa = wave_sum_all(accf); const double w0 = wave_sum_all_f64(sm.n), w1 = wave_sum_all_f64(sm.s1), w2 = wave_sum_all_f64(sm.s2);
            if (lane == 0) { S.wa[wave] = wa; S.wd[wave][0] = w0; S.wd[wave][1] = w1; S.wd[wave][2] = w2; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&S.wtag[wave], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == 0) {
                sp.n = 0; sp.t0 = 0;
                while (!dead) {
                    const unsigned t1 = __hip_atomic_load(&S.wtag[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), t2 = __hip_atomic_load(&S.wtag[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned t3 = __hip_atomic_load(&S.wtag[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (t1 == T && t2 == T && t3 == T) break;
                    if (spin_fail(sp, ctl, 5u, 0u, T)) dead = true;
                }
                dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 7) {
                    float s = 0.0f; double b0 = 0.0, b1 = 0.0, b2 = 0.0;
                    for (int w = 0; w < 4; ++w) { s += S.wa[w]; b0 += S.wd[w][0]; b1 += S.wd[w][1]; b2 += S.wd[w][2]; }
                    const double pick = lane < 3 ? b0 : lane < 5 ? b1 : b2;
                    const unsigned word = lane == 0 ? __float_as_uint(s) : (lane & 1) ? (unsigned)__double2hiint(pick) : (unsigned)__double2loint(pick);
                    st1g(RS_SUM, sumw(par, blockIdx.x) + 8 * lane, T, word);
                }
            }
        }
        SR_STAMP(k, 5);
    }
    // ---- what L launches would have left behind: r_{L-1}, p_{L-1}, A p_{L-1}, delta (without its last term); the last iteration's two words
    if (a.L > 0) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < nr && xout) {
                const long i = (long)(ya + j) * W + x0;
                *reinterpret_cast<float2*>(a.r_out + i) = make_float2(rr[j + 2].x, rr[j + 2].y);
                *reinterpret_cast<float2*>(a.p_out + i) = make_float2(pp[j + 2].x, pp[j + 2].y);
                *reinterpret_cast<float2*>(a.A_out + i) = make_float2(Ap[j + 2].x, Ap[j + 2].y);
                *reinterpret_cast<float2*>(a.delta + i) = make_float2(dl[j].x, dl[j].y);
            }
        }
        if (id == 0 || a.X != nullptr) {      // (uniform per workgroup: all four waves take part in the last sweep; with the update of the unknowns riding along every workgroup needs alpha_{L-1})
            const unsigned T = seq + (unsigned)a.L; const int par = (a.L - 1) & 1;
            unsigned w7[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) w7[c] = 0u;
            bool ok = !slot_live;
            sp.n = 0; sp.t0 = 0;
            while (!ok && !dead) {
                ok = true;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < 7; ++c) { const u32x2 v = ld1g(RS_SUM, sumw(par, slot) + 8 * c); w7[c] = v.x; ok = ok && v.y == T; }
                if (!ok && spin_fail(sp, ctl, 1u, (unsigned)slot, T)) dead = true;
            }
            float ad; double n, a1, b1;
            exchange_scalars(T, w7, ad, n, a1, b1);
            float aD, bN;
            scalars_from_sums(aN_prev, ad, n, a1, b1, aD, bN);
            if (writer) {
                a.words[2 * (a.L - 1)] = aD; a.words[2 * (a.L - 1) + 1] = bN;
                // the next launch's tags start behind this one's (seq + 1 .. seq + L were used): the counter lives on the device (replay-safe) and is advanced by the one thread
                // that is through only when every workgroup has published its last sums, i.e. has long read it
                __hip_atomic_store(ctl + SR_SEQ, seq + (unsigned)a.L + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (a.X != nullptr) {      // PCGLinearUpdate (gauss_newton.t:901-906) riding along: X += delta + alpha_{L-1} p_{L-1} on my rows (k_linear_update's expressions)
                const float al = safe_div<false>(aN_prev, aD);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (j < nr && xout) {
                        const long i = (long)(ya + j) * W + x0;
                        const float2 xo = *reinterpret_cast<const float2*>(a.X + i);
                        const float d0 = __builtin_fmaf(al, pp[j + 2].x, dl[j].x), d1 = __builtin_fmaf(al, pp[j + 2].y, dl[j].y);
                        *reinterpret_cast<float2*>(a.X + i) = make_float2(xo.x + d0, xo.y + d1);
                    }
                }
            }
        }
    }
}

inline SrGeo make_sr_geo(int W, int H, int yoff, int R)
{
    SrGeo g; g.W = W; g.H = H; g.yoff = yoff; g.R = R;
    g.nstrips = (W + PM_USE - 1) / PM_USE;
    g.nseg = (H + R - 1) / R;
    g.nwgrow = (g.nseg + SR_NT / 64 - 1) / (SR_NT / 64);
    g.total = g.nstrips * g.nwgrow;
    return g;
}

int g_sr_cap = 0;        // tests: workgroup budget (0 = the device's CU count: one workgroup per CU, all of them resident at once)
int g_sr_rows = 0;       // tests / tools: rows per segment (0 = automatic)

// rows per wave segment, or 0 = the shape does not fit
inline int sr_rows(int W, int H)
{
    if (W < 2 || (W & 1) || H < SR_MIN_R) return 0;
    if (16.0 * (double)W * (double)H >= 2147483648.0) return 0;
    // every workgroup must be RESIDENT (they wait for each other) and a wave's sums sweep covers slots 64 * wave + lane < 256: at most min(CUs, 256) workgroups, one per CU
    long cap = g_sr_cap > 0 ? g_sr_cap : thallo_hip_device_cu_count();
    if (cap > 256) cap = 256;
    const int nstrips = (W + PM_USE - 1) / PM_USE;
    const int R = g_sr_rows > 0 ? g_sr_rows : march_rows_per_segment(H, nstrips, SR_NT / 64, cap, SR_MIN_R);
    if (R < SR_MIN_R || R > SR_MAX_R) return 0;
    const SrGeo g = make_sr_geo(W, H, 0, R);
    if ((g.total + 7) / 8 * 8 > cap || (g.total + 7) / 8 * 8 > THALLO_MAX_PARTIALS) return 0;
    return R;
}

// the plan's exchange memory: [control words | sums records | column granules | row granules]   (u64 units)
struct SrLayout { long ctl, sums, colh, rowh, bytes; };
inline SrLayout sr_layout(const SrGeo& g)
{
    const long waves = (long)g.nstrips * g.nseg;
    SrLayout l;
    l.ctl = 0; l.sums = 32; l.colh = l.sums + 2L * 8 * THALLO_MAX_PARTIALS; l.rowh = l.colh + 2 * waves * 2 * 64;
    l.bytes = (l.rowh + 2 * waves * 2 * 64 * 4) * (long)sizeof(u64) + 256;
    return l;
}

template <int R> int sr_launch_r(const SrArgs& a, hipStream_t s)
{
    const int grid = (a.g.total + 7) / 8 * 8;
    {   // co-residency is a precondition, not an assumption: the kernel's workgroups wait for each other (asked once per instantiation)
        static int fits = 0;
        if (fits == 0) {
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sfs_resident<R>, SR_NT, 0);
            fits = (e == hipSuccess && per_cu >= 1) ? per_cu : -1;
        }
        if (fits < 0 || (long)fits * thallo_hip_device_cu_count() < grid) return -(int)hipErrorNotSupported;
    }
    hipLaunchKernelGGL((k_sfs_resident<R>), dim3(grid), dim3(SR_NT), 0, s, a);
    int e = check_launch(); return e ? e : grid;
}

}  // namespace

extern "C" {

#ifdef THALLO_MARCH_SWEEP
int thallo_hip_debug_stamps_sfs_resident(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_sr), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif

void thallo_hip_sfs_resident_debug_set(int what, int value) { if (what == 0) g_sr_rows = value; if (what == 1) g_sr_cap = value; }

/* rows per wave segment of the resident PCG kernel on a W x H image, or 0: the shape does not fit the chip's registers and the caller runs one launch per PCG iteration */
int thallo_hip_sfs_resident_rows(int W, int H) { return sr_rows(W, H); }

/* bytes of exchange memory a plan needs for the resident kernel (zero-filled by the caller once; layout private to this file) */
long thallo_hip_sfs_resident_bytes(int W, int H)
{
    if (W < 2 || (W & 1) || H < SR_MIN_R) return 0;
    return sr_layout(make_sr_geo(W, H, 0, SR_MIN_R)).bytes;       // (sized for the smallest R: the largest wave count)
}

/* The PCG loop of one Gauss-Newton step in one launch: L iterations from what thallo_hip_sfs_pcg_init left on PACKED planes (r_0 in r_in, zeros in p_in and delta,
 * alphaN_0), leaving what L launches of thallo_hip_sfs_pcg_iter_deferred leave: r_{L-1}, A p_{L-1}, p_{L-1} in the *_out planes, delta without its last term, and
 * words[2k] = alphaD_k, words[2k + 1] = betaN_k; with X != NULL also PCGLinearUpdate (X += delta + alpha_{L-1} p_{L-1}: thallo_hip_linear_update's result, bit for bit).
 * The *_out planes may be the *_in planes.  xbuf: thallo_hip_sfs_resident_bytes() bytes, zeroed once by the caller,
 * private to the plan.  Returns the number of workgroups (> 0), -hipErrorNotSupported when the shape does not fit, another negative hipError_t on failure.
 * Replaces gauss_newton.t:1615-1687 for shapes whose solver state fits the chip's registers. */
int thallo_hip_sfs_pcg_resident(int W, int H, int yoff, const float* host_params, const float* G, const float* Fw,
                                const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                                thallo_sum_t alphaN0, float* words, float* X, void* xbuf, int L, thallo_stream_t stream)
{
    if (H < 1 || (W & 1) || W < 2 || L < 1 || !host_params) return -(int)hipErrorInvalidValue;
    if (!G || !Fw || !r_in || !p_in || !r_out || !Ap_out || !p_out || !delta || !words || !xbuf || !alphaN0.partials) return -(int)hipErrorInvalidValue;
    const int R = sr_rows(W, H);
    if (R <= 0) return -(int)hipErrorNotSupported;
    SrArgs a; memset(&a, 0, sizeof(a));
    a.g = make_sr_geo(W, H, yoff, R);
    {
        const SrLayout l = sr_layout(a.g);
        u64* base = reinterpret_cast<u64*>(xbuf);
        a.b.rowh = base + l.rowh; a.b.colh = base + l.colh; a.b.sums = base + l.sums; a.b.ctl = reinterpret_cast<unsigned*>(base + l.ctl);
    }
    a.cm = cam_of(host_params);
    a.G = G; a.Fw = reinterpret_cast<const unsigned*>(Fw);
    a.r_in = r_in; a.p_in = p_in; a.r_out = r_out; a.A_out = Ap_out; a.p_out = p_out; a.delta = delta;
    a.aN0 = alphaN0; a.words = words; a.X = X; a.L = L;
    hipStream_t s = (hipStream_t)stream;
    switch (R) {
        case 2: return sr_launch_r<2>(a, s); case 3: return sr_launch_r<3>(a, s); case 4: return sr_launch_r<4>(a, s); case 5: return sr_launch_r<5>(a, s);
        case 6: return sr_launch_r<6>(a, s); case 7: return sr_launch_r<7>(a, s); case 8: return sr_launch_r<8>(a, s);
        default: return -(int)hipErrorNotSupported;
    }
}

/* the error word of a plan's resident launches: 1 = a bounded wait ran out (a workgroup was not resident, or a granule never arrived); clear != 0 resets it.
 * pm (5 words, may be NULL): what the first timed-out wait was for.  Synchronises the stream.  spin_ms >= 0 sets the bound (0 = the 2 s default). */
int thallo_hip_sfs_resident_status(void* xbuf, int clear, int spin_ms, unsigned* pm, thallo_stream_t stream)
{
    if (!xbuf) return -(int)hipErrorInvalidValue;
    unsigned* ctl = reinterpret_cast<unsigned*>(xbuf);
    hipStream_t s = (hipStream_t)stream;
    unsigned w[SR_CTL_WORDS];
    if (hipMemcpyAsync(w, ctl, sizeof(w), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    if (pm) for (int i = 0; i < 5; ++i) pm[i] = w[SR_PM + i];
    if (clear && w[SR_ERR]) { const unsigned z = 0; if (hipMemcpyAsync(ctl + SR_ERR, &z, sizeof(z), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (spin_ms >= 0) { const unsigned v = (unsigned)spin_ms; if (hipMemcpyAsync(ctl + SR_SPIN_MS, &v, sizeof(v), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    return (int)w[SR_ERR];
}

}  // extern "C"
