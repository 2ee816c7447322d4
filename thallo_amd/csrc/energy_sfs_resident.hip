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
constexpr int SR_MIN_R = 2, SR_MAX_R = 12, SR_MAX_R_LM = 7;       // (GN: 14 registers per held row and lane, LM: 18 + the own rows' CtC and b, beside the three rows of temporaries the rolling row step keeps: LM at 7 rows spills one register, 14 / 8 rows per wave tens) // rows per segment the kernel is instantiated for (two halo rows come from ONE neighbouring segment: R >= 2; 14 registers per held row and lane)

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
constexpr int SR_REC = 16;                // granules per sums record (7 used by GN, 13 by LM)

struct SrGeo { int W, H, yoff, R, nstrips, nseg, nwgrow, total, ab; };      // ab: A/B bits (tools): 1 = row granules row-major (a wave's 16-byte loads contiguous), 2 = 128-byte sums records in GN too; 8 = every halo row through global memory (none through LDS); 4 (tests) = FAULT INJECTION: workgroup 1 never publishes the sums of iteration 2 (what a workgroup that is not resident looks like to the others)

// exchange buffers of one plan (thallo_hip_sfs_resident_bytes); parity = iteration & 1
struct SrBufs {
    u64* rowh;        // [2 parity][waves][2 sides: 0 = the wave's FIRST two rows (for the wave above), 1 = its LAST two (for the wave below)][64 lanes][4: row a px 0, px 1, row b px 0, px 1]
    u64* colh;        // [2 parity][waves][2 sides: 0 = lane 1's pixels (for the strip to the left), 1 = lane 62's (for the strip to the right)][64: word 2 * row + pixel]
    u64* sums;        // [2 parity][1024 workgroups][8 (LM: 16): alphaD, N hi, N lo, S1 hi, S1 lo, S2 hi, S2 lo (LM: U, T1, T2 hi / lo), -]
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
    float* X;                                     // the unknowns, or NULL (GN only): PCGLinearUpdate stays a launch of its own
    int L;
    // LM
    const float* pre; const float* ctc;           // M^-1 and CtC of PCGFinalizeDiagonal (b = r_0)
    float* state; float q_tol;                    // lm state words ([0] Q0, [1] gate, [2] iterations done at the stop); the zeta test's tolerance
    float* prevX; float* t0_out; float* t1_out;   // savePreviousUnknowns' copy; per-workgroup partials of delta . J^T J delta and delta . b
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
    unsigned q[4][13][64];            // per wave: the 7 (LM: 13) words of the 64 slots it swept
    float wa[4]; double wd[4][6];     // per wave: alphaD part, {N, S1, S2} (LM: + {U, T1, T2}) parts
    float cst[4][2][64];              // per wave: lane 1's / lane 62's A p of its rows (word 2 * row + pixel), so that ONE store instruction publishes a column
    float crx[4][2][64];              // per wave: the received columns (word 2 * held row + pixel), for lanes 0 / 63 to pick up
    unsigned rtag[2][4][2];           // [parity][wave][side]: that wave's first two (0) / last two (1) rows of A p are in rrow, for the neighbouring wave of the same workgroup
    float rrow[2][4][2][4][64];       // the y halo between the stacked waves of a workgroup never leaves the CU (row a px 0, px 1, row b px 0, px 1 per lane)
    float red[32];                    // block_store_partials (the LM model cost's two sums)
};

// LM = the Levenberg-Marquardt loop (gauss_newton.t:1615-1687 with every UsesLambda() branch; k_pmarch<.., UPD, LMQ>'s iteration): A = J^T J + CtC, z = M^-1 r, blind
// divisions, the three sums of q's expansion beside {N, S1, S2}, the zeta test after every iteration (every workgroup for itself, from the same sums: the same decision
// everywhere) -- and behind the loop the update of delta it still owes, the model cost's J^T J delta and two dot products, savePreviousUnknowns and PCGLinearUpdate
// (k_pmarch<MODEL>'s launch), all from the registers the loop leaves.
template <int R, bool LM>
__global__ __launch_bounds__(SR_NT, 1) void k_sfs_resident(SrArgs a)
{
    constexpr int NR = R + 4;             // held rows: jj = 0, 1 the rows above, 2 .. R + 1 my own, R + 2, R + 3 the rows below (row t = ya - 2 + jj)
    constexpr int NQ = LM ? 6 : 3;        // double sums per record
    constexpr int NWD = 1 + 2 * NQ;       // words per record: alphaD, then (hi, lo) of every double
    constexpr int NLD = (NWD + 1) / 2;    // 16-byte loads that fetch a record
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
    if (!wg_id(blockIdx.x, id)) {                                         // (a slot without rows: its sums are zeros, the sweeps know; the model cost's partial slots are the caller's)
        if (LM && threadIdx.x == 0) { a.t0_out[blockIdx.x] = 0.0f; a.t1_out[blockIdx.x] = 0.0f; }
        return;
    }
    const bool writer = id == 0 && threadIdx.x == 0;
    if (threadIdx.x < 4) { S.qtag[threadIdx.x] = 0u; S.wtag[threadIdx.x] = 0u; }
    if (threadIdx.x < 16) (&S.rtag[0][0][0])[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < 4 * 2 * 64; i += SR_NT) { (&S.cst[0][0][0])[i] = 0.0f; (&S.crx[0][0][0])[i] = 0.0f; }
    __syncthreads();

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
    // which of my halo rows come through LDS (the neighbouring wave sits in my workgroup: three of four boundaries) and which through global memory (ab bit 3: all through global, A/B)
    const bool lds_rows = !(g.ab & 8);
    const bool up_lds = lds_rows && has_up && wave > 0, dn_lds = lds_rows && has_dn && wave < 3;
    const long waves = (long)g.nstrips * g.nseg;
    const rsrc_t RS_ROW = make_xrsrc(a.b.rowh), RS_COL = make_xrsrc(a.b.colh), RS_SUM = make_xrsrc(a.b.sums);
    // (a lane's two 16-byte pieces side by side: its two loads ask for ONE line.  Measured, same box, 640 x 480 GN through Thallo_ProblemStep: 8.23 us per PCG iteration against
    //  9.08 with the rows stored row-major -- a wave's 64 x 16 bytes of one load instruction contiguous, the lane's second piece 1 KB away; tools/sfs_resident_probe.py ab)
    const unsigned row2 = (g.ab & 1) ? 1024u : 16u, rowl = (g.ab & 1) ? 16u : 32u;
    auto rowh = [&](int par, int w, int side) { return (unsigned)((((long)par * waves + w) * 2 + side) * 2048 + lane * rowl); };      // this lane's four granules: the first row's two at +0, the second row's at +row2
    auto colh = [&](int par, int w, int side, int i) { return (unsigned)(((((long)par * waves + w) * 2 + side) * 64 + i) * 8); };
    // a workgroup's record: [par][slot][words]; 64 bytes (GN: two records per 128-byte line) or 128 (LM).  (Measured and dropped: the record stored word-pair-major, so that a
    // sweeping wave reads 1 KB of contiguous memory per load instruction -- a record's pieces then sit in 4 / 7 lines that eight workgroups of different XCDs write into:
    // GN 68.8 -> 84.2 us per 10-iteration launch at 640 x 480, LM 118 -> 114.7.  And a 128-byte stride for GN's 56 bytes: 68.8 -> 77: the cost is per 128-byte line touched.)
    auto sumw = [&](int par, int c, int slot) { return (unsigned)((((long)par * THALLO_MAX_PARTIALS + slot) * ((LM || (g.ab & 2)) ? SR_REC : SR_REC / 2)) * 8 + c * 16); };
    // which column word this lane fetches at the synchronisation point, and where it belongs (S.crx word 2 * jj + pixel):
    //   lanes 0 .. 2R-1: my own rows, from the strip beside me; 2R .. 2R+3: the two rows above, from the strip beside the wave above (its last two rows);
    //   2R+4 .. 2R+7: the two rows below, from the strip beside the wave below (its first two rows)
    int c_dw = 0, c_word = 0, c_dst = 0; bool c_any = false;
    if (lane < 2 * R) { c_dw = 0; c_word = lane; c_dst = lane + 4; c_any = nr > 0; }
    else if (lane < 2 * R + 4) { c_dw = -1; c_word = 2 * (R - 2) + (lane - 2 * R); c_dst = lane - 2 * R; c_any = has_up; }
    else if (lane < 2 * R + 8) { c_dw = 1; c_word = lane - 2 * R - 4; c_dst = lane; c_any = has_dn; }
    const bool need_cl = c_any && has_lf, need_cr = c_any && has_rt;
    const bool col_pub = lane < 2 * R && nr > 0;
    const v2f Z2 = { 0.f, 0.f };
    auto row_ok = [&](int jj) __attribute__((always_inline)) { const int t = ya - 2 + jj; return nr > 0 && xin && t >= 0 && t < H; };

    // ---- state.  delta: my own rows (GN); every held row (LM: the model cost's J^T J delta needs delta on the halo, and it is formed there from what the halo holds anyway)
    v2f rr[NR], pp[NR], Ap[NR], gx[NR], gy[NR], gz[NR], dl[NR], mi[LM ? NR : 1], ct[LM ? R : 1], bb[LM ? R : 1];
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
            rr[jj] = v2f{ r2.x, r2.y }; pp[jj] = v2f{ p2.x, p2.y }; Ap[jj] = Z2; dl[jj] = Z2;
            gx[jj] = v2f{ g0.x, g0.y }; gy[jj] = v2f{ g1.x, g1.y }; gz[jj] = v2f{ g2.x, g2.y };
            fwx[jj] = fw.x; fwy[jj] = fw.y;
            if (LM) { const float2 m2 = *reinterpret_cast<const float2*>(a.pre + i); mi[jj] = v2f{ m2.x, m2.y }; }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool mine = j < nr && xout;
            float2 d2 = make_float2(0.f, 0.f), c2 = make_float2(0.f, 0.f);
            if (mine) { const long i = (long)(ya + j) * W + x0; d2 = *reinterpret_cast<const float2*>(a.delta + i); if (LM) c2 = *reinterpret_cast<const float2*>(a.ctc + i); }
            dl[j + 2] = v2f{ d2.x, d2.y };
            if (LM) { ct[j] = v2f{ c2.x, c2.y }; bb[j] = rr[j + 2]; }      // b = r_0 (PCGFinalizeDiagonal: gauss_newton.t:962)
        }
    }

    const float aN0 = sum_partials(a.aN0.partials, a.aN0.count);
    float aN_prev = aN0;                  // alphaN_{k-1}
    float alpha = 0.0f, beta = 0.0f;
    float q_prev = LM ? a.state[0] : 0.0f;     // Q0 of the zeta test (0 behind thallo_hip_lm_state_reset: delta = 0, gauss_newton.t:965)
    bool stopped = false;
    Spin sp; sp.n = 0; sp.t0 = 0;
    bool dead = false;                    // a bounded wait ran out (here or elsewhere): no more waiting, the host raises
    const int slot = 64 * wave + lane;    // the sums slot this lane sweeps
    long sid;
    const bool slot_live = slot < grid && wg_id(slot, sid);

    // Publish my quarter of the sums of an iteration (the words of slot 64 w + lane) to the other waves of the workgroup, wait for theirs, and add all slots up in
    // the order the launch-per-iteration path uses (last_workgroup_totals: lane-strided over the slots, then the wave butterfly) -- same bits everywhere.
    auto exchange_scalars = [&](unsigned T, const unsigned (&wq)[NWD], float& ad_o, double (&tot)[NQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NWD; ++c) S.q[wave][c][lane] = wq[c];
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
        float t = 0.0f; double d[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) d[q] = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned* qq = &S.q[w][0][lane];
            t += __uint_as_float(qq[0]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) d[q] += __hiloint2double((int)qq[64 * (1 + 2 * q)], (int)qq[64 * (2 + 2 * q)]);
        }
        ad_o = wave_sum_all(t);
#pragma unroll
        for (int q = 0; q < NQ; ++q) tot[q] = wave_sum_all_f64(d[q]);
    };
    // The finish of iteration kf from its sums (k_pmarch's deferred finish / block_finish_sums_lm's last workgroup, their expressions): alpha_kf = alphaN_kf / alphaD_kf,
    // betaN_kf = N - 2 alpha S1 + alpha^2 S2; LM: q_{kf+1} = 0.5 [U + alpha (T1 - T2) - alpha^2 alphaD] and the zeta test (k_lm_zeta's rule, :1666-1686).  Leaves alpha, beta
    // for iteration kf + 1, the two words, and (LM) the decision to stop.
    auto finish = [&](int kf, float ad, const double (&tot)[NQ]) __attribute__((always_inline)) {
        const float al = safe_div<LM>(aN_prev, ad);
        double bnd = tot[0] - 2.0 * (double)al * tot[1] + (double)al * (double)al * tot[2];
        if (!(bnd > 0.0)) bnd = 0.0;
        const float bN = (float)bnd;
        if (writer) { a.words[2 * kf] = ad; a.words[2 * kf + 1] = bN; }
        if (LM) {
            const double U = tot[NQ - 3], T1 = tot[NQ - 2], T2 = tot[NQ - 1];
            const float Q1 = (float)(0.5 * (U + (double)al * (T1 - T2) - (double)al * (double)al * (double)ad));
            const float zt = (float)(kf + 1) * (Q1 - q_prev) / Q1;
            const bool stop = !isfinite(Q1) || !isfinite(zt) || zt < a.q_tol;
            if (stop) { stopped = true; if (writer) { reinterpret_cast<unsigned*>(a.state)[1] = 1u; reinterpret_cast<int*>(a.state)[2] = kf + 1; } }
            else { q_prev = Q1; if (writer) a.state[0] = Q1; }
        }
        alpha = al;
        beta = safe_div<LM>(bN, aN_prev);
        aN_prev = bN;
    };
    // J^T J v for my rows from the held rows V (k_pmarch's row step: dB -> U_h, U_v -> T -> J^T, the Laplacian rows; every lane takes part in the exchanges);
    // emit(jj, vc, s) for every output row jj = 2 .. nr + 1 on the output lanes
    auto stencil = [&](const v2f (&V)[NR], auto&& emit) __attribute__((always_inline)) {
        v2f dB[NR], Uh[NR], Uv[NR], Tt[NR], Rr[NR][3];
        unsigned Fl[NR]; float Cy[NR];
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            Fl[jj] = row_ok(jj) ? ((fwx[jj] & 0xffu) | ((fwy[jj] & 0xffu) << 8)) : 0u;
            Cy[jj] = coef1(cm, ya - 2 + jj + g.yoff);
            dB[jj] = Z2; Uh[jj] = Z2; Uv[jj] = Z2; Tt[jj] = Z2;
#pragma unroll
            for (int c = 0; c < 3; ++c) Rr[jj][c] = Z2;
        }
#pragma unroll
        for (int st = 1; st < NR; ++st) {
            // One ROLLING pass (the marching kernel's order of work): the step that takes row st forms dB(st), U_h(st); U_v(st-1), the Laplacian rows R(st-1), T(st-1); the output
            // of row st-2.  Indices are compile-time constants; what a stage leaves is dead two steps later, so the temporaries of three rows are live at a time, not
            // those of all R + 4 (the same expressions, the same bits as pass by pass).
            {   // dB(row st) from v(st), v(st - 1); U_h(st)
                const int jj = st;
                const bool ok = row_ok(jj);
                const v2f v0 = V[jj], v1 = V[jj - 1];
                const v2f vl0 = nbL(v0);
                const v2f dB0 = sel(ok, gx[jj] * v0 + gy[jj] * vl0 + gz[jj] * v1, Z2);
                const v2f dBr = nbR(dB0);
                const v2f wx = cm.wg * v2f{ (float)((fwx[jj] >> 8) & 0xffu), (float)((fwy[jj] >> 8) & 0xffu) };
                const v2f wy = cm.wg * v2f{ (float)((fwx[jj] >> 16) & 0xffu), (float)((fwy[jj] >> 16) & 0xffu) };
                const M2 wn0 = { ok && (wx.x != 0.0f || wy.x != 0.0f), ok && (wx.y != 0.0f || wy.y != 0.0f) };
                dB[jj] = dB0;
                Uh[jj] = sel(wn0, wx * (wx * (dB0 - dBr)), Z2);
            }
            if (st >= 2) {   // U_v(st - 1) from dB(st - 1), dB(st); the Laplacian rows R(st - 1) from v(st - 2), v(st - 1), v(st)
                const int jj = st - 1;
                const bool ok = row_ok(jj);
                const v2f wx = cm.wg * v2f{ (float)((fwx[jj] >> 8) & 0xffu), (float)((fwy[jj] >> 8) & 0xffu) };
                const v2f wy = cm.wg * v2f{ (float)((fwx[jj] >> 16) & 0xffu), (float)((fwy[jj] >> 16) & 0xffu) };
                const M2 wn = { ok && (wx.x != 0.0f || wy.x != 0.0f), ok && (wx.y != 0.0f || wy.y != 0.0f) };
                Uv[jj] = sel(wn, wy * (wy * (dB[jj] - dB[jj + 1])), Z2);
                const v2f v0 = V[jj + 1], v1 = V[jj], v2 = V[jj - 1];
                const v2f vl1 = nbL(v1), vr1 = nbR(v1);
                const float cy0 = Cy[jj + 1], cy1 = Cy[jj], cy2 = Cy[jj - 1];
                const M2 f2b = bit(Fl[jj], 2u);
                Rr[jj][0] = sel(f2b, cm.ws * (4.0f * (cxc * v1) - cxm * vl1 - cxc * v2 - cxp * vr1 - cxc * v0), Z2);
                Rr[jj][1] = sel(f2b, cm.ws * (4.0f * (cy1 * v1) - cy1 * vl1 - cy2 * v2 - cy1 * vr1 - cy0 * v0), Z2);
                Rr[jj][2] = sel(f2b, cm.ws * (4.0f * v1 - vl1 - v2 - vr1 - v0), Z2);
            }
            if (st >= 3) {   // T(st - 1)
                const int jj = st - 1;
                v2f T1 = Uh[jj] + Uv[jj];
                T1 -= nbL(Uh[jj]);
                T1 -= Uv[jj - 1];
                Tt[jj] = T1;
            }
            if (st >= 4 && st - 2 < R + 2) {   // output row y = ya + (st - 2) - 2
                const int jj = st - 2;
                const int y = ya - 2 + jj;
                const v2f T2 = Tt[jj], T1 = Tt[jj + 1];
                const v2f gT2r = nbR(gy[jj] * T2);
                v2f Rl[3], Rq[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { Rl[c] = nbL(Rr[jj][c]); Rq[c] = nbR(Rr[jj][c]); }
                if (jj - 2 < nr && xout) {
                    const v2f vc = V[jj];
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
                    emit(jj, vc, s);
                }
            }
        }
    };
    // my quarter of the sums records of an iteration, polled until every tag is there (the final sweep; inside the loop the same loads ride in the one polling loop)
    auto sweep_sums = [&](int par, unsigned T, unsigned (&wq)[NWD]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NWD; ++c) wq[c] = 0u;
        bool ok = !slot_live;
        sp.n = 0; sp.t0 = 0;
        while (!ok && !dead) {
            ok = true;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int c = 0; c < NWD; ++c) { const u32x2 v = ld1g(RS_SUM, sumw(par, c >> 1, slot) + 8 * (c & 1)); wq[c] = v.x; ok = ok && v.y == T; }
            if (!ok && spin_fail(sp, ctl, 1u, (unsigned)slot, T)) dead = true;
        }
    };

    for (int k = 0; k < a.L; ++k) {
        const unsigned T = seq + (unsigned)k + 1u, Tp = T - 1u;
        const int par = k & 1, parp = par ^ 1;
        SR_STAMP(k, 0);
        if (k > 0) {
            // ---- the one synchronisation point: my quarter of the sums of iteration k-1, the two rows of A p_{k-1} from the wave above and from the wave below, one
            // word per lane of the columns from the strips to the left / right.  Everything is polled in ONE loop, all loads of a pass in flight together.
            unsigned wq[NWD]; float cvl = 0.f, cvr = 0.f;
            v2f ru0 = Z2, ru1 = Z2, rd0 = Z2, rd1 = Z2;
#pragma unroll
            for (int c = 0; c < NWD; ++c) wq[c] = 0u;
            if (up_lds || dn_lds) {      // the rows of the waves of my own workgroup: in LDS, tagged when that wave's stencil was through -- earlier than anything that comes through global memory
                sp.n = 0; sp.t0 = 0;
                const unsigned* tu = &S.rtag[parp][up_lds ? wave - 1 : wave][1]; const unsigned* td = &S.rtag[parp][dn_lds ? wave + 1 : wave][0];
                while (!dead) {
                    const unsigned a0 = up_lds ? __hip_atomic_load(tu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : Tp, a1 = dn_lds ? __hip_atomic_load(td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : Tp;
                    if (a0 == Tp && a1 == Tp) break;
                    if (spin_fail(sp, ctl, 6u, (unsigned)wid, Tp)) dead = true;
                }
                dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (up_lds && xout) { const float* u = &S.rrow[parp][wave - 1][1][0][lane]; Ap[0] = v2f{ u[0], u[64] }; Ap[1] = v2f{ u[128], u[192] }; }
                if (dn_lds && xout) { const float* d = &S.rrow[parp][wave + 1][0][0][lane]; Ap[R + 2] = v2f{ d[0], d[64] }; Ap[R + 3] = v2f{ d[128], d[192] }; }
            }
            {
                const bool need_u = xout && has_up && !up_lds, need_d = xout && has_dn && !dn_lds, need_s = slot_live;
                const unsigned usrc = rowh(parp, has_up ? wid - 1 : wid, 1), dsrc = rowh(parp, has_dn ? wid + 1 : wid, 0);
                const unsigned clsrc = colh(parp, need_cl ? wid - g.nseg + c_dw : wid, 1, c_word), crsrc = colh(parp, need_cr ? wid + g.nseg + c_dw : wid, 0, c_word);
                bool ok_s = !need_s, ok_u = !need_u, ok_d = !need_d, ok_cl = !need_cl, ok_cr = !need_cr;
                sp.n = 0; sp.t0 = 0;
                while (!(ok_s && ok_u && ok_d && ok_cl && ok_cr) && !dead) {
                    asm volatile("" ::: "memory");                     // (every pass re-reads: nothing may be hoisted out of the loop)
                    u32x4 vs[NLD], vu[2], vd[2]; u32x2 vcl, vcr;
                    if (!ok_s) {
#pragma unroll
                        for (int c = 0; c < NLD; ++c) vs[c] = ld2g(RS_SUM, sumw(parp, c, slot));
                    }
                    if (!ok_u) { vu[0] = ld2g(RS_ROW, usrc); vu[1] = ld2g(RS_ROW, usrc + row2); }
                    if (!ok_d) { vd[0] = ld2g(RS_ROW, dsrc); vd[1] = ld2g(RS_ROW, dsrc + row2); }
                    if (!ok_cl) vcl = ld1g(RS_COL, clsrc);
                    if (!ok_cr) vcr = ld1g(RS_COL, crsrc);
                    if (!ok_s) {
                        bool all = true;
#pragma unroll
                        for (int c = 0; c < NLD; ++c) {
                            wq[2 * c] = vs[c].x; all = all && vs[c].y == Tp;
                            if (2 * c + 1 < NWD) { wq[2 * c + 1] = vs[c].z; all = all && vs[c].w == Tp; }
                        }
                        ok_s = all;
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
            float ad; double tot[NQ];
            exchange_scalars(Tp, wq, ad, tot);
            finish(k - 1, ad, tot);
            if (LM && stopped) break;                             // (every workgroup takes the same decision: nobody waits for an iteration that nobody runs)
        }
        SR_STAMP(k, 2);
        // ---- r_k = r_{k-1} - alpha A p_{k-1} ; delta += alpha p_{k-1} ; p_k = M^-1 r_k + beta p_{k-1}     (every row I hold, halo included; k_pmarch<UPD>'s expressions)
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) {
            const bool ok = row_ok(jj);
            v2f rk = rr[jj];
            if (k > 0) rk = fma2(-alpha, Ap[jj], rk);
            const v2f pv = pp[jj];
            if (k > 0) dl[jj] = fma2(alpha, pv, dl[jj]);
            const v2f zk = LM ? mi[LM ? jj : 0] * rk : rk;
            v2f v0 = zk + beta * pv;
            v0 = sel(ok, v0, Z2);
            rr[jj] = rk; pp[jj] = v0;
        }
        SR_STAMP(k, 3);
        // ---- A p_k for my rows, the sums
        v2f acc = Z2; Sums3 sm; SumsQ sq;
        stencil(pp, [&](int jj, v2f vc, v2f s) __attribute__((always_inline)) {
            if (LM) s += ct[LM ? jj - 2 : 0] * vc;
            acc += vc * s;
            const v2f rk = rr[jj], mk = LM ? mi[LM ? jj : 0] : splat(1.0f);
            sm.add(mk.x, rk.x, s.x); sm.add(mk.y, rk.y, s.y);
            if (LM) { const v2f b2 = bb[LM ? jj - 2 : 0], dk = dl[jj]; sq.add(dk.x, rk.x, b2.x, vc.x, s.x); sq.add(dk.y, rk.y, b2.y, vc.y, s.y); }
            Ap[jj] = s;
            if (lane == 1 || lane == 62) { float* d = &S.cst[wave][lane == 1 ? 0 : 1][2 * (jj - 2)]; d[0] = s.x; d[1] = s.y; }
        });
        SR_STAMP(k, 4);
        // ---- publish: the boundary rows, my two columns (one store instruction each), then the workgroup's sums (wave butterflies -> LDS -> wave 0 adds the four
        // waves up in order and publishes the record)
        {
            if (xout && has_up) {
                if (up_lds) { float* d = &S.rrow[par][wave][0][0][lane]; d[0] = Ap[2].x; d[64] = Ap[2].y; d[128] = Ap[3].x; d[192] = Ap[3].y; }
                else { const unsigned d = rowh(par, wid, 0); st2g(RS_ROW, d, T, Ap[2].x, Ap[2].y); st2g(RS_ROW, d + row2, T, Ap[3].x, Ap[3].y); }
            }
            if (xout && has_dn) {
                if (dn_lds) { float* d = &S.rrow[par][wave][1][0][lane]; d[0] = Ap[R].x; d[64] = Ap[R].y; d[128] = Ap[R + 1].x; d[192] = Ap[R + 1].y; }
                else { const unsigned d = rowh(par, wid, 1); st2g(RS_ROW, d, T, Ap[R].x, Ap[R].y); st2g(RS_ROW, d + row2, T, Ap[R + 1].x, Ap[R + 1].y); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the rows and columns written to LDS above)
            if (lane == 0) {
                if (up_lds) __hip_atomic_store(&S.rtag[par][wave][0], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (dn_lds) __hip_atomic_store(&S.rtag[par][wave][1], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (col_pub && has_lf) st1g(RS_COL, colh(par, wid, 0, lane), T, __float_as_uint(S.cst[wave][0][lane]));
            if (col_pub && has_rt) st1g(RS_COL, colh(par, wid, 1, lane), T, __float_as_uint(S.cst[wave][1][lane]));
            const float accf = acc.x + acc.y;
            const float wa = wave_sum_all(accf);
            double wsum[NQ];
            wsum[0] = wave_sum_all_f64(sm.n); wsum[1] = wave_sum_all_f64(sm.s1); wsum[2] = wave_sum_all_f64(sm.s2);
            if (LM) { wsum[NQ - 3] = wave_sum_all_f64(sq.u); wsum[NQ - 2] = wave_sum_all_f64(sq.t1); wsum[NQ - 1] = wave_sum_all_f64(sq.t2); }
            if (lane == 0) {
                S.wa[wave] = wa;
#pragma unroll
                for (int q = 0; q < NQ; ++q) S.wd[wave][q] = wsum[q];
            }
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
                if (lane < NWD) {
                    float s = 0.0f; double b[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) b[q] = 0.0;
                    for (int w = 0; w < 4; ++w) {
                        s += S.wa[w];
#pragma unroll
                        for (int q = 0; q < NQ; ++q) b[q] += S.wd[w][q];
                    }
                    double pick = b[0];
#pragma unroll
                    for (int q = 1; q < NQ; ++q) if (lane >= 1 + 2 * q) pick = b[q];
                    const unsigned word = lane == 0 ? __float_as_uint(s) : (lane & 1) ? (unsigned)__double2hiint(pick) : (unsigned)__double2loint(pick);
                    if (!((g.ab & 4) && id == 1 && k == 2)) st1g(RS_SUM, sumw(par, lane >> 1, blockIdx.x) + 8 * (lane & 1), T, word);
                }
            }
        }
        SR_STAMP(k, 5);
    }
    // ---- behind the loop.  Whoever needs alpha of the last iteration (the writer's workgroup: the two words; every workgroup when the update of the unknowns rides along,
    // and in LM) sweeps the last sums -- unless the zeta test has ended the loop, whose finish is done.
    const bool need_last = a.L > 0 && !stopped && (id == 0 || a.X != nullptr || LM);
    if (need_last) {      // (uniform per workgroup: all four waves take part in the sweep)
        const unsigned T = seq + (unsigned)a.L; const int par = (a.L - 1) & 1;
        unsigned wq[NWD];
        sweep_sums(par, T, wq);
        float ad; double tot[NQ];
        exchange_scalars(T, wq, ad, tot);
        finish(a.L - 1, ad, tot);
    }
    if (writer) {
        // the next launch's tags start behind this one's (seq + 1 .. seq + L were used): the counter lives on the device (replay-safe) and is advanced by the one thread
        // that is through only when every workgroup has published its last sums, i.e. has long read it
        __hip_atomic_store(ctl + SR_SEQ, seq + (unsigned)a.L + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!LM) {
        // what L launches would have left behind: r_{L-1}, p_{L-1}, A p_{L-1}, delta (without its last term); with X: PCGLinearUpdate (gauss_newton.t:901-906) riding along,
        // X += delta + alpha_{L-1} p_{L-1} on my rows (k_linear_update's expressions)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < nr && xout) {
                const long i = (long)(ya + j) * W + x0;
                *reinterpret_cast<float2*>(a.r_out + i) = make_float2(rr[j + 2].x, rr[j + 2].y);
                *reinterpret_cast<float2*>(a.p_out + i) = make_float2(pp[j + 2].x, pp[j + 2].y);
                *reinterpret_cast<float2*>(a.A_out + i) = make_float2(Ap[j + 2].x, Ap[j + 2].y);
                *reinterpret_cast<float2*>(a.delta + i) = make_float2(dl[j + 2].x, dl[j + 2].y);
                if (a.X != nullptr) {
                    const float2 xo = *reinterpret_cast<const float2*>(a.X + i);
                    const float d0 = __builtin_fmaf(alpha, pp[j + 2].x, dl[j + 2].x), d1 = __builtin_fmaf(alpha, pp[j + 2].y, dl[j + 2].y);
                    *reinterpret_cast<float2*>(a.X + i) = make_float2(xo.x + d0, xo.y + d1);
                }
            }
        }
    } else {
        // LM: the update of delta the loop still owes (thallo_hip_lm_owed_delta: the iteration the loop ended on), then the model cost (thallo.t:3845-3865 expanded:
        // delta . J^T J delta and delta . b), savePreviousUnknowns (:915-920) and PCGLinearUpdate (:901-906) -- k_pmarch<MODEL>'s launch, from the registers
        v2f dv[NR];
#pragma unroll
        for (int jj = 0; jj < NR; ++jj) { v2f v0 = fma2(alpha, pp[jj], dl[jj]); dv[jj] = sel(row_ok(jj), v0, Z2); }
        v2f acc = Z2, acc2 = Z2;
        stencil(dv, [&](int jj, v2f vc, v2f s) __attribute__((always_inline)) { acc += vc * s; acc2 += vc * bb[LM ? jj - 2 : 0]; });
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < nr && xout) {
                const long i = (long)(ya + j) * W + x0;
                const v2f v0 = dv[j + 2];
                *reinterpret_cast<float2*>(a.delta + i) = make_float2(v0.x, v0.y);
                const float2 xo2 = *reinterpret_cast<const float2*>(a.X + i);
                const v2f xo = { xo2.x, xo2.y }, xn = xo + v0;
                *reinterpret_cast<float2*>(a.prevX + i) = xo2;
                *reinterpret_cast<float2*>(a.X + i) = make_float2(xn.x, xn.y);
            }
        }
        float vv[2] = { acc.x + acc.y, acc2.x + acc2.y };
        float* __restrict__ const oo[2] = { a.t0_out, a.t1_out };
        block_store_partials<2>(vv, oo, S.red);                  // (a workgroup barrier inside: every wave of a live workgroup gets here)
    }
}

int g_sr_ab = 0;         // tools: A/B bits of the exchange layout (SrGeo::ab)
int g_sr_spin_ms = -1;   // tests: the bound of the kernel's waits in ms, written in front of the next launch (-1: whatever the plan's control words say; 0 there = 2 s)

inline SrGeo make_sr_geo(int W, int H, int yoff, int R)
{
    SrGeo g; g.W = W; g.H = H; g.yoff = yoff; g.R = R; g.ab = g_sr_ab;
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
    l.ctl = 0; l.sums = 32; l.colh = l.sums + 2L * SR_REC * THALLO_MAX_PARTIALS; l.rowh = l.colh + 2 * waves * 2 * 64;
    l.bytes = (l.rowh + 2 * waves * 2 * 64 * 4) * (long)sizeof(u64) + 256;
    return l;
}

template <int R, bool LM> int sr_launch_r(const SrArgs& a, hipStream_t s)
{
    const int grid = (a.g.total + 7) / 8 * 8;
    {   // co-residency is a precondition, not an assumption: the kernel's workgroups wait for each other (asked once per instantiation)
        static int fits = 0;
        if (fits == 0) {
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sfs_resident<R, LM>, SR_NT, 0);
            fits = (e == hipSuccess && per_cu >= 1) ? per_cu : -1;
        }
        if (fits < 0 || (long)fits * thallo_hip_device_cu_count() < grid) return -(int)hipErrorNotSupported;
    }
    if (g_sr_spin_ms >= 0) { const unsigned v = (unsigned)g_sr_spin_ms; if (hipMemcpyAsync(a.b.ctl + SR_SPIN_MS, &v, sizeof(v), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    hipLaunchKernelGGL((k_sfs_resident<R, LM>), dim3(grid), dim3(SR_NT), 0, s, a);
    int e = check_launch(); return e ? e : grid;
}
template <bool LM> int sr_launch(const SrArgs& a, int R, hipStream_t s)
{
    switch (R) {
        case 2: return sr_launch_r<2, LM>(a, s); case 3: return sr_launch_r<3, LM>(a, s); case 4: return sr_launch_r<4, LM>(a, s); case 5: return sr_launch_r<5, LM>(a, s);
        case 6: return sr_launch_r<6, LM>(a, s); case 7: return sr_launch_r<7, LM>(a, s);
        default: break;
    }
    if (!LM) switch (R) {
        case 8: return sr_launch_r<8, false>(a, s); case 9: return sr_launch_r<9, false>(a, s); case 10: return sr_launch_r<10, false>(a, s);
        case 11: return sr_launch_r<11, false>(a, s); case 12: return sr_launch_r<12, false>(a, s);
        default: break;
    }
    return -(int)hipErrorNotSupported;
}
inline void sr_bind(SrArgs& a, void* xbuf)
{
    const SrLayout l = sr_layout(a.g);
    u64* base = reinterpret_cast<u64*>(xbuf);
    a.b.rowh = base + l.rowh; a.b.colh = base + l.colh; a.b.sums = base + l.sums; a.b.ctl = reinterpret_cast<unsigned*>(base + l.ctl);
}

}  // namespace

extern "C" {

#ifdef THALLO_MARCH_SWEEP
int thallo_hip_debug_stamps_sfs_resident(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_sr), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif

void thallo_hip_sfs_resident_debug_set(int what, int value) { if (what == 0) g_sr_rows = value; if (what == 1) g_sr_cap = value; if (what == 2) g_sr_ab = value; if (what == 3) g_sr_spin_ms = value; }

/* rows per wave segment of the resident PCG kernel on a W x H image, or 0: the shape does not fit the chip's registers and the caller runs one launch per PCG iteration */
int thallo_hip_sfs_resident_rows(int W, int H) { return sr_rows(W, H); }
int thallo_hip_sfs_resident_rows_lm(int W, int H) { const int R = sr_rows(W, H); return R <= SR_MAX_R_LM ? R : 0; }

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
    sr_bind(a, xbuf);
    a.cm = cam_of(host_params);
    a.G = G; a.Fw = reinterpret_cast<const unsigned*>(Fw);
    a.r_in = r_in; a.p_in = p_in; a.r_out = r_out; a.A_out = Ap_out; a.p_out = p_out; a.delta = delta;
    a.aN0 = alphaN0; a.words = words; a.X = X; a.L = L;
    return sr_launch<false>(a, R, (hipStream_t)stream);
}

/* The same for a Levenberg-Marquardt step, with its tail: from what thallo_hip_sfs_pcg_init_lm left (r = b, M^-1 in pre, CtC, zeros in p_prev and delta, alphaN_0 = r . M^-1 r) and
 * a reset state (thallo_hip_lm_state_reset), at most L iterations of thallo_hip_sfs_pcg_iter_lm -- the zeta test (thallo_hip_lm_zeta's rule, q_tolerance) ends the loop on
 * the device, in every workgroup alike; lm_state[1] / [2] = gate / iterations done as the launches leave them -- then thallo_hip_sfs_lm_model_cost's launch: the owed update of
 * delta (written to `delta`), per-workgroup partials of delta . J^T J delta and delta . b (dJJd_out, db_out: the return value says how many), prevX = X, X += delta.
 * words[2k] = alphaD_k, words[2k + 1] = betaN_k for the iterations that ran.  L must not exceed the residual reset period (gauss_newton.t:1653-1657: no reset inside the
 * launch).  Bit for bit the launches' results when they run with the same rows per wave.  Replaces gauss_newton.t:1615-1687 + the model-cost part of :1707-1714. */
int thallo_hip_sfs_pcg_resident_lm(int W, int H, int yoff, const float* host_params, const float* G, const float* Fw,
                                   const float* r_in, const float* p_in, const float* pre, const float* CtC, float* delta,
                                   thallo_sum_t alphaN0, float* words, float* lm_state, float q_tolerance, float* dJJd_out, float* db_out,
                                   float* X, float* prevX, void* xbuf, int L, thallo_stream_t stream)
{
    if (H < 1 || (W & 1) || W < 2 || L < 1 || !host_params) return -(int)hipErrorInvalidValue;
    if (!G || !Fw || !r_in || !p_in || !pre || !CtC || !delta || !words || !lm_state || !dJJd_out || !db_out || !X || !prevX || !xbuf || !alphaN0.partials) return -(int)hipErrorInvalidValue;
    const int R = sr_rows(W, H);
    if (R <= 0 || R > SR_MAX_R_LM) return -(int)hipErrorNotSupported;
    SrArgs a; memset(&a, 0, sizeof(a));
    a.g = make_sr_geo(W, H, yoff, R);
    sr_bind(a, xbuf);
    a.cm = cam_of(host_params);
    a.G = G; a.Fw = reinterpret_cast<const unsigned*>(Fw);
    a.r_in = r_in; a.p_in = p_in; a.delta = delta;
    a.aN0 = alphaN0; a.words = words; a.X = X; a.L = L;
    a.pre = pre; a.ctc = CtC; a.state = lm_state; a.q_tol = q_tolerance; a.prevX = prevX; a.t0_out = dJJd_out; a.t1_out = db_out;
    return sr_launch<true>(a, R, (hipStream_t)stream);
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
