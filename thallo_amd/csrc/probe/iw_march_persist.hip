// probe/iw_march_persist.hip (was energy_image_warping_march_persist.hip; RESEARCH builds only since round 6: make VARIANT=research) -- the marching PCG iteration of image_warping as a PERSISTENT loop: iterations k0 .. k1-1 of a Gauss-Newton step
// in ONE launch, for images whose solver state does not fit the chip's registers (2048^2 on one GPU: 35 rows per wave).
//
// energy_image_warping_march_rc.hip pays a launch boundary, a ramp and a tail per iteration (~10 us of every ~49 at 2048^2).  Here the grid of that kernel --
// one workgroup of four waves per CU, the same strips, segments and placement -- stays on the chip and every wave loops over the iterations.  What one
// iteration needs from the others is
//   * everybody's sums of iteration k-1 (alpha_{k-1}, beta_{k-1}): a tagged 64-byte record per workgroup, swept by all workgroups (the resident kernel's
//     "allgather": each of a workgroup's four waves polls a quarter of the records, LDS carries the quarters to the other three);
//   * the halo of r_{k-1}, p_{k-1}: two rows above / below the segment and a pixel pair left / right of the strip, written by OTHER workgroups.  A wave's record
//     goes out after its stores of the iteration have left the CU (s_waitcnt vmcnt(0) of every wave, then the workgroup's barrier), and those stores are
//     write-through (sc1), so whoever holds every record of iteration k-1 may read every r_{k-1}, p_{k-1} -- with L1-bypassing (sc1) loads, or behind ONE
//     agent-scope acquire per wave (template parameter ACQ; MI355X_MICROARCH.md "inter-workgroup visibility").
// There is no grid barrier and no other hand-over: the sums are the synchronisation point of the iteration, as in the resident kernel.  r ping-pongs between
// two planes as behind the launches; p_k goes into plane k mod n of the plan's ring (the host updates delta from the ring, solver.cpp).
//
// MEASURED (round 5, 2048^2, profiles/r05/persist_*.txt) and NOT the default: 6.58 ms per GN step against 5.96 for a launch per iteration on the same box (THALLO_AB=persist=1
// runs it).  Where the time goes (tools/persist_probe.py, stamps build): a wave's march takes 42-43 us on average with r resident (46-48 without) but 47-50 for the
// slowest of the 1003 waves, and everybody waits for that one; the sums exchange behind the last arrival costs 1.6 us, the drain / barrier / publish steps another
// 2-3.  With the cross-workgroup wait compiled out (tools build -DPST_NOSYNC, garbage results) the loop runs 5.54 ms per step: the coupling costs ~10 us per
// iteration, as much as the launch boundary + ramp + tail it replaces.  Write-through stores cost ~5 us per iteration against non-temporal ones (which are not
// coherent: timing only), two workgroups per CU (half the rows per wave, twice the issue rate) lose another 5 us to halo traffic, prefetch depth 1 / 4 and a
// stagger between the waves of a workgroup change nothing.
//
// Same expressions on the same inputs as the launch-per-iteration kernel (jtjp_pair / iter_sums_pixel_masked of iw_march.hpp, -ffp-contract=on), same strips,
// segments and order of every sum: r, p and every alpha_k / beta_k are BIT-identical to it (tests/test_gpu_parity.py).  Replaces the loop of gauss_newton.t:1615-1687.
#include "iw_march.hpp"
#include "probe/thallo_hip_research.h"
#include <cstring>

using namespace thallo;

namespace {

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

typedef unsigned long long u64;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000); }
// AUX: 0 plain, 2 non-temporal, 16 sc1 (stores: write-through; loads: past L1)
template <int AUX> __device__ __forceinline__ u32x4 bld4(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, AUX); }
template <int AUX> __device__ __forceinline__ u32x2 bld2(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, AUX); }
template <int AUX> __device__ __forceinline__ unsigned bld1(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b32(r, vo, so, AUX); }
template <int AUX> __device__ __forceinline__ void bst4(rsrc_t r, unsigned vo, unsigned so, float a, float b, float c, float d)
{ u32x4 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); v.z = __float_as_uint(c); v.w = __float_as_uint(d); __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, AUX); }
template <int AUX> __device__ __forceinline__ void bst2(rsrc_t r, unsigned vo, unsigned so, float a, float b)
{ u32x2 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, AUX); }
__device__ __forceinline__ float uf(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float to_sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

}  // namespace

// control words of a plan's persistent launches (device memory, the first 256 bytes of its exchange buffer)
enum { PST_SEQ = 0, PST_ERR = 1, PST_SPIN_MS = 2, PST_NEXT = 3, PST_PM = 4, PST_CTL_WORDS = 16 };
constexpr int PST_MAX_PLANES = THALLO_HIP_MAX_UPDATE_TERMS + 2;
// r of up to this many of a wave's rows never leaves the CU between the iterations of a launch: 24 B per lane and row in LDS ([rows + 1 scratch row][256 threads] x {float4 | float2},
// 147,456 B of the CU's 160 KB next to the 15.3 KB of static words)
constexpr int PST_RES_ROWS = 23;

struct PersistArgs {
    MarchGeo g;
    const float* cs; const unsigned char* flags; float wf2, wr2;
    float* r[2];                          // iteration k reads r[k & 1], writes r[(k + 1) & 1]
    float* plane[PST_MAX_PLANES]; int n_planes;      // iteration k reads p_{k-1} from plane (k - 1) mod n, writes p_k into plane k mod n
    int k0, k1;                           // the iterations of this launch: k0 >= 1 (iteration 0 of a GN step has no A p_{k-1}: the stored-plane kernel runs it)
    float* parts; int slots; int B;       // the plan's reduction slots: partials of slot j at parts + j * THALLO_MAX_PARTIALS, its word at parts + slots * THALLO_MAX_PARTIALS + j;
                                          // alphaN_k = slot B + 2k, alphaD_k = B + 2k + 1, betaN_k = B + 2k + 2
    double* s12[2];                       // the double sums {N, S1, S2} per workgroup of iteration k live in s12[k & 1] (plain partials: iteration k0 - 1's are read, iteration k1 - 1's written)
    int nb_prev;                          // workgroups of the launch that ran iteration k0 - 1
    thallo_sum_t aN;                      // alphaN_{k0-1}: partials, or one word
    u64* sums;                            // [2 parity][THALLO_MAX_PARTIALS workgroups][8]: alphaD, N hi, N lo, S1 hi, S1 lo, S2 hi, S2 lo, - as {value | tag} granules
    unsigned* ctl;
    const int* irregular;
    int res_rows;                         // rows of r a wave keeps in LDS between the iterations of a launch (0 .. PST_RES_ROWS)
};

namespace {

#ifdef PST_STAMPS
// tools/persist_probe.py: where an iteration spends its time (100 MHz wall clock, lane 0 of every wave, iterations k0 + 4 .. k0 + 7 of a launch)
__device__ unsigned long long* g_stamps_p = nullptr;
#define PSTAMP(k, i) do { if ((threadIdx.x & 63) == 0 && g_stamps_p && (k) - a.k0 >= 4 && (k) - a.k0 < 8) g_stamps_p[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + ((k) - a.k0 - 4)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define PSTAMP(k, i) do { } while (0)
#endif
struct Spin { unsigned n; long long t0; };
// bounded wait bookkeeping: true = give up (this wave or somebody else timed out; every later wait of the wave falls through at once)
__device__ __forceinline__ bool spin_fail(Spin& sp, unsigned* ctl, unsigned what, unsigned idx, unsigned tag)
{
    __builtin_amdgcn_s_sleep(1);
    if (((++sp.n) & 127u) != 0u) return false;
    if (__hip_atomic_load(ctl + PST_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    const long long now = wall_clock64();
    if (sp.t0 == 0) { sp.t0 = now; return false; }
    const unsigned ms = __hip_atomic_load(ctl + PST_SPIN_MS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long bound = ms ? (long long)ms * 100000LL : 2LL * 100000000LL;        // default: 2 s of the 100 MHz wall clock
    if (now - sp.t0 <= bound) return false;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_exchange(ctl + PST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        unsigned* pm = ctl + PST_PM;        // first timeout of the launch: what was waited for
        pm[0] = what; pm[1] = blockIdx.x; pm[2] = threadIdx.x >> 6; pm[3] = idx; pm[4] = tag;
    }
    return true;
}

struct RawP {                               // what one step takes, for one lane (2 pixels)
    u32x4 po, cs; u32x2 pa; unsigned f;     // row t:   p_{k-1} (Offset part x0,y0,x1,y1 | Angle part a0,a1), (c0,s0,c1,s1), the dword holding the pair's flags bytes
    u32x4 ro; u32x2 ra;                     // row t-1: r_{k-1}
};
__device__ __forceinline__ void take4u(u32x4& d, const u32x4& s) { unsigned a, b, c, e; take1(a, s.x); take1(b, s.y); take1(c, s.z); take1(e, s.w); d.x = a; d.y = b; d.z = c; d.w = e; }
__device__ __forceinline__ void take2u(u32x2& d, const u32x2& s) { unsigned a, b; take1(a, s.x); take1(b, s.y); d.x = a; d.y = b; }
__device__ __forceinline__ void take(RawP& d, const RawP& s) { take4u(d.po, s.po); take2u(d.pa, s.pa); take4u(d.cs, s.cs); take1(d.f, s.f); take4u(d.ro, s.ro); take2u(d.ra, s.ra); }

struct PRow { float px[2], py[2], pa[2]; };                      // p of a lane's pixel pair in one row
struct GRow { float c[2], s[2], a[2]; unsigned f; };             // cos / sin of Angle, the active bits as 0 / 1, the two flags bytes
struct RRow { float rx[2], ry[2], ra[2]; };                      // r_k

struct PstLds {
    float4 lut[32];                   // by the 5-bit flags value: M^-1 of the Offset channels, of the Angle channel, w_fit^2 where the fit residual is valid
    float red[16]; double redd[48];
    unsigned q[4][2][7][64];          // per wave: the 7 words of the (up to) 2 x 64 record slots it swept
};

// arrival counters: [2 parity][8 shards], each on a 128-byte line of its own, behind the control words; monotonic within a launch, zeroed in front of it
constexpr int PST_CNT_STRIDE = 32;        // words
__global__ void k_persist_begin(unsigned* ctl, unsigned n)
{
    if (threadIdx.x == 0) { const unsigned s = ctl[PST_NEXT]; ctl[PST_SEQ] = s; ctl[PST_NEXT] = s + n + 1u; }
    if (threadIdx.x < 16) (ctl + 64)[threadIdx.x * PST_CNT_STRIDE] = 0u;
}

// ACQ: 0 = r / p loads with sc1 (past L1), no fence; 1 = one agent-scope acquire per wave and iteration (buffer_inv sc1), plain loads
// DEPTH: rows of prefetch.  OCC: register budget (workgroups of 4 waves per CU the kernel is compiled for; the grid is always one per CU)
template <int ACQ, int DEPTH, int OCC>
__global__ __launch_bounds__(MARCH_NT, OCC) void k_march_persist(PersistArgs a)
{
    static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "the prefetch slots rotate inside a trip of four rows");
    __shared__ PstLds S;
#ifndef PST_LD_AUX
#define PST_LD_AUX 16
#endif
#ifndef PST_ST_AUX
#define PST_ST_AUX 16
#endif
#ifndef PST_ST_AUX_P
#define PST_ST_AUX_P PST_ST_AUX
#endif
    constexpr int LD = ACQ ? 0 : PST_LD_AUX, ST = PST_ST_AUX, STP = PST_ST_AUX_P;      // (tools builds time other cache policies: anything but sc1 / sc1 is NOT coherent)
    const MarchGeo g = a.g;
    unsigned* const ctl = a.ctl;
    if (a.irregular != nullptr && __builtin_amdgcn_readfirstlane(a.irregular[0]) != 0) {      // not the unit pixel grid after all: poison, never the wrong Jacobian
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int k = a.k0; k < a.k1; ++k) { float* w = a.parts + (size_t)a.slots * THALLO_MAX_PARTIALS; w[a.B + 2 * k - 1] = __builtin_nanf(""); w[a.B + 2 * k] = __builtin_nanf(""); }
        if (threadIdx.x == 0) { (a.parts + (size_t)(a.B + 2 * (a.k1 - 1) + 1) * THALLO_MAX_PARTIALS)[blockIdx.x] = __builtin_nanf(""); }
        return;
    }
    // which workgroup slots carry rows (march_place: slot b -> id).  A slot without rows never publishes -- the sweeps of the others know, its sums are zeros --
    // and leaves the zeros of the launch's last iteration where a launch per iteration would
    const int grid = (int)gridDim.x;
    const int G = (grid % 8) == 0 ? 8 : 1;
    auto slot_has_rows = [&](int b) { const int grp = b % G, l = b / G; const long lo = (long)g.total * grp / G, hi = (long)g.total * (grp + 1) / G; return lo + l < hi; };
    if (!slot_has_rows((int)blockIdx.x)) {
        if (threadIdx.x == 0) {
            float* aD_out = a.parts + (size_t)(a.B + 2 * (a.k1 - 1) + 1) * THALLO_MAX_PARTIALS; double* s12_out = a.s12[(a.k1 - 1) & 1];
            aD_out[blockIdx.x] = 0.0f; s12_out[3 * blockIdx.x] = 0.0; s12_out[3 * blockIdx.x + 1] = 0.0; s12_out[3 * blockIdx.x + 2] = 0.0;
        }
        return;
    }
    if (threadIdx.x < 32) { float mo, ma; pre_from_flags((unsigned char)threadIdx.x, a.wf2, a.wr2, mo, ma); S.lut[threadIdx.x] = make_float4(mo, ma, (threadIdx.x & 2) ? a.wf2 : 0.f, 0.f); }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned N = (unsigned)g.W * (unsigned)g.H;         // (12 N < 2^32: host-checked)
    const int W2 = g.W >> 1;                                  // pixel pairs per row
    int strip, ya, yb;
    march_place(g, wave, strip, ya, yb);
    const bool work = ya < yb;
    const int x0 = strip * MARCH_USE - 2 + 2 * lane;          // first of this lane's two pixels
    const bool xin = x0 >= 0 && x0 < g.W;                     // W even: both pixels exist or neither
    const bool xout = xin && lane >= 1 && lane <= 62;         // this lane's pixels are outputs of this wave
    const int xc = x0 < 0 ? 0 : x0 > g.W - 2 ? g.W - 2 : x0;
    const unsigned h = (unsigned)xc >> 1;                     // the lane's pixel pair in its row
    const unsigned vo16 = h * 16u, vo8 = h * 8u;
    const unsigned vf0 = (h >> 1) * 4u, vf1 = ((h + 1u) >> 1) * 4u, sh0 = (h & 1u) * 16u;
    const unsigned angle0 = 8u * N;                           // byte offset of a vector's Angle part
    const unsigned mxin = xin ? 0xffffu : 0u;
    const rsrc_t RS_CS = make_rsrc(a.cs), RS_F = make_rsrc(a.flags), RS_SUM = make_rsrc(a.sums);

    // the records this lane sweeps: slots lane + 64 (4 j + wave), j = 0, 1 -- so that the four waves' quarters, read back in the order wave 0 .. 3 of j = 0, then of
    // j = 1, are the lane-strided order lane, lane + 64, lane + 128, ... of the launch-per-iteration sums (grids of up to 512 workgroups: two per CU)
    const int slot0 = 64 * wave + lane, slot1 = 256 + slot0;
    const bool live0 = slot0 < grid && slot_has_rows(slot0), live1 = slot1 < grid && slot_has_rows(slot1);
    (void)live0; (void)live1;
    const bool two = grid > 256;
    // Waiting is QUIET: a workgroup that has published its record adds one to its shard (blockIdx % 8: its XCD under round-robin placement) of the iteration's arrival
    // counter; whoever waits polls the eight shard words -- eight lanes, one dword each -- until all have their full count, and only then sweeps the records, once.
    // (Sweeping the records themselves in the waiting loop -- 16 KB per workgroup and pass, from every workgroup that is through -- took the memory system away from
    //  the waves still marching: their march went from ~46 to ~57 us, tools/persist_probe.py.)
    unsigned* const cnt = ctl + 64;
    unsigned shard_live = 0;                                  // lanes 0 .. 7: workgroups with rows in shard `lane`
    if (lane < 8) for (int b = lane; b < grid; b += 8) shard_live += slot_has_rows(b) ? 1u : 0u;
    auto sumw = [&](int par, int sl) { return (unsigned)((((long)par * THALLO_MAX_PARTIALS + sl) * 8) * 8); };       // that workgroup's 64-byte record
    const unsigned seq = __hip_atomic_load(ctl + PST_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // tag of iteration k: seq + (k - k0) + 1
    // the words of iteration k-1 are left behind by the one wave that owns the first segment of strip 0
    const bool scal_writer = work && strip == 0 && ya == g.row0 && lane == 0;
    float* const words = a.parts + (size_t)a.slots * THALLO_MAX_PARTIALS;

    Spin sp; sp.n = 0; sp.t0 = 0;
    bool dead = false;
    float aN_prev = 0.0f;                 // alphaN_{k-1}

    // r RESIDENT in LDS: r_k is this wave's private state except where another wave reads it -- its first and last row (the y halo of the waves above / below) and
    // lanes 1 / 62 of every row (the x halo of the strips left / right).  Rows [ra0, ra0 + nres) of the segment keep r in LDS between the iterations of a launch:
    // their r loads fetch lanes 0 / 63 only (what the neighbouring strips stored; the other lanes carry a byte offset beyond the buffer: no request, zeros) and their r
    // stores write lanes 1 / 62 only.  The memory instruction stream of a row step is the same for every row -- no branch, no second loop --, resident or not; an r row
    // costs 24 B per pixel and iteration in HBM traffic when streamed, ~1 B when resident.  The launch's first iteration reads every r from memory, its last one
    // writes every r back (the next launch, or nobody, reads it).
    extern __shared__ __attribute__((aligned(16))) unsigned char pst_dyn[];
    float4* const lds_r4 = reinterpret_cast<float4*>(pst_dyn) + threadIdx.x;                                        // [a.res_rows + 1][256]
    float2* const lds_r2 = reinterpret_cast<float2*>(pst_dyn + (a.res_rows + 1) * MARCH_NT * sizeof(float4)) + threadIdx.x;
    const int ra0 = ya + 1;
    const int nres = __builtin_amdgcn_readfirstlane(yb - ya - 2 < 0 ? 0 : yb - ya - 2 > a.res_rows ? a.res_rows : yb - ya - 2);
    const bool lane_in = lane >= 1 && lane <= 62;
    const unsigned OOB = 0xffffff00u;                          // beyond num_records of every plane's descriptor
    const unsigned vo16_res = lane_in ? OOB : vo16, vo8_res = lane_in ? OOB : vo8;
    const bool lane_edge = lane == 1 || lane == 62;

    typedef RawP RawT;
    for (int k = a.k0; k < a.k1; ++k) {
        const int par = k & 1;
        const unsigned T = seq + (unsigned)(k - a.k0) + 1u, Tp = T - 1u;
        float alpha, beta;
        PSTAMP(k, 0);
        // ---- alpha_{k-1}, beta_{k-1}: the sums of iteration k-1, added up in the launch-per-iteration order (lane-strided over the workgroups, then the wave butterfly)
        {
            float t = 0.0f; double n = 0.0, a1 = 0.0, b1 = 0.0; float ad;
            if (k == a.k0) {              // ... left by an earlier launch as plain partials (rc_iteration_scalars' loop)
                const float* aD_part = a.parts + (size_t)(a.B + 2 * (k - 1) + 1) * THALLO_MAX_PARTIALS;
                const double* s12_part = a.s12[(k - 1) & 1];
                if (a.aN.count == 1) aN_prev = a.aN.partials[0];
                else { float u = 0.0f; for (int i = lane; i < a.aN.count; i += THALLO_WAVE) u += a.aN.partials[i]; aN_prev = wave_sum_all(u); }       // (rc_sum's order)
                const int nb = a.nb_prev;
                for (int i = lane; i < nb; i += THALLO_WAVE) { t += aD_part[i]; n += s12_part[3 * i]; a1 += s12_part[3 * i + 1]; b1 += s12_part[3 * i + 2]; }
                ad = nb == 1 ? aD_part[0] : wave_sum_all(t);
            } else {                      // ... published by every workgroup of this launch as a tagged record: my quarter, the other three through LDS
                unsigned w7[7];
#pragma unroll
                for (int c = 0; c < 7; ++c) w7[c] = 0u;
                {   // every workgroup's arrival of iteration k-1 (that parity's counter has taken one per workgroup and iteration of the parity so far)
                    const unsigned want = shard_live * (unsigned)((k - 1 - a.k0) / 2 + 1);
                    const unsigned* c = cnt + ((par ^ 1) * 8 + (lane & 7)) * PST_CNT_STRIDE;
                    sp.n = 0; sp.t0 = 0;
#ifdef PST_NOSYNC
                    bool in = true;          // (tools, TIMING ONLY: nobody waits for anybody -- what the loop costs without its synchronisation point; results are garbage)
#else
                    bool in = lane >= 8;
#endif
                    while (!dead) {
                        if (!in) in = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
                        if (__all(in)) break;
                        __builtin_amdgcn_s_sleep(8);
                        if (spin_fail(sp, ctl, 2u, (unsigned)(lane & 7), want)) dead = true;
                    }
                    dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
                }
                unsigned w7b[7];
#pragma unroll
                for (int c = 0; c < 7; ++c) w7b[c] = 0u;
#ifdef PST_NOSYNC
                bool ok0 = true, ok1 = true;
#else
                bool ok0 = !live0, ok1 = !live1;
#endif
                const unsigned src0 = sumw(par ^ 1, slot0), src1 = sumw(par ^ 1, slot1);
                sp.n = 0; sp.t0 = 0;
                while (!(ok0 && ok1) && !dead) {
                    asm volatile("" ::: "memory");                     // (every pass re-reads)
                    u32x4 vs[4], vt[4];
                    if (!ok0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) vs[c] = __builtin_amdgcn_raw_buffer_load_b128(RS_SUM, src0 + 16 * c, 0, 16);
                    }
                    if (!ok1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) vt[c] = __builtin_amdgcn_raw_buffer_load_b128(RS_SUM, src1 + 16 * c, 0, 16);
                    }
                    if (!ok0) {
                        w7[0] = vs[0].x; w7[1] = vs[0].z; w7[2] = vs[1].x; w7[3] = vs[1].z; w7[4] = vs[2].x; w7[5] = vs[2].z; w7[6] = vs[3].x;
                        ok0 = vs[0].y == Tp && vs[0].w == Tp && vs[1].y == Tp && vs[1].w == Tp && vs[2].y == Tp && vs[2].w == Tp && vs[3].y == Tp;
                    }
                    if (!ok1) {
                        w7b[0] = vt[0].x; w7b[1] = vt[0].z; w7b[2] = vt[1].x; w7b[3] = vt[1].z; w7b[4] = vt[2].x; w7b[5] = vt[2].z; w7b[6] = vt[3].x;
                        ok1 = vt[0].y == Tp && vt[0].w == Tp && vt[1].y == Tp && vt[1].w == Tp && vt[2].y == Tp && vt[2].w == Tp && vt[3].y == Tp;
                    }
                    if (!(ok0 && ok1) && spin_fail(sp, ctl, 1u, (unsigned)slot0, Tp)) dead = true;
                }
                dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
                PSTAMP(k, 1);
#pragma unroll
                for (int c = 0; c < 7; ++c) { S.q[wave][0][c][lane] = w7[c]; S.q[wave][1][c][lane] = w7b[c]; }
                __syncthreads();
                PSTAMP(k, 2);          // every record of iteration k-1 is in: every workgroup's r_{k-1}, p_{k-1} have left its CU
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (j == 1 && !two) break;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned* qq = &S.q[w][j][0][lane];
                        t += __uint_as_float(qq[0]);
                        n += __hiloint2double((int)qq[64], (int)qq[128]);
                        a1 += __hiloint2double((int)qq[192], (int)qq[256]);
                        b1 += __hiloint2double((int)qq[320], (int)qq[384]);
                    }
                }
                ad = wave_sum_all(t);
            }
            n = wave_sum_all_d(n); a1 = wave_sum_all_d(a1); b1 = wave_sum_all_d(b1);
            alpha = safe_div<false>(aN_prev, ad);
            double bd = n - 2.0 * (double)alpha * a1 + (double)alpha * (double)alpha * b1;
            if (!(bd > 0.0)) bd = 0.0;
            const float bn = (float)bd;
            if (scal_writer) { words[a.B + 2 * (k - 1) + 1] = ad; words[a.B + 2 * k] = bn; }      // alphaD_{k-1}, betaN_{k-1} (= alphaN_k)
            beta = safe_div<false>(bn, aN_prev);
            aN_prev = bn;
            alpha = to_sgpr(alpha); beta = to_sgpr(beta);
        }
        PSTAMP(k, 3);
#ifdef PST_STAGGER
        if (wave == 1) __builtin_amdgcn_s_sleep(PST_STAGGER); else if (wave == 2) __builtin_amdgcn_s_sleep(2 * PST_STAGGER); else if (wave == 3) __builtin_amdgcn_s_sleep(3 * PST_STAGGER);      // (tools: the four waves of a workgroup leave the barrier together; a quarter of a row step apart they do not issue their loads in the same cycles)
#endif
        if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // this wave's loads below come behind it (in order: no wait, no barrier needed for the wave's own loads)

        const rsrc_t RS_P = make_rsrc(a.plane[(k - 1) % a.n_planes]), RS_Q = make_rsrc(a.plane[k % a.n_planes]);
        const rsrc_t RS_R = __builtin_amdgcn_make_buffer_rsrc(a.r[par], 0, 12u * N, 0x00020000), RS_RO = make_rsrc(a.r[par ^ 1]);      // (num_records = the plane: the resident rows' masked lanes are out of range)
        const bool ld_phase = k > a.k0, st_phase = k < a.k1 - 1;
        float acc = 0.0f; double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (work) {
            RawT slot_[DEPTH];
            auto issue = [&](RawT& s, int t) {
                const unsigned tc = (unsigned)(t < 0 ? 0 : t > g.H - 1 ? g.H - 1 : t), row = tc * (unsigned)W2;
                s.po = bld4<LD>(RS_P, vo16, row * 16u); s.pa = bld2<LD>(RS_P, vo8, angle0 + row * 8u);
                s.cs = bld4<0>(RS_CS, vo16, row * 16u);
                const unsigned parf = row & 1u;
                s.f = bld1<0>(RS_F, parf ? vf1 : vf0, (row - parf) * 2u);
                const unsigned tr = (unsigned)(t - 1 < 0 ? 0 : t - 1 > g.H - 1 ? g.H - 1 : t - 1), rrow = tr * (unsigned)W2;
                const bool lr = ld_phase && (unsigned)(t - 1 - ra0) < (unsigned)nres;                  // that row's r is in LDS: lanes 0 / 63 only
                s.ro = bld4<LD>(RS_R, lr ? vo16_res : vo16, rrow * 16u); s.ra = bld2<LD>(RS_R, lr ? vo8_res : vo8, angle0 + rrow * 8u);
            };
            // rings of four rows, indexed by the row modulo 4 (compile-time inside a trip of four rows)
            PRow pp[4], pk[4]; GRow gg[4]; RRow rr[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { pp[i].px[q] = 0.f; pp[i].py[q] = 0.f; pp[i].pa[q] = 0.f; pk[i].px[q] = 0.f; pk[i].py[q] = 0.f; pk[i].pa[q] = 0.f;
                                              gg[i].c[q] = 1.f; gg[i].s[q] = 0.f; gg[i].a[q] = 0.f; }
                gg[i].f = 0u;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) { rr[i].rx[q] = 0.f; rr[i].ry[q] = 0.f; rr[i].ra[q] = 0.f; }
            const int t_first = ya - 2, t_last = yb + 1;          // rows of p_{k-1} / cs / flags to take
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) slot_[j] = RawT{};
            const int t_begin = t_first - DEPTH;
            for (int t0 = t_begin; t0 <= t_last; t0 += 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = t0 + j;
                    // ring roles at this step: row t -> index j, t-1 -> j+3, t-2 -> j+2, t-3 -> j+1 (mod 4)
                    PRow& p0 = pp[j % 4]; const PRow& p1 = pp[(j + 3) % 4]; const PRow& p2 = pp[(j + 2) % 4];
                    GRow& g0 = gg[j % 4]; const GRow& g1 = gg[(j + 3) % 4]; const GRow& g2 = gg[(j + 2) % 4]; const GRow& g3 = gg[(j + 1) % 4];
                    PRow& k1 = pk[(j + 3) % 4]; const PRow& k2 = pk[(j + 2) % 4]; const PRow& k3 = pk[(j + 1) % 4];
                    RRow& r1 = rr[(j + 1) % 2]; const RRow& r2 = rr[j % 2];
                    RawT cur;
                    take(cur, slot_[j % DEPTH]);                 // (the only place that waits for memory)
                    fence_order();                               // the refill stays behind the moves ...
                    issue(slot_[j % DEPTH], t + DEPTH > t_last ? t_last : t + DEPTH);
                    fence_order();                               // ... and in front of the arithmetic
                    // ---- row t enters the p_{k-1} / geometry rings (outside the image: inactive)
                    {
                        const bool rowok = t >= 0 && t < g.H;
                        const unsigned parf = ((unsigned)t * (unsigned)W2) & 1u;
                        const unsigned fl = (cur.f >> (parf ? 16u - sh0 : sh0)) & (rowok ? mxin : 0u);
                        p0.px[0] = uf(cur.po.x); p0.py[0] = uf(cur.po.y); p0.px[1] = uf(cur.po.z); p0.py[1] = uf(cur.po.w); p0.pa[0] = uf(cur.pa.x); p0.pa[1] = uf(cur.pa.y);
                        g0.c[0] = uf(cur.cs.x); g0.s[0] = uf(cur.cs.y); g0.c[1] = uf(cur.cs.z); g0.s[1] = uf(cur.cs.w); g0.f = fl;
                        g0.a[0] = (float)(fl & 1u); g0.a[1] = (float)((fl >> 8) & 1u);
                    }
                    // ---- row u = t-1: A p_{k-1}(u) -> r_k(u), p_k(u)
                    const int u = t - 1;
                    {
                        const float4 m0 = S.lut[g1.f & 31u], m1 = S.lut[(g1.f >> 8) & 31u];      // (M^-1 offsets, M^-1 angle, w_fit^2 or 0)
                        const float wfit[2] = { m0.z, m1.z };
                        float ax[2], ay[2], av[2];
                        jtjp_pair(p2, p1, p0, g2, g1, g0, g2.a, g1.a, g0.a, wfit, a.wr2, ax, ay, av);
                        // r_{k-1}(u): from memory (every lane of a streamed row, lanes 0 / 63 of a resident one) or from this thread's own words in LDS
                        const int ru = u - ra0;
                        const bool row_res = (unsigned)ru < (unsigned)nres;
                        const bool ldr = row_res && ld_phase, str = row_res && st_phase;
                        const int sl_ld = (ldr ? ru : a.res_rows) * MARCH_NT, sl_st = (str ? ru : a.res_rows) * MARCH_NT;      // (the scratch row takes what nobody reads)
                        const float4 l4 = lds_r4[sl_ld]; const float2 l2 = lds_r2[sl_ld];
                        const bool ul = ldr && lane_in;
                        float rx[2] = { ul ? l4.x : uf(cur.ro.x), ul ? l4.z : uf(cur.ro.z) }, ry[2] = { ul ? l4.y : uf(cur.ro.y), ul ? l4.w : uf(cur.ro.w) }, rq[2] = { ul ? l2.x : uf(cur.ra.x), ul ? l2.y : uf(cur.ra.y) };
                        rx[0] = __builtin_fmaf(-alpha, ax[0], rx[0]); ry[0] = __builtin_fmaf(-alpha, ay[0], ry[0]);
                        rx[1] = __builtin_fmaf(-alpha, ax[1], rx[1]); ry[1] = __builtin_fmaf(-alpha, ay[1], ry[1]);
                        rq[0] = __builtin_fmaf(-alpha, av[0], rq[0]); rq[1] = __builtin_fmaf(-alpha, av[1], rq[1]);
                        const float mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            k1.px[q] = mo[q] * rx[q] + beta * p1.px[q]; k1.py[q] = mo[q] * ry[q] + beta * p1.py[q]; k1.pa[q] = ma[q] * rq[q] + beta * p1.pa[q];
                            r1.rx[q] = rx[q]; r1.ry[q] = ry[q]; r1.ra[q] = rq[q];
                        }
                        lds_r4[sl_st] = make_float4(rx[0], ry[0], rx[1], ry[1]); lds_r2[sl_st] = make_float2(rq[0], rq[1]);
                        const bool mine = xout && u >= ya && u < yb;
                        if (mine && (!str || lane_edge)) {    // r_k: this wave's own rows, write-through (other workgroups read them in the next iteration); resident rows: what the strips left / right read
                            const unsigned row = (unsigned)u * (unsigned)W2;
                            bst4<ST>(RS_RO, vo16, row * 16u, rx[0], ry[0], rx[1], ry[1]); bst2<ST>(RS_RO, vo8, angle0 + row * 8u, rq[0], rq[1]);
                        }
                        if (mine) {                           // p_k: into the ring's plane (the next iteration's halo reads; the host's delta update)
                            const unsigned row = (unsigned)u * (unsigned)W2;
                            bst4<STP>(RS_Q, vo16, row * 16u, k1.px[0], k1.py[0], k1.px[1], k1.py[1]); bst2<STP>(RS_Q, vo8, angle0 + row * 8u, k1.pa[0], k1.pa[1]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- row v = t-2: A p_k(v) and the iteration's sums (the wave's own rows and output lanes only)
                    {
                        const int v = t - 2;
                        const bool on = xout && v >= ya && v < yb;
                        const float4 m0 = S.lut[on ? g2.f & 31u : 0u], m1 = S.lut[on ? (g2.f >> 8) & 31u : 0u];
                        const float wfit[2] = { m0.z, m1.z }, mo[2] = { m0.x, m1.x }, ma[2] = { m0.y, m1.y };
                        float ax[2], ay[2], av[2];
                        jtjp_pair(k3, k2, k1, g3, g2, g1, g3.a, g2.a, g1.a, wfit, a.wr2, ax, ay, av);
                        const float msum = on ? 1.0f : 0.0f;
                        __builtin_amdgcn_sched_barrier(0);
                        iter_sums_pixel_masked(msum, k2.px[0], k2.py[0], k2.pa[0], ax[0], ay[0], av[0], r2.rx[0], r2.ry[0], r2.ra[0], mo[0], ma[0], acc, s0, s1, s2);
                        __builtin_amdgcn_sched_barrier(0);
                        iter_sums_pixel_masked(msum, k2.px[1], k2.py[1], k2.pa[1], ax[1], ay[1], av[1], r2.rx[1], r2.ry[1], r2.ra[1], mo[1], ma[1], acc, s0, s1, s2);
                    }
                    asm volatile("" : "+v"(acc), "+v"(s0), "+v"(s1), "+v"(s2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- the iteration's sums of this workgroup (iter_tail's order), published when every wave's stores have left the CU
        PSTAMP(k, 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PSTAMP(k, 5);
        const float wa = wave_sum_all(acc); const double w0 = wave_sum_all_d(s0), w1 = wave_sum_all_d(s1), w2 = wave_sum_all_d(s2);
        if (lane == 0) { S.red[wave] = wa; S.redd[3 * wave] = w0; S.redd[3 * wave + 1] = w1; S.redd[3 * wave + 2] = w2; }
        __syncthreads();
        PSTAMP(k, 6);
        if (threadIdx.x == 0) {
            float sa = 0.0f; double b0 = 0.0, b1 = 0.0, b2 = 0.0;
            for (int w = 0; w < MARCH_NT / THALLO_WAVE; ++w) { sa += S.red[w]; b0 += S.redd[3 * w]; b1 += S.redd[3 * w + 1]; b2 += S.redd[3 * w + 2]; }
            if (k == a.k1 - 1) {          // the launch's last iteration: plain partials, for the launch (or the one-wave finish) behind this one
                float* aD_out = a.parts + (size_t)(a.B + 2 * k + 1) * THALLO_MAX_PARTIALS; double* s12_out = a.s12[par];
                aD_out[blockIdx.x] = sa; s12_out[3 * blockIdx.x] = b0; s12_out[3 * blockIdx.x + 1] = b1; s12_out[3 * blockIdx.x + 2] = b2;
            } else {
                const u64 u0 = (u64)__double_as_longlong(b0), u1 = (u64)__double_as_longlong(b1), u2 = (u64)__double_as_longlong(b2);
                const unsigned dst = sumw(par, (int)blockIdx.x);
                u32x4 d;
                d.x = __float_as_uint(sa); d.y = T; d.z = (unsigned)(u0 >> 32); d.w = T; __builtin_amdgcn_raw_buffer_store_b128(d, RS_SUM, dst, 0, 16);
                d.x = (unsigned)u0; d.z = (unsigned)(u1 >> 32);                           __builtin_amdgcn_raw_buffer_store_b128(d, RS_SUM, dst + 16, 0, 16);
                d.x = (unsigned)u1; d.z = (unsigned)(u2 >> 32);                           __builtin_amdgcn_raw_buffer_store_b128(d, RS_SUM, dst + 32, 0, 16);
                d.x = (unsigned)u2; d.z = 0u;                                             __builtin_amdgcn_raw_buffer_store_b128(d, RS_SUM, dst + 48, 0, 16);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the record has left the CU before the arrival
                __hip_atomic_fetch_add(cnt + (par * 8 + (int)(blockIdx.x & 7)) * PST_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// product configuration
#ifndef PST_DEPTH_V
#define PST_DEPTH_V 2
#endif
constexpr int PST_DEPTH = PST_DEPTH_V, PST_OCC = 2;
int g_pst_acq = 0;                       // tools: 1 = the acquire form
int g_pst_res_rows = PST_RES_ROWS;       // tools / tests: rows of r per wave kept in LDS (0: every r row streams)
int g_pst_occ = 1;                       // workgroups per CU the grid is sized for (1: 4 waves per CU; 2: 8 -- twice the instruction issue rate, half the LDS per workgroup)
inline int pst_res_rows(int occ) { const int m = occ >= 2 ? 9 : PST_RES_ROWS; return g_pst_res_rows < m ? g_pst_res_rows : m; }      // (2 per CU: 9 + 1 rows x 6 KB + 15.3 KB static <= 80 KB)
inline size_t pst_lds_bytes(int rows) { return (size_t)(rows + 1) * MARCH_NT * (sizeof(float4) + sizeof(float2)); }

inline size_t pst_bytes() { return 256 + 16 * PST_CNT_STRIDE * sizeof(unsigned) + (size_t)2 * THALLO_MAX_PARTIALS * 8 * sizeof(u64); }      // control words | arrival counters | records

}  // namespace

extern "C" {

/* bytes of exchange memory a plan needs for the persistent marching loop (control words + the tagged sums records); zero-filled by the caller once */
long thallo_hip_iw_march_persist_bytes(void) { return (long)pst_bytes(); }

/* rows per wave the persistent loop runs a W x H image with, or 0: it does not (odd width, a vector beyond one buffer descriptor, more workgroups than the
 * device holds at one per CU -- they wait for each other -- or than a wave's sweep of the sums records covers) */
int thallo_hip_iw_march_persist_rows(int W, int H)
{
    if (W < 2 || (W & 1) || H < 1 || 12.0 * (double)W * (double)H >= 4294967296.0) return 0;
    const int R = march_pick_rows(W, H, g_pst_occ);
    if (R <= 0) return 0;
    const MarchGeo g = make_march_geo(W, H, 0, H, R);
    const int grid = (g.total + 7) / 8 * 8;
    long cap = g_march_cap > 0 ? g_march_cap : (long)thallo_hip_device_cu_count() * g_pst_occ;
    if (cap > 512) cap = 512;               // (a wave's sweep covers two records per lane)
    return grid <= cap ? R : 0;
}

void thallo_hip_iw_march_persist_debug_set(int what, int value) { if (what == 0) g_pst_acq = value; if (what == 1) g_pst_res_rows = value < 0 ? 0 : value > PST_RES_ROWS ? PST_RES_ROWS : value;
  if (what == 2) g_pst_occ = value >= 2 ? 2 : 1; }
#ifdef PST_STAMPS
int thallo_hip_debug_stamps_persist(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_p), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif

/* Iterations k0 .. k1-1 (1 <= k0 < k1) of the PCG loop of a GN step in ONE launch; whole images on the unit pixel grid.  What a launch per iteration
 * (thallo_hip_iw_pcg_iter_march_rc_deferred, delta mode "none") reads and leaves behind, in the plan's own layout:
 *   r[k & 1] -> r[(k + 1) & 1];  p_{k-1} in planes[(k - 1) % n_planes] -> p_k in planes[k % n_planes]  (n_planes >= k1 - k0 + 1, so that no plane of the launch is written twice or read after its overwrite);
 *   reduction slots at parts + j * THALLO_HIP_MAX_PARTIALS with their words at parts + slots * THALLO_HIP_MAX_PARTIALS + j, alphaN_k = slot B + 2k, alphaD_k = B + 2k + 1, betaN_k = B + 2k + 2:
 *   read  alphaN_{k0-1} (alphaN_prev: partials or one word), the nb_prev alphaD partials of iteration k0 - 1 and its double sums in s12[(k0 - 1) & 1];
 *   write the words alphaD_{k-1}, betaN_{k-1} for k0 <= k < k1, and the partials of iteration k1 - 1 (alphaD slot, s12[(k1 - 1) & 1]; as many as the return value).
 * xbuf: thallo_hip_iw_march_persist_bytes() bytes, zeroed once, private to the plan.  Every workgroup must be resident (checked against the device's occupancy answer);
 * every wait inside is bounded, thallo_hip_iw_march_persist_status tells.  Returns the number of workgroups (> 0), -hipErrorNotSupported when the shape does not fit,
 * another negative hipError_t on failure.  Replaces the loop of gauss_newton.t:1615-1687. */
int thallo_hip_iw_pcg_march_persist(int W, int H, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    float* r0, float* r1, float* const* planes, int n_planes, int k0, int k1,
                                    float* parts, int slots, int B, double* s12_0, double* s12_1, int nb_prev, thallo_sum_t alphaN_prev,
                                    const int* irregular, void* xbuf, thallo_stream_t stream)
{
    if (!cs || !flags || !r0 || !r1 || !planes || !parts || !s12_0 || !s12_1 || !xbuf) return -(int)hipErrorInvalidValue;
    if (k0 < 1 || k1 <= k0 || n_planes < 2 || n_planes > PST_MAX_PLANES || k1 - k0 + 1 > n_planes || nb_prev < 1 || nb_prev > THALLO_MAX_PARTIALS || !alphaN_prev.partials || alphaN_prev.count < 1 || alphaN_prev.count > THALLO_MAX_PARTIALS ||
        B < 0 || slots < B + 2 * k1 + 1) return -(int)hipErrorInvalidValue;
    const int R = thallo_hip_iw_march_persist_rows(W, H);
    if (R <= 0) return -(int)hipErrorNotSupported;
    PersistArgs a; memset(&a, 0, sizeof(a));
    a.g = make_march_geo(W, H, 0, H, R);
    a.cs = cs; a.flags = flags; a.wf2 = w_fit * w_fit; a.wr2 = w_reg * w_reg;
    a.r[0] = r0; a.r[1] = r1;
    for (int i = 0; i < n_planes; ++i) { if (!planes[i]) return -(int)hipErrorInvalidValue; a.plane[i] = planes[i]; }
    a.n_planes = n_planes; a.k0 = k0; a.k1 = k1;
    a.parts = parts; a.slots = slots; a.B = B; a.s12[0] = s12_0; a.s12[1] = s12_1; a.nb_prev = nb_prev; a.aN = alphaN_prev;
    a.ctl = reinterpret_cast<unsigned*>(xbuf); a.sums = reinterpret_cast<u64*>(reinterpret_cast<char*>(xbuf) + 256 + 16 * PST_CNT_STRIDE * sizeof(unsigned));
    a.irregular = irregular;
    const int occ = g_pst_occ;
    a.res_rows = pst_res_rows(occ);
    const size_t lds = pst_lds_bytes(a.res_rows);
    const int grid = (a.g.total + 7) / 8 * 8;
    hipStream_t s = (hipStream_t)stream;
    const int acq = g_pst_acq ? 1 : 0;
    {   // co-residency is a precondition: the workgroups wait for each other
        int per_cu = 0;
        const hipError_t e = acq ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_march_persist<1, PST_DEPTH, PST_OCC>, MARCH_NT, lds)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_march_persist<0, PST_DEPTH, PST_OCC>, MARCH_NT, lds);
        if (e != hipSuccess || per_cu < 1 || (long)(per_cu < occ ? per_cu : occ) * thallo_hip_device_cu_count() < grid) return -(int)hipErrorNotSupported;
    }
    hipLaunchKernelGGL(k_persist_begin, dim3(1), dim3(64), 0, s, a.ctl, (unsigned)(k1 - k0));
    if (acq) hipLaunchKernelGGL((k_march_persist<1, PST_DEPTH, PST_OCC>), dim3(grid), dim3(MARCH_NT), lds, s, a);
    else     hipLaunchKernelGGL((k_march_persist<0, PST_DEPTH, PST_OCC>), dim3(grid), dim3(MARCH_NT), lds, s, a);
    int e = check_launch(); return e ? e : grid;
}

/* the error word of a plan's persistent launches: 1 = a bounded wait ran out (a workgroup was not resident, or a record never arrived); clear != 0 resets it.
 * pm (5 words, may be NULL): what the first timed-out wait was for.  Synchronises the stream.  spin_ms >= 0 sets the bound (0 = the 2 s default). */
int thallo_hip_iw_march_persist_status(void* xbuf, int clear, int spin_ms, unsigned* pm, thallo_stream_t stream)
{
    if (!xbuf) return -(int)hipErrorInvalidValue;
    unsigned* ctl = reinterpret_cast<unsigned*>(xbuf);
    hipStream_t s = (hipStream_t)stream;
    unsigned w[PST_CTL_WORDS];
    if (hipMemcpyAsync(w, ctl, sizeof(w), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    if (pm) for (int i = 0; i < 5; ++i) pm[i] = w[PST_PM + i];
    if (clear && w[PST_ERR]) { const unsigned z = 0; if (hipMemcpyAsync(ctl + PST_ERR, &z, sizeof(z), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (spin_ms >= 0) { const unsigned v = (unsigned)spin_ms; if (hipMemcpyAsync(ctl + PST_SPIN_MS, &v, sizeof(v), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    return (int)w[PST_ERR];
}

}  // extern "C"
