// energy_image_warping_resident.hip -- the whole PCG loop of one Gauss-Newton step of image_warping in ONE launch, for working sets that fit
// the register files of the chip (512^2, the 2048 x 256 slab of an 8-GPU run: BASELINE config 1 and the 1/8 slab of the benchmark).
//
// Why: at those sizes a launch per PCG iteration (energy_image_warping_march.hip, energy_image_warping.hip) is bounded by what sits BETWEEN the
// arithmetic -- 2.7 us of launch boundary, a burst that loads r / Ap / p / cs / flags from L2, a burst that stores them again, a reduction tail --
// 10.7 us per iteration at 512^2 and 16.4 us on the slab, of which ~2 us are arithmetic.  Here a wave keeps its pixels'
// r, p, A p (and cos / sin / M^-1 / flags) in REGISTERS and delta in LDS for all L iterations; what moves per iteration is only what another
// wave needs:
//   * the boundary of A p_k (first / last row of a segment, lanes 1 / 62 of a strip) to the four neighbouring waves, as 8-byte
//     {value | tag} granules in global memory (write-through, agent scope: MI355X_MICROARCH.md "handoff-1to1"; the data IS the flag);
//   * the workgroup's four sums {alphaD | N, S1, S2} to EVERY workgroup (7 granules per workgroup, swept by all: "allgather").
// Both are published at the same point -- the end of the iteration's arithmetic -- and consumed at the same point -- in front of the next
// iteration's r / p update -- so an iteration has ONE synchronisation point, and it is neighbour-and-scalars only: no grid barrier.  (Exchanging
// A p_{k-1} instead of p_k is what makes that possible: p_k on the halo needs alpha / beta, i.e. the global sums, first; A p_{k-1} does not, and
// r, p on the halo are then recomputed locally -- the same trick the multi-GPU slabs use for their ghost rows, solver_dist.cpp.)
//
// Geometry, arithmetic and summation order are the marching kernel's (a wave = a 124-pixel-wide column strip x R rows, lane l the pixels
// x0 + 2l, x0 + 2l + 1, lanes 0 / 63 the x halo; workgroup = 4 stacked segments; the same per-lane accumulation order, wave butterfly, workgroup
// sum and lane-strided sum of the per-workgroup partials), and both files are compiled with -ffp-contract=on (contraction decided per source
// expression, never across statements): r, p, delta, A p and every alpha / beta come out bit-identical to the launch-per-iteration marching
// kernel run with the same R (tests/test_gpu_parity.py).  Replaces gauss_newton.t:1615-1687 (the PCG loop) for these shapes.
#include "iw_march.hpp"
#include <cstring>

using namespace thallo;

namespace {

constexpr int RES_USE = 124;              // output pixels per wave row (lanes 1..62 x 2)
constexpr int RES_NT = 256;               // 4 waves = 4 vertically adjacent segments of one strip (one workgroup per CU)
constexpr int RES_MAX_R = 10;             // rows per segment the kernel is instantiated for (register budget: 19 words per row and lane stay in registers; 6 R <= 64: a column's words fit a wave)

inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

typedef unsigned long long u64;

// The exchange buffers are addressed as raw buffers (one descriptor in SGPRs + a 32-bit byte offset per lane: no 64-bit address arithmetic in the loop);
// aux 16 = sc1: write-through stores / L1-bypassing loads, the agent-scope forms of MI355X_MICROARCH.md "inter-workgroup visibility".  A 16-byte access
// moves TWO granules; each 8-byte half carries its own tag, so a torn 16-byte access is still two whole granules.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ u32x4 ld2g(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16); }
__device__ __forceinline__ u32x2 ld1g(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16); }
__device__ __forceinline__ void st2g(rsrc_t r, unsigned off, unsigned tag, float v0, float v1)
{ u32x4 d; d.x = __float_as_uint(v0); d.y = tag; d.z = __float_as_uint(v1); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 16); }
__device__ __forceinline__ void st1g(rsrc_t r, unsigned off, unsigned tag, unsigned v) { u32x2 d; d.x = v; d.y = tag; __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 16); }

// system scope (sc0 sc1): granules a PEER GPU stored into my memory / that I store into a peer's
__device__ __forceinline__ u32x4 ld2s(rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 17); }
__device__ __forceinline__ void st2s(rsrc_t r, unsigned off, unsigned tag, float v0, float v1)
{ u32x4 d; d.x = __float_as_uint(v0); d.y = tag; d.z = __float_as_uint(v1); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 17); }

}  // namespace

// control words of a resident launch (device memory, RES_CTL_WORDS of them)
enum { RES_SEQ = 0, RES_ERR = 1, RES_SPIN_MS = 2, RES_NEXT = 3, RES_PM = 4, RES_CTL_WORDS = 16 };

struct ResGeo { int W, H, row0, row1, R, nstrips, nseg, nwgrow, total; };

// exchange buffers of one plan (thallo_hip_iw_resident_bytes); parity = iteration & 1
struct ResBufs {
    u64* rowh;        // [2 parity][waves][2 sides: 0 = the wave's FIRST row (for the wave above), 1 = its LAST row (for the wave below)][64 lanes][6 components]
    u64* colh;        // [2 parity][waves][2 sides: 0 = lane 1's pixels (for the strip to the left), 1 = lane 62's (for the strip to the right)][64: word 6 * row + component]
    u64* sums;        // [2 parity][1024 workgroups][8: alphaD, N hi, N lo, S1 hi, S1 lo, S2 hi, S2 lo, -]   (a workgroup's record = one 64-byte line, one store instruction)
    u64* gs;          // [2 parity][2]: alphaD_k, betaN_k over ALL ranks, published by workgroup 0 (multi-GPU form: only that workgroup sweeps the sums)
    unsigned* ctl;    // RES_CTL_WORDS
};

// one rank's row slab of a multi-GPU run (solver_dist.cpp): the mailbox block of every rank carries, behind its scalar granules, a GHOST area
// [2 parity][strips][2: 0 = the row that comes from the rank above, 1 = from the rank below][64 lanes][6 granules] that the neighbouring rank's
// boundary waves store into directly (peer-to-peer over xGMI, system scope)
struct ResDist { thallo_dist_t d; unsigned ghost_off; int slot0; int above, below; };

struct ResArgs {
    ResGeo g; ResBufs b; ResDist x;
    const float* cs; const unsigned char* flags; float wf2, wr2;
    const float* r_in; const float* p_in;         // r_0 and p_{-1} (zeros): what PCGInit1 wrote
    float* r_out; float* A_out; float* p_out;     // r_{L-1}, A p_{L-1}, p_{L-1}: what L launches of the marching kernel leave behind
    float* delta;                                 // in: 0; out: sum_{k < L-1} alpha_k p_k (PCGLinearUpdate adds the last term, like behind the launches)
    thallo_sum_t aN0;                             // alphaN_0
    float* words;                                 // words[2k] = alphaD_k, words[2k + 1] = betaN_k  (the plan's scal(B + 2k + 1), scal(B + 2k + 2))
    const int* irregular;
    int L;
    float* X0; float* X1;                         // the unknowns (Offset: 2 floats per pixel, Angle: 1), or NULL: PCGLinearUpdate stays a launch of its own (always NULL across ranks)
    int fault;                                    // tests: workgroup 1 never publishes the sums of iteration 2 (what a workgroup that is not resident looks like to the others)
};

namespace {

#ifdef THALLO_MARCH_SWEEP
// tools/resident_probe.py RP_STAMPS=1: where an iteration spends its time (100 MHz wall clock, lane 0 of every wave, iterations 8..11)
__device__ unsigned long long* g_stamps_r = nullptr;
#define RES_STAMP(k, i) do { if ((threadIdx.x & 63) == 0 && g_stamps_r && (k) >= 8 && (k) < 12) g_stamps_r[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + ((k) - 8)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define RES_STAMP(k, i) do { } while (0)
#endif
#ifdef THALLO_MARCH_SWEEP
#define RES_NOTE(k, i, v) do { if ((threadIdx.x & 63) == 0 && g_stamps_r && (k) >= 8 && (k) < 12) g_stamps_r[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + ((k) - 8)) * 16 + (i)] = (unsigned long long)(v); } while (0)
#else
#define RES_NOTE(k, i, v) do { } while (0)
#endif
struct Spin { unsigned n; long long t0; };
// bounded wait bookkeeping: true = give up (this wave or somebody else timed out; every later wait of the wave falls through at once)
__device__ __forceinline__ bool spin_fail(Spin& sp, unsigned* ctl, unsigned what, unsigned idx, unsigned tag)
{
    __builtin_amdgcn_s_sleep(1);
    if (((++sp.n) & 127u) != 0u) return false;
    if (__hip_atomic_load(ctl + RES_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    const long long now = wall_clock64();
    if (sp.t0 == 0) { sp.t0 = now; return false; }
    const unsigned ms = __hip_atomic_load(ctl + RES_SPIN_MS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long bound = ms ? (long long)ms * 100000LL : 2LL * 100000000LL;        // default: 2 s of the 100 MHz wall clock
    if (now - sp.t0 <= bound) return false;
    if ((threadIdx.x & 63) == 0 && __hip_atomic_exchange(ctl + RES_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        unsigned* pm = ctl + RES_PM;        // first timeout of the launch: what was waited for
        pm[0] = what; pm[1] = blockIdx.x; pm[2] = threadIdx.x >> 6; pm[3] = idx; pm[4] = tag;
    }
    return true;
}

// LDS words shared by the four waves of a workgroup
struct ResLds {
    float2 lut[32];
    unsigned qtag[4];                 // quarter-sweep exchange: wave w's column is complete for tag ...
    unsigned wtag[4];                 // wave sums of an iteration are in place
    unsigned rtag[2][4][2];           // [parity][wave][side]: that wave's first (0) / last (1) row of A p is in rrow, for the neighbouring wave of the same workgroup
    unsigned q[4][7][64];             // per wave: the 7 words of the 64 slots it swept
    float wa[4]; double wd[4][3];     // per wave: alphaD part, {N, S1, S2} parts
    float rrow[2][4][2][6][64];       // the y halo between the stacked waves of the workgroup never leaves the CU
    float cst[4][2][64];              // per wave: lane 1's / lane 62's A p of its rows, transposed so that ONE store instruction publishes a column
    float crx[4][2][64];              // per wave: the received columns (from the strip to the left / right), for lanes 0 / 63 to pick up
};

// dist_exchange_iter_wave (dist_device.hpp) with the two results in registers: ONE full wave, converged.  This rank's sums go out as 7 granules to every rank's
// mailbox slots slot0 .. slot0 + 6, the wave waits (bounded) for everybody's and adds them in rank order.
__device__ __forceinline__ void res_exchange_ranks(const thallo_dist_t& d, int slot0, float ad, double q0, double q1, double q2, float an, float& gad_o, float& bn_o)
{
    const int lane = threadIdx.x & (THALLO_WAVE - 1);
    unsigned w[7];
    w[0] = __float_as_uint(ad);
    { const u64 b = (u64)__double_as_longlong(q0); w[1] = (unsigned)(b >> 32); w[2] = (unsigned)b; }
    { const u64 b = (u64)__double_as_longlong(q1); w[3] = (unsigned)(b >> 32); w[4] = (unsigned)b; }
    { const u64 b = (u64)__double_as_longlong(q2); w[5] = (unsigned)(b >> 32); w[6] = (unsigned)b; }
    const unsigned seq = ld_agent(d.ctl + DIST_SEQ);
    if (lane < d.world) {
#pragma unroll
        for (int j = 0; j < 7; ++j) st_sys(d.peer_mail[lane] + (long)(slot0 + j) * d.world + d.rank, ((u64)seq << 32) | (u64)w[j]);
    }
    unsigned mine = 0;
    if (lane < 7 * d.world) {                       // lane -> (granule j, source rank r)
        const int j = lane / d.world, r = lane - j * d.world;
        const u64* g = d.mail + (long)(slot0 + j) * d.world + r;
        u64 v = ld_sys(g);
        int it = 0; long long t0 = 0;
        const long long bound = dist_spin_ticks(d);
        while ((unsigned)(v >> 32) != seq) {
            if ((it & 1023) == 0) { if (ld_agent(d.ctl + DIST_ERR) != 0) break; if (it == 0) t0 = wall_clock64(); }
            ++it;
            if ((it & 1023) == 0 && wall_clock64() - t0 > bound) {
                if (__hip_atomic_exchange(d.ctl + DIST_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    unsigned* pm = d.ctl + DIST_POST_MORTEM;
                    pm[0] = (unsigned)(slot0 + j); pm[1] = (unsigned)r; pm[2] = seq; pm[3] = (unsigned)(v >> 32); pm[4] = (unsigned)v;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = ld_sys(g);
        }
        mine = (unsigned)v;
    }
    float gad = 0.0f; double gq[3] = { 0.0, 0.0, 0.0 };
    for (int r = 0; r < d.world; ++r) {
        gad += __uint_as_float(__shfl(mine, r, THALLO_WAVE));
        for (int j = 0; j < 3; ++j) {
            const unsigned hi = __shfl(mine, (1 + 2 * j) * d.world + r, THALLO_WAVE), lo = __shfl(mine, (2 + 2 * j) * d.world + r, THALLO_WAVE);
            gq[j] += __longlong_as_double((long long)(((u64)hi << 32) | (u64)lo));
        }
    }
    const float alpha = safe_div<false>(an, gad);
    double bn = gq[0] - 2.0 * (double)alpha * gq[1] + (double)alpha * (double)alpha * gq[2];
    if (!(bn > 0.0)) bn = 0.0;
    gad_o = gad; bn_o = (float)bn;
}

template <int R, bool DIST>
__global__ __launch_bounds__(RES_NT, 1) void k_pcg_resident(ResArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ResLds& S = *reinterpret_cast<ResLds*>(smem);
    float* dl = reinterpret_cast<float*>(smem + ((sizeof(ResLds) + 15) & ~(size_t)15));      // delta: [R][6][256], a thread's own words only
    float4* csl = reinterpret_cast<float4*>(dl + R * 6 * RES_NT) + threadIdx.x;               // cos / sin of the rows I hold: [R + 2][256] x {c0, s0, c1, s1}, a thread's own words only
    const ResGeo g = a.g;
    // (the wave index through readfirstlane: the compiler then KNOWS that segment, rows, neighbour flags and buffer offsets are wave-uniform -- scalar registers and
    //  scalar branches instead of exec-masked code for every `if (has_up)`)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned* const ctl = a.b.ctl;

    // workgroup -> (strip, first segment): the marching kernel's XCD-aware placement (workgroups b and b + 8 share an XCD)
    const int grid = (int)gridDim.x;
    const int G = (grid % 8) == 0 ? 8 : 1;
    auto wg_id = [&](int b, long& id) { const int grp = b % G, l = b / G; const long lo = (long)g.total * grp / G, hi = (long)g.total * (grp + 1) / G; id = lo + l; return id < hi; };
    long id;
    if (!wg_id(blockIdx.x, id)) return;                                  // (a slot without rows: its sums are zeros, the sweeps know)
    const bool writer = id == 0 && threadIdx.x == 0;                     // leaves alphaD_k / betaN_k behind as words
    if (a.irregular != nullptr && __builtin_amdgcn_readfirstlane(a.irregular[0]) != 0) {      // not the unit pixel grid after all: poison, never the wrong Jacobian
        if (writer) for (int k = 0; k < 2 * a.L; ++k) a.words[k] = __builtin_nanf("");
        return;
    }
    if (threadIdx.x < 32) { float mo, ma; pre_from_flags((unsigned char)threadIdx.x, a.wf2, a.wr2, mo, ma); S.lut[threadIdx.x] = make_float2(mo, ma); }
    if (threadIdx.x < 4) { S.qtag[threadIdx.x] = 0u; S.wtag[threadIdx.x] = 0u; }
    if (threadIdx.x < 16) (&S.rtag[0][0][0])[threadIdx.x] = 0u;
    __syncthreads();                                                      // (the only barrier of the launch)

    const int strip = (int)(id % g.nstrips), seg = (int)(id / g.nstrips) * (RES_NT / 64) + wave;
    int ya = g.row0 + seg * g.R, yb = ya + g.R;
    if (yb > g.row1) yb = g.row1;
    if (ya > g.row1) ya = g.row1;
    const int nr = yb - ya;                                               // rows of this wave (0: a wave of the last workgroup row without a segment)
    const int x0 = strip * RES_USE - 2 + 2 * lane;
    const bool xin = x0 >= 0 && x0 < g.W;
    const bool xout = xin && lane >= 1 && lane <= 62;
    const long N = (long)g.W * g.H;
    const int W2 = g.W >> 1;
    const int xc = x0 < 0 ? 0 : x0 > g.W - 2 ? g.W - 2 : x0;
    const unsigned seq = __hip_atomic_load(ctl + RES_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // tag of iteration k: seq + k + 1

    // ---- who my neighbours are (a neighbour exists = somebody publishes the granules I would wait for)
    const int wid = strip * g.nseg + seg;                                 // wave id: (strip, segment)
    // (a segment with fewer than R rows is the last of its strip -- nothing below it, or the slab's ghost row only if the host made sure R divides the rows --
    //  so the row below a wave that HAS one is always row jj = R + 1)
    const bool rank_up = DIST && a.x.above != 0, rank_dn = DIST && a.x.below != 0;     // my top / bottom ghost row is the last / first owned row of another rank
    const bool up_gh = nr > 0 && seg == 0 && rank_up, dn_gh = nr == R && yb == g.row1 && rank_dn;
    const bool has_up = nr > 0 && (seg > 0 || rank_up), has_dn = nr == R && (yb < g.row1 || rank_dn);
    const bool has_lf = nr > 0 && strip > 0, has_rt = nr > 0 && strip + 1 < g.nstrips;        // (then lane 63's pixels -- x = 124 (strip + 1), + 1 -- are inside the image)
    const long waves = (long)g.nstrips * g.nseg;
    // byte offsets into the three exchange buffers (raw-buffer addressing: descriptor + 32-bit offset)
    const rsrc_t RS_ROW = make_rsrc(a.b.rowh), RS_COL = make_rsrc(a.b.colh), RS_SUM = make_rsrc(a.b.sums), RS_GS = make_rsrc(a.b.gs);
    // multi-GPU: my ghost area (peers store into it) and the two neighbours' (I store into them)
    const rsrc_t RS_GH = make_rsrc(DIST ? (const void*)a.x.d.mail : (const void*)a.b.gs);
    const rsrc_t RS_PA = make_rsrc(rank_up ? (const void*)a.x.d.peer_mail[a.x.d.rank - 1] : (const void*)a.b.gs), RS_PB = make_rsrc(rank_dn ? (const void*)a.x.d.peer_mail[a.x.d.rank + 1] : (const void*)a.b.gs);
    auto ghost = [&](int par, int dir) { return a.x.ghost_off + (unsigned)(((((long)par * g.nstrips + strip) * 2 + dir) * 64 + lane) * 48); };
    const unsigned seqx = DIST ? __hip_atomic_load(a.x.d.ctl + DIST_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 12 : 0u;       // cross-rank tag of iteration k: seqx + k + 1 (the GN step counter is the same on every rank)
    auto rowh = [&](int par, int w, int side) { return (unsigned)(((((long)par * waves + w) * 2 + side) * 64 + lane) * 48); };       // this lane's 6 granules of that row
    auto colh = [&](int par, int w, int side, int i) { return (unsigned)(((((long)par * waves + w) * 2 + side) * 64 + i) * 8); };     // granule i = 6 * row + component
    auto sumw = [&](int par, int slot) { return (unsigned)((((long)par * THALLO_MAX_PARTIALS + slot) * 8) * 8); };                    // that workgroup's 64-byte record
    // ---- state: rows t = ya - 1 + jj, jj = 0 .. R + 1 (jj = 0 and jj = nr + 1: the y halo; lanes 0 / 63: the x halo)
    // (x, y) of a pixel as a register pair: the stencil and the vector updates run on packed fp32 instructions (iw_march.hpp jtjp_pair_xy; round 5)
    v2f rxy[R + 2][2], pxy[R + 2][2], axy[R + 2][2]; float ra[R + 2][2], pa[R + 2][2], av[R + 2][2];
    unsigned fl[R + 2];
    const float4* ro4 = reinterpret_cast<const float4*>(a.r_in); const float2* ra2 = reinterpret_cast<const float2*>(a.r_in + 2 * N);
    const float4* po4 = reinterpret_cast<const float4*>(a.p_in); const float2* pa2 = reinterpret_cast<const float2*>(a.p_in + 2 * N);
    const float4* cs4 = reinterpret_cast<const float4*>(a.cs);
    const unsigned* f4 = reinterpret_cast<const unsigned*>(a.flags);
    float4* dl4 = reinterpret_cast<float4*>(a.delta); float2* dl2 = reinterpret_cast<float2*>(a.delta + 2 * N);
#pragma unroll
    for (int jj = 0; jj < R + 2; ++jj) {
        const int t = ya - 1 + jj;
        const bool ok = nr > 0 && jj <= nr + 1 && xin && t >= 0 && t < g.H;
        const int tc = t < 0 ? 0 : t > g.H - 1 ? g.H - 1 : t;
        const long i2 = (long)tc * W2 + (xc >> 1);
        const float4 r4 = ro4[i2]; const float2 r2 = ra2[i2]; const float4 p4 = po4[i2]; const float2 p2 = pa2[i2]; const float4 c4 = cs4[i2];
        const unsigned fw = f4[i2 >> 1];
        const unsigned f = ok ? (fw >> ((((long)t * W2 + (x0 >> 1)) & 1) != 0 ? 16 : 0)) & 0xffffu : 0u;
        fl[jj] = f;
        rxy[jj][0].x = r4.x; rxy[jj][0].y = r4.y; rxy[jj][1].x = r4.z; rxy[jj][1].y = r4.w; ra[jj][0] = r2.x; ra[jj][1] = r2.y;
        pxy[jj][0].x = p4.x; pxy[jj][0].y = p4.y; pxy[jj][1].x = p4.z; pxy[jj][1].y = p4.w; pa[jj][0] = p2.x; pa[jj][1] = p2.y;
        csl[jj * RES_NT] = c4;
#pragma unroll
        for (int q = 0; q < 2; ++q) { axy[jj][q].x = 0.f; axy[jj][q].y = 0.f; av[jj][q] = 0.f; }
        if (jj >= 1 && jj <= R) {       // delta of my rows into LDS
            const int j = jj - 1;
            const bool mine = j < nr && xout;
            float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f); float2 d2 = make_float2(0.f, 0.f);
            if (mine) { const long i = (long)t * W2 + (x0 >> 1); d4 = dl4[i]; d2 = dl2[i]; }
            float* d = dl + (j * 6) * RES_NT + threadIdx.x;
            d[0] = d4.x; d[RES_NT] = d4.y; d[2 * RES_NT] = d4.z; d[3 * RES_NT] = d4.w; d[4 * RES_NT] = d2.x; d[5 * RES_NT] = d2.y;
        }
    }

    const float aN0 = sum_partials(a.aN0.partials, a.aN0.count);
    float aN_prev = aN0;                  // alphaN_{k-1}
    float alpha = 0.0f, beta = 0.0f;
    Spin sp; sp.n = 0; sp.t0 = 0;
    bool dead = false;                    // a bounded wait ran out (here or elsewhere): no more waiting, the host raises

    // which of my halo rows comes through LDS (the neighbouring wave sits in my workgroup) and which through global memory (another workgroup)
    const bool up_lds = has_up && wave > 0, up_glb = has_up && wave == 0 && !up_gh;
    const bool dn_lds = has_dn && wave < 3 && !dn_gh, dn_glb = has_dn && wave == 3 && !dn_gh;
    const bool sweeper = !DIST || id == 0;                    // multi-GPU: only workgroup 0 adds the sums up (and talks to the other ranks); everybody else waits for its two words
    const int slot = 64 * wave + lane;    // the sums slot this lane sweeps
    long sid;
    const bool slot_live = slot < grid && wg_id(slot, sid);
    const bool col_lane = lane < 6 * nr;  // columns: lane = word 6 * row + component of the LEFT and of the RIGHT exchange

    // Publish my quarter of the sums of an iteration (the 7 words of slot 64 w + lane) to the other waves of the workgroup, wait for theirs, and add all
    // slots up in the order the launch-per-iteration path uses (lane-strided over the slots, then the wave butterfly): alphaD_k, betaN_k -- same bits everywhere.
    auto exchange_scalars = [&](int kk, unsigned T, const unsigned (&w7)[7], float& ad_o, double& n_o, double& s1_o, double& s2_o) {
        RES_STAMP(kk, 8);
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
        RES_STAMP(kk, 9);
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
        n_o = wave_sum_all_d(n); s1_o = wave_sum_all_d(a1); s2_o = wave_sum_all_d(b1);
        RES_STAMP(kk, 10);
    };
    // alphaD_k, betaN_k from the sums: on one GPU alpha_k = alphaN_k / alphaD_k, betaN_k = N - 2 alpha S1 + alpha^2 S2 (iteration_scalars' expressions); across
    // ranks wave 0 of workgroup 0 sends this rank's four sums to every rank's mailbox, adds everybody's in rank order (dist_exchange_iter_wave's granules, slots and
    // order: same bits as the launch-per-iteration transport) and publishes the two words for the rest of the chip
    auto scalars_from_sums = [&](int k_of, unsigned T, int par, float aN, float ad, double n, double a1, double b1, float& aD_o, float& bN_o) {
        if (!DIST) {
            const float al = safe_div<false>(aN, ad);
            double bd = n - 2.0 * (double)al * a1 + (double)al * (double)al * b1;
            if (!(bd > 0.0)) bd = 0.0;
            aD_o = ad; bN_o = (float)bd;
        } else if (wave == 0) {
            float gad, gbn;
            res_exchange_ranks(a.x.d, a.x.slot0 + 7 * k_of, ad, n, a1, b1, aN, gad, gbn);
            if (lane == 0) st2g(RS_GS, (unsigned)par * 16u, T, gad, gbn);
            aD_o = gad; bN_o = gbn;
        } else {
            u32x4 v; bool ok = false;
            sp.n = 0; sp.t0 = 0;
            while (!ok && !dead) {
                asm volatile("" ::: "memory");
                v = ld2g(RS_GS, (unsigned)par * 16u);
                ok = v.y == T && v.w == T;
                if (!ok && spin_fail(sp, ctl, 8u, 0u, T)) dead = true;
            }
            aD_o = __uint_as_float(v.x); bN_o = __uint_as_float(v.z);
        }
    };

    for (int k = 0; k < a.L; ++k) {
        const unsigned T = seq + (unsigned)k + 1u, Tp = T - 1u;
        const int par = k & 1, parp = par ^ 1;
        // (the validity masks derived from the flags are loop-invariant; hoisted they are ~40 SGPR pairs, i.e. spilled to VGPR lanes and read back with two
        //  v_readlane + hazard nops at every use -- recomputing the compare where it is used is cheaper, so the flags are made opaque once per iteration)
#pragma unroll
        for (int jj = 0; jj < R + 2; ++jj) asm volatile("" : "+v"(fl[jj]));
        RES_STAMP(k, 0);
        if (k > 0) {
            // ---- the one synchronisation point.  Everything that comes through global memory is polled in ONE loop, all loads of a pass in flight together:
            // my quarter of the sums of iteration k-1, the row of A p_{k-1} from the workgroup above (wave 0) / below (wave 3), and one word per lane of the
            // two columns from the strips to the left / right.
            // y halo from the waves of my own workgroup: their rows are in LDS, tagged when that wave's stencil was through -- earlier than anything that
            // comes through global memory, so they are taken first and the polling below overlaps nothing with them
            if (up_lds || dn_lds) {
                sp.n = 0; sp.t0 = 0;
                const unsigned* tu = &S.rtag[parp][up_lds ? wave - 1 : wave][1]; const unsigned* td = &S.rtag[parp][dn_lds ? wave + 1 : wave][0];
                while (!dead) {
                    const unsigned a0 = up_lds ? __hip_atomic_load(tu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : Tp, a1 = dn_lds ? __hip_atomic_load(td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : Tp;
                    if (a0 == Tp && a1 == Tp) break;
                    if (spin_fail(sp, ctl, 6u, (unsigned)wid, Tp)) dead = true;
                }
                dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (up_lds && xout) {
                    const float* u = &S.rrow[parp][wave - 1][1][0][lane];
                    axy[0][0].x = u[0]; axy[0][0].y = u[64]; av[0][0] = u[128]; axy[0][1].x = u[192]; axy[0][1].y = u[256]; av[0][1] = u[320];
                }
                if (dn_lds && xout) {
                    const float* d = &S.rrow[parp][wave + 1][0][0][lane];
                    axy[R + 1][0].x = d[0]; axy[R + 1][0].y = d[64]; av[R + 1][0] = d[128]; axy[R + 1][1].x = d[192]; axy[R + 1][1].y = d[256]; av[R + 1][1] = d[320];
                }
            }
            unsigned w7[7]; float rowu[6], rowd[6]; float cvl = 0.f, cvr = 0.f; float g_ad = 0.f, g_bn = 0.f;
#pragma unroll
            for (int c = 0; c < 7; ++c) w7[c] = 0u;
#pragma unroll
            for (int c = 0; c < 6; ++c) { rowu[c] = 0.f; rowd[c] = 0.f; }
            {
                const unsigned Txp = seqx + (unsigned)k;              // cross-rank tag of iteration k - 1
                const bool need_u = xout && (up_glb || up_gh), need_d = xout && (dn_glb || dn_gh), need_cl = col_lane && has_lf, need_cr = col_lane && has_rt;
                const bool need_s = sweeper && slot_live, need_g = !sweeper;
                const unsigned usrc = up_gh ? ghost(parp, 0) : rowh(parp, up_glb ? wid - 1 : wid, 1), dsrc = dn_gh ? ghost(parp, 1) : rowh(parp, dn_glb ? wid + 1 : wid, 0);
                const unsigned ssrc = sumw(parp, slot);
                const unsigned clsrc = colh(parp, has_lf ? wid - g.nseg : wid, 1, lane), crsrc = colh(parp, has_rt ? wid + g.nseg : wid, 0, lane);
                bool ok_s = !need_s, ok_g = !need_g, ok_u = !need_u, ok_d = !need_d, ok_cl = !need_cl, ok_cr = !need_cr;
                sp.n = 0; sp.t0 = 0;
                unsigned npass = 0;
                while (!(ok_s && ok_g && ok_u && ok_d && ok_cl && ok_cr) && !dead) {
                    ++npass;
                    asm volatile("" ::: "memory");                     // (every pass re-reads: nothing may be hoisted out of the loop)
                    u32x4 vs[4], vu[3], vd[3], vg; u32x2 vcl, vcr;
                    if (!ok_s) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) vs[c] = ld2g(RS_SUM, ssrc + 16 * c);
                    }
                    if (!ok_g) vg = ld2g(RS_GS, (unsigned)parp * 16u);
                    if (!ok_u) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) vu[c] = (DIST && up_gh) ? ld2s(RS_GH, usrc + 16 * c) : ld2g(RS_ROW, usrc + 16 * c);
                    }
                    if (!ok_d) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) vd[c] = (DIST && dn_gh) ? ld2s(RS_GH, dsrc + 16 * c) : ld2g(RS_ROW, dsrc + 16 * c);
                    }
                    if (!ok_cl) vcl = ld1g(RS_COL, clsrc);
                    if (!ok_cr) vcr = ld1g(RS_COL, crsrc);
                    if (!ok_s) {
                        w7[0] = vs[0].x; w7[1] = vs[0].z; w7[2] = vs[1].x; w7[3] = vs[1].z; w7[4] = vs[2].x; w7[5] = vs[2].z; w7[6] = vs[3].x;
                        ok_s = vs[0].y == Tp && vs[0].w == Tp && vs[1].y == Tp && vs[1].w == Tp && vs[2].y == Tp && vs[2].w == Tp && vs[3].y == Tp;
                    }
                    if (!ok_g) { g_ad = __uint_as_float(vg.x); g_bn = __uint_as_float(vg.z); ok_g = vg.y == Tp && vg.w == Tp; }
                    if (!ok_u) {
                        const unsigned tt = (DIST && up_gh) ? Txp : Tp;
#pragma unroll
                        for (int c = 0; c < 3; ++c) { rowu[2 * c] = __uint_as_float(vu[c].x); rowu[2 * c + 1] = __uint_as_float(vu[c].z); }
                        ok_u = vu[0].y == tt && vu[0].w == tt && vu[1].y == tt && vu[1].w == tt && vu[2].y == tt && vu[2].w == tt;
                    }
                    if (!ok_d) {
                        const unsigned tt = (DIST && dn_gh) ? Txp : Tp;
#pragma unroll
                        for (int c = 0; c < 3; ++c) { rowd[2 * c] = __uint_as_float(vd[c].x); rowd[2 * c + 1] = __uint_as_float(vd[c].z); }
                        ok_d = vd[0].y == tt && vd[0].w == tt && vd[1].y == tt && vd[1].w == tt && vd[2].y == tt && vd[2].w == tt;
                    }
                    if (!ok_cl) { cvl = __uint_as_float(vcl.x); ok_cl = vcl.y == Tp; }
                    if (!ok_cr) { cvr = __uint_as_float(vcr.x); ok_cr = vcr.y == Tp; }
                    if (!(ok_s && ok_g && ok_u && ok_d && ok_cl && ok_cr) && spin_fail(sp, ctl, !ok_s ? 1u : !ok_g ? 8u : !(ok_u && ok_d) ? 3u : 4u, (unsigned)wid, Tp)) dead = true;
                }
                RES_NOTE(k, 7, npass); (void)npass;
            }
            dead = __builtin_amdgcn_readfirstlane(__any(dead) ? 1 : 0) != 0;
            RES_STAMP(k, 1);
            // the columns go through LDS to the two lanes that hold them (lane 0 / 63); same wave: program order + lgkmcnt(0)
            if (col_lane) { S.crx[wave][0][lane] = cvl; S.crx[wave][1][lane] = cvr; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (the rows from the waves of my own workgroup were taken from LDS in front of the polling loop)
            if ((up_glb || up_gh) && xout) { axy[0][0].x = rowu[0]; axy[0][0].y = rowu[1]; av[0][0] = rowu[2]; axy[0][1].x = rowu[3]; axy[0][1].y = rowu[4]; av[0][1] = rowu[5]; }
            if ((dn_glb || dn_gh) && xout) { axy[R + 1][0].x = rowd[0]; axy[R + 1][0].y = rowd[1]; av[R + 1][0] = rowd[2]; axy[R + 1][1].x = rowd[3]; axy[R + 1][1].y = rowd[4]; av[R + 1][1] = rowd[5]; }
            // x halo: lane 0 takes lane 62's pixels of the strip to the left, lane 63 lane 1's pixels of the strip to the right, for each of my rows
            if ((lane == 0 && has_lf) || (lane == 63 && has_rt)) {
                const float* cx = &S.crx[wave][lane == 0 ? 0 : 1][0];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (j < nr) {
                        axy[j + 1][0].x = cx[6 * j]; axy[j + 1][0].y = cx[6 * j + 1]; av[j + 1][0] = cx[6 * j + 2];
                        axy[j + 1][1].x = cx[6 * j + 3]; axy[j + 1][1].y = cx[6 * j + 4]; av[j + 1][1] = cx[6 * j + 5];
                    }
                }
            }
            float aD = g_ad, bN = g_bn;
            if (sweeper) {
                float ad; double n, a1, b1;
                exchange_scalars(k, Tp, w7, ad, n, a1, b1);
                scalars_from_sums(k - 1, Tp, parp, aN_prev, ad, n, a1, b1, aD, bN);
            }
            alpha = safe_div<false>(aN_prev, aD);
            beta = safe_div<false>(bN, aN_prev);
            if (writer) { a.words[2 * (k - 1)] = aD; a.words[2 * (k - 1) + 1] = bN; }
            aN_prev = bN;
            RES_STAMP(k, 2);
        }
        RES_STAMP(k, 3);
        // ---- r_k = r_{k-1} - alpha A p_{k-1} ; delta += alpha p_{k-1} ; p_k = M^-1 r_k + beta p_{k-1}     (every row I hold, halo included)
#pragma unroll
        for (int jj = 0; jj < R + 2; ++jj) {
            if (k > 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    rxy[jj][q] = fma2(-alpha, axy[jj][q], rxy[jj][q]);
                    ra[jj][q] = __builtin_fmaf(-alpha, av[jj][q], ra[jj][q]);
                }
                if (jj >= 1 && jj <= R) {
                    float* d = dl + ((jj - 1) * 6) * RES_NT + threadIdx.x;
                    d[0] = __builtin_fmaf(alpha, pxy[jj][0].x, d[0]); d[RES_NT] = __builtin_fmaf(alpha, pxy[jj][0].y, d[RES_NT]);
                    d[2 * RES_NT] = __builtin_fmaf(alpha, pxy[jj][1].x, d[2 * RES_NT]); d[3 * RES_NT] = __builtin_fmaf(alpha, pxy[jj][1].y, d[3 * RES_NT]);
                    d[4 * RES_NT] = __builtin_fmaf(alpha, pa[jj][0], d[4 * RES_NT]); d[5 * RES_NT] = __builtin_fmaf(alpha, pa[jj][1], d[5 * RES_NT]);
                }
            }
            const bool act = (jj <= nr + 1) && nr > 0 && xin && (ya - 1 + jj) >= 0 && (ya - 1 + jj) < g.H;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float2 m = S.lut[(fl[jj] >> (8 * q)) & 31u];           // M^-1 by the pixel's flags, looked up where it is used: not a register per row
                const v2f pnew = m.x * rxy[jj][q] + beta * pxy[jj][q];
                pxy[jj][q] = act ? pnew : v2f{ 0.f, 0.f };
                pa[jj][q] = act ? m.y * ra[jj][q] + beta * pa[jj][q] : 0.f;
            }
        }
        RES_STAMP(k, 4);
        // ---- A p_k for my rows; the four sums; the boundary of A p_k to my neighbours
        float acc = 0.0f; double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < nr) {
                const int jm = j, jc = j + 1, jn = j + 2;
                const float4 csm = csl[jm * RES_NT], csc = csl[jc * RES_NT], csn = csl[jn * RES_NT];        // {c0, s0, c1, s1} of the rows above, here, below (my own words in LDS)
                struct PRow { v2f xy[2]; float pa[2]; }; struct GRow { v2f cs[2], gx[2]; };
                const PRow pm_ = { { pxy[jm][0], pxy[jm][1] }, { pa[jm][0], pa[jm][1] } }, pc_ = { { pxy[jc][0], pxy[jc][1] }, { pa[jc][0], pa[jc][1] } };
                const PRow pn_ = { { pxy[jn][0], pxy[jn][1] }, { pa[jn][0], pa[jn][1] } };
                // ((sin, -cos) is needed of the centre row only: the x-direction terms; the rows above / below enter through (cos, sin))
                const GRow gm_ = { { v2f{ csm.x, csm.y }, v2f{ csm.z, csm.w } }, { v2f{ 0.f, 0.f }, v2f{ 0.f, 0.f } } }, gn_ = { { v2f{ csn.x, csn.y }, v2f{ csn.z, csn.w } }, { v2f{ 0.f, 0.f }, v2f{ 0.f, 0.f } } };
                const GRow gc_ = { { v2f{ csc.x, csc.y }, v2f{ csc.z, csc.w } }, { v2f{ csc.y, -csc.x }, v2f{ csc.w, -csc.z } } };
                float am[2], ac[2], an[2], wfit[2], wdum[2];
                flag_masks(fl[jm], a.wf2, am, wdum); flag_masks(fl[jc], a.wf2, ac, wfit); flag_masks(fl[jn], a.wf2, an, wdum);
                v2f b2[2]; float bv[2];
                jtjp_pair_xy(pm_, pc_, pn_, gm_, gc_, gn_, am, ac, an, wfit, a.wr2, b2, bv);        // (every lane: the x neighbours come through wave shifts; iw_march.hpp, branch-free, on (x, y) pairs)
                if (xout) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned fq = (fl[jc] >> (8 * q)) & 255u;
                        const float pxi = pxy[jc][q].x, pyi = pxy[jc][q].y, pai = pa[jc][q];
                        acc += pxi * b2[q].x + pyi * b2[q].y + pai * bv[q];
                        const float2 mq = S.lut[fq & 31u];
                        const double dmo = mq.x, dma = mq.y, drx = rxy[jc][q].x, dry = rxy[jc][q].y, dra = ra[jc][q], dax = b2[q].x, day = b2[q].y, daa = bv[q];
                        s0 = __builtin_fma(dmo, __builtin_fma(dry, dry, drx * drx), __builtin_fma(dma, dra * dra, s0));
                        s1 = __builtin_fma(dmo, __builtin_fma(dry, day, drx * dax), __builtin_fma(dma, dra * daa, s1));
                        s2 = __builtin_fma(dmo, __builtin_fma(day, day, dax * dax), __builtin_fma(dma, daa * daa, s2));
                        axy[jc][q] = b2[q]; av[jc][q] = bv[q];
                    }
                    // the boundary goes out as soon as it exists: to another workgroup as granules, to a wave of my workgroup through LDS (tagged below)
                    if (DIST && j == 0 && up_gh) {        // my first owned row is the rank above's bottom ghost row
                        const unsigned d = ghost(par, 1), Tx = seqx + (unsigned)k + 1u;
                        st2s(RS_PA, d, Tx, b2[0].x, b2[0].y); st2s(RS_PA, d + 16, Tx, bv[0], b2[1].x); st2s(RS_PA, d + 32, Tx, b2[1].y, bv[1]);
                    }
                    if (DIST && j == R - 1 && dn_gh) {    // my last owned row is the rank below's top ghost row
                        const unsigned d = ghost(par, 0), Tx = seqx + (unsigned)k + 1u;
                        st2s(RS_PB, d, Tx, b2[0].x, b2[0].y); st2s(RS_PB, d + 16, Tx, bv[0], b2[1].x); st2s(RS_PB, d + 32, Tx, b2[1].y, bv[1]);
                    }
                    if (j == 0 && has_up && !up_gh) {
                        if (up_glb) {
                            const unsigned d = rowh(par, wid, 0);
                            st2g(RS_ROW, d, T, b2[0].x, b2[0].y); st2g(RS_ROW, d + 16, T, bv[0], b2[1].x); st2g(RS_ROW, d + 32, T, b2[1].y, bv[1]);
                        } else {
                            float* d = &S.rrow[par][wave][0][0][lane];
                            d[0] = b2[0].x; d[64] = b2[0].y; d[128] = bv[0]; d[192] = b2[1].x; d[256] = b2[1].y; d[320] = bv[1];
                        }
                    }
                    if (j == nr - 1 && has_dn && !dn_gh) {
                        if (dn_glb) {
                            const unsigned d = rowh(par, wid, 1);
                            st2g(RS_ROW, d, T, b2[0].x, b2[0].y); st2g(RS_ROW, d + 16, T, bv[0], b2[1].x); st2g(RS_ROW, d + 32, T, b2[1].y, bv[1]);
                        } else {
                            float* d = &S.rrow[par][wave][1][0][lane];
                            d[0] = b2[0].x; d[64] = b2[0].y; d[128] = bv[0]; d[192] = b2[1].x; d[256] = b2[1].y; d[320] = bv[1];
                        }
                    }
                    if (lane == 1 || lane == 62) {
                        float* d = &S.cst[wave][lane == 1 ? 0 : 1][6 * j];
                        d[0] = b2[0].x; d[1] = b2[0].y; d[2] = bv[0]; d[3] = b2[1].x; d[4] = b2[1].y; d[5] = bv[1];
                    }
                }
            }
        }
        RES_STAMP(k, 5);
        // ---- publish: the LDS rows' tags, my two columns (one store instruction each), then the workgroup's sums (wave butterflies -> LDS -> wave 0 adds the
        // four waves up in order and publishes 7 granules)
        {
            // what the neighbours wait for goes out first: the tags of the rows in LDS, the two columns; the wave butterflies come behind
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the rows and columns written to LDS above)
            if (lane == 0) {
                if (up_lds) __hip_atomic_store(&S.rtag[par][wave][0], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (dn_lds) __hip_atomic_store(&S.rtag[par][wave][1], T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (col_lane && has_lf) st1g(RS_COL, colh(par, wid, 0, lane), T, __float_as_uint(S.cst[wave][0][lane]));
            if (col_lane && has_rt) st1g(RS_COL, colh(par, wid, 1, lane), T, __float_as_uint(S.cst[wave][1][lane]));
            const float wa = wave_sum_all(acc); const double w0 = wave_sum_all_d(s0), w1 = wave_sum_all_d(s1), w2 = wave_sum_all_d(s2);
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
                    if (!(a.fault && id == 1 && k == 2)) st1g(RS_SUM, sumw(par, blockIdx.x) + 8 * lane, T, word);
                }
            }
        }
        RES_STAMP(k, 6);
    }
    // ---- what L launches would have left behind: r_{L-1}, p_{L-1}, A p_{L-1}, delta (without its last term); the last iteration's two words
    if (a.L > 0) {
        float4* Ro4 = reinterpret_cast<float4*>(a.r_out); float2* Ra2 = reinterpret_cast<float2*>(a.r_out + 2 * N);
        float4* Ao4 = reinterpret_cast<float4*>(a.A_out); float2* Aa2 = reinterpret_cast<float2*>(a.A_out + 2 * N);
        float4* qo4 = reinterpret_cast<float4*>(a.p_out); float2* qa2 = reinterpret_cast<float2*>(a.p_out + 2 * N);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j < nr && xout) {
                const int jc = j + 1;
                const long i = (long)(ya + j) * W2 + (x0 >> 1);
                Ro4[i] = make_float4(rxy[jc][0].x, rxy[jc][0].y, rxy[jc][1].x, rxy[jc][1].y); Ra2[i] = make_float2(ra[jc][0], ra[jc][1]);
                qo4[i] = make_float4(pxy[jc][0].x, pxy[jc][0].y, pxy[jc][1].x, pxy[jc][1].y); qa2[i] = make_float2(pa[jc][0], pa[jc][1]);
                Ao4[i] = make_float4(axy[jc][0].x, axy[jc][0].y, axy[jc][1].x, axy[jc][1].y); Aa2[i] = make_float2(av[jc][0], av[jc][1]);
                const float* d = dl + (j * 6) * RES_NT + threadIdx.x;
                dl4[i] = make_float4(d[0], d[RES_NT], d[2 * RES_NT], d[3 * RES_NT]); dl2[i] = make_float2(d[4 * RES_NT], d[5 * RES_NT]);
            }
        }
        const bool fold = !DIST && a.X0 != nullptr;      // PCGLinearUpdate (gauss_newton.t:901-906) rides along: every workgroup needs alpha_{L-1}
        if (id == 0 || fold) {      // (uniform per workgroup: all four waves take part in the last sweep)
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
            exchange_scalars(-1, T, w7, ad, n, a1, b1);
            float aD, bN;
            scalars_from_sums(a.L - 1, T, par, aN_prev, ad, n, a1, b1, aD, bN);
            if (writer) { a.words[2 * (a.L - 1)] = aD; a.words[2 * (a.L - 1) + 1] = bN; }
            if (fold) {      // X += delta + alpha_{L-1} p_{L-1} on my rows (k_linear_update's expressions: one fma, one add per element)
                const float al = safe_div<false>(aN_prev, aD);
                float4* X4 = reinterpret_cast<float4*>(a.X0); float2* X2 = reinterpret_cast<float2*>(a.X1);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (j < nr && xout) {
                        const int jc = j + 1;
                        const long i = (long)(ya + j) * W2 + (x0 >> 1);
                        const float* d = dl + (j * 6) * RES_NT + threadIdx.x;
                        float4 x4 = X4[i]; float2 x2 = X2[i];
                        x4.x = x4.x + __builtin_fmaf(al, pxy[jc][0].x, d[0]); x4.y = x4.y + __builtin_fmaf(al, pxy[jc][0].y, d[RES_NT]);
                        x4.z = x4.z + __builtin_fmaf(al, pxy[jc][1].x, d[2 * RES_NT]); x4.w = x4.w + __builtin_fmaf(al, pxy[jc][1].y, d[3 * RES_NT]);
                        x2.x = x2.x + __builtin_fmaf(al, pa[jc][0], d[4 * RES_NT]); x2.y = x2.y + __builtin_fmaf(al, pa[jc][1], d[5 * RES_NT]);
                        X4[i] = x4; X2[i] = x2;
                    }
                }
            }
        }
        // the next launch's tags start behind this one's (the counter lives on the device: replay-safe; advanced by the one thread that is through only when every workgroup
        // has published its last sums, i.e. has long read it)
        if (writer) __hip_atomic_store(ctl + RES_SEQ, seq + (unsigned)a.L + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


inline ResGeo make_res_geo(int W, int H, int row0, int row1, int R)
{
    ResGeo g; g.W = W; g.H = H; g.row0 = row0; g.row1 = row1; g.R = R;
    g.nstrips = (W + RES_USE - 1) / RES_USE;
    g.nseg = (row1 - row0 + R - 1) / R;
    g.nwgrow = (g.nseg + RES_NT / 64 - 1) / (RES_NT / 64);
    g.total = g.nstrips * g.nwgrow;
    return g;
}

int g_res_cap = 0;       // tests: workgroup budget (0 = the device's CU count: one workgroup per CU, all of them resident at once)
int g_res_rows = 0;      // tests / tools: rows per segment (0 = automatic)
int g_res_fault = 0;     // tests: fault injection (ResArgs::fault)
int g_res_spin_ms = -1;  // tests: the bound of the kernel's waits in ms, written in front of the next launch (-1: leave the plan's control word)

// rows per wave segment, or 0 = the shape does not fit.  below: the slab has a ghost row under its last owned row (a rank below) -- the last segment must
// then be a full one (its row below sits at a fixed place in the wave's registers)
inline int res_rows(int W, int rows, bool below = false)
{
    if (W < 2 || (W & 1) || rows < 1) return 0;
    // every workgroup must be RESIDENT (they wait for each other) and a wave's sums sweep covers slots 64 * wave + lane < 256: at most
    // min(CUs, 256) workgroups, one per CU
    long cap = g_res_cap > 0 ? g_res_cap : thallo_hip_device_cu_count();
    if (cap > 256) cap = 256;
    const int nstrips = (W + RES_USE - 1) / RES_USE;
    int R = g_res_rows > 0 ? g_res_rows : march_rows_per_segment(rows, nstrips, RES_NT / 64, cap, 2);
    if (below) while (R > 0 && R <= RES_MAX_R && rows % R != 0) ++R;
    if (R <= 0 || R > RES_MAX_R) return 0;
    const ResGeo g = make_res_geo(W, rows, 0, rows, R);
    if ((g.total + 7) / 8 * 8 > cap || (g.total + 7) / 8 * 8 > THALLO_MAX_PARTIALS) return 0;      // every workgroup must be resident: they wait for each other
    return R;
}

inline size_t res_lds_bytes(int R) { return ((sizeof(ResLds) + 15) & ~(size_t)15) + ((size_t)R * 6 + (size_t)(R + 2) * 4) * RES_NT * sizeof(float); }

// the plan's exchange memory: [control words | the two global words | sums records | column granules | row granules]  (the control words come first: their
// place does not depend on the rows per segment)
struct ResLayout { long ctl, gs, sums, colh, rowh, bytes; };
inline ResLayout res_layout(const ResGeo& g)
{
    const long waves = (long)g.nstrips * g.nseg;
    ResLayout l;
    l.ctl = 0; l.gs = 32; l.sums = l.gs + 8; l.colh = l.sums + 2L * 8 * THALLO_MAX_PARTIALS; l.rowh = l.colh + 2 * waves * 2 * 64;       // (u64 units; 32 u64 = 256 bytes of control words)
    l.bytes = (l.rowh + 2 * waves * 2 * 64 * 6) * (long)sizeof(u64) + 256;
    return l;
}
inline void res_bufs(void* xbuf, const ResGeo& g, ResBufs& b)
{
    const ResLayout l = res_layout(g);
    u64* base = reinterpret_cast<u64*>(xbuf);
    b.rowh = base + l.rowh; b.colh = base + l.colh; b.sums = base + l.sums; b.gs = base + l.gs; b.ctl = reinterpret_cast<unsigned*>(base + l.ctl);
}

template <bool DIST>
int res_launch(const ResArgs& a, int R, hipStream_t s)
{
    const int grid = (a.g.total + 7) / 8 * 8;
    const size_t lds = res_lds_bytes(R);
    {   // co-residency is a precondition, not an assumption: the kernel's workgroups wait for each other, so what the device says it can hold of THIS
        // instantiation with THIS much LDS must cover the grid (asked once per instantiation)
        static int fits[RES_MAX_R + 1][2] = {};
        int& f = fits[R][DIST ? 1 : 0];
        if (f == 0) {
            int per_cu = 0; hipError_t e = hipErrorNotSupported;
#define RES_OCC(RR) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_resident<RR, DIST>, RES_NT, lds)
            switch (R) { case 1: RES_OCC(1); break; case 2: RES_OCC(2); break; case 3: RES_OCC(3); break; case 4: RES_OCC(4); break; case 5: RES_OCC(5); break;
                         case 6: RES_OCC(6); break; case 7: RES_OCC(7); break; case 8: RES_OCC(8); break; case 9: RES_OCC(9); break; case 10: RES_OCC(10); break; default: break; }
#undef RES_OCC
            f = (e == hipSuccess && per_cu >= 1) ? per_cu : -1;
        }
        if (f < 0 || (long)f * thallo_hip_device_cu_count() < grid) return -(int)hipErrorNotSupported;
    }
    if (g_res_spin_ms >= 0) { const unsigned v = (unsigned)g_res_spin_ms; if (hipMemcpyAsync(a.b.ctl + RES_SPIN_MS, &v, sizeof(v), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
#define RES_LAUNCH(RR) hipLaunchKernelGGL((k_pcg_resident<RR, DIST>), dim3(grid), dim3(RES_NT), lds, s, a)
    switch (R) {
        case 1: RES_LAUNCH(1); break; case 2: RES_LAUNCH(2); break; case 3: RES_LAUNCH(3); break;
        case 4: RES_LAUNCH(4); break; case 5: RES_LAUNCH(5); break; case 6: RES_LAUNCH(6); break; case 7: RES_LAUNCH(7); break;
        case 8: RES_LAUNCH(8); break; case 9: RES_LAUNCH(9); break; case 10: RES_LAUNCH(10); break;
        default: return -(int)hipErrorNotSupported;
    }
#undef RES_LAUNCH
    int e = check_launch(); return e ? e : grid;
}

}  // namespace

extern "C" {

#ifdef THALLO_MARCH_SWEEP
int thallo_hip_debug_stamps_resident(unsigned long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_r), &buf, sizeof buf) == hipSuccess ? 0 : -1; }
#endif

void thallo_hip_resident_debug_set(int what, int value) { if (what == 0) g_res_rows = value; if (what == 1) g_res_cap = value; if (what == 2) g_res_fault = value; if (what == 3) g_res_spin_ms = value; }

/* rows per wave segment of the resident PCG kernel on `rows` owned rows of a W-wide image, or 0: the shape does not fit the chip's registers
 * (more than RES_MAX_R rows per wave at one workgroup per CU) and the caller runs one launch per PCG iteration.  below != 0: a row slab with a rank below. */
int thallo_hip_iw_resident_rows(int W, int rows) { return res_rows(W, rows); }
int thallo_hip_iw_resident_rows_slab(int W, int rows, int below) { return res_rows(W, rows, below != 0); }

/* bytes of exchange memory a plan needs for the resident kernel (zero-filled by the caller once; layout private to this file) */
long thallo_hip_iw_resident_bytes(int W, int rows)
{
    // (sized for the smallest admissible R: the largest wave count; a slab whose last segment must be full may run with a larger one)
    const int R = res_rows(W, rows);
    if (R <= 0) return 0;
    return res_layout(make_res_geo(W, rows, 0, rows, R)).bytes;
}
/* bytes of the ghost area a rank's mailbox block carries behind its scalar granules (multi-GPU form) */
long thallo_hip_iw_resident_ghost_bytes(int W) { return W < 2 ? 0 : 2L * ((W + RES_USE - 1) / RES_USE) * 2 * 64 * 48; }

/* The PCG loop of one Gauss-Newton step in one launch: L iterations from what thallo_hip_iw_pcg_init left (r_0 in r_in, zeros in p_in and delta, cs / flags,
 * alphaN_0), leaving what L launches of thallo_hip_iw_pcg_iter_march leave: r_{L-1}, A p_{L-1}, p_{L-1} in the *_out planes, delta without its last term, and
 * words[2k] = alphaD_k, words[2k + 1] = betaN_k; with X_offset / X_angle (round 6) PCGLinearUpdate rides along: X += delta + alpha_{L-1} p_{L-1}, thallo_hip_linear_update's
 * bits.  The *_out planes may be the *_in planes (every workgroup has read its rows and halo before any workgroup can
 * be through its L iterations: each iteration needs every workgroup's sums).  xbuf: thallo_hip_iw_resident_bytes() bytes, zeroed once by the caller, private to the plan.
 * Returns the number of workgroups (> 0), -hipErrorNotSupported when the shape does not fit, another negative hipError_t on failure.
 * Replaces gauss_newton.t:1615-1687 for shapes whose solver state fits the chip's registers. */
int thallo_hip_iw_pcg_resident(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                               const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                               thallo_sum_t alphaN0, float* words, const int* irregular, float* X_offset, float* X_angle, void* xbuf, int L, thallo_stream_t stream)
{
    if ((X_offset == nullptr) != (X_angle == nullptr)) return -(int)hipErrorInvalidValue;
    if (row0 != 0 || row1 != H || H < 1 || (W & 1) || W < 2 || L < 1) return -(int)hipErrorInvalidValue;      // (whole images; the row-slab form is thallo_hip_iw_pcg_resident_dist)
    if (!cs || !flags || !r_in || !p_in || !r_out || !Ap_out || !p_out || !delta || !words || !xbuf || !alphaN0.partials) return -(int)hipErrorInvalidValue;
    const int R = res_rows(W, row1 - row0);
    if (R <= 0) return -(int)hipErrorNotSupported;
    ResArgs a; memset(&a, 0, sizeof(a));
    a.g = make_res_geo(W, H, row0, row1, R);
    res_bufs(xbuf, a.g, a.b);
    a.cs = cs; a.flags = flags; a.wf2 = w_fit * w_fit; a.wr2 = w_reg * w_reg;
    a.r_in = r_in; a.p_in = p_in; a.r_out = r_out; a.A_out = Ap_out; a.p_out = p_out; a.delta = delta;
    a.aN0 = alphaN0; a.words = words; a.irregular = irregular; a.L = L; a.fault = g_res_fault; a.X0 = X_offset; a.X1 = X_angle;
    return res_launch<false>(a, R, (hipStream_t)stream);
}

/* The same for ONE RANK's row slab of a multi-GPU run (local image W x H with its ghost rows, owned rows [row0, row1)): the first / last owned row of A p_k
 * goes straight into the neighbouring ranks' ghost areas (d.peer_mail[rank -+ 1] + ghost_off bytes: thallo_hip_iw_resident_ghost_bytes() bytes behind the scalar
 * granules of every rank's mailbox block), workgroup 0 adds this rank's sums up, exchanges them with the other ranks through the mailbox slots slot0 + 7 k ..
 * (the granules, slots and rank order of thallo_hip_iw_pcg_iter_march_dist: same bits) and publishes alphaD_k / betaN_k for the rest of the chip.  alphaN0: the
 * global alphaN_0 (one word).  L <= 4095.  The caller has advanced the GN step counter (thallo_hip_dist_begin_step). */
int thallo_hip_iw_pcg_resident_dist(int W, int H, int row0, int row1, const float* cs, const unsigned char* flags, float w_fit, float w_reg,
                                    const float* r_in, const float* p_in, float* r_out, float* Ap_out, float* p_out, float* delta,
                                    thallo_sum_t alphaN0, float* words, const int* irregular, void* xbuf, thallo_dist_t d, long ghost_off, int slot0, int L, thallo_stream_t stream)
{
    if (row0 < 0 || row1 > H || row0 >= row1 || (W & 1) || W < 2 || L < 1 || L > 4095) return -(int)hipErrorInvalidValue;
    if (!cs || !flags || !r_in || !p_in || !r_out || !Ap_out || !p_out || !delta || !words || !xbuf || !alphaN0.partials) return -(int)hipErrorInvalidValue;
    if (d.world < 1 || d.world > THALLO_DIST_MAX_WORLD || 7 * d.world > 64 || !d.mail || !d.ctl || ghost_off < 0 || (ghost_off & 15) || slot0 < 0) return -(int)hipErrorInvalidValue;
    const bool above = row0 > 0, below = row1 < H;
    if ((above && (d.rank < 1 || !d.peer_mail[d.rank - 1])) || (below && (d.rank + 1 >= d.world || !d.peer_mail[d.rank + 1])) || row0 > 1 || H - row1 > 1) return -(int)hipErrorInvalidValue;
    const int R = res_rows(W, row1 - row0, below);
    if (R <= 0) return -(int)hipErrorNotSupported;
    ResArgs a; memset(&a, 0, sizeof(a));
    a.g = make_res_geo(W, H, row0, row1, R);
    res_bufs(xbuf, a.g, a.b);
    a.x.d = d; a.x.ghost_off = (unsigned)ghost_off; a.x.slot0 = slot0; a.x.above = above; a.x.below = below;
    a.cs = cs; a.flags = flags; a.wf2 = w_fit * w_fit; a.wr2 = w_reg * w_reg;
    a.r_in = r_in; a.p_in = p_in; a.r_out = r_out; a.A_out = Ap_out; a.p_out = p_out; a.delta = delta;
    a.aN0 = alphaN0; a.words = words; a.irregular = irregular; a.L = L;
    return res_launch<true>(a, R, (hipStream_t)stream);
}

/* the error word of a plan's resident launches: 1 = a bounded wait ran out (a workgroup was not resident, or a granule never arrived); clear != 0 resets
 * it.  pm (5 words, may be NULL): what the first timed-out wait was for.  Synchronises the stream.  spin_ms >= 0 sets the bound (0 = the 2 s default). */
int thallo_hip_iw_resident_status(void* xbuf, int clear, int spin_ms, unsigned* pm, thallo_stream_t stream)
{
    if (!xbuf) return -(int)hipErrorInvalidValue;
    unsigned* ctl = reinterpret_cast<unsigned*>(xbuf);
    hipStream_t s = (hipStream_t)stream;
    unsigned w[RES_CTL_WORDS];
    if (hipMemcpyAsync(w, ctl, sizeof(w), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    if (pm) for (int i = 0; i < 5; ++i) pm[i] = w[RES_PM + i];
    if (clear && w[RES_ERR]) { const unsigned z = 0; if (hipMemcpyAsync(ctl + RES_ERR, &z, sizeof(z), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (spin_ms >= 0) { const unsigned v = (unsigned)spin_ms; if (hipMemcpyAsync(ctl + RES_SPIN_MS, &v, sizeof(v), hipMemcpyHostToDevice, s) != hipSuccess) return -(int)hipErrorUnknown; }
    if (hipStreamSynchronize(s) != hipSuccess) return -(int)hipErrorUnknown;
    return (int)w[RES_ERR];
}

}  // extern "C"
