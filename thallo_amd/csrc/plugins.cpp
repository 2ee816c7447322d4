// plugins.cpp -- the bundled energy plugins (host side).  Each one binds the caller's void**
// (API/src/util.t:609-643) and forwards to the C-ABI kernel shim (include/thallo_hip.h).
#include "plugin.hpp"
#ifdef THALLO_RESEARCH
#include "probe/thallo_hip_research.h"      // research builds only: the persistent marching loop, bundle adjustment's one-launch loop (measured slower; not in the product)
#endif
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace thallo {

int DeviceBuffer::alloc(size_t n)
{
    release();
    if (n == 0) return 0;
    hipError_t e = hipMalloc(&ptr, n);
    if (e != hipSuccess) { ptr = nullptr; set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); return -(int)e; }
    bytes = n;
    e = hipMemset(ptr, 0, n);
    return e == hipSuccess ? 0 : -(int)e;
}
void DeviceBuffer::release() { if (ptr) { hipFree(ptr); ptr = nullptr; bytes = 0; } }

// ------------------------------------------------------------------ materialized schedules (spmv_kernels.hip)
// Host-built CSR Jacobian of an energy whose J is constant (the two Laplacian known-answer energies), its transpose and,
// for `[[Jt][J]]p`, the product J^T J -- what precomputeJ + csr2csc (+ csrgemm) produce in the reference
// (gauss_newton.t:327-487,1332-1446).  mode: 0 matrix-free, 1 = `[Jt][[J]p]`, 2 = `[[Jt][J]]p`.
struct HostCsr {
    int rows = 0, cols = 0;
    std::vector<int> ptr, col; std::vector<float> val;
    void begin(int ncols) { rows = 0; cols = ncols; ptr.assign(1, 0); col.clear(); val.clear(); }
    void add(int c, float v) { col.push_back(c); val.push_back(v); }
    void end_row() { ptr.push_back((int)col.size()); ++rows; }
    HostCsr transposed() const
    {
        HostCsr t; t.rows = cols; t.cols = rows; t.ptr.assign(cols + 1, 0); t.col.resize(col.size()); t.val.resize(val.size());
        for (int c : col) ++t.ptr[c + 1];
        for (int i = 0; i < cols; ++i) t.ptr[i + 1] += t.ptr[i];
        std::vector<int> fill(t.ptr.begin(), t.ptr.end() - 1);
        for (int r = 0; r < rows; ++r) for (int k = ptr[r]; k < ptr[r + 1]; ++k) { const int d = fill[col[k]]++; t.col[d] = r; t.val[d] = val[k]; }
        return t;
    }
    // this (n x m) times b (m x n'): row-by-row with a dense accumulator over the few columns a row touches
    HostCsr times(const HostCsr& b) const
    {
        HostCsr c; c.begin(b.cols);
        std::vector<float> acc(b.cols, 0.0f); std::vector<int> mark(b.cols, -1), touched;
        for (int r = 0; r < rows; ++r) {
            touched.clear();
            for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
                const int m = col[k]; const float a = val[k];
                for (int j = b.ptr[m]; j < b.ptr[m + 1]; ++j) {
                    const int cc = b.col[j];
                    if (mark[cc] != r) { mark[cc] = r; acc[cc] = 0.0f; touched.push_back(cc); }
                    acc[cc] += a * b.val[j];
                }
            }
            std::sort(touched.begin(), touched.end());
            for (int cc : touched) c.add(cc, acc[cc]);
            c.end_row();
        }
        return c;
    }
};

struct DeviceCsr {
    int rows = 0; DeviceBuffer ptr, col, val;
    int upload(const HostCsr& h)
    {
        rows = h.rows;
        if (ptr.alloc(h.ptr.size() * sizeof(int)) || col.alloc(std::max<size_t>(1, h.col.size()) * sizeof(int)) || val.alloc(std::max<size_t>(1, h.val.size()) * sizeof(float))) return -1;
        if (hipMemcpy(ptr.ptr, h.ptr.data(), h.ptr.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return -1;
        if (!h.col.empty() && (hipMemcpy(col.ptr, h.col.data(), h.col.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                               hipMemcpy(val.ptr, h.val.data(), h.val.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)) return -1;
        return 0;
    }
    int spmv(const float* x, float* y, const float* dot_with, float* dot_out, hipStream_t s) const
    { return thallo_hip_csr_spmv(rows, (const int*)ptr.ptr, (const int*)col.ptr, (const float*)val.ptr, x, y, dot_with, dot_out, s); }
};

struct MaterializedJ {
    int mode = 0;
    DeviceCsr J, Jt, JtJ; DeviceBuffer Jp;
    int build(const HostCsr& j, int mode_)
    {
        mode = mode_;
        if (mode == 1) {
            if (J.upload(j) || Jt.upload(j.transposed()) || Jp.alloc(std::max(1, j.rows) * sizeof(float))) return -1;
        } else if (mode == 2) {
            const HostCsr jt = j.transposed();
            if (JtJ.upload(jt.times(j))) return -1;
        }
        return 0;
    }
    // Ap = J^T J p ; partials of p . Ap
    int apply(LaunchCtx& c, const float* p, float* Ap, float* out) const
    {
        if (mode == 2) { TimedLaunch t(c, "PCGStep1_JtJ"); return JtJ.spmv(p, Ap, p, out, c.stream); }
        { TimedLaunch t(c, "PCGStep1_J"); int rc = J.spmv(p, (float*)Jp.ptr, nullptr, nullptr, c.stream); if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1_Jt");
        return Jt.spmv((const float*)Jp.ptr, Ap, p, out, c.stream);
    }
};

// ------------------------------------------------------------------ tests/minimal/laplacian.t
class LaplacianImagePlugin : public EnergyPlugin {
    int W, H; float w_fit; int xguard;
    int mat_mode = 0; MaterializedJ mat;
    std::vector<UnknownImage> imgs;
    float* X = nullptr; const float* A = nullptr;
public:
    LaplacianImagePlugin(const unsigned* dims, float w, int xg, int mat) : W((int)dims[0]), H((int)dims[1]), w_fit(w), xguard(xg), mat_mode(mat)
    { imgs.push_back({ 0, (long)W * H }); }
    int prepare(LaunchCtx&) override
    {   // tests/minimal/laplacian.t:16-20 asks for materialized J / JtJ; J is constant: rows = fit(x,y), then the guarded differences
        if (!mat_mode || mat.mode) return 0;
        HostCsr j; j.begin(W * H);
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) { j.add(y * W + x, w_fit); j.end_row(); }
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            const int i = y * W + x;
            if (x + 1 < W && (xguard || y + 1 < H)) { j.add(i, 1.0f); j.add(i + 1, -1.0f); }      // InBounds(x+1,y) or the shipped InBounds(x+1,y+1)
            j.end_row();
            if (y + 1 < H) { j.add(i, 1.0f); j.add(i + W, -1.0f); }
            j.end_row();
        }
        return mat.build(j, mat_mode);
    }
    const char* name() const override { return "laplacian_image"; }
    long n_unknowns() const override { return (long)W * H; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return false; }
    int bind(void** p) override { X = (float*)p[0]; A = (const float*)p[1]; return (X && A) ? 0 : -1; }
    float* unknown_ptr(int) override { return X; }
    int cost(LaunchCtx& c, float* out) override
    { TimedLaunch t(c, "computeCost"); return thallo_hip_lapimg_cost(W, H, X, A, w_fit, xguard, out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    { TimedLaunch t(c, "PCGInit1"); return thallo_hip_lapimg_pcg_init(W, H, X, A, w_fit, xguard, v.r, v.z, v.p[cur], v.delta, v.diag, aN, c.stream); }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        if (mat.mode) return mat.apply(c, p, Ap, out);
        TimedLaunch t(c, "PCGStep1"); return thallo_hip_lapimg_apply_jtj(W, H, w_fit, xguard, p, Ap, out, c.stream);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        if (mat.mode) {
            { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
            return mat.apply(c, v.p[cur ^ 1], v.Ap, out);
        }
        TimedLaunch t(c, "PCGStep1"); return thallo_hip_lapimg_pcg_step1(W, H, w_fit, xguard, v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.Ap, first ? 1 : 0, aN, aD, bN, out, c.stream);
    }
};

// ------------------------------------------------------------------ examples/image_warping/image_warping.t
class ImageWarpingPlugin : public EnergyPlugin {
    int W, H;
    std::vector<UnknownImage> imgs;
    float *offset = nullptr, *angle = nullptr;
    const float *urshape = nullptr, *constraints = nullptr, *mask = nullptr;
    float w_fit = 0, w_reg = 0;
    DeviceBuffer cs, flags, irregular;     // per-GN-iteration planes: (cos,sin) float2, validity bits; UrShape-is-grid word
    DeviceBuffer xres;                     // exchange memory of the resident PCG kernel (granule buffers + control words)
    bool resident_ = false;                // the shape fits the resident kernel (whole image, unit pixel grid, even W, few enough rows per wave)
    bool resident_slab_ = false;           // ... as one rank's row slab of a multi-GPU run (thallo_hip_iw_pcg_resident_dist)
    bool resident_broken_ = false;         // a bounded wait of the resident kernel ran out on this plan (something kept its workgroups from being co-resident): never again
    bool march_ = false;                   // UrShape verified (at Init) to be the unit pixel grid and W even: the marching one-kernel iteration
    DeviceBuffer xpst;                     // exchange memory of the persistent marching loop (control words + tagged sums records)
    bool persist_ = false;                 // ... and iterations 1 .. L-1 of a GN step as ONE launch of it (whole image, every workgroup resident; THALLO_AB=persist=1)
    bool march_rc_ = false;                // ... in its form without an A p plane (whole image on one GPU; THALLO_MARCH=3: the stored-plane form, A/B)
    bool grid_ = false;                    // UrShape is the unit pixel grid (host-checked at Init)
    int row0_ = 0, row1_ = 0;              // owned rows (all of them unless the Plan is one row slab of a multi-GPU run)
public:
    ImageWarpingPlugin(const unsigned* dims) : W((int)dims[0]), H((int)dims[1]), row1_((int)dims[1])
    {
        const long N = (long)W * H;
        imgs.push_back({ 0, 2 * N }); imgs.push_back({ 1, N });
    }
    const char* name() const override { return "image_warping"; }
    long n_unknowns() const override { return 3L * W * H; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return true; }           // image_warping.t:11
    int bind(void** p) override
    {
        offset = (float*)p[0]; angle = (float*)p[1]; urshape = (const float*)p[2];
        constraints = (const float*)p[3]; mask = (const float*)p[4];
        if (!offset || !angle || !urshape || !constraints || !mask || !p[5] || !p[6]) { set_error("image_warping: null problem parameter"); return -1; }
        w_fit = *(const float*)p[5]; w_reg = *(const float*)p[6];       // host scalars, re-read every Init/Step
        const long N = (long)W * H;
        if (!cs.ptr) { if (cs.alloc(N * 8)) return -1; if (flags.alloc((N + 255) / 256 * 256)) return -1; if (irregular.alloc(64)) return -1; }
        return 0;
    }
    int prepare(LaunchCtx& c) override
    {   // UrShape is a constant input: establish once per Init, on the host, whether it is the unit pixel grid (what the reference's
        // harness passes, CombinedSolver.h:158-176) -- that selects thallo_hip_iw_pcg_iter_march; pcg_init still re-verifies on the device
        march_ = grid_ = march_rc_ = false;
        const char* e = env_switch("THALLO_MARCH");
        int* word = (int*)irregular.ptr + 8;
        int rc = thallo_hip_iw_urshape_irregular(W, H, urshape, word, c.stream);
        if (rc < 0) { set_error("image_warping: UrShape check failed (%d)", rc); return -1; }
        int bad = 1;
        if (hipMemcpyAsync(&bad, word, sizeof(int), hipMemcpyDeviceToHost, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess) return -1;
        // measured (tools/march_probe.py MB_MODE=rows, profiles/r02): 2048^2 82 vs 90 us, 1024^2 29.8 vs 31.5 us, 2048x256 21.1 vs 21.4 us, but 512^2 14.9 vs
        // 13.5 us -- below ~0.4 Mpixel a wave's short march is all lead-in and tail, the LDS-tiled kernel wins
        grid_ = bad == 0;
        if ((e && e[0] == '0') || (W & 1)) return 0;
        march_ = grid_ && ((long)W * H >= 400000 || (e && (e[0] == '2' || e[0] == '4')));      // THALLO_MARCH=2: the marching kernel at every size (tests); 3: with the stored A p plane (A/B); 4: both
        // an image with more 124-pixel column strips than the device has workgroup slots stays on the tile kernel (which loops over its tiles)
        if (march_ && thallo_hip_iw_march_rows(W, H) <= 0) march_ = false;
        march_rc_ = march_ && !(e && (e[0] == '3' || e[0] == '4')) && 12.0 * (double)W * (double)H < 4294967296.0;      // (whole images and row slabs alike)
        // small working sets: the whole PCG loop in one launch (state in registers); THALLO_RESIDENT=0: one launch per PCG iteration (A/B)
        resident_ = resident_slab_ = false;
        const char* er = env_switch("THALLO_RESIDENT");
        const bool whole = row0_ == 0 && row1_ == H;
        const int rr = whole ? thallo_hip_iw_resident_rows(W, H) : thallo_hip_iw_resident_rows_slab(W, row1_ - row0_, row1_ < H ? 1 : 0);
        if (grid_ && !(er && er[0] == '0') && rr > 0 && !resident_broken_) {
            const long need = thallo_hip_iw_resident_bytes(W, row1_ - row0_);
            if (need > 0 && (long)xres.bytes < need) {
                if (xres.alloc((size_t)need) || hipMemsetAsync(xres.ptr, 0, (size_t)need, c.stream) != hipSuccess) { set_error("image_warping: out of device memory for the resident kernel's exchange buffers"); return -1; }
            }
            if (need > 0) { resident_ = whole; resident_slab_ = true; }
        }
        // larger whole images: iterations 1 .. L-1 of a GN step as ONE persistent launch of the marching kernel's grid -- RESEARCH builds with THALLO_AB=persist=1 only:
        // measured 8-10 % SLOWER than a launch per iteration at 2048^2 (probe/iw_march_persist.hip, profiles/r05/persist_ab.txt); bit-identical; not in the product library
        persist_ = false;
#ifdef THALLO_RESEARCH
        const char* ep = env_switch("THALLO_PERSIST");
        if (march_rc_ && whole && !resident_broken_ && ep && ep[0] == '1' && thallo_hip_iw_march_persist_rows(W, H) > 0) {
            const long need = thallo_hip_iw_march_persist_bytes();
            if ((long)xpst.bytes < need && xpst.alloc((size_t)need)) { set_error("image_warping: out of device memory for the persistent loop's exchange buffer"); return -1; }
            persist_ = true;
        }
#endif
        return 0;
    }
    // ---- one row slab of a multi-GPU run (solver_dist.cpp)
    bool supports_row_slabs() const override { return true; }
    int set_row_slab(int row0, int row1) override
    {
        if (row0 < 0 || row1 > H || row0 >= row1 || (W & 3)) { set_error("image_warping: row slab [%d,%d) of %d rows, W = %d (W %% 4 must be 0)", row0, row1, H, W); return -1; }
        row0_ = row0; row1_ = row1; return 0;
    }
    int slab_width() const override { return W; }
    bool slab_grid_ok() const override { return grid_; }
    unsigned char* slab_flags() override { return (unsigned char*)flags.ptr; }
    int pcg_iter_dist(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, thallo_sum_t aN2, thallo_sum_t aD2,
                      const thallo_dist_t& d, float* out, int slot0, float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGIteration");
        if (march_rc_ && !(mode & 1))
            return thallo_hip_iw_pcg_iter_march_rc_dist(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                        v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                        aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, d, out, v.s12, v.fin_tickets, slot0, aD_word, bN_word, c.stream);
        if (march_)
            return thallo_hip_iw_pcg_iter_march_dist(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                     v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                     aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, d, out, v.s12, v.fin_tickets, slot0, aD_word, bN_word, c.stream);
        return thallo_hip_iw_pcg_iter_dist(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, v.pre, w_fit, w_reg,
                                           v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                           aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, d, out, v.s12, v.fin_tickets, slot0, aD_word, bN_word, c.stream);
    }
    bool dist_defers_finish() const override { return march_rc_ && march_ && thallo_hip_iw_march_rc_deferred_fits(W, row1_ - row0_) != 0; }      // (every workgroup of the launch resident)
    int pcg_iter_dist_deferred(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, thallo_sum_t aN2, thallo_sum_t aD2,
                               const thallo_prev_t& prev, int prev_slot0, unsigned long long* gs, const thallo_dist_t& d, float* out, double* s12_out) override
    {
        TimedLaunch t(c, "PCGIteration");
        if (mode & 1)       // a GN step's first iteration: the stored-plane kernel, partials only (no tickets: nothing is exchanged at its end)
            return thallo_hip_iw_pcg_iter_march_dist(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                     v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                     aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, d, out, s12_out, nullptr, 0, nullptr, nullptr, c.stream);
        return thallo_hip_iw_pcg_iter_march_rc_dist_deferred(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                             v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                             aN, aN2, aD2, prev, prev_slot0, gs, (const int*)irregular.ptr, d, out, s12_out, c.stream);
    }
    int pcg_iter_dist_finish(LaunchCtx& c, const thallo_prev_t& prev, int prev_slot0, thallo_sum_t aN, const thallo_dist_t& d, unsigned long long* gs) override
    { TimedLaunch t(c, "PCGScalars"); return thallo_hip_iw_dist_finish_deferred(prev, prev_slot0, aN, d, gs, c.stream); }
    float* unknown_ptr(int k) override { return k == 0 ? offset : angle; }
    int cost(LaunchCtx& c, float* out) override
    { TimedLaunch t(c, "computeCost"); return thallo_hip_iw_cost(W, H, row0_, row1_, offset, angle, urshape, constraints, mask, w_fit, w_reg, out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_iw_pcg_init(W, H, row0_, row1_, offset, angle, urshape, constraints, mask, w_fit, w_reg,
                                      v.r, v.pre, v.z, v.p[cur], v.delta, (float*)cs.ptr, (unsigned char*)flags.ptr, v.diag, (int*)irregular.ptr, aN, c.stream);
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_iw_apply_jtj(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, w_fit, w_reg, p, Ap, (const int*)irregular.ptr, out, c.stream);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        const thallo_sum_t none = { nullptr, 0 };
        return thallo_hip_iw_pcg_step1(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                       v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.Ap, first ? 1 : 0, aN, aD, bN, none, none, (const int*)irregular.ptr, v.r, out, c.stream);
    }
    bool batches_delta() const override { return true; }
    bool takes_any_p_plane() const override { return march_; }      // (whole images and -- round 6 -- row slabs: the marching kernels keep the ghost rows of whatever plane they write current)
    bool persist_ok() const override { return persist_; }
#ifdef THALLO_RESEARCH
    int pcg_persist(LaunchCtx& c, SolverVectors& v, float* const* planes, int n_planes, int k0, int k1, float* parts, int slots, int B, int nb_prev, thallo_sum_t alphaN_prev) override
    {
        TimedLaunch t(c, "PCGLoopPersistent");
        return thallo_hip_iw_pcg_march_persist(W, H, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg, v.rbuf(0), v.rbuf(1), planes, n_planes, k0, k1,
                                               parts, slots, B, v.s12, v.s12b, nb_prev, alphaN_prev, (const int*)irregular.ptr, xpst.ptr, c.stream);
    }
#endif      // (the marching kernels; the mode-1 launch of a GN step's first iteration included)
    bool dist_batches_delta() const override { return march_rc_; }      // (the stored-plane marching kernel's multi-GPU variant has no such form: it would spill)
    bool one_kernel_iteration() const override { return true; }
    int pcg_iter(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, thallo_sum_t aN2, thallo_sum_t aD2, float* out,
                 float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGIteration");
        if (march_rc_ && !(mode & 1))
            return thallo_hip_iw_pcg_iter_march_rc(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                   v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                   aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, out, v.s12,
                                                   aD_word ? v.fin_tickets : nullptr, aD_word, bN_word, c.stream);
        if (march_)
            return thallo_hip_iw_pcg_iter_march(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, out, v.s12,
                                                aD_word ? v.fin_tickets : nullptr, aD_word, bN_word, c.stream);
        return thallo_hip_iw_pcg_iter(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, v.pre, w_fit, w_reg,
                                      v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                      aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, out, v.s12,
                                      aD_word ? v.fin_tickets : nullptr, aD_word, bN_word, c.stream);
    }
    int pcg_iter_finish(LaunchCtx& c, SolverVectors& v, const float* part, int count, thallo_sum_t aN, float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGScalars");
        return thallo_hip_iw_pcg_iter_finish(part, v.s12, count, aN, aD_word, bN_word, c.stream);
    }
    bool resident_ok() const override { return resident_; }
    int pcg_resident(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words) override
    {
        TimedLaunch t(c, "PCGLoopResident");
        return thallo_hip_iw_pcg_resident(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg, v.rbuf(0), v.p[0],
                                          v.rbuf(L & 1), v.Abuf(L & 1), v.p[L & 1], v.delta, aN0, words, (const int*)irregular.ptr,
                                          resident_updates_unknowns() ? offset : nullptr, resident_updates_unknowns() ? angle : nullptr, xres.ptr, L, c.stream);
    }
    // round 6: PCGLinearUpdate rides in the resident launch (one GPU; THALLO_AB=iw_resident_fold=0: a launch of its own, A/B)
    bool resident_updates_unknowns() const override { const char* e = env_switch("THALLO_IW_RESIDENT_FOLD"); return resident_ && !(e && e[0] == '0'); }
    bool resident_slab_ok() const override { return resident_slab_; }
    long resident_ghost_bytes() const override { return (W & 1) ? 0 : thallo_hip_iw_resident_ghost_bytes(W); }
    int pcg_resident_dist(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words, const thallo_dist_t& d, long ghost_off, int slot0) override
    {
        if (!resident_slab_) return -(int)hipErrorNotSupported;      // (disabled since the ranks agreed -- a bounded wait ran out, THALLO_RESIDENT=0 at a later Init: the caller fails the step)
        TimedLaunch t(c, "PCGLoopResident");
        return thallo_hip_iw_pcg_resident_dist(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg, v.rbuf(0), v.p[0],
                                               v.rbuf(L & 1), v.Abuf(L & 1), v.p[L & 1], v.delta, aN0, words, (const int*)irregular.ptr, xres.ptr, d, ghost_off, slot0, L, c.stream);
    }
    int resident_status(LaunchCtx& c, int clear, unsigned* pm) override
    {
        const int a = xres.ptr ? thallo_hip_iw_resident_status(xres.ptr, clear, -1, pm, c.stream) : 0;
#ifdef THALLO_RESEARCH
        if (a == 0 && xpst.ptr) return thallo_hip_iw_march_persist_status(xpst.ptr, clear, -1, pm, c.stream);
#endif
        return a;
    }
    void resident_disable() override { resident_ = resident_slab_ = persist_ = false; resident_broken_ = true; }
    bool iter_defers_finish() const override { return true; }
    int pcg_iter_deferred(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aN2, thallo_sum_t aD2, const thallo_prev_t& prev,
                          float* out, double* s12_out) override
    {
        TimedLaunch t(c, "PCGIteration");
        if (march_rc_ && !(mode & 1))
            return thallo_hip_iw_pcg_iter_march_rc_deferred(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                            v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                            aN, aN2, aD2, prev, (const int*)irregular.ptr, out, s12_out, c.stream);
        if (march_)
            return thallo_hip_iw_pcg_iter_march_deferred(W, H, row0_, row1_, (const float*)cs.ptr, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                                         v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                                         aN, aN2, aD2, prev, (const int*)irregular.ptr, out, s12_out, c.stream);
        return thallo_hip_iw_pcg_iter_deferred(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, v.pre, w_fit, w_reg,
                                               v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, mode,
                                               aN, aN2, aD2, prev, (const int*)irregular.ptr, out, s12_out, c.stream);
    }
    int pcg_iter_finish_from(LaunchCtx& c, const float* part, const double* s12p, int count, thallo_sum_t aN, float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGScalars");
        return thallo_hip_iw_pcg_iter_finish(part, s12p, count, aN, aD_word, bN_word, c.stream);
    }
    int pcg_step1_mode(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN,
                       thallo_sum_t aN2, thallo_sum_t aD2, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_iw_pcg_step1(W, H, row0_, row1_, (const float*)cs.ptr, urshape, (const unsigned char*)flags.ptr, w_fit, w_reg,
                                       v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.Ap, mode, aN, aD, bN, aN2, aD2, (const int*)irregular.ptr, v.r, out, c.stream);
    }
    int pcg_step2(LaunchCtx& c, SolverVectors& v, thallo_sum_t aN, thallo_sum_t aD, float* out) override
    {
        TimedLaunch t(c, "PCGStep2");
        return thallo_hip_iw_pcg_step2(W, H, row0_, row1_, (const unsigned char*)flags.ptr, w_fit, w_reg, v.r, v.Ap, v.pre, v.z, aN, aD, (const int*)irregular.ptr, out, c.stream);
    }
};

// ------------------------------------------------------------------ graph incidence (host-built, once per Init)
struct GraphIncidence {
    int N = 0, E = 0;
    DeviceBuffer out_ptr, out_v1, in_ptr, in_edge, in_src;
    long ell_stride = 0;          // > 0: out_v1 / in_edge are in the ELL layout of thallo_hip.h (position j*N + n), stride maxdeg*N
    int in_slots = 1;             // ... and the in-lists have this many edge slots (in_edge / in_src: in_slots * N entries)
    const int* bound_v0 = nullptr; const int* bound_v1 = nullptr;
    // per workgroup of 256 consecutive vertices: the vertices of OTHER workgroups its vertices share an edge with (either direction), ascending: {count, 0, 0, 0, ids ...},
    // `cap` ids at most (a longer list is cut: wg_ghosts tells) -- what the resident ARAP loop stages; wg_ghosts = the longest list
    std::vector<int> wg_list; int wg_ghosts = 0;
    void build_wg_ghost_lists(const std::vector<int>& v0, const std::vector<int>& v1, int cap)
    {
        const int nwg = (N + 255) / 256, W = 4 + cap;
        std::vector<std::vector<int>> need(nwg);
        for (int e = 0; e < E; ++e) { const int a = v0[e] / 256, b = v1[e] / 256; if (a != b) { need[a].push_back(v1[e]); need[b].push_back(v0[e]); } }
        wg_list.assign((size_t)W * nwg, 0); wg_ghosts = 0;
        for (int w = 0; w < nwg; ++w) {
            std::vector<int>& s = need[w];
            std::sort(s.begin(), s.end()); s.erase(std::unique(s.begin(), s.end()), s.end());
            wg_ghosts = std::max(wg_ghosts, (int)s.size());
            const int n = std::min((int)s.size(), cap);
            wg_list[(size_t)W * w] = n;
            std::copy(s.begin(), s.begin() + n, wg_list.begin() + (size_t)W * w + 4);
        }
    }
    // the caller's sparse maps on the host, checked
    static int read_edges(int N, int E, const int* d_v0, const int* d_v1, std::vector<int>& v0, std::vector<int>& v1)
    {
        v0.resize(E); v1.resize(E);
        if (hipMemcpy(v0.data(), d_v0, sizeof(int) * E, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(v1.data(), d_v1, sizeof(int) * E, hipMemcpyDeviceToHost) != hipSuccess) { set_error("graph: cannot read the sparse maps"); return -1; }
        for (int e = 0; e < E; ++e)
            if (v0[e] < 0 || v0[e] >= N || v1[e] < 0 || v1[e] >= N) { set_error("graph: edge %d = (%d,%d) outside [0,%d)", e, v0[e], v1[e], N); return -1; }
        return 0;
    }
    // the most vertices of other workgroups (256 consecutive vertices each) that one workgroup's vertices share an edge with
    static int ghost_count(int N, const std::vector<int>& v0, const std::vector<int>& v1)
    {
        const int nwg = (N + 255) / 256;
        std::vector<std::vector<int>> need(nwg);
        for (size_t e = 0; e < v0.size(); ++e) { const int a = v0[e] / 256, b = v1[e] / 256; if (a != b) { need[a].push_back(v1[e]); need[b].push_back(v0[e]); } }
        size_t most = 0;
        for (auto& s : need) { std::sort(s.begin(), s.end()); most = std::max(most, (size_t)(std::unique(s.begin(), s.end()) - s.begin())); }
        return (int)most;
    }
    int build(int N_, int E_, const int* d_v0, const int* d_v1, bool want_ell = false)
    {
        std::vector<int> v0, v1;
        if (int rc = read_edges(N_, E_, d_v0, d_v1, v0, v1)) return rc;
        bound_v0 = d_v0; bound_v1 = d_v1;
        return build_from_host(N_, E_, v0, v1, want_ell);
    }
    int build_from_host(int N_, int E_, const std::vector<int>& v0, const std::vector<int>& v1, bool want_ell)
    {
        N = N_; E = E_;
        if (want_ell) build_wg_ghost_lists(v0, v1, thallo_hip_arap_resident_max_ghosts());
        std::vector<int> optr(N + 1, 0), iptr(N + 1, 0), ov1(E), pos(E), iedge(E), isrc(E);
        for (int e = 0; e < E; ++e) { optr[v0[e] + 1]++; iptr[v1[e] + 1]++; }
        for (int n = 0; n < N; ++n) { optr[n + 1] += optr[n]; iptr[n + 1] += iptr[n]; }
        std::vector<int> oc(optr.begin(), optr.end() - 1), ic(iptr.begin(), iptr.end() - 1);
        for (int e = 0; e < E; ++e) { pos[e] = oc[v0[e]]++; ov1[pos[e]] = v1[e]; }                 // stable: input order within a vertex
        int maxdeg = 0;
        for (int n = 0; n < N; ++n) maxdeg = std::max(maxdeg, optr[n + 1] - optr[n]);
        ell_stride = 0;
        if (want_ell && maxdeg >= 1 && maxdeg <= 32 && (long)maxdeg * N <= 3L * E + N) {          // bounded padding: ELL positions j*N + n
            ell_stride = (long)maxdeg * N;
            std::vector<int> ell_v1((size_t)ell_stride, 0);
            for (int e = 0; e < E; ++e) { const long q = (long)(pos[e] - optr[v0[e]]) * N + v0[e]; ell_v1[q] = v1[e]; pos[e] = (int)q; }
            ov1.swap(ell_v1);
        }
        int maxin = 0;
        for (int n = 0; n < N; ++n) maxin = std::max(maxin, iptr[n + 1] - iptr[n]);
        in_slots = std::max(1, maxin);
        if (ell_stride && (maxin > 32 || (long)maxin * N > 3L * E + N)) {                          // in-lists would pad too much: back to CSR everywhere
            ell_stride = 0;
            std::vector<int> oc2(optr.begin(), optr.end() - 1);
            ov1.assign(E, 0);
            for (int e = 0; e < E; ++e) { pos[e] = oc2[v0[e]]++; ov1[pos[e]] = v1[e]; }
        }
        if (ell_stride) { iedge.assign((size_t)std::max(1, maxin) * N, 0); isrc.assign((size_t)std::max(1, maxin) * N, 0); }
        for (int e = 0; e < E; ++e) {
            const int k = ic[v1[e]]++;
            const long q = ell_stride ? (long)(k - iptr[v1[e]]) * N + v1[e] : k;
            iedge[q] = pos[e]; isrc[q] = v0[e];
        }
        auto up = [&](DeviceBuffer& b, const std::vector<int>& h) {
            if (b.alloc(sizeof(int) * (h.size() + 4))) return -1;
            return hipMemcpy(b.ptr, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
        };
        if (up(out_ptr, optr) || up(out_v1, ov1) || up(in_ptr, iptr) || up(in_edge, iedge) || up(in_src, isrc)) { set_error("graph: upload failed"); return -1; }
        return 0;
    }
};

// ------------------------------------------------------------------ tests/minimal_graph/laplacian.t
class LaplacianGraphPlugin : public EnergyPlugin {
    int N, E; float w_fit;
    std::vector<UnknownImage> imgs;
    float* X = nullptr; const float* A = nullptr; const int *v0 = nullptr, *v1 = nullptr;
    GraphIncidence g;
    int mat_mode = 0; MaterializedJ mat; const int *mat_v0 = nullptr, *mat_v1 = nullptr;
    int build_materialized()
    {   // rows = fit(n), then reg(e) = X(v0(e)) - X(v1(e)); rebuilt when the caller binds other edge lists
        if (!mat_mode || (mat.mode && mat_v0 == v0 && mat_v1 == v1)) return 0;
        std::vector<int> h0(E), h1(E);
        if (E && (hipMemcpy(h0.data(), v0, (size_t)E * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
                  hipMemcpy(h1.data(), v1, (size_t)E * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)) { set_error("cannot read the edge lists"); return -1; }
        HostCsr j; j.begin(N);
        for (int n = 0; n < N; ++n) { j.add(n, w_fit); j.end_row(); }
        for (int e = 0; e < E; ++e) {
            if (h0[e] < 0 || h0[e] >= N || h1[e] < 0 || h1[e] >= N) { set_error("edge %d references vertex outside [0,%d)", e, N); return -1; }
            if (h0[e] != h1[e]) { j.add(h0[e], 1.0f); j.add(h1[e], -1.0f); }
            j.end_row();
        }
        mat = MaterializedJ();
        mat_v0 = v0; mat_v1 = v1;
        return mat.build(j, mat_mode);
    }
public:
    LaplacianGraphPlugin(const unsigned* dims, float w, int matm) : N((int)dims[0]), E((int)dims[1]), w_fit(w), mat_mode(matm) { imgs.push_back({ 0, (long)N }); }
    const char* name() const override { return "laplacian_graph"; }
    long n_unknowns() const override { return N; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return false; }
    int bind(void** p) override { X = (float*)p[0]; A = (const float*)p[1]; v0 = (const int*)p[2]; v1 = (const int*)p[3]; return (X && A && v0 && v1) ? 0 : -1; }
    int prepare(LaunchCtx&) override { if (int rc = g.build(N, E, v0, v1)) return rc; return build_materialized(); }
    float* unknown_ptr(int) override { return X; }
    int cost(LaunchCtx& c, float* out) override
    { TimedLaunch t(c, "computeCost"); return thallo_hip_lapgraph_cost(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, X, A, w_fit, out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_lapgraph_pcg_init(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, (const int*)g.in_ptr.ptr, (const int*)g.in_src.ptr,
                                            X, A, w_fit, v.r, v.z, v.p[cur], v.delta, v.diag, aN, c.stream);
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        if (mat.mode) return mat.apply(c, p, Ap, out);
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_lapgraph_apply_jtj(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, (const int*)g.in_ptr.ptr, (const int*)g.in_src.ptr, w_fit, p, Ap, out, c.stream);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        if (mat.mode) return mat.apply(c, v.p[cur ^ 1], v.Ap, out);
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_lapgraph_apply_jtj(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, (const int*)g.in_ptr.ptr, (const int*)g.in_src.ptr,
                                             w_fit, v.p[cur ^ 1], v.Ap, out, c.stream);
    }
};

int g_arap_reorder = 1;        // tools / tests: 0 = the caller's vertex numbering always
extern "C" void thallo_hip_arap_debug_reorder(int on) { g_arap_reorder = on; }
// ------------------------------------------------------------------ examples/arap_mesh_deformation/arap_mesh_deformation.t
class ArapPlugin : public EnergyPlugin {
    int N, E;
    std::vector<UnknownImage> imgs;
    float *position = nullptr, *angle = nullptr;
    const float *original = nullptr, *constraints = nullptr; const int *v0 = nullptr, *v1 = nullptr;
    float w_fit = 0, w_reg = 0;
    GraphIncidence g;
    DeviceBuffer F, G;       // per-GN-iteration edge residuals / rotation-derivative blocks (out-CSR order)
    int n0_ = 0, n1_ = 0;    // owned vertices (all of them unless the Plan is one rank of a vertex-partitioned multi-GPU run)
public:
    ArapPlugin(const unsigned* dims) : N((int)dims[0]), E((int)dims[1]), n1_((int)dims[0]) { imgs.push_back({ 2, 3L * N }); imgs.push_back({ 3, 3L * N }); }
    long range_units() const override { return N; }
    int set_owned_range(long u0, long u1) override
    {
        if (u0 < 0 || u1 > N || u0 >= u1) { set_error("arap: vertex range [%ld,%ld) of %d", u0, u1, N); return -1; }
        n0_ = (int)u0; n1_ = (int)u1; return 0;
    }
    const char* name() const override { return "arap_mesh"; }
    long n_unknowns() const override { return 6L * N; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return true; }            // arap_mesh_deformation.t:13
    int bind(void** p) override
    {
        if (!p[0] || !p[1]) { set_error("arap: null weight parameter"); return -1; }
        w_fit = *(const float*)p[0]; w_reg = *(const float*)p[1];
        position = (float*)p[2]; angle = (float*)p[3]; original = (const float*)p[4]; constraints = (const float*)p[5];
        v0 = (const int*)p[6]; v1 = (const int*)p[7];
        if (!position || !angle || !original || !constraints || !v0 || !v1) { set_error("arap: null problem parameter"); return -1; }
        return 0;
    }
    // ---- the plan's own vertex numbering (round 4).  The resident PCG loop stages, per workgroup of 256 consecutive vertices, the vertices of OTHER workgroups they share
    // an edge with (at most 768), and every iteration moves their A p through the fabric: 516 per workgroup for the 320-wide torus in its natural order, and far more
    // than fit for a mesh numbered in file order.  Recursive coordinate bisection of `Original` into patches of 256 vertices cuts that to the patch's rim (~70), so:
    // when the whole problem is on this GPU and the bisection's numbering has fewer ghosts than the caller's, the plan works in its own numbering -- the graph is built
    // from renumbered edges, Original / Constraints are gathered once per Init, the unknowns are gathered when the caller may have written them and scattered back
    // whenever the solver has (PCGLinearUpdate, an LM revert): the caller's arrays are current when Thallo_ProblemStep returns, as the reference's are.
    bool perm_ = false, import_pending_ = true;
    std::vector<int> new2old_;                                   // cached per (Original, V0, V1) pointers
    const void *perm_key_[3] = { nullptr, nullptr, nullptr };
    DeviceBuffer posP, angP, origP, consP, d_new2old;
    hipStream_t stream_ = nullptr;
    float* Pos() { return perm_ ? (float*)posP.ptr : position; }
    float* Ang() { return perm_ ? (float*)angP.ptr : angle; }
    const float* Org() const { return perm_ ? (const float*)origP.ptr : original; }
    const float* Cns() const { return perm_ ? (const float*)consP.ptr : constraints; }
    static void bisect(std::vector<int>& idx, int lo, int hi, const float* P, int blocks)
    {   // idx[lo, hi): vertices of `blocks` workgroups; split at the longest extent so that the left part is a whole number of workgroups
        if (blocks <= 1) { std::sort(idx.begin() + lo, idx.begin() + hi); return; }
        float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
        for (int i = lo; i < hi; ++i) for (int c = 0; c < 3; ++c) { const float x = P[3L * idx[i] + c]; mn[c] = std::min(mn[c], x); mx[c] = std::max(mx[c], x); }
        int ax = 0; for (int c = 1; c < 3; ++c) if (mx[c] - mn[c] > mx[ax] - mn[ax]) ax = c;
        const int lb = blocks / 2, nl = lb * 256;
        std::nth_element(idx.begin() + lo, idx.begin() + lo + nl, idx.begin() + hi,
                         [&](int a, int b) { const float x = P[3L * a + ax], y = P[3L * b + ax]; return x < y || (x == y && a < b); });
        bisect(idx, lo, lo + nl, P, lb); bisect(idx, lo + nl, hi, P, blocks - lb);
    }
    const float *orig_src_ = nullptr, *cons_src_ = nullptr;      // whose contents origP / consP hold
    int import_unknowns(hipStream_t s)
    {
        if (!perm_) return 0;
        if (orig_src_ != original || cons_src_ != constraints) {      // the caller bound other arrays between two steps (parameters are dereferenced at every step: util.t:609-643)
            if (thallo_hip_permute3(N, (const int*)d_new2old.ptr, original, (float*)origP.ptr, 0, s) < 0 || thallo_hip_permute3(N, (const int*)d_new2old.ptr, constraints, (float*)consP.ptr, 0, s) < 0) return -1;
            orig_src_ = original; cons_src_ = constraints;
        }
        if (!import_pending_) return 0;
        if (thallo_hip_permute3(N, (const int*)d_new2old.ptr, position, (float*)posP.ptr, 0, s) < 0 || thallo_hip_permute3(N, (const int*)d_new2old.ptr, angle, (float*)angP.ptr, 0, s) < 0) return -1;
        import_pending_ = false;
        return 0;
    }
public:
    void unknowns_changed() override { import_pending_ = true; }
    void unknowns_written() override
    {
        if (!perm_) return;
        (void)thallo_hip_permute3(N, (const int*)d_new2old.ptr, (const float*)posP.ptr, position, 1, stream_);
        (void)thallo_hip_permute3(N, (const int*)d_new2old.ptr, (const float*)angP.ptr, angle, 1, stream_);
    }
    // what the incidence lists (and the renumbering) were built from: a second Init with the same sparse maps behind the same pointers skips 40-70 ms of host work
    struct Built { const void *o = nullptr, *a = nullptr, *b = nullptr; unsigned long long ck = 0; int n0 = -1, n1 = -1, reorder = -1; bool valid = false; } built_;
    DeviceBuffer ckbuf_;
    int prepare(LaunchCtx& c) override
    {
        stream_ = c.stream;
        unsigned long long ck[2] = { 0, 0 };
        if ((!ckbuf_.ptr && ckbuf_.alloc(64)) || hipMemsetAsync(ckbuf_.ptr, 0, 16, c.stream) != hipSuccess ||
            thallo_hip_checksum_i32(E, v0, (unsigned long long*)ckbuf_.ptr, c.stream) < 0 || thallo_hip_checksum_i32(E, v1, (unsigned long long*)ckbuf_.ptr + 1, c.stream) < 0 ||
            hipMemcpyAsync(ck, ckbuf_.ptr, 16, hipMemcpyDeviceToHost, c.stream) != hipSuccess || hipStreamSynchronize(c.stream) != hipSuccess) { set_error("arap: cannot read the sparse maps"); return -1; }
        const unsigned long long key = ck[0] * 0x100000001b3ull + ck[1];
        if (built_.valid && built_.o == original && built_.a == v0 && built_.b == v1 && built_.ck == key && built_.n0 == n0_ && built_.n1 == n1_ && built_.reorder == g_arap_reorder) {
            import_pending_ = true;
            if (perm_ && (thallo_hip_permute3(N, (const int*)d_new2old.ptr, original, (float*)origP.ptr, 0, c.stream) < 0 ||
                          thallo_hip_permute3(N, (const int*)d_new2old.ptr, constraints, (float*)consP.ptr, 0, c.stream) < 0)) return -1;
            orig_src_ = original; cons_src_ = constraints;
            update_resident();
            return 0;
        }
        built_.valid = false;
        std::vector<int> h0, h1;
        if (int rc = GraphIncidence::read_edges(N, E, v0, v1, h0, h1)) return rc;
        g.bound_v0 = v0; g.bound_v1 = v1;
        perm_ = false; import_pending_ = true;
        if (g_arap_reorder && n0_ == 0 && n1_ == N && N >= 512) {
            if (perm_key_[0] != original || perm_key_[1] != v0 || perm_key_[2] != v1 || (int)new2old_.size() != N) {
                std::vector<float> O((size_t)3 * N);
                if (hipMemcpy(O.data(), original, O.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { set_error("arap: cannot read Original"); return -1; }
                new2old_.resize(N); for (int i = 0; i < N; ++i) new2old_[i] = i;
                bool finite = true;
                for (float x : O) if (!std::isfinite(x)) { finite = false; break; }       // (a NaN coordinate has no place in an ordering: such a mesh keeps the caller's numbering)
                if (finite) bisect(new2old_, 0, N, O.data(), (N + 255) / 256);
                perm_key_[0] = original; perm_key_[1] = v0; perm_key_[2] = v1;
            }
            std::vector<int> old2new(N), p0(E), p1(E);
            for (int i = 0; i < N; ++i) old2new[new2old_[i]] = i;
            for (int e = 0; e < E; ++e) { p0[e] = old2new[h0[e]]; p1[e] = old2new[h1[e]]; }
            if (GraphIncidence::ghost_count(N, p0, p1) < GraphIncidence::ghost_count(N, h0, h1)) {
                const size_t bytes = sizeof(float) * 3 * (size_t)N + 64;
                if ((posP.bytes < bytes && (posP.alloc(bytes) || angP.alloc(bytes) || origP.alloc(bytes) || consP.alloc(bytes) || d_new2old.alloc(sizeof(int) * (size_t)N + 64))) ||
                    hipMemcpy(d_new2old.ptr, new2old_.data(), sizeof(int) * (size_t)N, hipMemcpyHostToDevice) != hipSuccess ||
                    thallo_hip_permute3(N, (const int*)d_new2old.ptr, original, (float*)origP.ptr, 0, c.stream) < 0 ||
                    thallo_hip_permute3(N, (const int*)d_new2old.ptr, constraints, (float*)consP.ptr, 0, c.stream) < 0) { set_error("arap: out of device memory for the renumbered vertex arrays"); return -1; }
                perm_ = true; orig_src_ = original; cons_src_ = constraints;
                h0.swap(p0); h1.swap(p1);
            }
        }
        if (int rc = g.build_from_host(N, E, h0, h1, true)) return rc;
        const size_t edges = g.ell_stride ? (size_t)g.ell_stride : (size_t)E;      // ELL layout (thallo_hip.h) when its padding is bounded
        if (F.bytes < sizeof(float) * 3 * edges + 64 && (F.alloc(sizeof(float) * 3 * edges + 64) || G.alloc(sizeof(float) * 9 * edges + 64))) return -1;
        rc_ = thallo_hip_arap_recompute_supported(N, g.ell_stride) != 0;      // applyJTJ rebuilds G_e from per-vertex sines / cosines instead of reading it (thallo_hip.h)
        sc_ = rc_ || (g.ell_stride > 0 && n0_ == 0 && n1_ == N);              // (the resident loop needs them for any ELL layout)
        if (sc_ && SC.bytes < sizeof(float) * 6 * (size_t)N + 64 && SC.alloc(sizeof(float) * 6 * (size_t)N + 64)) return -1;
        if (int rc = prepare_resident()) return rc;
        built_.o = original; built_.a = v0; built_.b = v1; built_.ck = key; built_.n0 = n0_; built_.n1 = n1_; built_.reorder = g_arap_reorder; built_.valid = true;
        return 0;
    }
    bool resident_fits_ = false;
    bool rc_ = false, sc_ = false;
    DeviceBuffer ovf_;                     // the resident loop's overflow edges (ELL slots beyond 6)
    DeviceBuffer SC;
    DeviceBuffer xres_;                    // exchange memory of the resident PCG loop (thallo_hip_arap_pcg_resident)
    bool resident_ = false, resident_broken_ = false;
    int prepare_resident()
    {   // the whole PCG loop in one launch: whole problem on one GPU, the recomputing applyJTJ, every workgroup resident, at most 768 ghosts per workgroup
        resident_fits_ = false; resident_ = false;
        if (!sc_ || n0_ != 0 || n1_ != N || !thallo_hip_arap_resident_fits(N, g.ell_stride)) return 0;
        if (g.wg_ghosts > thallo_hip_arap_resident_max_ghosts()) return 0;      // (a vertex order that scatters the neighbours: one launch per iteration)
        const long need = thallo_hip_arap_resident_bytes(N);
        if ((long)xres_.bytes < need && xres_.alloc((size_t)need)) { set_error("arap: out of device memory for the resident loop's exchange buffers"); return -1; }
        if (hipMemcpy((char*)xres_.ptr + thallo_hip_arap_resident_lists_offset(N), g.wg_list.data(), g.wg_list.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { set_error("arap: ghost list upload failed"); return -1; }
        {   const long of = thallo_hip_arap_resident_overflow_floats(N, (int)(g.ell_stride / N), g.in_slots);
            if (of > 0 && (long)ovf_.bytes < of * (long)sizeof(float) && ovf_.alloc((size_t)of * sizeof(float) + 64)) { set_error("arap: out of device memory for the resident loop's overflow edges"); return -1; } }
        resident_fits_ = true;
        update_resident();
        return 0;
    }
    void update_resident()
    {
        const char* er = env_switch("THALLO_RESIDENT");
        resident_ = resident_fits_ && !(er && er[0] == '0') && !resident_broken_;
    }
public:
    bool resident_ok() const override { return resident_; }
    int pcg_resident(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words) override
    {
        TimedLaunch t(c, "PCGLoopResident");
        return thallo_hip_arap_pcg_resident(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, (const int*)g.in_ptr.ptr, (const int*)g.in_src.ptr, Cns(), Org(),
                                            (const float*)SC.ptr, w_fit, w_reg, g.ell_stride, v.r, v.Ap, v.pre, v.p[0], v.p[1], v.delta, aN0, words, xres_.ptr, (float*)ovf_.ptr, g.in_slots, L, c.stream);
    }
    int resident_status(LaunchCtx& c, int clear, unsigned* pm) override { return xres_.ptr ? thallo_hip_arap_resident_status(xres_.ptr, clear, pm, c.stream) : 0; }
    void resident_disable() override { resident_ = false; resident_broken_ = true; }
    int apply_any(LaunchCtx& c, const float* p, float* Ap, float* out, const float* r, const float* pre, double* s3, const thallo_fin_t& fin)
    {
        const int* op = (const int*)g.out_ptr.ptr; const int* ov = (const int*)g.out_v1.ptr; const int* ip = (const int*)g.in_ptr.ptr;
        const int* ie = (const int*)g.in_edge.ptr; const int* is = (const int*)g.in_src.ptr;
        if (rc_) return thallo_hip_arap_apply_jtj_rc(N, n0_, n1_, op, ov, ip, is, Cns(), Org(), (const float*)SC.ptr, w_fit, w_reg, p, Ap, out, g.ell_stride, r, pre, s3, fin, c.stream);
        if (s3) return thallo_hip_arap_apply_jtj_sums_fin(N, n0_, n1_, op, ov, ip, ie, is, Cns(), (const float*)G.ptr, w_fit, w_reg, p, Ap, out, g.ell_stride, r, pre, s3, fin, c.stream);
        return thallo_hip_arap_apply_jtj(N, n0_, n1_, op, ov, ip, ie, is, Cns(), (const float*)G.ptr, w_fit, w_reg, p, Ap, out, g.ell_stride, c.stream);
    }
    float* unknown_ptr(int k) override { return k == 0 ? Pos() : Ang(); }
    int cost(LaunchCtx& c, float* out) override
    {
        if (import_unknowns(c.stream)) return -1;
        TimedLaunch t(c, "computeCost");
        return thallo_hip_arap_cost(N, n0_, n1_, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, Pos(), Ang(), Org(), Cns(), w_fit, w_reg, out, g.ell_stride, c.stream);
    }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        if (import_unknowns(c.stream)) return -1;
        { TimedLaunch t(c, "precompute");
          int rc = thallo_hip_arap_precompute2(N, (const int*)g.out_ptr.ptr, (const int*)g.out_v1.ptr, Pos(), Ang(), Org(), w_reg, (float*)F.ptr, (float*)G.ptr, sc_ ? (float*)SC.ptr : nullptr, g.ell_stride, c.stream);
          if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_arap_pcg_init(N, n0_, n1_, (const int*)g.out_ptr.ptr, (const int*)g.in_ptr.ptr, (const int*)g.in_edge.ptr, Pos(), Cns(),
                                        (const float*)F.ptr, (const float*)G.ptr, w_fit, w_reg, v.r, v.pre, v.z, v.p[cur], v.delta, v.diag, aN, g.ell_stride, c.stream);
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
        return apply_any(c, p, Ap, out, nullptr, nullptr, nullptr, none);
    }
    bool apply_returns_sums() const override { return true; }
    int apply_jtj_sums(LaunchCtx& c, SolverVectors& v, const float* p, float* Ap, float* out, const thallo_fin_t& fin) override
    {
        TimedLaunch t(c, "PCGStep1");
        return apply_any(c, p, Ap, out, v.r, v.pre, v.s12, fin);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1");
        const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
        return apply_any(c, v.p[cur ^ 1], v.Ap, out, nullptr, nullptr, nullptr, none);
    }
};

// ------------------------------------------------------------------ examples/bundle_adjustment/bundle_adjustment.t
class BundleAdjustmentPlugin : public EnergyPlugin {
    int C, P, O;
    std::vector<UnknownImage> imgs;
    float *cameras = nullptr, *points = nullptr; const float* obs = nullptr; const int *oToC = nullptr, *oToP = nullptr;
    DeviceBuffer cam_ptr, cam_obs, q_cam, q_pt, pt_ptr, pt_pos, Jb, F;
    DeviceBuffer q_ptk, JP, JpP;          // J^T (J p) with J p formed once (thallo_hip_ba_apply_jtj2): point-order position per observation, packed point blocks, J p
    DeviceBuffer xres_;                   // control words, arrival counters and partial slots of the resident PCG loop (thallo_hip_ba_pcg_resident)
    bool resident_ = false, resident_broken_ = false;
    // Round 6: a plan-side POINT ORDER.  A camera gathers the 12-byte vectors of its ~400 points; where the caller numbers the points without regard to
    // who sees them every gather pulls a 128-byte line of its own (ladybug-1723 shape with shuffled point ids: 41.6 against 31.0 us per PCG iteration).  prepare() sorts
    // the points by the first camera that observes them when the caller's order is far from that (mean jump of the first-observing camera between consecutive points > 64);
    // the solver then works on an internal copy of the points in that order (gathered when the caller may have written them, scattered back whenever the solver writes
    // them), every index list is built in internal ids, and the caller's arrays keep the caller's order.  Not across ranks (shard form): the point block is all-reduced
    // between ranks, each of which would choose another order.  THALLO_AB=ba_renumber=0 / 1: never / always.
    bool perm_ = false, import_pending_ = true, no_renumber_ = false;
    DeviceBuffer ptsP, d_new2old, oToP_int;
    hipStream_t stream_ = nullptr;
    float* Pts() { return perm_ ? (float*)ptsP.ptr : points; }
    const int* OToP() const { return perm_ ? (const int*)oToP_int.ptr : oToP; }
    int import_points(hipStream_t s)
    {
        stream_ = s;
        if (!perm_ || !import_pending_) return 0;
        if (thallo_hip_permute3(P, (const int*)d_new2old.ptr, points, (float*)ptsP.ptr, 0, s) < 0) return -1;
        import_pending_ = false;
        return 0;
    }
    int apply2(LaunchCtx& c, SolverVectors* v, const float* p, float* Ap, float* out, const thallo_fin_t& fin = thallo_fin_t{ { nullptr, 0 }, nullptr, nullptr, nullptr })
    {
        return thallo_hip_ba_apply_jtj2_fin(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_pos.ptr, (const int*)pt_ptr.ptr,
                                            cameras, Pts(), (const float*)JP.ptr, (float*)JpP.ptr, p, Ap, out,
                                            v ? v->r : nullptr, v ? v->pre : nullptr, v ? v->s12 : nullptr, c.gate, fin, c.stream);
    }
public:
    BundleAdjustmentPlugin(const unsigned* dims) : C((int)dims[0]), P((int)dims[1]), O((int)dims[2])
    { imgs.push_back({ 0, 9L * C }); imgs.push_back({ 1, 3L * P }); }
    const char* name() const override { return "bundle_adjustment"; }
    long n_unknowns() const override { return 9L * C + 3L * P; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return true; }            // bundle_adjustment.t:9
    int bind(void** p) override
    {
        cameras = (float*)p[0]; points = (float*)p[1]; obs = (const float*)p[2]; oToC = (const int*)p[3]; oToP = (const int*)p[4];
        if (!cameras || !points || !obs || !oToC || !oToP) { set_error("bundle_adjustment: null problem parameter"); return -1; }
        return 0;
    }
    int prepare(LaunchCtx&) override
    {   // incidence lists (the reference sorts / transposes the CSR with cuSPARSE every GN iteration: gauss_newton.t:1349-1378)
        std::vector<int> oc(O), op(O);
        if (hipMemcpy(oc.data(), oToC, sizeof(int) * O, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(op.data(), oToP, sizeof(int) * O, hipMemcpyDeviceToHost) != hipSuccess) { set_error("bundle_adjustment: cannot read the sparse maps"); return -1; }
        for (int o = 0; o < O; ++o)
            if (oc[o] < 0 || oc[o] >= C || op[o] < 0 || op[o] >= P) { set_error("bundle_adjustment: observation %d -> (camera %d, point %d) out of range", o, oc[o], op[o]); return -1; }
        {   // the point order (see above)
            std::vector<int> first(P, C);
            for (int o = 0; o < O; ++o) if (oc[o] < first[op[o]]) first[op[o]] = oc[o];
            double tv = 0.0; for (int j = 0; j + 1 < P; ++j) tv += std::abs(first[j + 1] - first[j]);
            const char* e = env_switch("THALLO_BA_RENUMBER");
            perm_ = !no_renumber_ && P > 1 && ((e && e[0] == '1') || (!(e && e[0] == '0') && tv > 64.0 * (double)P));
            import_pending_ = true;
            if (perm_) {
                std::vector<int> new2old(P), old2new(P);
                for (int j = 0; j < P; ++j) new2old[j] = j;
                std::stable_sort(new2old.begin(), new2old.end(), [&](int x, int y) { return first[x] < first[y]; });
                for (int j = 0; j < P; ++j) old2new[new2old[j]] = j;
                for (int o = 0; o < O; ++o) op[o] = old2new[op[o]];
                if (ptsP.alloc(sizeof(float) * 3 * (size_t)P + 64) || d_new2old.alloc(sizeof(int) * ((size_t)P + 4)) || oToP_int.alloc(sizeof(int) * ((size_t)O + 4)) ||
                    hipMemcpy(d_new2old.ptr, new2old.data(), sizeof(int) * P, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(oToP_int.ptr, op.data(), sizeof(int) * O, hipMemcpyHostToDevice) != hipSuccess) { set_error("bundle_adjustment: out of device memory for the renumbered points"); return -1; }
            }
        }
        std::vector<int> cp(C + 1, 0), pp(P + 1, 0), cobs(O), qc(O), qp(O), pos(O), ppos(O);
        for (int o = 0; o < O; ++o) { cp[oc[o] + 1]++; pp[op[o] + 1]++; }
        for (int c = 0; c < C; ++c) cp[c + 1] += cp[c];
        for (int j = 0; j < P; ++j) pp[j + 1] += pp[j];
        std::vector<int> cc(cp.begin(), cp.end() - 1), pc(pp.begin(), pp.end() - 1);
        for (int o = 0; o < O; ++o) cobs[cc[oc[o]]++] = o;
        if (perm_)          // a renumbered plan also walks each camera's observations in the order of the (internal) point ids: its gathers then run through ptsP front to back
            for (int c = 0; c < C; ++c) std::sort(cobs.begin() + cp[c], cobs.begin() + cp[c + 1], [&](int a, int b) { return op[a] != op[b] ? op[a] < op[b] : a < b; });
        for (int q = 0; q < O; ++q) { const int o = cobs[q]; pos[o] = q; qc[q] = oc[o]; qp[q] = op[o]; }
        if (perm_) { for (int q = 0; q < O; ++q) ppos[pc[qp[q]]++] = q; }          // ... and each point's observations in camera order
        else for (int o = 0; o < O; ++o) ppos[pc[op[o]]++] = pos[o];
        auto up = [&](DeviceBuffer& b, const std::vector<int>& h) {
            if (b.alloc(sizeof(int) * (h.size() + 4))) return -1;
            return hipMemcpy(b.ptr, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
        };
        if (up(cam_ptr, cp) || up(cam_obs, cobs) || up(q_cam, qc) || up(q_pt, qp) || up(pt_ptr, pp) || up(pt_pos, ppos)) { set_error("bundle_adjustment: upload failed"); return -1; }
        if (!Jb.ptr && (Jb.alloc(sizeof(float) * 24 * (size_t)O + 64) || F.alloc(sizeof(float) * 2 * (size_t)O + 64))) return -1;
        {
            if (!q_ptk.ptr && (q_ptk.alloc(sizeof(int) * (size_t)O + 64) || JP.alloc(sizeof(float) * 6 * (size_t)O + 64) || JpP.alloc(sizeof(float) * 2 * (size_t)O + 64))) return -1;
            {   const int rc = thallo_hip_ba_point_order(O, (const int*)pt_pos.ptr, (int*)q_ptk.ptr, nullptr);
                const hipError_t se = rc < 0 ? hipSuccess : hipDeviceSynchronize();
                if (rc < 0 || se != hipSuccess) { set_error("bundle_adjustment: point order failed (launch %d, sync %d: %s)", rc, (int)se, hipGetErrorString(rc < 0 ? (hipError_t)(-rc) : se)); return -1; }
            }
        }
        // the PCG loop of a GN step in one launch (whole problem on one GPU) -- RESEARCH builds with THALLO_RESIDENT=2 only: bit-identical to three launches per iteration
        // but SLOWER at the ladybug shape (40.8 against 31.0 us per PCG iteration, profiles/r05/ba_resident_phases.txt: three grid-wide barriers of ~3 us each cost what the
        // launch boundaries did, and a static share of the camera blocks per workgroup balances worse than the hardware's dispatch of 431 of them); not in the product library
        resident_ = false;
#ifdef THALLO_RESEARCH
        {   const char* er = env_switch("THALLO_RESIDENT");
            if (er && er[0] == '2' && !resident_broken_) {
                const long need = thallo_hip_ba_resident_bytes();
                if ((long)xres_.bytes < need && xres_.alloc((size_t)need)) { set_error("bundle_adjustment: out of device memory for the resident loop's exchange buffer"); return -1; }
                resident_ = true;
            }
        }
#endif
        return 0;
    }
    bool resident_ok() const override { return resident_; }
#ifdef THALLO_RESEARCH
    int pcg_resident(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words) override
    {
        TimedLaunch t(c, "PCGLoopResident");
        const int rc = thallo_hip_ba_pcg_resident(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_pos.ptr, (const int*)pt_ptr.ptr, cameras, Pts(),
                                                  (const float*)JP.ptr, (float*)JpP.ptr, v.r, v.Ap, v.pre, v.p[0], v.p[1], v.delta, aN0, words, xres_.ptr, L, c.stream);
        return rc;
    }
    int resident_status(LaunchCtx& c, int clear, unsigned* pm) override { return xres_.ptr ? thallo_hip_ba_resident_status(xres_.ptr, clear, pm, c.stream) : 0; }
#endif
    void resident_disable() override { resident_ = false; resident_broken_ = true; }
    float* unknown_ptr(int k) override { return k == 0 ? cameras : Pts(); }
    void forbid_renumbering() override { no_renumber_ = true; }
    void unknowns_changed() override { import_pending_ = true; }
    void unknowns_written() override { if (perm_) (void)thallo_hip_permute3(P, (const int*)d_new2old.ptr, (const float*)ptsP.ptr, points, 1, stream_); }
    const char* schedule_name() const override { return perm_ ? "materialized J blocks, closed form; points renumbered by first observing camera" : "materialized J blocks, closed form"; }
    int cost(LaunchCtx& c, float* out) override
    { if (import_points(c.stream) < 0) return -1; TimedLaunch t(c, "computeCost"); return thallo_hip_ba_cost(C, P, O, cameras, Pts(), obs, oToC, OToP(), out, c.stream); }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        if (import_points(c.stream) < 0) return -1;
        { TimedLaunch t(c, "precomputeJ");
          int rc = thallo_hip_ba_compute_j(O, cameras, Pts(), obs, (const int*)cam_obs.ptr, (const int*)q_cam.ptr, (const int*)q_pt.ptr, (float*)Jb.ptr, (float*)F.ptr, c.stream);
          if (rc < 0) return rc;
          if ((rc = thallo_hip_ba_pack_point_blocks(O, (const float*)Jb.ptr, (const int*)q_ptk.ptr, (float*)JP.ptr, c.stream)) < 0) return rc; }
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_ba_pcg_init(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_ptr.ptr, (const int*)pt_pos.ptr, (const int*)q_cam.ptr,
                                      (const float*)Jb.ptr, (const float*)F.ptr, v.r, v.pre, v.z, v.p[cur], v.delta, v.diag, aN, c.stream);
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        if (c.lm_ctc)
            return thallo_hip_ba_apply_jtj2_lm(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_pos.ptr, (const int*)pt_ptr.ptr,
                                               cameras, Pts(), (const float*)JP.ptr, (float*)JpP.ptr, p, c.lm_ctc, Ap, out, c.gate, c.stream);
        return apply2(c, nullptr, p, Ap, out);
    }
    long shared_block_offset() const override { return 9L * C; }
    long shared_block_floats() const override { return 3L * P; }
    int shared_split_slots() const override { return thallo_hip_ba_apply2_camera_slots(C, P); }
    bool apply_adds_ctc() const override { return true; }
    // (PCGStep3 folded into this apply was measured: forming p_k = z + beta p_{k-1} at every gather costs the camera kernel 8 us, the launch it saves 7: not kept)
    // LM in the single-reduction form: the flat vector update, then the two gather launches with all sums; the point launch's last workgroup finishes the scalars
    // and applies the zeta test (three launches per iteration instead of four)
    bool lm_one_kernel() const override { return true; }
    bool lm_iter_after_reset() const override { return true; }
    bool lm_iter_defers_finish() const override { return true; }
    int lm_reset_residual(LaunchCtx& c, SolverVectors& v, float* bN_out) override
    {
        TimedLaunch t(c, "PCGStep2");
        return thallo_hip_ba_lm_reset_residual(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_pos.ptr, (const int*)pt_ptr.ptr, cameras, Pts(),
                                               (const float*)JP.ptr, (float*)JpP.ptr, v.delta, v.CtC, v.b, v.pre, v.r, bN_out, c.gate, c.stream);
    }
    int pcg_iter_lm(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out, const thallo_fin_t& fin, float* lm_state, int k,
                    float q_tol) override
    {
        { TimedLaunch t(c, "PCGUpdate");
          int rc;
          if (c.lm_defer_aD_word)      // the finish of iteration k - 1 rides in this launch (aD: its partials)
              rc = thallo_hip_pcg_update_lm_fin(v.r, v.Ap, v.pre, v.p[cur], v.p[cur ^ 1], v.delta, v.n, aN, aD.partials, v.s12, v.s12b, aD.count, c.lm_defer_aD_word, c.lm_defer_bN_word,
                                                lm_state, k - 1, q_tol, ((k - 1) & 1) ? 6 : 0, (k & 1) ? 6 : 0, c.stream);
          else rc = thallo_hip_pcg_update_lm(v.r, v.Ap, v.pre, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : c.lm_reset_bn_word ? 2 : 0, aN, aD, bN, c.lm_reset_bn_word, lm_state, c.stream);
          if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_ba_pcg_apply_lm(C, P, (const int*)cam_ptr.ptr, (const int*)q_pt.ptr, (const int*)pt_pos.ptr, (const int*)pt_ptr.ptr, cameras, Pts(),
                                          (const float*)JP.ptr, (float*)JpP.ptr, v.p[cur ^ 1], v.CtC, v.Ap, out, v.r, v.pre, v.delta, v.b, v.s12, v.s12b, fin, lm_state, k, q_tol,
                                          c.lm_q_in, c.lm_q_out, c.stream);
    }
    bool apply_returns_sums() const override { return true; }
    int apply_jtj_sums(LaunchCtx& c, SolverVectors& v, const float* p, float* Ap, float* out, const thallo_fin_t& fin) override
    {
        TimedLaunch t(c, "PCGStep1");
        return apply2(c, &v, p, Ap, out, fin);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        TimedLaunch t(c, "PCGStep1");
        return apply2(c, nullptr, v.p[cur ^ 1], v.Ap, out);
    }
};

// ------------------------------------------------------------------ examples/shape_from_shading/shape_from_shading.t
class ShapeFromShadingPlugin : public EnergyPlugin {
    int W, H;
    std::vector<UnknownImage> imgs;
    float hp[16];
    float* X = nullptr; const float *D = nullptr, *Im = nullptr; const unsigned char *mR = nullptr, *mC = nullptr;
    DeviceBuffer G, Wt, fl, U, R;
    int row0_ = 0, row1_ = 0, yoff_ = 0, Hg_ = 0;      // owned rows of the local image; its first row's global index; global image height (all of it on one GPU)
    // The precomputed planes (G, Wt, fl) are a function of the unknowns, the inputs and the scalar parameters.  computeCost and PCGInit1 both need them; LM
    // evaluates the cost of the updated unknowns at the END of a step and, if the step is accepted, starts the next step from the same unknowns: the planes of
    // that cost evaluation are the next PCGInit1's (the reference precomputes per GN iteration, gauss_newton.t:979-986, and has no precomputed planes in its
    // cost kernel).  `planes_valid_` drops when the driver writes the unknowns (unknowns_changed), when a parameter or a pointer changes (bind) or the slab does.
    bool planes_valid_ = false;
    int precompute(LaunchCtx& c)
    {
        if (planes_valid_) return 0;
        TimedLaunch t(c, "precompute");
        const int rc = thallo_hip_sfs_precompute(W, H, 0, H, yoff_, Hg_, hp, X, D, Im, mR, mC, (float*)G.ptr, (float*)Wt.ptr, (unsigned char*)fl.ptr, c.stream);
        planes_valid_ = rc >= 0;
        return rc;
    }
public:
    ShapeFromShadingPlugin(const unsigned* dims) : W((int)dims[0]), H((int)dims[1]), row1_((int)dims[1]), Hg_((int)dims[1]) { imgs.push_back({ 16, (long)W * H }); }
    // ---- one row slab of a multi-GPU run (solver_dist.cpp): the chain B_I -> shading row -> J^T gather has radius 2; pixel coordinates and the
    // image-border guard are global, so the kernels also get the slab's global row offset and the global height
    bool supports_row_slabs() const override { return true; }
    int slab_ghost_rows() const override { return 2; }
    int slab_width() const override { return W; }
    int set_row_slab(int row0, int row1) override
    {
        if (row0 < 0 || row1 > H || row0 >= row1 || (W & 3)) { set_error("shape_from_shading: row slab [%d,%d) of %d rows, W = %d (W %% 4 must be 0)", row0, row1, H, W); return -1; }
        row0_ = row0; row1_ = row1; return 0;
    }
    int set_slab_global(int global_row0, int global_rows) override
    {
        if (global_row0 < 0 || global_rows < global_row0 + H) { set_error("shape_from_shading: local rows [%d,%d) outside the %d global rows", global_row0, global_row0 + H, global_rows); return -1; }
        yoff_ = global_row0; Hg_ = global_rows; planes_valid_ = false; return 0;
    }
    const char* name() const override { return "shape_from_shading"; }
    long n_unknowns() const override { return (long)W * H; }
    const std::vector<UnknownImage>& unknown_images() const override { return imgs; }
    bool use_preconditioner() const override { return false; }            // no UsePreconditioner() in the .t
    int bind(void** p) override
    {
        for (int k = 0; k < 16; ++k) {
            if (!p[k]) { set_error("shape_from_shading: null scalar parameter %d", k); return -1; }
            const float v = *(const float*)p[k];
            if (memcmp(&v, &hp[k], sizeof v)) planes_valid_ = false;
            hp[k] = v;
        }
        if (X != (float*)p[16] || D != (const float*)p[17] || Im != (const float*)p[18] || mR != (const unsigned char*)p[19] || mC != (const unsigned char*)p[20]) planes_valid_ = false;
        X = (float*)p[16]; D = (const float*)p[17]; Im = (const float*)p[18]; mR = (const unsigned char*)p[19]; mC = (const unsigned char*)p[20];
        if (!X || !D || !Im || !mR || !mC) { set_error("shape_from_shading: null image parameter"); return -1; }
        const size_t N = (size_t)W * H;
        if (!G.ptr && (G.alloc(16 * N + 64) || Wt.alloc(8 * N + 64) || fl.alloc(N + 256) || U.alloc(8 * N + 64) || R.alloc(12 * N + 64))) return -1;
        return 0;
    }
    float* unknown_ptr(int) override { return X; }
    void unknowns_changed() override { planes_valid_ = false; }
    int cost(LaunchCtx& c, float* out) override
    {
        {   // planes and cost in ONE launch (marching kernel; elsewhere -hipErrorNotSupported: the two launches below).  Also when the planes are valid: the
            // cost of given unknowns then always comes out of the same kernel, bit for bit (a poll between two steps and the step's own evaluation agree),
            // for 7 us more than k_cost alone.
            TimedLaunch t(c, "precompute+computeCost");
            const int rc = thallo_hip_sfs_precompute_cost(W, H, 0, H, yoff_, Hg_, hp, X, D, Im, mR, mC, (float*)G.ptr, (float*)Wt.ptr, (unsigned char*)fl.ptr, row0_, row1_, out, c.stream);
            if (rc > 0) { planes_valid_ = true; return rc; }
            if (rc != -(int)hipErrorNotSupported) return rc;
        }
        int rc = precompute(c); if (rc < 0) return rc;
        TimedLaunch t(c, "computeCost");
        return thallo_hip_sfs_cost(W, H, row0_, row1_, yoff_, Hg_, hp, X, D, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, out, c.stream);
    }
    int pcg_init(LaunchCtx& c, SolverVectors& v, int cur, float* aN) override
    {
        int rc = precompute(c); if (rc < 0) return rc;
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_sfs_pcg_init(W, H, row0_, row1_, yoff_, Hg_, hp, X, D, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, (float*)U.ptr, (float*)R.ptr,
                                       v.r, v.z, v.p[cur], v.delta, v.diag, aN, c.stream);
    }
    int apply_jtj(LaunchCtx& c, const float* p, float* Ap, float* out) override
    {
        TimedLaunch t(c, "PCGStep1");
        if (c.lm_ctc)
            return thallo_hip_sfs_apply_jtj_lm(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, (float*)U.ptr, (float*)R.ptr,
                                               p, c.lm_ctc, Ap, out, c.gate, c.stream);
        return thallo_hip_sfs_apply_jtj_gated(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, (float*)U.ptr, (float*)R.ptr, p, Ap, out,
                                              c.gate, c.stream);
    }
    bool apply_adds_ctc() const override { return true; }
    // round 6, packed planes, the whole image on one GPU: PCGFinalizeDiagonal inside PCGInit1's launch; owed delta update + model-cost applyJTJ + dot in one launch
    bool packed() const { return thallo_hip_sfs_planes_layout(W, H) == 1 && row0_ == 0 && row1_ == H; }
    bool init_folds_lm_diagonal() const override { return packed(); }
    int pcg_init_lm(LaunchCtx& c, SolverVectors& v, int cur, float radius, float min_lm, float max_lm, int save_ssq, float* aN) override
    {
        int rc = precompute(c); if (rc < 0) return rc;
        TimedLaunch t(c, "PCGInit1");
        return thallo_hip_sfs_pcg_init_lm(W, H, row0_, row1_, yoff_, Hg_, hp, X, D, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, v.r, v.z, v.p[cur], v.delta,
                                          v.SSq, v.CtC, v.pre, v.b, radius, min_lm, max_lm, save_ssq, aN, c.stream);
    }
    bool lm_model_cost_one_launch() const override { return packed(); }
    int lm_model_cost(LaunchCtx& c, SolverVectors& v, const float* aN_words, const float* aD_words, int stride, const float* lm_state, int L, float* dJJd_out, float* db_out, bool upd) override
    {
        TimedLaunch t(c, "PCGModelCost");
        const int rc = thallo_hip_sfs_lm_model_cost(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, v.delta, v.Adelta, v.p[1], v.p[0], v.b,
                                                    aN_words, aD_words, stride, lm_state, L, dJJd_out, db_out, upd ? X : nullptr, upd ? v.prevX : nullptr, c.stream);
        if (rc >= 0 && upd) unknowns_written();
        return rc;
    }
    bool apply_folds_pupdate() const override { return thallo_hip_sfs_march_fits(W) != 0; }
    int apply_jtj_pupdate(LaunchCtx& c, const float* z, const float* p_in, float* p_out, float* Ap, float* out, bool first, thallo_sum_t aN, thallo_sum_t bN) override
    {
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_sfs_apply_jtj_lm_pupdate(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, z, p_in, p_out, c.lm_ctc, Ap, out,
                                                   first ? 1 : 0, aN, bN, c.gate, c.stream);
    }
    // GN on one GPU, small images (the reference's 640 x 480): the whole PCG loop of a step in ONE launch, state in registers (energy_sfs_resident.hip; round 6).
    // THALLO_RESIDENT=0: one launch per PCG iteration (A/B)
    DeviceBuffer xres_; bool resident_ = false, resident_broken_ = false;
    int prepare(LaunchCtx& c) override
    {
        resident_ = false;
        const char* er = env_switch("THALLO_RESIDENT");
        if (packed() && !(er && er[0] == '0') && !resident_broken_ && thallo_hip_sfs_resident_rows(W, H) > 0) {
            const long need = thallo_hip_sfs_resident_bytes(W, H);
            if (need > 0 && (long)xres_.bytes < need) {
                if (xres_.alloc((size_t)need) || hipMemsetAsync(xres_.ptr, 0, (size_t)need, c.stream) != hipSuccess) { set_error("shape_from_shading: out of device memory for the resident kernel's exchange buffers"); return -1; }
            }
            resident_ = need > 0;
        }
        return 0;
    }
    bool resident_ok() const override { return resident_ && packed(); }
    int pcg_resident(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words) override
    {
        TimedLaunch t(c, "PCGLoopResident");
        return thallo_hip_sfs_pcg_resident(W, H, yoff_, hp, (const float*)G.ptr, (const float*)Wt.ptr, v.rbuf(0), v.p[0], v.rbuf(L & 1), v.Abuf(L & 1), v.p[L & 1], v.delta, aN0, words,
                                           resident_updates_unknowns() ? X : nullptr, xres_.ptr, L, c.stream);
    }
    bool resident_updates_unknowns() const override { const char* e = env_switch("THALLO_SFS_RESIDENT_FOLD"); return !(e && e[0] == '0'); }
    bool resident_lm_ok() const override { return resident_ && packed() && thallo_hip_sfs_resident_rows_lm(W, H) > 0; }
    int pcg_resident_lm(LaunchCtx& c, SolverVectors& v, int L, thallo_sum_t aN0, float* words, float* lm_state, float q_tol, float* dJJd_out, float* db_out) override
    {
        TimedLaunch t(c, "PCGLoopResident");
        const int rc = thallo_hip_sfs_pcg_resident_lm(W, H, yoff_, hp, (const float*)G.ptr, (const float*)Wt.ptr, v.r, v.p[0], v.pre, v.CtC, v.delta, aN0, words, lm_state, q_tol, dJJd_out, db_out,
                                                      X, v.prevX, xres_.ptr, L, c.stream);
        if (rc >= 0) unknowns_written();
        return rc;
    }
    int resident_status(LaunchCtx& c, int clear, unsigned* pm) override { return xres_.ptr ? thallo_hip_sfs_resident_status(xres_.ptr, clear, -1, pm, c.stream) : 0; }
    void resident_disable() override { resident_ = false; resident_broken_ = true; }
    // GN on one GPU: one launch per PCG iteration (the marching kernel with PCGUpdate riding along; r, Ap, p ping-pong).  Across ranks: the flat form.
    bool one_kernel_iteration() const override { return thallo_hip_sfs_march_fits(W) != 0 && row0_ == 0 && row1_ == H; }
    bool takes_any_p_plane() const override { return one_kernel_iteration(); }          // round 5: p_k into a ring of planes, delta from the ring (solver.cpp ring_planes)
    bool dist_flat_form() const override { return true; }
    int pcg_iter(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, thallo_sum_t, thallo_sum_t, float* out,
                 float* aD_word, float* bN_word) override
    {
        if ((mode & ~3) || (mode & 3) == 3) return -1;              // bit 0: first; bit 1: delta left alone (the ring of p planes); no batched delta updates here (batches_delta() is false)
        TimedLaunch t(c, "PCGIteration");
        const thallo_fin_t fin = { bN, aD_word ? v.fin_tickets : nullptr, aD_word, bN_word };
        return thallo_hip_sfs_pcg_iter(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr,
                                       v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], (mode & 2) ? nullptr : v.delta, mode & 1, aN, aD, bN, out, v.s12, fin, c.stream);
    }
    // the finish of iteration k-1 inside the launch of iteration k (one GPU; THALLO_FIN_IN_KERNEL=1: the launch's own last workgroup finishes, A/B)
    bool iter_defers_finish() const override { const char* e = env_switch("THALLO_FIN_IN_KERNEL"); return one_kernel_iteration() && !(e && e[0]); }
    int pcg_iter_deferred(LaunchCtx& c, SolverVectors& v, int cur, int mode, thallo_sum_t aN, thallo_sum_t, thallo_sum_t, const thallo_prev_t& prev, float* out, double* s12_out) override
    {
        if ((mode & ~3) || (mode & 3) == 3) return -1;
        TimedLaunch t(c, "PCGIteration");
        return thallo_hip_sfs_pcg_iter_deferred(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr,
                                                v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], (mode & 2) ? nullptr : v.delta, mode & 1, aN, prev, out, s12_out, c.stream);
    }
    int pcg_iter_finish_from(LaunchCtx& c, const float* part, const double* s12p, int count, thallo_sum_t aN, float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGScalars");
        return thallo_hip_pcg_scalars_finish(part, s12p, count, aN, aD_word, bN_word, c.stream);
    }
    bool one_kernel_slab() const override { return thallo_hip_sfs_march_fits(W) != 0; }
    bool lm_one_kernel() const override { return one_kernel_iteration(); }
    bool lm_one_kernel_slab() const override { return thallo_hip_sfs_march_fits(W) != 0; }
    int pcg_iter_lm(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out, const thallo_fin_t& fin, float* lm_state, int k,
                    float q_tol) override
    {
        TimedLaunch t(c, "PCGIteration");
        return thallo_hip_sfs_pcg_iter_lm(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr,
                                          v.rbuf(cur), v.rbuf(cur ^ 1), v.Abuf(cur), v.Abuf(cur ^ 1), v.p[cur], v.p[cur ^ 1], v.delta, v.CtC, v.b, v.pre, first ? 1 : 0, aN, aD, bN, out, v.s12, v.s12b,
                                          fin, lm_state, k, q_tol, c.stream);
    }
    int pcg_iter_finish(LaunchCtx& c, SolverVectors& v, const float* part, int count, thallo_sum_t aN, float* aD_word, float* bN_word) override
    {
        TimedLaunch t(c, "PCGScalars");
        return thallo_hip_pcg_scalars_finish(part, v.s12, count, aN, aD_word, bN_word, c.stream);
    }
    bool apply_returns_sums() const override { return true; }
    int apply_jtj_sums(LaunchCtx& c, SolverVectors& v, const float* p, float* Ap, float* out, const thallo_fin_t& fin) override
    {
        TimedLaunch t(c, "PCGStep1");
        return thallo_hip_sfs_apply_jtj_sums_fin(W, H, row0_, row1_, yoff_, Hg_, hp, (const float*)G.ptr, (const float*)Wt.ptr, (const unsigned char*)fl.ptr, (float*)U.ptr, (float*)R.ptr, p, Ap, out,
                                                 v.r, v.s12, fin, c.stream);
    }
    int pcg_step1(LaunchCtx& c, SolverVectors& v, int cur, bool first, thallo_sum_t aN, thallo_sum_t aD, thallo_sum_t bN, float* out) override
    {
        { TimedLaunch t(c, "PCGStep3"); int rc = thallo_hip_pcg_pupdate(v.z, v.p[cur], v.p[cur ^ 1], v.delta, v.n, first ? 1 : 0, aN, aD, bN, c.stream); if (rc < 0) return rc; }
        return apply_jtj(c, v.p[cur ^ 1], v.Ap, out);
    }
};

EnergyPlugin* make_plugin(const ProblemSpec& spec, const unsigned* dims)
{
    auto cst = [&](const char* k, double dflt) { auto it = spec.constants.find(k); return it == spec.constants.end() ? dflt : it->second; };
    // schedule lines of the .t (thallo.t:5661-5690): J / JtJ materialized on every residual -> `[Jt][[J]p]` / `[[Jt][J]]p`
    const int mat = cst("materialize_JtJ", 0) > 0 ? 2 : cst("materialize_J", 0) > 0 ? 1 : 0;
    if (spec.energy == "laplacian_image") return new LaplacianImagePlugin(dims, (float)cst("w_fit", 0.2), (int)cst("xguard", 0), mat);
    if (spec.energy == "image_warping")   return new ImageWarpingPlugin(dims);
    if (spec.energy == "laplacian_graph") return new LaplacianGraphPlugin(dims, (float)cst("w_fit", 0.5), mat);
    if (spec.energy == "arap_mesh")       return new ArapPlugin(dims);
    if (spec.energy == "bundle_adjustment") return new BundleAdjustmentPlugin(dims);
    if (spec.energy == "shape_from_shading") return new ShapeFromShadingPlugin(dims);
    set_error("no gfx950 plugin for energy '%s' (%s)", spec.energy.c_str(), spec.file.c_str());
    return nullptr;
}

}  // namespace thallo
