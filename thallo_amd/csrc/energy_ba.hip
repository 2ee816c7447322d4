// energy_ba.hip -- bundle adjustment (examples/bundle_adjustment/bundle_adjustment.t:1-34) on the
// MATERIALIZED sparse-J path.
//
// Reference schedule for this energy ([Jt][[J]p]): precomputeJ dumps J as CSR
// (generateDumpJ gauss_newton.t:327-487: 2 rows per observation, 12 nonzeros per row), cuSPARSE sorts it,
// transposes it (csr2csc :1375-1378) and every PCG iteration runs two csrmv: Jp = J p, Ap = J^T (Jp)
// (:1493-1517).  282.7 MB of values+column indices per iteration at ladybug-1723 size.
//
// Here J is materialised once per GN iteration as ONE 96-byte block per observation
//   Jb[q] = { dr0/dcam[9], dr0/dpt[3], dr1/dcam[9], dr1/dpt[3] }      (24 floats, 16-byte aligned)
// stored in camera-sorted order q (camera incidence CSR).  The column indices are implicit (camera block,
// point block), so no colInd array and no transposed copy exist: J^T(Jp) is a GATHER
//   camera c : one wave walks the camera's contiguous blocks, wave-reduces 9 sums
//   point  p : one thread walks the point's ~4 blocks through pt_pos[]
// and Jp_q = Jc.p_cam + Jp.p_pt is recomputed on the fly by both sides (2 x 24 flops) instead of being
// written and re-read.  No atomics, no sort, no csr2csc; 192 B/observation/iteration instead of 416.
// Derivatives: forward-mode dual numbers over the expression of lib.t:514-555 -- the same partials the
// reference's symbolic AD generates (Select differentiates the taken branch, ad.t:800-809).
//
// Flat vector layout: [cameras 9*c+k | points 9*C + 3*p + k]  (thallo.t:1104-1125).
#include "device_common.hpp"
#include <cstring>
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {

constexpr int BLOCK = 256;
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

struct Jet {
    float v; float d[12];
};
__device__ __forceinline__ Jet jc(float c) { Jet r; r.v = c; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = 0.0f; return r; }
__device__ __forceinline__ Jet jvar(float c, int k) { Jet r = jc(c); r.d[k] = 1.0f; return r; }
__device__ __forceinline__ Jet operator+(const Jet& a, const Jet& b) { Jet r; r.v = a.v + b.v; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ Jet operator-(const Jet& a, const Jet& b) { Jet r; r.v = a.v - b.v; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ Jet operator*(const Jet& a, const Jet& b) { Jet r; r.v = a.v * b.v; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ Jet operator/(const Jet& a, const Jet& b) { Jet r; const float ib = 1.0f / b.v; r.v = a.v * ib; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
__device__ __forceinline__ Jet operator-(const Jet& a) { Jet r; r.v = -a.v; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ Jet jsqrt(const Jet& a) { Jet r; r.v = sqrtf(a.v); const float k = 0.5f / r.v; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * k; return r; }
__device__ __forceinline__ Jet jsin(const Jet& a, float s, float c) { Jet r; r.v = s; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * c; return r; }
__device__ __forceinline__ Jet jcos(const Jet& a, float s, float c) { Jet r; r.v = c; _Pragma("unroll")
    for (int i = 0; i < 12; ++i) r.d[i] = -a.d[i] * s; return r; }

// residual of one observation; T = float (cost) or Jet (Jacobian)
template <typename T> struct Ops;
template <> struct Ops<float> {
    static __device__ __forceinline__ float var(float c, int) { return c; }
    static __device__ __forceinline__ float cst(float c) { return c; }
    static __device__ __forceinline__ float val(float a) { return a; }
    static __device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
    static __device__ __forceinline__ void sincos_(float a, float& s, float& c) { sincosf(a, &s, &c); }
};
template <> struct Ops<Jet> {
    static __device__ __forceinline__ Jet var(float c, int k) { return jvar(c, k); }
    static __device__ __forceinline__ Jet cst(float c) { return jc(c); }
    static __device__ __forceinline__ float val(const Jet& a) { return a.v; }
    static __device__ __forceinline__ Jet sqrt_(const Jet& a) { return jsqrt(a); }
    static __device__ __forceinline__ void sincos_(const Jet& a, Jet& s, Jet& c) { float sv, cv; sincosf(a.v, &sv, &cv); s = jsin(a, sv, cv); c = jcos(a, sv, cv); }
};

template <typename T>
__device__ __forceinline__ void ba_residual(const float* __restrict__ cam, const float* __restrict__ pt, float ox, float oy, T& r0, T& r1)
{
    using O = Ops<T>;
    T c[9], X[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) c[k] = O::var(cam[k], k);
#pragma unroll
    for (int k = 0; k < 3; ++k) X[k] = O::var(pt[k], 9 + k);
    const T theta2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    T p[3];
    if (O::val(theta2) > 1e-8f) {                         // bundle_adjustment.t via lib.t:516-535
        const T theta = O::sqrt_(theta2);
        T st, ct; O::sincos_(theta, st, ct);
        const T ti = O::cst(1.0f) / theta;
        const T w0 = c[0] * ti, w1 = c[1] * ti, w2 = c[2] * ti;
        const T wx0 = w1 * X[2] - w2 * X[1], wx1 = w2 * X[0] - w0 * X[2], wx2 = w0 * X[1] - w1 * X[0];
        const T tmp = (w0 * X[0] + w1 * X[1] + w2 * X[2]) * (O::cst(1.0f) - ct);
        p[0] = X[0] * ct + wx0 * st + w0 * tmp; p[1] = X[1] * ct + wx1 * st + w1 * tmp; p[2] = X[2] * ct + wx2 * st + w2 * tmp;
    } else {                                              // lib.t:537-553: R = I + hat(w)
        p[0] = X[0] + (c[1] * X[2] - c[2] * X[1]); p[1] = X[1] + (c[2] * X[0] - c[0] * X[2]); p[2] = X[2] + (c[0] * X[1] - c[1] * X[0]);
    }
    p[0] = p[0] + c[3]; p[1] = p[1] + c[4]; p[2] = p[2] + c[5];
    const T cx = -p[0] / p[2], cy = -p[1] / p[2];
    const T r2 = cx * cx + cy * cy;
    const T dist = O::cst(1.0f) + r2 * (c[7] + c[8] * r2);
    const T fd = c[6] * dist;
    r0 = O::cst(ox) - cx * fd; r1 = O::cst(oy) - cy * fd;
}

// ------------------------------------------------------------------------------------------ the block of one observation in closed form (round 4)
// The same partials as ba_residual<Jet> (tests/test_gpu_parity.py compares the two), hand-derived so that a block costs ~150 flops from a point and 45 per-camera
// constants instead of a 96-byte load: the camera kernel of J^T (J p) rebuilds it per observation instead of streaming the 65 MB of J every PCG iteration (what
// energy_graph.hip's k_arap_apply_rc does for ARAP's per-edge blocks).  With P = R(w) X + t, c = -P.xy / P.z, pred = f (1 + r2 (l1 + l2 r2)) c, residual = obs - pred:
//   d pred / d P = A dc/dP,  A = f [dist I + 2 (l1 + 2 l2 r2) c c^T];  J_t = -d pred / d P =: D;  J_X = D R;  J_w[k] = D (dR/dw_k X);  J_f, J_l1, J_l2 = -dist c, -f r2 c, -f r2^2 c
// and dR/dw_k from R X = cos X + sin (u x X) + (1 - cos) u (u . X), u = w / |w| (lib.t:516-535; the small-angle branch R = I + hat(w), lib.t:537-553, has dR/dw_k = hat(e_k)).
struct CamPre { float R[9], B[27], t[3], f, l1, l2; };
__device__ __forceinline__ CamPre ba_cam_pre(const float* __restrict__ cam)
{
    CamPre o;
    const float w0 = cam[0], w1 = cam[1], w2 = cam[2];
    o.t[0] = cam[3]; o.t[1] = cam[4]; o.t[2] = cam[5]; o.f = cam[6]; o.l1 = cam[7]; o.l2 = cam[8];
    const float th2 = w0 * w0 + w1 * w1 + w2 * w2;
    if (th2 > 1e-8f) {
        const float th = sqrtf(th2), it = 1.0f / th;
        float s, c; sincosf(th, &s, &c);
        const float u[3] = { w0 * it, w1 * it, w2 * it }, oc = 1.0f - c;
        o.R[0] = c + oc * u[0] * u[0];        o.R[1] = oc * u[0] * u[1] - s * u[2]; o.R[2] = oc * u[0] * u[2] + s * u[1];
        o.R[3] = oc * u[1] * u[0] + s * u[2]; o.R[4] = c + oc * u[1] * u[1];        o.R[5] = oc * u[1] * u[2] - s * u[0];
        o.R[6] = oc * u[2] * u[0] - s * u[1]; o.R[7] = oc * u[2] * u[1] + s * u[0]; o.R[8] = c + oc * u[2] * u[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float du[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) du[i] = ((i == k ? 1.0f : 0.0f) - u[k] * u[i]) * it;
            float* B = o.B + 9 * k;
            // -s u_k I + s hat(du) + c u_k hat(u) + (1 - c)(du u^T + u du^T) + s u_k u u^T
            const float a = -s * u[k], b = c * u[k], e = s * u[k];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) B[3 * i + j] = oc * (du[i] * u[j] + u[i] * du[j]) + e * u[i] * u[j] + (i == j ? a : 0.0f);
            B[1] += -s * du[2] - b * u[2]; B[2] += s * du[1] + b * u[1];
            B[3] += s * du[2] + b * u[2];  B[5] += -s * du[0] - b * u[0];
            B[6] += -s * du[1] - b * u[1]; B[7] += s * du[0] + b * u[0];
        }
    } else {
        o.R[0] = 1.f; o.R[1] = -w2; o.R[2] = w1; o.R[3] = w2; o.R[4] = 1.f; o.R[5] = -w0; o.R[6] = -w1; o.R[7] = w0; o.R[8] = 1.f;
#pragma unroll
        for (int i = 0; i < 27; ++i) o.B[i] = 0.0f;
        o.B[5] = -1.f; o.B[7] = 1.f;            // hat(e_0)
        o.B[9 + 2] = 1.f; o.B[9 + 6] = -1.f;    // hat(e_1)
        o.B[18 + 1] = -1.f; o.B[18 + 3] = 1.f;  // hat(e_2)
    }
    return o;
}
struct Blk { float a[24]; };
__device__ __forceinline__ Blk ba_block(const CamPre& c, float X0, float X1, float X2)
{
    Blk b;
    const float P0 = c.R[0] * X0 + c.R[1] * X1 + c.R[2] * X2 + c.t[0], P1 = c.R[3] * X0 + c.R[4] * X1 + c.R[5] * X2 + c.t[1], P2 = c.R[6] * X0 + c.R[7] * X1 + c.R[8] * X2 + c.t[2];
    const float iz = 1.0f / P2, cx = -P0 * iz, cy = -P1 * iz;
    const float r2 = cx * cx + cy * cy, h = c.l1 + 2.0f * c.l2 * r2, dist = 1.0f + r2 * (c.l1 + c.l2 * r2), fd = c.f * dist, fh2 = 2.0f * c.f * h;
    const float a00 = fd + fh2 * cx * cx, a01 = fh2 * cx * cy, a11 = fd + fh2 * cy * cy;
    const float D0[3] = { a00 * iz, a01 * iz, (a00 * cx + a01 * cy) * iz }, D1[3] = { a01 * iz, a11 * iz, (a01 * cx + a11 * cy) * iz };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* B = c.B + 9 * k;
        const float v0 = B[0] * X0 + B[1] * X1 + B[2] * X2, v1 = B[3] * X0 + B[4] * X1 + B[5] * X2, v2 = B[6] * X0 + B[7] * X1 + B[8] * X2;
        b.a[k] = D0[0] * v0 + D0[1] * v1 + D0[2] * v2; b.a[12 + k] = D1[0] * v0 + D1[1] * v1 + D1[2] * v2;
        b.a[3 + k] = D0[k]; b.a[15 + k] = D1[k];
        b.a[9 + k] = D0[0] * c.R[k] + D0[1] * c.R[3 + k] + D0[2] * c.R[6 + k]; b.a[21 + k] = D1[0] * c.R[k] + D1[1] * c.R[3 + k] + D1[2] * c.R[6 + k];
    }
    const float fr2 = c.f * r2;
    b.a[6] = -cx * dist; b.a[7] = -cx * fr2; b.a[8] = -cx * fr2 * r2;
    b.a[18] = -cy * dist; b.a[19] = -cy * fr2; b.a[20] = -cy * fr2 * r2;
    return b;
}

// computeCost: per observation (any order)
__global__ __launch_bounds__(BLOCK) void k_cost(int O_, const float* __restrict__ cams, const float* __restrict__ pts,
                                                const float2* __restrict__ obs, const int* __restrict__ oToC, const int* __restrict__ oToP,
                                                float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (int o = blockIdx.x * BLOCK + threadIdx.x; o < O_; o += gridDim.x * BLOCK) {
        float r0, r1; const float2 ob = obs[o];
        ba_residual<float>(cams + 9L * oToC[o], pts + 3L * oToP[o], ob.x, ob.y, r0, r1);
        acc += 0.5f * (r0 * r0 + r1 * r1);
    }
    block_store_partial(acc, out, red);
}

// precomputeJ: J blocks + residuals in camera-sorted order q.  AD = the forward-mode duals (what rounds 1-3 stored; kept as the closed form's test reference)
template <bool AD>
__global__ __launch_bounds__(BLOCK) void k_compute_j(int O_, const float* __restrict__ cams, const float* __restrict__ pts,
                                                     const float2* __restrict__ obs, const int* __restrict__ cam_obs,
                                                     const int* __restrict__ q_cam, const int* __restrict__ q_pt,
                                                     float4* __restrict__ Jb, float2* __restrict__ F)
{
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < O_; q += gridDim.x * BLOCK) {
        const float2 ob = obs[cam_obs[q]];
        float4* b = Jb + 6L * q;
        if (AD) {
            Jet r0, r1;
            ba_residual<Jet>(cams + 9L * q_cam[q], pts + 3L * q_pt[q], ob.x, ob.y, r0, r1);
            b[0] = make_float4(r0.d[0], r0.d[1], r0.d[2], r0.d[3]); b[1] = make_float4(r0.d[4], r0.d[5], r0.d[6], r0.d[7]);
            b[2] = make_float4(r0.d[8], r0.d[9], r0.d[10], r0.d[11]);
            b[3] = make_float4(r1.d[0], r1.d[1], r1.d[2], r1.d[3]); b[4] = make_float4(r1.d[4], r1.d[5], r1.d[6], r1.d[7]);
            b[5] = make_float4(r1.d[8], r1.d[9], r1.d[10], r1.d[11]);
            F[q] = make_float2(r0.v, r1.v);
        } else {
            const float* X = pts + 3L * q_pt[q];
            float r0, r1;
            ba_residual<float>(cams + 9L * q_cam[q], X, ob.x, ob.y, r0, r1);
            const CamPre cp = ba_cam_pre(cams + 9L * q_cam[q]);
            const Blk k = ba_block(cp, X[0], X[1], X[2]);
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = make_float4(k.a[4 * i], k.a[4 * i + 1], k.a[4 * i + 2], k.a[4 * i + 3]);
            F[q] = make_float2(r0, r1);
        }
    }
}

__device__ __forceinline__ Blk ld_blk(const float4* __restrict__ Jb, long q)
{
    Blk b; const float4* s = Jb + 6 * q;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float4 v = s[k]; b.a[4 * k] = v.x; b.a[4 * k + 1] = v.y; b.a[4 * k + 2] = v.z; b.a[4 * k + 3] = v.w; }
    return b;
}

// Gather kernels: workgroups [0, cam_blocks) = 4 cameras each (one wave per camera); the rest = one thread per point.
// MODE 0: PCGInit1 (+_Finish): r = -J^T F, pre = guardedInvert(diag J^T J), z, p_prev = 0, delta = 0, alphaN partials
// MODE 1: PCGStep1: Ap = J^T (J p), alphaD partials
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_gather(int C_, int P_, int cam_blocks, const int* __restrict__ cam_ptr, const int* __restrict__ q_pt,
                                                  const int* __restrict__ pt_ptr, const int* __restrict__ pt_pos, const int* __restrict__ q_cam,
                                                  const float4* __restrict__ Jb, const float2* __restrict__ F, const float* __restrict__ p,
                                                  float* __restrict__ o0, float* __restrict__ pre, float* __restrict__ z,
                                                  float* __restrict__ p_prev, float* __restrict__ delta, float* __restrict__ diag_out,
                                                  float* __restrict__ part_out, const float* __restrict__ rs = nullptr, const float* __restrict__ prs = nullptr,
                                                  double* __restrict__ s3_out = nullptr)
{   // MODE 1 with rs / prs / s3_out: also the Sums3 of the single-reduction PCG form (r and M^-1 at the unknowns this thread writes)
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    float acc = 0.0f; Sums3 sm;
    const long PB = 9L * C_;                       // start of the point block in the flat vectors
    if ((int)blockIdx.x < cam_blocks) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int c = blockIdx.x * 4 + wave; c < C_; c += cam_blocks * 4) {
            float s[9], dg[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) { s[k] = 0.0f; dg[k] = 0.0f; }
            float pc[9];
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 9; ++k) pc[k] = p[9L * c + k];
            }
            for (int q = cam_ptr[c] + lane; q < cam_ptr[c + 1]; q += 64) {
                const Blk b = ld_blk(Jb, q);
                float j0, j1;
                if (MODE == 0) { const float2 f = F[q]; j0 = f.x; j1 = f.y; }
                else {
                    const float* pp = p + PB + 3L * q_pt[q];
                    const float p0 = pp[0], p1 = pp[1], p2 = pp[2];
                    j0 = b.a[9] * p0 + b.a[10] * p1 + b.a[11] * p2; j1 = b.a[21] * p0 + b.a[22] * p1 + b.a[23] * p2;
#pragma unroll
                    for (int k = 0; k < 9; ++k) { j0 += b.a[k] * pc[k]; j1 += b.a[12 + k] * pc[k]; }
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    s[k] += b.a[k] * j0 + b.a[12 + k] * j1;
                    if (MODE == 0) dg[k] += b.a[k] * b.a[k] + b.a[12 + k] * b.a[12 + k];
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) { s[k] = wave_sum_all(s[k]); if (MODE == 0) dg[k] = wave_sum_all(dg[k]); }
            if (lane < 9) {
                // pick element `lane` without dynamic register indexing
                float sv = 0.0f, dv = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) if (lane == k) { sv = s[k]; dv = dg[k]; }
                const long i = 9L * c + lane;
                if (MODE == 0) {
                    const float rr = -sv, m = guarded_invert(dv), zz = m * rr;
                    o0[i] = rr; pre[i] = m; z[i] = zz; p_prev[i] = 0.0f; delta[i] = 0.0f;
                    if (diag_out) diag_out[i] = dv;
                    acc += rr * zz;
                } else {
                    o0[i] = sv;
                    acc += p[i] * sv;
                    if (s3_out) sm.add(prs[i], rs[i], sv);
                }
            }
        }
    } else {
        const int nb = gridDim.x - cam_blocks;
        for (int j = (blockIdx.x - cam_blocks) * BLOCK + threadIdx.x; j < P_; j += nb * BLOCK) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
            float pp0 = 0.f, pp1 = 0.f, pp2 = 0.f;
            const long i = PB + 3L * j;
            if (MODE == 1) { pp0 = p[i]; pp1 = p[i + 1]; pp2 = p[i + 2]; }
            for (int k = pt_ptr[j]; k < pt_ptr[j + 1]; ++k) {
                const int q = pt_pos[k];
                const Blk b = ld_blk(Jb, q);
                float j0, j1;
                if (MODE == 0) { const float2 f = F[q]; j0 = f.x; j1 = f.y; }
                else {
                    const float* pc = p + 9L * q_cam[q];
                    j0 = b.a[9] * pp0 + b.a[10] * pp1 + b.a[11] * pp2; j1 = b.a[21] * pp0 + b.a[22] * pp1 + b.a[23] * pp2;
#pragma unroll
                    for (int m = 0; m < 9; ++m) { const float v = pc[m]; j0 += b.a[m] * v; j1 += b.a[12 + m] * v; }
                }
                s0 += b.a[9] * j0 + b.a[21] * j1; s1 += b.a[10] * j0 + b.a[22] * j1; s2 += b.a[11] * j0 + b.a[23] * j1;
                if (MODE == 0) { d0 += b.a[9] * b.a[9] + b.a[21] * b.a[21]; d1 += b.a[10] * b.a[10] + b.a[22] * b.a[22]; d2 += b.a[11] * b.a[11] + b.a[23] * b.a[23]; }
            }
            if (MODE == 0) {
                const float r0 = -s0, r1 = -s1, r2 = -s2;
                const float m0 = guarded_invert(d0), m1 = guarded_invert(d1), m2 = guarded_invert(d2);
                o0[i] = r0; o0[i + 1] = r1; o0[i + 2] = r2; pre[i] = m0; pre[i + 1] = m1; pre[i + 2] = m2;
                z[i] = m0 * r0; z[i + 1] = m1 * r1; z[i + 2] = m2 * r2;
                if (diag_out) { diag_out[i] = d0; diag_out[i + 1] = d1; diag_out[i + 2] = d2; }
                p_prev[i] = 0.f; p_prev[i + 1] = 0.f; p_prev[i + 2] = 0.f; delta[i] = 0.f; delta[i + 1] = 0.f; delta[i + 2] = 0.f;
                acc += r0 * (m0 * r0) + r1 * (m1 * r1) + r2 * (m2 * r2);
            } else {
                o0[i] = s0; o0[i + 1] = s1; o0[i + 2] = s2;
                acc += pp0 * s0 + pp1 * s1 + pp2 * s2;
                if (s3_out) { sm.add(prs[i], rs[i], s0); sm.add(prs[i + 1], rs[i + 1], s1); sm.add(prs[i + 2], rs[i + 2], s2); }
            }
        }
    }
    block_store_partial(acc, part_out, red);
    if (MODE == 1 && s3_out) block_store_sums3(sm, s3_out, redd);
}

// ------------------------------------------------------------------------------------------ J^T (J p) with J p computed ONCE
// k_gather<1> above reads every 96-byte J block twice and forms J p twice (once for the camera that owns the observation, once for
// its point, gathered through pt_pos: uncoalesced 96-byte reads + 36 bytes of the camera's p per observation).  Here:
//   k_cam2: one wave per camera (as before): full block, J p = J_cam p_cam + J_pt p_pt, camera part of J^T (J p); and (J p) goes out
//           in CAMERA order, coalesced (round 4; rounds 2-3 scattered it into point order through q_ptk: 8-byte writes, 20.3 MB of write traffic for 5.4 MB
//           of data per launch at the ladybug shape -- profiles/r04/ba_camera_kernel_pmc.json);
//   k_pt2 : one thread per point: its observations' point blocks (6 floats each, packed once per GN iteration in point order: JP, contiguous per point) and
//           their J p, gathered through pt_pos (the observation's place in camera order: an index load + an 8-byte gather from a 5.4 MB array that the L2s hold)
//           give the point part of J^T (J p).  1.3-1.5 us per PCG iteration less than the scatter, same bits (A/B of the two forms on one box: GN 42.6 -> 41.1, LM 49.6 -> 48.2 us).
// Per observation: 96 + 12 (p of the point) + 8 (J p out) in the camera kernel, 24 + 4 + 8 in the point kernel = 152 B instead of
// 2 x 96 + 48 of gathered p; two launches instead of one (the point kernel needs every camera's J p).
__global__ __launch_bounds__(BLOCK) void k_inverse_perm(int O_, const int* __restrict__ pt_pos, int* __restrict__ q_ptk)
{
    for (int k = blockIdx.x * BLOCK + threadIdx.x; k < O_; k += gridDim.x * BLOCK) q_ptk[pt_pos[k]] = k;
}
__global__ __launch_bounds__(BLOCK) void k_pack_point_blocks(int O_, const float4* __restrict__ Jb, const int* __restrict__ q_ptk, float2* __restrict__ JP)
{
    for (int q = blockIdx.x * BLOCK + threadIdx.x; q < O_; q += gridDim.x * BLOCK) {
        const float4 a = Jb[6L * q + 2], b = Jb[6L * q + 5];       // (r0.d8, r0.d9, r0.d10, r0.d11), (r1.d8 .. r1.d11): entries 9..11 are the point's
        float2* d = JP + 3L * q_ptk[q];
        d[0] = make_float2(a.y, a.z); d[1] = make_float2(a.w, b.y); d[2] = make_float2(b.z, b.w);
    }
}

// LM residual reset (gauss_newton.t:1653-1657) as the epilogue of the two launches: with p = delta and ctc set, r = b - (J^T J + CtC) delta and the partials of
// betaN = r . M^-1 r instead of A p and p . A p (nothing else is read from a reset in the single-reduction loop: M^-1 r is formed where p is)
struct ResetArgs { float* r; const float* b; const float* pre; };

__global__ __launch_bounds__(BLOCK) void k_cam2(int C_, const int* __restrict__ cam_ptr, const int* __restrict__ q_pt,
                                                const float* __restrict__ cams, const float* __restrict__ pts, const float* __restrict__ p, float* __restrict__ Ap, float2* __restrict__ JpC,
                                                float* __restrict__ part_out, const float* __restrict__ rs, const float* __restrict__ prs, double* __restrict__ s3_out,
                                                const unsigned* __restrict__ gate, const float* __restrict__ ctc, const float* __restrict__ delta, LmFin lm, ResetArgs rst)
{   // lm.b: the launch of a one-reduction LM iteration (thallo_hip_ba_pcg_apply_lm) -- also {U, T1, T2} of q's expansion in alpha (device_common.hpp SumsQ) per workgroup
    __shared__ float red[16];
    __shared__ double redd[6 * BLOCK / 64];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;
    float acc = 0.0f; Sums3 sm; SumsQ sq;
    const long PB = 9L * C_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = blockIdx.x * 4 + wave; c < C_; c += gridDim.x * 4) {
        float s[9], pc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { s[k] = 0.0f; pc[k] = p[9L * c + k]; }
        // the words of the epilogue (lanes 0 .. 8: one camera unknown each), asked for before the observation loop instead of behind the wave reduction
        const long ie = 9L * c + (lane < 9 ? lane : 8);
        const float e_p = p[ie], e_c = ctc ? ctc[ie] : 0.f, e_r = s3_out ? rs[ie] : 0.f, e_m = s3_out ? prs[ie] : rst.r ? rst.pre[ie] : 0.f,
                    e_d = lm.b ? delta[ie] : 0.f, e_b = lm.b ? lm.b[ie] : rst.r ? rst.b[ie] : 0.f;
        const CamPre cp = ba_cam_pre(cams + 9L * c);              // (every lane for itself: ~200 flops per camera, against ~6 observations per lane)
        // TWO observations per trip: both index loads, then both gathers, are in flight before the first block is rebuilt, and the two blocks' arithmetic is independent -- at
        // 1.7 waves per SIMD a wave is mostly alone and issues a DEPENDENT instruction every ~5 cycles (profiles/r04/issue_rates_pk_rate.txt).  Added into s in the list's order.
        // Measured in one process per box, GN per PCG iteration: 1 / 2 / 3 / 4 per trip = 33.2 / 31.9 / 32.5 / 32.9 us (BA_CAM_OBS to rebuild with another count).
#ifndef BA_CAM_OBS
#define BA_CAM_OBS 2
#endif
        constexpr int NO = BA_CAM_OBS;
        const int q1e = cam_ptr[c + 1];
        for (int q = cam_ptr[c] + lane; q < q1e; q += 64 * NO) {
            long pi[NO]; float x[NO][3], pp[NO][3]; bool h[NO];
#pragma unroll
            for (int u = 0; u < NO; ++u) { h[u] = q + 64 * u < q1e; pi[u] = q_pt[h[u] ? q + 64 * u : q]; }
#pragma unroll
            for (int u = 0; u < NO; ++u) { const float* X = pts + 3L * pi[u]; const float* P3 = p + PB + 3L * pi[u]; x[u][0] = X[0]; x[u][1] = X[1]; x[u][2] = X[2]; pp[u][0] = P3[0]; pp[u][1] = P3[1]; pp[u][2] = P3[2]; }
            Blk b[NO]; float j0[NO], j1[NO];
#pragma unroll
            for (int u = 0; u < NO; ++u) b[u] = ba_block(cp, x[u][0], x[u][1], x[u][2]);
#pragma unroll
            for (int u = 0; u < NO; ++u) {
                j0[u] = b[u].a[9] * pp[u][0] + b[u].a[10] * pp[u][1] + b[u].a[11] * pp[u][2]; j1[u] = b[u].a[21] * pp[u][0] + b[u].a[22] * pp[u][1] + b[u].a[23] * pp[u][2];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int u = 0; u < NO; ++u) { j0[u] += b[u].a[k] * pc[k]; j1[u] += b[u].a[12 + k] * pc[k]; }
            }
#pragma unroll
            for (int u = 0; u < NO; ++u) if (h[u]) JpC[q + 64 * u] = make_float2(j0[u], j1[u]);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
#pragma unroll
                for (int u = 0; u < NO; ++u) if (h[u]) s[k] += b[u].a[k] * j0[u] + b[u].a[12 + k] * j1[u];
            }
        }

#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] = wave_sum_all(s[k]);
        if (lane < 9) {
            float sv = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) if (lane == k) sv = s[k];
            const long i = 9L * c + lane;
            if (ctc) sv += e_c * e_p;                     // LM: (J^T J + CtC) p, PCGStep1_Finish (gauss_newton.t:774-787) folded in
            if (rst.r) { const float rv = e_b - sv; rst.r[i] = rv; acc += (e_m * rv) * rv; }
            else {
                Ap[i] = sv;
                acc += e_p * sv;
                if (s3_out) sm.add(e_m, e_r, sv);
                if (lm.b) sq.add(e_d, e_r, e_b, e_p, sv);
            }
        }
    }
    // one pass for the float partial and the three / six double sums (one barrier; the same additions in the same order as one block_store_* call per quantity group)
    const FinArgs nf{ thallo_sum_t{ nullptr, 0 }, nullptr, nullptr, nullptr, 0, 0 };
    if (lm.b) block_finish_sums_lm(acc, sm, sq, part_out, s3_out, nf, lm, red, redd);
    else if (s3_out) block_finish_sums(acc, sm, part_out, s3_out, nf, red, redd);
    else block_store_partial(acc, part_out, red);
}

__global__ __launch_bounds__(BLOCK) void k_pt2(int C_, int P_, const int* __restrict__ pt_ptr, const int* __restrict__ pt_pos, const float2* __restrict__ JP, const float2* __restrict__ JpC,
                                               const float* __restrict__ p, float* __restrict__ Ap, float* __restrict__ part_out,
                                               const float* __restrict__ rs, const float* __restrict__ prs, double* __restrict__ s3_out,
                                               const unsigned* __restrict__ gate, FinArgs fin, const float* __restrict__ ctc, const float* __restrict__ delta, LmFin lm, ResetArgs rst)
{   // part_out / s3_out: the slot arrays of the whole applyJTJ (k_cam2 filled slots [0, fin.blk_off)); this launch's workgroups use the slots behind them
    __shared__ float red[16];
    __shared__ double redd[6 * BLOCK / 64];
    if (gate != nullptr && __builtin_amdgcn_readfirstlane((int)gate[0]) != 0) return;
    float acc = 0.0f; Sums3 sm; SumsQ sq;
    const long PB = 9L * C_;
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < P_; j += gridDim.x * BLOCK) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        // the point's own words -- p, CtC, r, M^-1, delta, b -- are asked for BEFORE the observation loop: behind it they were one more round trip at the end of every thread
        // (the LM launch with its six sums: 16.8 us against 11.5 for the plain one, most of it this)
        const long i = PB + 3L * j;
        float pv[3], cv[3] = { 0.f, 0.f, 0.f }, rv[3] = { 0.f, 0.f, 0.f }, mv[3] = { 0.f, 0.f, 0.f }, dv[3] = { 0.f, 0.f, 0.f }, bv[3] = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            pv[u] = p[i + u];
            if (ctc) cv[u] = ctc[i + u];
            if (s3_out) { rv[u] = rs[i + u]; mv[u] = prs[i + u]; }
            if (lm.b) { dv[u] = delta[i + u]; bv[u] = lm.b[i + u]; }
            if (rst.r) { bv[u] = rst.b[i + u]; mv[u] = rst.pre[i + u]; }
        }
        // four observations per trip, all their loads in flight together (a point has ~4 observations: one round of latencies instead of four; the index load and
        // the gather behind it are the kernel's critical path); added up in the list's order, as the one-at-a-time loop did
        const int k1 = pt_ptr[j + 1];
        for (int k0 = pt_ptr[j]; k0 < k1; k0 += 4) {
            float2 a[4], b[4], c[4], jp[4]; int q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = min(k0 + u, k1 - 1); q[u] = pt_pos[k]; a[u] = JP[3L * k]; b[u] = JP[3L * k + 1]; c[u] = JP[3L * k + 2]; }     // (r0.d9, r0.d10), (r0.d11, r1.d9), (r1.d10, r1.d11)
#pragma unroll
            for (int u = 0; u < 4; ++u) jp[u] = JpC[q[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < k1) { s0 += a[u].x * jp[u].x + b[u].y * jp[u].y; s1 += a[u].y * jp[u].x + c[u].x * jp[u].y; s2 += b[u].x * jp[u].x + c[u].y * jp[u].y; }
        }
        if (ctc) { s0 += cv[0] * pv[0]; s1 += cv[1] * pv[1]; s2 += cv[2] * pv[2]; }
        if (rst.r) {
            const float r0 = bv[0] - s0, r1 = bv[1] - s1, r2 = bv[2] - s2;
            rst.r[i] = r0; rst.r[i + 1] = r1; rst.r[i + 2] = r2;
            acc += (mv[0] * r0) * r0 + (mv[1] * r1) * r1 + (mv[2] * r2) * r2;
            continue;
        }
        Ap[i] = s0; Ap[i + 1] = s1; Ap[i + 2] = s2;
        acc += pv[0] * s0 + pv[1] * s1 + pv[2] * s2;
        if (s3_out) { sm.add(mv[0], rv[0], s0); sm.add(mv[1], rv[1], s1); sm.add(mv[2], rv[2], s2); }
        if (lm.b) { sq.add(dv[0], rv[0], bv[0], pv[0], s0); sq.add(dv[1], rv[1], bv[1], pv[1], s1); sq.add(dv[2], rv[2], bv[2], pv[2], s2); }
    }
    if (lm.b) block_finish_sums_lm(acc, sm, sq, part_out, s3_out, fin, lm, red, redd);
    else if (s3_out) block_finish_sums(acc, sm, part_out, s3_out, fin, red, redd);
    else block_store_partial(acc, part_out + fin.blk_off, red);
}

#ifdef THALLO_RESEARCH
#include "probe/ba_resident_device.inc"      // the one-launch PCG loop (round 5: bit-identical, slower; research builds only)
#endif

inline void gather_shape(int C_, int P_, int& cam_blocks, int& grid)
{
    cam_blocks = (C_ + 3) / 4; if (cam_blocks > 448) cam_blocks = 448;
    int pb = (P_ + BLOCK - 1) / BLOCK; if (pb > 512) pb = 512; if (pb < 1) pb = 1;
    grid = cam_blocks + pb;                        // <= 960 partials
}

}  // namespace

extern "C" {

int thallo_hip_ba_cost(int C_, int P_, int O_, const float* cameras, const float* points, const float* observations,
                       const int* oToC, const int* oToP, float* cost_out, thallo_stream_t stream)
{
    (void)C_; (void)P_;
    int grid = (O_ + BLOCK - 1) / BLOCK; if (grid > THALLO_MAX_PARTIALS) grid = THALLO_MAX_PARTIALS; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, O_, cameras, points, (const float2*)observations, oToC, oToP, cost_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_ba_compute_j(int O_, const float* cameras, const float* points, const float* observations,
                            const int* cam_obs, const int* q_cam, const int* q_pt, float* Jb, float* F, thallo_stream_t stream)
{
    int grid = (O_ + BLOCK - 1) / BLOCK; if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_compute_j<false>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, O_, cameras, points, (const float2*)observations,
                       cam_obs, q_cam, q_pt, (float4*)Jb, (float2*)F);
    return check_launch();
}
/* the same blocks from forward-mode dual numbers over the residual's expression (rounds 1-3's J): the closed form's reference in the tests */
int thallo_hip_ba_compute_j_ad(int O_, const float* cameras, const float* points, const float* observations,
                               const int* cam_obs, const int* q_cam, const int* q_pt, float* Jb, float* F, thallo_stream_t stream)
{
    int grid = (O_ + BLOCK - 1) / BLOCK; if (grid > 2048) grid = 2048; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_compute_j<true>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, O_, cameras, points, (const float2*)observations,
                       cam_obs, q_cam, q_pt, (float4*)Jb, (float2*)F);
    return check_launch();
}

int thallo_hip_ba_pcg_init(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_ptr, const int* pt_pos, const int* q_cam,
                           const float* Jb, const float* F, float* r, float* pre, float* z, float* p_prev, float* delta,
                           float* diag_out, float* aN_out, thallo_stream_t stream)
{
    int cb, grid; gather_shape(C_, P_, cb, grid);
    hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, C_, P_, cb, cam_ptr, q_pt, pt_ptr, pt_pos, q_cam,
                       (const float4*)Jb, (const float2*)F, (const float*)nullptr, r, pre, z, p_prev, delta, diag_out, aN_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_ba_point_order(int O_, const int* pt_pos, int* q_ptk, thallo_stream_t stream)
{
    if (O_ < 1 || !pt_pos || !q_ptk) return -(int)hipErrorInvalidValue;
    int grid = (O_ + BLOCK - 1) / BLOCK; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_inverse_perm, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, O_, pt_pos, q_ptk);
    return check_launch();
}

int thallo_hip_ba_pack_point_blocks(int O_, const float* Jb, const int* q_ptk, float* JP, thallo_stream_t stream)
{
    if (O_ < 1 || !Jb || !q_ptk || !JP) return -(int)hipErrorInvalidValue;
    int grid = (O_ + BLOCK - 1) / BLOCK; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_pack_point_blocks, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, O_, (const float4*)Jb, q_ptk, (float2*)JP);
    return check_launch();
}

static int ba_apply2(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                     const float* cameras, const float* points, const float* JP, float* JpC, const float* p, float* Ap, float* aD_out,
                     const float* r, const float* pre, double* s3_out, const unsigned* gate, thallo_fin_t fin, const float* ctc, thallo_stream_t stream,
                     const float* delta = nullptr, LmFin lm = LmFin{ nullptr, nullptr, nullptr, 0, 0.0f }, ResetArgs rst = ResetArgs{ nullptr, nullptr, nullptr })
{
    if (rst.r && (!rst.b || !rst.pre || !ctc || s3_out || lm.b || fin.tickets)) return -(int)hipErrorInvalidValue;
    if (!cam_ptr || !q_pt || !pt_pos || !pt_ptr || !cameras || !points || !JP || !JpC || !p || !Ap || !aD_out) return -(int)hipErrorInvalidValue;
    if (s3_out && (!r || !pre)) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!s3_out || (gate && !lm.b) || !fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    if (lm.b && (!ctc || !delta || !lm.q3_out || !lm.state)) return -(int)hipErrorInvalidValue;      // (fin.tickets NULL: partials only -- the next thallo_hip_pcg_update_lm_fin finishes)
    int cb, grid; gather_shape(C_, P_, cb, grid);
    // the camera launch fills slots [0, cb); the point launch the rest, and (fin) its last workgroup adds up all `grid` of them
    hipLaunchKernelGGL(k_cam2, dim3(cb), dim3(BLOCK), 0, (hipStream_t)stream, C_, cam_ptr, q_pt, cameras, points, p, Ap, (float2*)JpC, aD_out, r, pre, s3_out, gate, ctc, delta, lm, rst);
    hipLaunchKernelGGL(k_pt2, dim3(grid - cb), dim3(BLOCK), 0, (hipStream_t)stream, C_, P_, pt_ptr, pt_pos, (const float2*)JP, (const float2*)JpC, p, Ap, aD_out, r, pre,
                       s3_out, gate, FinArgs{ fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, cb, grid }, ctc, delta, lm, rst);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_ba_apply2_camera_slots(int C_, int P_)
{   // how many of thallo_hip_ba_apply_jtj2*'s partial slots belong to the camera launch (the first ones)
    int cb, grid; gather_shape(C_, P_, cb, grid); return cb;
}
int thallo_hip_ba_apply_jtj2_fin(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                 const float* cameras, const float* points, const float* JP, float* JpC, const float* p, float* Ap, float* aD_out,
                                 const float* r, const float* pre, double* s3_out, const unsigned* gate, thallo_fin_t fin, thallo_stream_t stream)
{ return ba_apply2(C_, P_, cam_ptr, q_pt, pt_pos, pt_ptr, cameras, points, JP, JpC, p, Ap, aD_out, r, pre, s3_out, gate, fin, nullptr, stream); }
int thallo_hip_ba_apply_jtj2_lm(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                const float* cameras, const float* points, const float* JP, float* JpC, const float* p, const float* CtC, float* Ap, float* aD_out,
                                const unsigned* gate, thallo_stream_t stream)
{
    if (!CtC) return -(int)hipErrorInvalidValue;
    const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
    return ba_apply2(C_, P_, cam_ptr, q_pt, pt_pos, pt_ptr, cameras, points, JP, JpC, p, Ap, aD_out, nullptr, nullptr, nullptr, gate, none, CtC, stream);
}
int thallo_hip_ba_pcg_apply_lm(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                               const float* cameras, const float* points, const float* JP, float* JpC, const float* p, const float* CtC, float* Ap, float* aD_out,
                               const float* r, const float* pre, const float* delta, const float* b, double* s3_out, double* q3_out, thallo_fin_t fin,
                               float* lm_state, int k, float q_tolerance, int q_in, int q_out, thallo_stream_t stream)
{
    if (!CtC || !b || !delta || !lm_state || !q3_out || !s3_out || q_in < 0 || q_in > 7 || q_out < 0 || q_out > 7) return -(int)hipErrorInvalidValue;
    return ba_apply2(C_, P_, cam_ptr, q_pt, pt_pos, pt_ptr, cameras, points, JP, JpC, p, Ap, aD_out, r, pre, s3_out, reinterpret_cast<const unsigned*>(lm_state) + 1, fin, CtC, stream,
                     delta, LmFin{ b, q3_out, lm_state, k, q_tolerance, q_in, q_out });
}
int thallo_hip_ba_lm_reset_residual(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                                    const float* cameras, const float* points, const float* JP, float* JpC, const float* delta, const float* CtC, const float* b, const float* pre,
                                    float* r, float* betaN_out, const unsigned* gate, thallo_stream_t stream)
{
    if (!delta || !CtC || !b || !pre || !r || !betaN_out) return -(int)hipErrorInvalidValue;
    const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
    return ba_apply2(C_, P_, cam_ptr, q_pt, pt_pos, pt_ptr, cameras, points, JP, JpC, delta, r /* (not written in this mode) */, betaN_out, nullptr, nullptr, nullptr, gate, none, CtC, stream,
                     nullptr, LmFin{ nullptr, nullptr, nullptr, 0, 0.0f }, ResetArgs{ r, b, pre });
}
#ifdef THALLO_RESEARCH
#include "probe/ba_resident_host.inc"
#endif
int thallo_hip_ba_apply_jtj2(int C_, int P_, const int* cam_ptr, const int* q_pt, const int* pt_pos, const int* pt_ptr,
                             const float* cameras, const float* points, const float* JP, float* JpC, const float* p, float* Ap, float* aD_out,
                             const float* r, const float* pre, double* s3_out, const unsigned* gate, thallo_stream_t stream)
{
    const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
    return thallo_hip_ba_apply_jtj2_fin(C_, P_, cam_ptr, q_pt, pt_pos, pt_ptr, cameras, points, JP, JpC, p, Ap, aD_out, r, pre, s3_out, gate, none, stream);
}

}  // extern "C"
