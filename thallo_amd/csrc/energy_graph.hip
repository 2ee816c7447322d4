// energy_graph.hip -- graph-edge iteration domains: the graph-Laplacian known-answer energy
// (tests/minimal_graph/laplacian.t) and ARAP mesh deformation (examples/arap_mesh_deformation/
// arap_mesh_deformation.t:1-21).
//
// The reference runs edge domains residual-wise: one thread per directed edge computes Jp and scatters
// J^T(Jp) with float atomics (createapplyjtjResidualwise thallo.t:3536-3569, scatter lowering :3352-3403,
// kernels gauss_newton.t:998-1015), needs Ap_X cleared first and a separate PCGStep1_Finish for p.Ap.
// Here the same sums are GATHERED per vertex over its incident edges -- no atomics, no clear, the dot
// product fused, bitwise reproducible.  The incidence lists are built once per Init from V0/V1
// (the sparse maps are constant during a solve):
//   out CSR : edges sorted by source vertex; position e' in that order is the edge's id from now on;
//             out_v1[e'] = target vertex
//   in  CSR : for each vertex the ids e' of the edges that END there, and their source vertices
//
// ARAP, edge e = (n -> m), dv = O_n - O_m, R = ZYX Euler rotation of Angle_n (lib.t:123-137):
//   F_e = w_reg ((P_n - P_m) - R dv)                 in R^3
//   dF_e/dP_n = w I, dF_e/dP_m = -w I, dF_e/dA_n = -w G_e,  G_e = [dR/da dv | dR/db dv | dR/dg dv]
//   fit_n = [C_n.x >= -999999.9] w_fit (P_n - C_n)
// Per GN iteration a precompute kernel stores F_e (3 floats) and G_e (9 floats) per edge in out-CSR
// order (each vertex's edges contiguous); evalJTF / applyJTJ then need, per vertex,
//   own edges:      F/G rows (contiguous) + the neighbour's P-part of p (gather)
//   incoming edges: F_e or (G_e, p_A(src), p_P(src)) (gather)
// ~100k vertices / 600k edges = a 13 MB working set: L2/Infinity-Cache resident, latency-bound
// (SURVEY.md 8d), so the kernels are one-thread-per-vertex with everything else kept simple.
#include <cstring>
#include <algorithm>
#include "device_common.hpp"
#include "../../include/thallo_hip.h"

using namespace thallo;

namespace {

constexpr int BLOCK = 256;
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }
inline int vgrid(long n)
{
    long g = (n + BLOCK - 1) / BLOCK;
    if (g > THALLO_MAX_PARTIALS) g = THALLO_MAX_PARTIALS;
    if (g < 1) g = 1;
    return (int)g;
}

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 ld3(const float* p, long i) { f3 v; v.x = p[3 * i]; v.y = p[3 * i + 1]; v.z = p[3 * i + 2]; return v; }
__device__ __forceinline__ void st3(float* p, long i, f3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// ------------------------------------------------------------------------------------------ graph Laplacian (E6)
// fit_n = w (X_n - A_n) ; reg_e = X_v0 - X_v1
__global__ __launch_bounds__(BLOCK) void k_lapg_cost(int N, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                      const float* __restrict__ X, const float* __restrict__ A, float w, float* __restrict__ out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < N; n += gridDim.x * BLOCK) {
        const float x = X[n];
        const float f = w * (x - A[n]);
        float s = f * f;
        for (int k = out_ptr[n]; k < out_ptr[n + 1]; ++k) { const float d = x - X[out_v1[k]]; s += d * d; }
        acc += 0.5f * s;
    }
    block_store_partial(acc, out, red);
}

// (J^T J v)_n, or J^T F with v = X and the fit term supplied by the caller
__device__ __forceinline__ float lapg_apply(int n, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                            const int* __restrict__ in_ptr, const int* __restrict__ in_src, const float* __restrict__ v, float c)
{
    float s = 0.0f;
    for (int k = out_ptr[n]; k < out_ptr[n + 1]; ++k) s += c - v[out_v1[k]];
    for (int k = in_ptr[n]; k < in_ptr[n + 1]; ++k) s -= v[in_src[k]] - c;
    return s;
}

__global__ __launch_bounds__(BLOCK) void k_lapg_init(int N, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                      const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                      const float* __restrict__ X, const float* __restrict__ A, float w,
                                                      float* __restrict__ r, float* __restrict__ z, float* __restrict__ p_prev,
                                                      float* __restrict__ delta, float* __restrict__ diag_out, float* __restrict__ aN_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < N; n += gridDim.x * BLOCK) {
        const float x = X[n];
        if (diag_out) diag_out[n] = w * w + (float)(out_ptr[n + 1] - out_ptr[n]) + (float)(in_ptr[n + 1] - in_ptr[n]);
        const float res = -(w * (w * (x - A[n])) + lapg_apply(n, out_ptr, out_v1, in_ptr, in_src, X, x));
        r[n] = res; z[n] = res; p_prev[n] = 0.0f; delta[n] = 0.0f;      // identity preconditioner
        acc += res * res;
    }
    block_store_partial(acc, aN_out, red);
}

__global__ __launch_bounds__(BLOCK) void k_lapg_apply(int N, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                       const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                       float w, const float* __restrict__ p, float* __restrict__ Ap, float* __restrict__ aD_out)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < N; n += gridDim.x * BLOCK) {
        const float c = p[n];
        const float a = w * (w * c) + lapg_apply(n, out_ptr, out_v1, in_ptr, in_src, p, c);
        Ap[n] = a;
        acc += c * a;
    }
    block_store_partial(acc, aD_out, red);
}

// ------------------------------------------------------------------------------------------ ARAP (E2)
// Edge-data layout.  S == 0: out-CSR order, array of structs (edge k's 3 / 9 floats contiguous) -- what the slab / partition drivers
// use.  S > 0 ("ELL"): the j-th edge of vertex n lives at position j*N + n and component c at c*S + position, S = maxdeg*N: the
// threads of a wave (consecutive vertices) then read consecutive addresses in every edge array -- index lists, F, all 9 planes of G
// -- instead of addresses 13 floats apart (a wave instruction touches 2 cache lines instead of up to 64).
struct ELay { long S; int N; };
__device__ __forceinline__ long epos(const ELay& L, int base, int n, int j) { return L.S ? (long)j * L.N + n : (long)base + j; }
__device__ __forceinline__ f3 ldE(const float* __restrict__ F, const ELay& L, long pos)
{
    if (L.S) { f3 v; v.x = F[pos]; v.y = F[L.S + pos]; v.z = F[2 * L.S + pos]; return v; }
    return ld3(F, pos);
}
__device__ __forceinline__ void stE(float* __restrict__ F, const ELay& L, long pos, f3 v)
{
    if (L.S) { F[pos] = v.x; F[L.S + pos] = v.y; F[2 * L.S + pos] = v.z; } else st3(F, pos, v);
}
// column c (0..2) of the 3x3 block G_e
__device__ __forceinline__ f3 ldG(const float* __restrict__ G, const ELay& L, long pos, int c)
{
    if (L.S) { f3 v; v.x = G[(3 * c) * L.S + pos]; v.y = G[(3 * c + 1) * L.S + pos]; v.z = G[(3 * c + 2) * L.S + pos]; return v; }
    return ld3(G, 3 * pos + c);
}
__device__ __forceinline__ void stG(float* __restrict__ G, const ELay& L, long pos, int c, f3 v)
{
    if (L.S) { G[(3 * c) * L.S + pos] = v.x; G[(3 * c + 1) * L.S + pos] = v.y; G[(3 * c + 2) * L.S + pos] = v.z; } else st3(G, 3 * pos + c, v);
}

struct Rot { float R[9], dA[9], dB[9], dG[9]; };
__device__ __forceinline__ void rot3(f3 a, Rot& o)
{   // lib.t:123-137 and its three angle derivatives
    float sa, ca, sb, cb, sg, cg;
    sincosf(a.x, &sa, &ca); sincosf(a.y, &sb, &cb); sincosf(a.z, &sg, &cg);
    o.R[0] = cg * cb;  o.R[1] = -sg * ca + cg * sb * sa;  o.R[2] = sg * sa + cg * sb * ca;
    o.R[3] = sg * cb;  o.R[4] = cg * ca + sg * sb * sa;   o.R[5] = -cg * sa + sg * sb * ca;
    o.R[6] = -sb;      o.R[7] = cb * sa;                  o.R[8] = cb * ca;
    o.dA[0] = 0.0f;     o.dA[1] = sg * sa + cg * sb * ca;   o.dA[2] = sg * ca - cg * sb * sa;
    o.dA[3] = 0.0f;     o.dA[4] = -cg * sa + sg * sb * ca;  o.dA[5] = -cg * ca - sg * sb * sa;
    o.dA[6] = 0.0f;     o.dA[7] = cb * ca;                  o.dA[8] = -cb * sa;
    o.dB[0] = -cg * sb; o.dB[1] = cg * cb * sa;             o.dB[2] = cg * cb * ca;
    o.dB[3] = -sg * sb; o.dB[4] = sg * cb * sa;             o.dB[5] = sg * cb * ca;
    o.dB[6] = -cb;      o.dB[7] = -sb * sa;                 o.dB[8] = -sb * ca;
    o.dG[0] = -sg * cb; o.dG[1] = -cg * ca - sg * sb * sa;  o.dG[2] = cg * sa - sg * sb * ca;
    o.dG[3] = cg * cb;  o.dG[4] = -sg * ca + cg * sb * sa;  o.dG[5] = sg * sa + cg * sb * ca;
    o.dG[6] = 0.0f;     o.dG[7] = 0.0f;                     o.dG[8] = 0.0f;
}
__device__ __forceinline__ f3 mv(const float* M, f3 v)
{
    f3 o; o.x = M[0] * v.x + M[1] * v.y + M[2] * v.z; o.y = M[3] * v.x + M[4] * v.y + M[5] * v.z; o.z = M[6] * v.x + M[7] * v.y + M[8] * v.z;
    return o;
}

__global__ __launch_bounds__(BLOCK) void k_arap_cost(int N, int n0, int n1, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                      const float* __restrict__ P, const float* __restrict__ Ang, const float* __restrict__ O,
                                                      const float* __restrict__ Cn, float wf, float wr, float* __restrict__ out, ELay L)
{
    __shared__ float red[16];
    float acc = 0.0f;
    for (int n = n0 + blockIdx.x * BLOCK + threadIdx.x; n < n1; n += gridDim.x * BLOCK) {
        const f3 p = ld3(P, n), o = ld3(O, n), c = ld3(Cn, n);
        Rot rt; rot3(ld3(Ang, n), rt);
        float s = 0.0f;
        if (c.x >= -999999.9f) { const float fx = wf * (p.x - c.x), fy = wf * (p.y - c.y), fz = wf * (p.z - c.z); s += fx * fx + fy * fy + fz * fz; }
        const int ob = out_ptr[n], deg = out_ptr[n + 1] - ob;
        for (int j = 0; j < deg; ++j) {
            const int m = out_v1[epos(L, ob, n, j)];
            const f3 pm = ld3(P, m), om = ld3(O, m);
            f3 dv; dv.x = o.x - om.x; dv.y = o.y - om.y; dv.z = o.z - om.z;
            const f3 rv = mv(rt.R, dv);
            const float ex = wr * ((p.x - pm.x) - rv.x), ey = wr * ((p.y - pm.y) - rv.y), ez = wr * ((p.z - pm.z) - rv.z);
            s += ex * ex + ey * ey + ez * ez;
        }
        acc += 0.5f * s;
    }
    block_store_partial(acc, out, red);
}

// per GN iteration: F_e (3) and G_e (9, column-major: [dR/da dv | dR/db dv | dR/dg dv]) per edge, out-CSR order
__global__ __launch_bounds__(BLOCK) void k_arap_precompute(int N, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                            const float* __restrict__ P, const float* __restrict__ Ang, const float* __restrict__ O,
                                                            float wr, float* __restrict__ F, float* __restrict__ G, float* __restrict__ SC, ELay L)
{
    for (int n = blockIdx.x * BLOCK + threadIdx.x; n < N; n += gridDim.x * BLOCK) {
        const f3 p = ld3(P, n), o = ld3(O, n);
        const f3 ang = ld3(Ang, n);
        Rot rt; rot3(ang, rt);
        if (SC) {                                  // sines | cosines of the vertex's three angles: what k_arap_apply_rc rebuilds dR/da, dR/db, dR/dg from
            f3 sn, cs; sincosf(ang.x, &sn.x, &cs.x); sincosf(ang.y, &sn.y, &cs.y); sincosf(ang.z, &sn.z, &cs.z);
            st3(SC, n, sn); st3(SC, (long)N + n, cs);
        }
        const int ob = out_ptr[n], deg = out_ptr[n + 1] - ob;
        for (int j = 0; j < deg; ++j) {
            const long k = epos(L, ob, n, j);
            const int m = out_v1[k];
            const f3 pm = ld3(P, m), om = ld3(O, m);
            f3 dv; dv.x = o.x - om.x; dv.y = o.y - om.y; dv.z = o.z - om.z;
            const f3 rv = mv(rt.R, dv);
            f3 f; f.x = wr * ((p.x - pm.x) - rv.x); f.y = wr * ((p.y - pm.y) - rv.y); f.z = wr * ((p.z - pm.z) - rv.z);
            stE(F, L, k, f);
            stG(G, L, k, 0, mv(rt.dA, dv)); stG(G, L, k, 1, mv(rt.dB, dv)); stG(G, L, k, 2, mv(rt.dG, dv));
        }
    }
}

// PCGInit1 (+_Finish), gather form.  flat layout: [Position 3n+c | Angle 3N+3n+c]
__global__ __launch_bounds__(BLOCK) void k_arap_init(int N, int n0, int n1, const int* __restrict__ out_ptr, const int* __restrict__ in_ptr,
                                                      const int* __restrict__ in_edge, const float* __restrict__ P, const float* __restrict__ Cn,
                                                      const float* __restrict__ F, const float* __restrict__ G, float wf, float wr,
                                                      float* __restrict__ r, float* __restrict__ pre, float* __restrict__ z,
                                                      float* __restrict__ p_prev, float* __restrict__ delta,
                                                      float* __restrict__ diag_out, float* __restrict__ aN_out, ELay L)
{
    __shared__ float red[16];
    float acc = 0.0f;
    const float wr2 = wr * wr;
    for (int n = n0 + blockIdx.x * BLOCK + threadIdx.x; n < n1; n += gridDim.x * BLOCK) {
        f3 jp = { 0.f, 0.f, 0.f }, ja = { 0.f, 0.f, 0.f }, da = { 0.f, 0.f, 0.f };
        float dp = 0.0f;
        const int ob = out_ptr[n], deg = out_ptr[n + 1] - ob;
        for (int j = 0; j < deg; ++j) {                              // own edges: dF/dP_n = w I, dF/dA_n = -w G
            const long k = epos(L, ob, n, j);
            const f3 f = ldE(F, L, k);
            const f3 g0 = ldG(G, L, k, 0), g1 = ldG(G, L, k, 1), g2 = ldG(G, L, k, 2);
            jp.x += wr * f.x; jp.y += wr * f.y; jp.z += wr * f.z;
            ja.x -= wr * (g0.x * f.x + g0.y * f.y + g0.z * f.z);
            ja.y -= wr * (g1.x * f.x + g1.y * f.y + g1.z * f.z);
            ja.z -= wr * (g2.x * f.x + g2.y * f.y + g2.z * f.z);
            dp += wr2;
            da.x += wr2 * (g0.x * g0.x + g0.y * g0.y + g0.z * g0.z);
            da.y += wr2 * (g1.x * g1.x + g1.y * g1.y + g1.z * g1.z);
            da.z += wr2 * (g2.x * g2.x + g2.y * g2.y + g2.z * g2.z);
        }
        const int ib = in_ptr[n], ideg = in_ptr[n + 1] - ib;
        for (int j = 0; j < ideg; ++j) {                             // incoming edges: dF/dP_n = -w I
            const f3 f = ldE(F, L, in_edge[epos(L, ib, n, j)]);
            jp.x -= wr * f.x; jp.y -= wr * f.y; jp.z -= wr * f.z;
            dp += wr2;
        }
        const f3 c = ld3(Cn, n);
        if (c.x >= -999999.9f) {
            const f3 p = ld3(P, n);
            jp.x += wf * (wf * (p.x - c.x)); jp.y += wf * (wf * (p.y - c.y)); jp.z += wf * (wf * (p.z - c.z));
            dp += wf * wf;
        }
        f3 rp = { -jp.x, -jp.y, -jp.z }, ra = { -ja.x, -ja.y, -ja.z };
        const float mp = guarded_invert(dp);
        f3 mpv = { mp, mp, mp }, mav = { guarded_invert(da.x), guarded_invert(da.y), guarded_invert(da.z) };
        f3 zp = { mp * rp.x, mp * rp.y, mp * rp.z }, za = { mav.x * ra.x, mav.y * ra.y, mav.z * ra.z };
        const f3 zero = { 0.f, 0.f, 0.f };
        st3(r, n, rp); st3(r, (long)N + n, ra);
        st3(pre, n, mpv); st3(pre, (long)N + n, mav);
        if (diag_out) { const f3 dpv = { dp, dp, dp }; st3(diag_out, n, dpv); st3(diag_out, (long)N + n, da); }
        st3(z, n, zp); st3(z, (long)N + n, za);
        st3(p_prev, n, zero); st3(p_prev, (long)N + n, zero);
        st3(delta, n, zero); st3(delta, (long)N + n, zero);
        acc += rp.x * zp.x + rp.y * zp.y + rp.z * zp.z + ra.x * za.x + ra.y * za.y + ra.z * za.z;
    }
    block_store_partial(acc, aN_out, red);
}

// PCGStep1: Ap = J^T J p (gather), alphaD partials
__global__ __launch_bounds__(BLOCK) void k_arap_apply(int N, int n0, int n1, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                       const int* __restrict__ in_ptr, const int* __restrict__ in_edge, const int* __restrict__ in_src,
                                                       const float* __restrict__ Cn, const float* __restrict__ G, float wf, float wr,
                                                       const float* __restrict__ p, float* __restrict__ Ap, float* __restrict__ aD_out, ELay L,
                                                       const float* __restrict__ rs, const float* __restrict__ pre, double* __restrict__ s3_out, FinArgs fin)
{   // rs / pre / s3_out (all or none): also the Sums3 of the single-reduction PCG form over the unknowns of [n0,n1)
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    float acc = 0.0f; Sums3 sm;
    const float wr2 = wr * wr;
    for (int n = n0 + blockIdx.x * BLOCK + threadIdx.x; n < n1; n += gridDim.x * BLOCK) {
        const f3 pp = ld3(p, n), pa = ld3(p, (long)N + n);
        f3 ap = { 0.f, 0.f, 0.f }, aa = { 0.f, 0.f, 0.f };
        const int ob = out_ptr[n], deg = out_ptr[n + 1] - ob;
#pragma unroll 4
        for (int j = 0; j < deg; ++j) {
            const long k = epos(L, ob, n, j);
            const f3 pm = ld3(p, out_v1[k]);
            const f3 g0 = ldG(G, L, k, 0), g1 = ldG(G, L, k, 1), g2 = ldG(G, L, k, 2);
            // Jp / w = (pP_n - pP_m) - G pA_n
            const float jx = (pp.x - pm.x) - (g0.x * pa.x + g1.x * pa.y + g2.x * pa.z);
            const float jy = (pp.y - pm.y) - (g0.y * pa.x + g1.y * pa.y + g2.y * pa.z);
            const float jz = (pp.z - pm.z) - (g0.z * pa.x + g1.z * pa.y + g2.z * pa.z);
            ap.x += jx; ap.y += jy; ap.z += jz;
            aa.x -= g0.x * jx + g0.y * jy + g0.z * jz;
            aa.y -= g1.x * jx + g1.y * jy + g1.z * jz;
            aa.z -= g2.x * jx + g2.y * jy + g2.z * jz;
        }
        const int ib = in_ptr[n], ideg = in_ptr[n + 1] - ib;
#pragma unroll 4
        for (int j = 0; j < ideg; ++j) {                             // edge (m -> n): contributes -w * Jp to P_n
            const long k = epos(L, ib, n, j);
            const int e = in_edge[k], m = in_src[k];
            const f3 pm = ld3(p, m), am = ld3(p, (long)N + m);
            const f3 g0 = ldG(G, L, e, 0), g1 = ldG(G, L, e, 1), g2 = ldG(G, L, e, 2);
            ap.x -= (pm.x - pp.x) - (g0.x * am.x + g1.x * am.y + g2.x * am.z);
            ap.y -= (pm.y - pp.y) - (g0.y * am.x + g1.y * am.y + g2.y * am.z);
            ap.z -= (pm.z - pp.z) - (g0.z * am.x + g1.z * am.y + g2.z * am.z);
        }
        ap.x *= wr2; ap.y *= wr2; ap.z *= wr2; aa.x *= wr2; aa.y *= wr2; aa.z *= wr2;
        if (Cn[3 * n] >= -999999.9f) { ap.x += wf * wf * pp.x; ap.y += wf * wf * pp.y; ap.z += wf * wf * pp.z; }
        st3(Ap, n, ap); st3(Ap, (long)N + n, aa);
        acc += pp.x * ap.x + pp.y * ap.y + pp.z * ap.z + pa.x * aa.x + pa.y * aa.y + pa.z * aa.z;
        if (s3_out) {
            const f3 rp = ld3(rs, n), ra = ld3(rs, (long)N + n), mp = ld3(pre, n), ma = ld3(pre, (long)N + n);
            sm.add(mp.x, rp.x, ap.x); sm.add(mp.y, rp.y, ap.y); sm.add(mp.z, rp.z, ap.z);
            sm.add(ma.x, ra.x, aa.x); sm.add(ma.y, ra.y, aa.y); sm.add(ma.z, ra.z, aa.z);
        }
    }
    if (s3_out) block_finish_sums(acc, sm, aD_out, s3_out, fin, red, redd);
    else block_store_partial(acc, aD_out, red);
}

// The same J^T J p for the ELL layout with at most MD edge slots per vertex (S = maxdeg * N, maxdeg <= MD): both edge loops fully unrolled with
// the slots beyond a vertex's degree predicated off, so that every index load of a vertex is issued first, then every gather that depends on
// them, then the arithmetic -- two dependent memory round trips per vertex instead of one pair per group of four edges (the kernel is latency-bound:
// 13 MB working set, 1.6 waves per SIMD).  Same terms in the same order: bitwise the loop form's output.
template <int MD>
__global__ __launch_bounds__(BLOCK) void k_arap_apply_ell(int N, int n0, int n1, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                           const int* __restrict__ in_ptr, const int* __restrict__ in_edge, const int* __restrict__ in_src,
                                                           const float* __restrict__ Cn, const float* __restrict__ G, float wf, float wr,
                                                           const float* __restrict__ p, float* __restrict__ Ap, float* __restrict__ aD_out, ELay L,
                                                           const float* __restrict__ rs, const float* __restrict__ pre, double* __restrict__ s3_out, FinArgs fin)
{
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    float acc = 0.0f; Sums3 sm;
    const float wr2 = wr * wr;
    for (int n = n0 + blockIdx.x * BLOCK + threadIdx.x; n < n1; n += gridDim.x * BLOCK) {
        const int deg = out_ptr[n + 1] - out_ptr[n], ideg = in_ptr[n + 1] - in_ptr[n];
        int vo[MD], ei[MD], vi[MD];
#pragma unroll
        for (int j = 0; j < MD; ++j) {       // (slots beyond the degree exist -- the planes hold maxdeg * N entries -- but their contents are not used)
            const long k = (long)j * L.N + n;
            const bool a = j < deg, b = j < ideg;
            vo[j] = a ? out_v1[k] : n; ei[j] = b ? in_edge[k] : n; vi[j] = b ? in_src[k] : n;
        }
        const f3 pp = ld3(p, n), pa = ld3(p, (long)N + n);
        f3 pmo[MD], pmi[MD], ami[MD];
#pragma unroll
        for (int j = 0; j < MD; ++j) { pmo[j] = ld3(p, vo[j]); pmi[j] = ld3(p, vi[j]); ami[j] = ld3(p, (long)N + vi[j]); }
        f3 ap = { 0.f, 0.f, 0.f }, aa = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const long k = (long)j * L.N + n;
            const f3 pm = pmo[j];
            const f3 g0 = ldG(G, L, k, 0), g1 = ldG(G, L, k, 1), g2 = ldG(G, L, k, 2);
            const float jx = (pp.x - pm.x) - (g0.x * pa.x + g1.x * pa.y + g2.x * pa.z);
            const float jy = (pp.y - pm.y) - (g0.y * pa.x + g1.y * pa.y + g2.y * pa.z);
            const float jz = (pp.z - pm.z) - (g0.z * pa.x + g1.z * pa.y + g2.z * pa.z);
            if (j < deg) {
                ap.x += jx; ap.y += jy; ap.z += jz;
                aa.x -= g0.x * jx + g0.y * jy + g0.z * jz;
                aa.y -= g1.x * jx + g1.y * jy + g1.z * jz;
                aa.z -= g2.x * jx + g2.y * jy + g2.z * jz;
            }
        }
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const f3 pm = pmi[j], am = ami[j];
            const f3 g0 = ldG(G, L, ei[j], 0), g1 = ldG(G, L, ei[j], 1), g2 = ldG(G, L, ei[j], 2);
            if (j < ideg) {
                ap.x -= (pm.x - pp.x) - (g0.x * am.x + g1.x * am.y + g2.x * am.z);
                ap.y -= (pm.y - pp.y) - (g0.y * am.x + g1.y * am.y + g2.y * am.z);
                ap.z -= (pm.z - pp.z) - (g0.z * am.x + g1.z * am.y + g2.z * am.z);
            }
        }
        ap.x *= wr2; ap.y *= wr2; ap.z *= wr2; aa.x *= wr2; aa.y *= wr2; aa.z *= wr2;
        if (Cn[3 * n] >= -999999.9f) { ap.x += wf * wf * pp.x; ap.y += wf * wf * pp.y; ap.z += wf * wf * pp.z; }
        st3(Ap, n, ap); st3(Ap, (long)N + n, aa);
        acc += pp.x * ap.x + pp.y * ap.y + pp.z * ap.z + pa.x * aa.x + pa.y * aa.y + pa.z * aa.z;
        if (s3_out) {
            const f3 rp = ld3(rs, n), ra = ld3(rs, (long)N + n), mp = ld3(pre, n), ma = ld3(pre, (long)N + n);
            sm.add(mp.x, rp.x, ap.x); sm.add(mp.y, rp.y, ap.y); sm.add(mp.z, rp.z, ap.z);
            sm.add(ma.x, ra.x, aa.x); sm.add(ma.y, ra.y, aa.y); sm.add(ma.z, ra.z, aa.z);
        }
    }
    if (s3_out) block_finish_sums(acc, sm, aD_out, s3_out, fin, red, redd);
    else block_store_partial(acc, aD_out, red);
}

// J^T J p with G_e RECOMPUTED instead of read (round 3; VERDICT r2 item 6: the per-edge G plane is 36 B/edge, read once for a vertex's own edges and once more,
// gathered, for its incoming ones -- 43 MB of the kernel's 64 MB per launch at 102,400 vertices -- while 13.5 MB are algorithmic).  G_e = [dR/da dv | dR/db dv | dR/dg dv]
// depends on the SOURCE vertex's angles and dv = O_src - O_dst only: per vertex the precompute stores the six sines / cosines (SC, 2.4 MB, L2-resident like p and O),
// and an edge costs a 12-byte gather of O (incoming: + 24 bytes of SC) and ~70 flops instead of 36 streamed bytes.  ELL layout, at most MD edge slots, same gather
// structure as k_arap_apply_ell (two dependent round trips per vertex); in_edge is not needed.  Same formulas as rot3 / k_arap_precompute.
struct DRot { float a1, a2, a4, a5, a7, a8, b0, b1, b2, b3, b4, b5, b6, b7, b8, g0, g1, g2, g3, g4, g5; };      // the nonzero entries of dR/da, dR/db, dR/dg (row-major)
__device__ __forceinline__ DRot drot(f3 s, f3 c)
{   // s = (sin a, sin b, sin g), c = cosines; lib.t:123-137 differentiated, as in rot3
    const float sa = s.x, sb = s.y, sg = s.z, ca = c.x, cb = c.y, cg = c.z;
    DRot o;
    o.a1 = sg * sa + cg * sb * ca;   o.a2 = sg * ca - cg * sb * sa;
    o.a4 = -cg * sa + sg * sb * ca;  o.a5 = -cg * ca - sg * sb * sa;
    o.a7 = cb * ca;                  o.a8 = -cb * sa;
    o.b0 = -cg * sb; o.b1 = cg * cb * sa;  o.b2 = cg * cb * ca;
    o.b3 = -sg * sb; o.b4 = sg * cb * sa;  o.b5 = sg * cb * ca;
    o.b6 = -cb;      o.b7 = -sb * sa;      o.b8 = -sb * ca;
    o.g0 = -sg * cb; o.g1 = -cg * ca - sg * sb * sa;  o.g2 = cg * sa - sg * sb * ca;
    o.g3 = cg * cb;  o.g4 = -sg * ca + cg * sb * sa;  o.g5 = sg * sa + cg * sb * ca;
    return o;
}
__device__ __forceinline__ void gcols(const DRot& d, f3 dv, f3& g0, f3& g1, f3& g2)
{
    g0.x = d.a1 * dv.y + d.a2 * dv.z;               g0.y = d.a4 * dv.y + d.a5 * dv.z;               g0.z = d.a7 * dv.y + d.a8 * dv.z;
    g1.x = d.b0 * dv.x + d.b1 * dv.y + d.b2 * dv.z; g1.y = d.b3 * dv.x + d.b4 * dv.y + d.b5 * dv.z; g1.z = d.b6 * dv.x + d.b7 * dv.y + d.b8 * dv.z;
    g2.x = d.g0 * dv.x + d.g1 * dv.y + d.g2 * dv.z; g2.y = d.g3 * dv.x + d.g4 * dv.y + d.g5 * dv.z; g2.z = 0.0f;
}
template <int MD>
__global__ __launch_bounds__(BLOCK) void k_arap_apply_rc(int N, int n0, int n1, const int* __restrict__ out_ptr, const int* __restrict__ out_v1,
                                                          const int* __restrict__ in_ptr, const int* __restrict__ in_src,
                                                          const float* __restrict__ Cn, const float* __restrict__ O, const float* __restrict__ SC, float wf, float wr,
                                                          const float* __restrict__ p, float* __restrict__ Ap, float* __restrict__ aD_out, ELay L,
                                                          const float* __restrict__ rs, const float* __restrict__ pre, double* __restrict__ s3_out, FinArgs fin)
{
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    float acc = 0.0f; Sums3 sm;
    const float wr2 = wr * wr;
    for (int n = n0 + blockIdx.x * BLOCK + threadIdx.x; n < n1; n += gridDim.x * BLOCK) {
        const int deg = out_ptr[n + 1] - out_ptr[n], ideg = in_ptr[n + 1] - in_ptr[n];
        int vo[MD], vi[MD];
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const long k = (long)j * L.N + n;
            vo[j] = j < deg ? out_v1[k] : n; vi[j] = j < ideg ? in_src[k] : n;
        }
        const f3 pp = ld3(p, n), pa = ld3(p, (long)N + n), on = ld3(O, n);
        const DRot dn = drot(ld3(SC, n), ld3(SC, (long)N + n));
        f3 pmo[MD], omo[MD], pmi[MD], ami[MD], omi[MD], smi[MD], cmi[MD];
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            pmo[j] = ld3(p, vo[j]); omo[j] = ld3(O, vo[j]);
            pmi[j] = ld3(p, vi[j]); ami[j] = ld3(p, (long)N + vi[j]); omi[j] = ld3(O, vi[j]); smi[j] = ld3(SC, vi[j]); cmi[j] = ld3(SC, (long)N + vi[j]);
        }
        f3 ap = { 0.f, 0.f, 0.f }, aa = { 0.f, 0.f, 0.f };
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const f3 pm = pmo[j];
            f3 dv; dv.x = on.x - omo[j].x; dv.y = on.y - omo[j].y; dv.z = on.z - omo[j].z;
            f3 g0, g1, g2; gcols(dn, dv, g0, g1, g2);
            const float jx = (pp.x - pm.x) - (g0.x * pa.x + g1.x * pa.y + g2.x * pa.z);
            const float jy = (pp.y - pm.y) - (g0.y * pa.x + g1.y * pa.y + g2.y * pa.z);
            const float jz = (pp.z - pm.z) - (g0.z * pa.x + g1.z * pa.y + g2.z * pa.z);
            if (j < deg) {
                ap.x += jx; ap.y += jy; ap.z += jz;
                aa.x -= g0.x * jx + g0.y * jy + g0.z * jz;
                aa.y -= g1.x * jx + g1.y * jy + g1.z * jz;
                aa.z -= g2.x * jx + g2.y * jy + g2.z * jz;
            }
        }
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const f3 pm = pmi[j], am = ami[j];
            const DRot dm = drot(smi[j], cmi[j]);
            f3 dv; dv.x = omi[j].x - on.x; dv.y = omi[j].y - on.y; dv.z = omi[j].z - on.z;
            f3 g0, g1, g2; gcols(dm, dv, g0, g1, g2);
            if (j < ideg) {
                ap.x -= (pm.x - pp.x) - (g0.x * am.x + g1.x * am.y + g2.x * am.z);
                ap.y -= (pm.y - pp.y) - (g0.y * am.x + g1.y * am.y + g2.y * am.z);
                ap.z -= (pm.z - pp.z) - (g0.z * am.x + g1.z * am.y + g2.z * am.z);
            }
        }
        ap.x *= wr2; ap.y *= wr2; ap.z *= wr2; aa.x *= wr2; aa.y *= wr2; aa.z *= wr2;
        if (Cn[3 * n] >= -999999.9f) { ap.x += wf * wf * pp.x; ap.y += wf * wf * pp.y; ap.z += wf * wf * pp.z; }
        st3(Ap, n, ap); st3(Ap, (long)N + n, aa);
        acc += pp.x * ap.x + pp.y * ap.y + pp.z * ap.z + pa.x * aa.x + pa.y * aa.y + pa.z * aa.z;
        if (s3_out) {
            const f3 rp = ld3(rs, n), ra = ld3(rs, (long)N + n), mp = ld3(pre, n), ma = ld3(pre, (long)N + n);
            sm.add(mp.x, rp.x, ap.x); sm.add(mp.y, rp.y, ap.y); sm.add(mp.z, rp.z, ap.z);
            sm.add(ma.x, ra.x, aa.x); sm.add(ma.y, ra.y, aa.y); sm.add(ma.z, ra.z, aa.z);
        }
    }
    if (s3_out) block_finish_sums(acc, sm, aD_out, s3_out, fin, red, redd);
    else block_store_partial(acc, aD_out, red);
}

// ------------------------------------------------------------------------------------------ the whole PCG loop of a Gauss-Newton step in ONE launch (round 4)
// ARAP at 100k vertices is a 10 MB state walked by two latency-bound launches per PCG iteration (the flat vector update 7.6 us + the gather 13 us + their
// boundaries = 19.5 us).  Here one thread keeps its vertex's r, p, A p, M^-1, delta -- and the G matrices of its edges, which a Gauss-Newton step does not change --
// in registers for all L iterations (the idea of energy_image_warping_resident.hip on an index-list domain), and an iteration has ONE hand-over
// between workgroups.  What this chip makes of a hand-over: an agent-scope access goes past the XCD's L2 to the fabric, ~1.5-2 us each way, so "store, then the
// reader's poll sees it" is ~5 us however little is handed over (phase stamps, tools/arap_resident_probe.py), and 400 workgroups polling 400 records each is
// volume-bound on top (10 us with one 16-byte request per lane and record part).  The first two versions of this kernel had two hand-overs per iteration (p_k to the
// neighbours, the sums to everybody) and were no faster than two launches (24.9 and 19.3 us per iteration).  Hence:
//   * same vertex -> (workgroup, thread) map as the launch-per-iteration kernels (vertex n = 256 b + t), so a workgroup's partial sums are the same numbers;
//     every workgroup must be RESIDENT (they wait for each other): the host checks what the device can hold (2 workgroups per CU);
//   * a workgroup keeps p_k of its own vertices AND of the vertices they share an edge with ("ghosts": a host-built ascending list, at most ARAP_RES_GHOSTS
//     vertices, else the plan runs one launch per iteration) in LDS, and r, M^-1 of the ghosts too: it updates the ghosts' r and p itself, with the owner's
//     expressions on the owner's inputs (bit-identical), and all it needs from the owner is A p_k at the ghost -- which goes out TOGETHER with the workgroup's
//     sums: A p_k as {value | tag} granules (write-through sc1 stores: the data is the flag, nothing to drain), the {alphaD | N, S1, S2} record as four tagged
//     16-byte parts in part-major order (a wave's load of one part of 64 records is 1 KB of whole lines);
//   * the sums go up a tree shaped like load_iteration_sums' order (lane l adds partials l, l + 64, ... ascending, then the butterflies): with more than 64 workgroups,
//     workgroup l < 64 adds the records l, l + 64, ... into "lane record" l; the <= 64 lane records (up to 64 workgroups: the records themselves) are read and put
//     through the butterflies by the last workgroup, which publishes the four totals for everybody else to poll (up to 192 workgroups), or by every workgroup for
//     itself (beyond) -- bit-identical alpha_k / beta_k everywhere.  (Every workgroup sweeping every record is W^2 fabric requests per poll round: 11 us of an
//     iteration at 400 workgroups, 1 us at 25);
//   * both are double-buffered by the iteration's parity: a workgroup publishes iteration k + 2 only after it has every record of k + 1, which a reader writes
//     after it is through with k.
// Every wait is bounded (2 s; an error word, later waits fall through, the host reports it at the next cost evaluation).  Bit-identical to PCGUpdate + applyJTJ per
// iteration (Plan::step_gn_expanded): same expressions (this file is built with -ffp-contract=on; the flat update's fmas are spelled out), same summation order.
// Replaces the loop of gauss_newton.t:1615-1687 for this plugin.
constexpr int ARAP_RES_GHOSTS = 768;                      // vertices of other workgroups a workgroup may share edges with
constexpr int ARAP_RES_SPAN = ARAP_RES_GHOSTS + BLOCK;    // staged vertices: LDS 2 parts x 3 floats each
constexpr int ARAP_RES_COPIES = 8;                        // copies of the iteration's totals (a reader takes copy workgroup % 8: its XCD's, with round-robin placement)
constexpr int ARAP_RES_ROOT = 192;                        // up to this many workgroups ONE workgroup reads the lane records and publishes the totals
constexpr int ARAP_RES_LISTW = 4 + ARAP_RES_GHOSTS;       // ints per workgroup in the exchange memory: {count, -, -, -, ghost vertices in ascending order}
enum { ARES_SEQ = 0, ARES_ERR = 1, ARES_SPIN_MS = 2, ARES_PM = 4, ARES_CTL_WORDS = 16 };
typedef unsigned long long u64r;
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));
typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_r;
__device__ __forceinline__ rsrc_r ares_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000); }
struct ArapResArgs {
    int N, nwg, L; long ell;
    const int *out_ptr, *out_v1, *in_ptr, *in_src;
    const float *Cn, *O, *SC; float wf, wr;
    float *r, *Ap; const float* pre; float *p0, *p1, *delta;      // p_{L-1} ends in p[L & 1] like behind L launches of the flat update (p0 = the plan's p[0]: zeros at the start)
    thallo_sum_t aN0; float* words;
    u64r* rec;            // [2 parity][4 parts][nwg] x 16 bytes
    u64r* lrec;           // [2 parity][ARAP_RES_COPIES][4 parts][64] x 16 bytes: the lane records; behind them [2 parity][ARAP_RES_COPIES][4 parts] x 16 bytes: the totals (copies spread the readers)
    u64r* ag;             // [2 parity][2 parts: Position, Angle][N][3] granules of A p_k
    float* ovf;           // overflow edges (slots >= 6 of the ELL lists): [(out: j - 6 | in: maxo - 6 + j - 6)][N] x 9 words {staged slot, G8}, written by the launch itself
    int maxo, maxi;       // edge slots of the out / in ELL lists
    const int* wlist;     // [nwg][ARAP_RES_LISTW]: the vertices of other workgroups that the workgroup's vertices share an edge with, ascending
    unsigned* ctl;
    unsigned* stamps;     // research build: [4 points][512 workgroups]
};
struct ASpin { unsigned n; long long t0; };
__device__ __forceinline__ bool ares_spin_fail(ASpin& sp, unsigned* ctl, unsigned what, unsigned idx, unsigned tag)
{   // bounded wait bookkeeping (as energy_image_warping_resident.hip: spin_fail): true = give up
    __builtin_amdgcn_s_sleep(1);
    if (((++sp.n) & 63u) != 0u) return false;
    if (__hip_atomic_load(ctl + ARES_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    const long long now = wall_clock64();
    if (sp.t0 == 0) { sp.t0 = now; return false; }
    const unsigned ms = __hip_atomic_load(ctl + ARES_SPIN_MS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const long long bound = ms ? (long long)ms * 100000LL : 2LL * 100000000LL;        // default: 2 s of the 100 MHz wall clock
    if (now - sp.t0 <= bound) return false;
    if (__hip_atomic_exchange(ctl + ARES_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        unsigned* pm = ctl + ARES_PM; pm[0] = what; pm[1] = blockIdx.x; pm[2] = threadIdx.x; pm[3] = idx; pm[4] = tag;
    }
    return true;
}
__global__ void k_arap_res_begin(unsigned* ctl, unsigned L) { if (threadIdx.x == 0) ctl[ARES_SEQ] += L + 2u; }       // tags of this launch: seq + 1 .. seq + L (never 0, never an earlier launch's)

#ifdef ARAP_STAMPS            // research builds only (tools/arap_resident_probe.py): phase stamps of iteration 10, workgroups 0 and 200, into the control words
#define ASTAMP(i) do { if (k == 10 && tid == 0 && (wg == 0 || wg == 200)) a.ctl[16 + (wg ? 12 : 0) + (i)] = (unsigned)wall_clock64(); \
                       if (k == 10 && tid == 0 && ((i) == 0 || (i) == 2 || (i) == 3 || (i) == 6)) a.stamps[((i) == 0 ? 0 : (i) == 2 ? 1 : (i) == 3 ? 2 : 3) * 512 + wg] = (unsigned)wall_clock64(); } while (0)
static void* g_arap_dbg_xbuf = nullptr;
extern "C" void* thallo_hip_arap_debug_last_xbuf(void) { return g_arap_dbg_xbuf; }
#else
#define ASTAMP(i) do { } while (0)
#endif
struct G8 { float a, b, c, d, e, f, g, h; };          // gcols' three columns without the constant zero: g0 = (a, b, c), g1 = (d, e, f), g2 = (g, h, 0)
template <int MD, bool OVF>
__global__ __launch_bounds__(BLOCK, 2) void k_arap_resident(ArapResArgs a)
{
    constexpr int GH = ARAP_RES_GHOSTS / BLOCK;                       // ghosts a thread looks after
    __shared__ float4 lp[2 * ARAP_RES_SPAN];                          // [2 parts][span]: p_k of the staged vertices (slot order; one 16-byte read per neighbour)
    __shared__ float gr[6 * ARAP_RES_GHOSTS], gm[6 * ARAP_RES_GHOSTS]; // [6 components][ghost]: r_k and M^-1 of the ghosts
    __shared__ int s_gl[ARAP_RES_GHOSTS];                             // ghost -> vertex (ascending)
    __shared__ float red[16];
    __shared__ double redd[3 * BLOCK / 64];
    __shared__ float s_tot_f; __shared__ double s_tot[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int N = a.N, n = wg * BLOCK + tid;
    const bool live = n < N;
    const int nc = live ? n : 0;                                     // (threads past the last vertex run on vertex 0's data and contribute nothing)
    const float wr2 = a.wr * a.wr;
    const rsrc_r RREC = ares_rsrc(a.rec), RLREC = ares_rsrc(a.lrec), RAG = ares_rsrc(a.ag);
    const int nlane = min(64, a.nwg);                                // lane records in use
    // staged slots: my workgroup's vertices first (slot = thread), the ghosts behind them in the list's order
    const int nown = min(BLOCK, N - wg * BLOCK), own_s = 0;
    const int nghost = min(a.wlist[(long)ARAP_RES_LISTW * wg], ARAP_RES_GHOSTS);
    for (int g = tid; g < nghost; g += BLOCK) s_gl[g] = a.wlist[(long)ARAP_RES_LISTW * wg + 4 + g];
    __syncthreads();
    auto slot_of = [&](int v) {
        if (v / BLOCK == wg) return v - wg * BLOCK;
        int lo = 0, hi = nghost;                                      // (first entry >= v: the list holds every neighbour of my vertices)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_gl[mid] < v) lo = mid + 1; else hi = mid; }
        return nown + min(lo, nghost > 0 ? nghost - 1 : 0);
    };
    // the vertex's constants: where its neighbours are staged, the G matrices of its edges (k_arap_apply_rc's expressions, evaluated once), the fit flag
    const int deg = live ? a.out_ptr[nc + 1] - a.out_ptr[nc] : 0, ideg = live ? a.in_ptr[nc + 1] - a.in_ptr[nc] : 0;
    const f3 on = ld3(a.O, nc), sn = ld3(a.SC, nc), cn = ld3(a.SC, (long)N + nc);
    const DRot dn = drot(sn, cn);
    int qo[MD], qi[MD]; G8 go[MD], gi[MD];
#pragma unroll
    for (int j = 0; j < MD; ++j) {
        const long k = (long)j * N + nc;
        const int vo = j < deg ? a.out_v1[k] : nc, vi = j < ideg ? a.in_src[k] : nc;
        qo[j] = slot_of(vo); qi[j] = slot_of(vi);
        {
            const f3 om = ld3(a.O, vo);
            f3 dv; dv.x = on.x - om.x; dv.y = on.y - om.y; dv.z = on.z - om.z;
            f3 g0, g1, g2; gcols(dn, dv, g0, g1, g2);
            go[j].a = g0.x; go[j].b = g0.y; go[j].c = g0.z; go[j].d = g1.x; go[j].e = g1.y; go[j].f = g1.z; go[j].g = g2.x; go[j].h = g2.y;
        }
        {
            const f3 om = ld3(a.O, vi);
            const DRot dm = drot(ld3(a.SC, vi), ld3(a.SC, (long)N + vi));
            f3 dv; dv.x = om.x - on.x; dv.y = om.y - on.y; dv.z = om.z - on.z;
            f3 g0, g1, g2; gcols(dm, dv, g0, g1, g2);
            gi[j].a = g0.x; gi[j].b = g0.y; gi[j].c = g0.z; gi[j].d = g1.x; gi[j].e = g1.y; gi[j].f = g1.z; gi[j].g = g2.x; gi[j].h = g2.y;
        }
    }
    // edge slots beyond MD (a real mesh has vertices of degree 10 and more; its AVERAGE is 6): slot and G of such an edge go to memory once per launch and are read back
    // every iteration -- a few per cent of the edges, L2-resident -- in the same place of the summation order (ascending slot) as the launch-per-iteration kernels have them
    for (int j = MD; OVF && j < deg; ++j) {
        const int vo = a.out_v1[(long)j * N + nc];
        const f3 om = ld3(a.O, vo);
        f3 dv; dv.x = on.x - om.x; dv.y = on.y - om.y; dv.z = on.z - om.z;
        f3 g0, g1, g2; gcols(dn, dv, g0, g1, g2);
        float* o = a.ovf + 9L * ((long)(j - MD) * N + nc);
        o[0] = __int_as_float(slot_of(vo)); o[1] = g0.x; o[2] = g0.y; o[3] = g0.z; o[4] = g1.x; o[5] = g1.y; o[6] = g1.z; o[7] = g2.x; o[8] = g2.y;
    }
    for (int j = MD; OVF && j < ideg; ++j) {
        const int vi = a.in_src[(long)j * N + nc];
        const f3 om = ld3(a.O, vi);
        const DRot dm = drot(ld3(a.SC, vi), ld3(a.SC, (long)N + vi));
        f3 dv; dv.x = om.x - on.x; dv.y = om.y - on.y; dv.z = om.z - on.z;
        f3 g0, g1, g2; gcols(dm, dv, g0, g1, g2);
        float* o = a.ovf + 9L * ((long)(a.maxo - MD + j - MD) * N + nc);
        o[0] = __int_as_float(slot_of(vi)); o[1] = g0.x; o[2] = g0.y; o[3] = g0.z; o[4] = g1.x; o[5] = g1.y; o[6] = g1.z; o[7] = g2.x; o[8] = g2.y;
    }
    const int qn = own_s + tid;
    const bool fit = a.Cn[3 * nc] >= -999999.9f;
    // the solver state of the vertex; p_0 = M^-1 r_0 + 0 p (PCGUpdate's expression with alpha = beta = 0), for the ghosts too
    f3 rp = ld3(a.r, nc), ra = ld3(a.r, (long)N + nc), pp = ld3(a.p0, nc), pa = ld3(a.p0, (long)N + nc);
    const f3 mp = ld3(a.pre, nc), ma = ld3(a.pre, (long)N + nc);
    f3 dp = ld3(a.delta, nc), da = ld3(a.delta, (long)N + nc);
    f3 ap = { 0.f, 0.f, 0.f }, aa = { 0.f, 0.f, 0.f };
    float an = sum_partials(a.aN0.partials, a.aN0.count);
    __syncthreads();                                                  // (s_vid)
    {
        pp.x = __builtin_fmaf(0.0f, pp.x, rp.x * mp.x); pp.y = __builtin_fmaf(0.0f, pp.y, rp.y * mp.y); pp.z = __builtin_fmaf(0.0f, pp.z, rp.z * mp.z);
        pa.x = __builtin_fmaf(0.0f, pa.x, ra.x * ma.x); pa.y = __builtin_fmaf(0.0f, pa.y, ra.y * ma.y); pa.z = __builtin_fmaf(0.0f, pa.z, ra.z * ma.z);
        if (live) { lp[qn] = make_float4(pp.x, pp.y, pp.z, 0.0f); lp[ARAP_RES_SPAN + qn] = make_float4(pa.x, pa.y, pa.z, 0.0f); }
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            const int g = tid + h * BLOCK;
            if (g < nghost) {
                const int u = g < own_s ? g : g + nown, v = s_gl[g];
                const f3 r0 = ld3(a.r, v), r1 = ld3(a.r, (long)N + v), m0 = ld3(a.pre, v), m1 = ld3(a.pre, (long)N + v), z0 = ld3(a.p0, v), z1 = ld3(a.p0, (long)N + v);
                gr[g] = r0.x; gr[ARAP_RES_GHOSTS + g] = r0.y; gr[2 * ARAP_RES_GHOSTS + g] = r0.z; gr[3 * ARAP_RES_GHOSTS + g] = r1.x; gr[4 * ARAP_RES_GHOSTS + g] = r1.y; gr[5 * ARAP_RES_GHOSTS + g] = r1.z;
                gm[g] = m0.x; gm[ARAP_RES_GHOSTS + g] = m0.y; gm[2 * ARAP_RES_GHOSTS + g] = m0.z; gm[3 * ARAP_RES_GHOSTS + g] = m1.x; gm[4 * ARAP_RES_GHOSTS + g] = m1.y; gm[5 * ARAP_RES_GHOSTS + g] = m1.z;
                lp[u] = make_float4(__builtin_fmaf(0.0f, z0.x, r0.x * m0.x), __builtin_fmaf(0.0f, z0.y, r0.y * m0.y), __builtin_fmaf(0.0f, z0.z, r0.z * m0.z), 0.0f);
                lp[ARAP_RES_SPAN + u] = make_float4(__builtin_fmaf(0.0f, z1.x, r1.x * m1.x), __builtin_fmaf(0.0f, z1.y, r1.y * m1.y), __builtin_fmaf(0.0f, z1.z, r1.z * m1.z), 0.0f);
            }
        }
    }
    const unsigned seq0 = a.ctl[ARES_SEQ];
    ASpin sp = { 0u, 0 };
    bool dead = false;                                                // a bounded wait ran out (here or elsewhere): fall through to the end
    __syncthreads();
    for (int k = 0; k < a.L; ++k) {
        const unsigned tag = seq0 + (unsigned)k + 1u;
        const unsigned par = (unsigned)(k & 1);
        ASTAMP(0);
        // ---- J^T J p_k at my vertex (k_arap_apply_rc, expression for expression; the neighbours' p from LDS, the G matrices from registers)
        ap.x = 0.f; ap.y = 0.f; ap.z = 0.f; aa.x = 0.f; aa.y = 0.f; aa.z = 0.f;
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const float4 t4 = lp[qo[j]]; f3 pm; pm.x = t4.x; pm.y = t4.y; pm.z = t4.z;
            f3 g0, g1, g2; g0.x = go[j].a; g0.y = go[j].b; g0.z = go[j].c; g1.x = go[j].d; g1.y = go[j].e; g1.z = go[j].f; g2.x = go[j].g; g2.y = go[j].h; g2.z = 0.0f;
            const float jx = (pp.x - pm.x) - (g0.x * pa.x + g1.x * pa.y + g2.x * pa.z);
            const float jy = (pp.y - pm.y) - (g0.y * pa.x + g1.y * pa.y + g2.y * pa.z);
            const float jz = (pp.z - pm.z) - (g0.z * pa.x + g1.z * pa.y + g2.z * pa.z);
            if (j < deg) {
                ap.x += jx; ap.y += jy; ap.z += jz;
                aa.x -= g0.x * jx + g0.y * jy + g0.z * jz;
                aa.y -= g1.x * jx + g1.y * jy + g1.z * jz;
                aa.z -= g2.x * jx + g2.y * jy + g2.z * jz;
            }
        }
        for (int j = MD; OVF && j < deg; ++j) {                      // (the thread's own stores of the prologue: program order)
            const float* o = a.ovf + 9L * ((long)(j - MD) * N + nc);
            const float4 t4 = lp[__float_as_int(o[0])]; f3 pm; pm.x = t4.x; pm.y = t4.y; pm.z = t4.z;
            f3 g0, g1, g2; g0.x = o[1]; g0.y = o[2]; g0.z = o[3]; g1.x = o[4]; g1.y = o[5]; g1.z = o[6]; g2.x = o[7]; g2.y = o[8]; g2.z = 0.0f;
            const float jx = (pp.x - pm.x) - (g0.x * pa.x + g1.x * pa.y + g2.x * pa.z);
            const float jy = (pp.y - pm.y) - (g0.y * pa.x + g1.y * pa.y + g2.y * pa.z);
            const float jz = (pp.z - pm.z) - (g0.z * pa.x + g1.z * pa.y + g2.z * pa.z);
            ap.x += jx; ap.y += jy; ap.z += jz;
            aa.x -= g0.x * jx + g0.y * jy + g0.z * jz;
            aa.y -= g1.x * jx + g1.y * jy + g1.z * jz;
            aa.z -= g2.x * jx + g2.y * jy + g2.z * jz;
        }
#pragma unroll
        for (int j = 0; j < MD; ++j) {
            const float4 t4 = lp[qi[j]], u4 = lp[ARAP_RES_SPAN + qi[j]]; f3 pm, am; pm.x = t4.x; pm.y = t4.y; pm.z = t4.z; am.x = u4.x; am.y = u4.y; am.z = u4.z;
            f3 g0, g1, g2; g0.x = gi[j].a; g0.y = gi[j].b; g0.z = gi[j].c; g1.x = gi[j].d; g1.y = gi[j].e; g1.z = gi[j].f; g2.x = gi[j].g; g2.y = gi[j].h; g2.z = 0.0f;
            if (j < ideg) {
                ap.x -= (pm.x - pp.x) - (g0.x * am.x + g1.x * am.y + g2.x * am.z);
                ap.y -= (pm.y - pp.y) - (g0.y * am.x + g1.y * am.y + g2.y * am.z);
                ap.z -= (pm.z - pp.z) - (g0.z * am.x + g1.z * am.y + g2.z * am.z);
            }
        }
        for (int j = MD; OVF && j < ideg; ++j) {
            const float* o = a.ovf + 9L * ((long)(a.maxo - MD + j - MD) * N + nc);
            const int q = __float_as_int(o[0]);
            const float4 t4 = lp[q], u4 = lp[ARAP_RES_SPAN + q]; f3 pm, am; pm.x = t4.x; pm.y = t4.y; pm.z = t4.z; am.x = u4.x; am.y = u4.y; am.z = u4.z;
            f3 g0, g1, g2; g0.x = o[1]; g0.y = o[2]; g0.z = o[3]; g1.x = o[4]; g1.y = o[5]; g1.z = o[6]; g2.x = o[7]; g2.y = o[8]; g2.z = 0.0f;
            ap.x -= (pm.x - pp.x) - (g0.x * am.x + g1.x * am.y + g2.x * am.z);
            ap.y -= (pm.y - pp.y) - (g0.y * am.x + g1.y * am.y + g2.y * am.z);
            ap.z -= (pm.z - pp.z) - (g0.z * am.x + g1.z * am.y + g2.z * am.z);
        }
        ap.x *= wr2; ap.y *= wr2; ap.z *= wr2; aa.x *= wr2; aa.y *= wr2; aa.z *= wr2;
        if (fit) { ap.x += a.wf * a.wf * pp.x; ap.y += a.wf * a.wf * pp.y; ap.z += a.wf * a.wf * pp.z; }
        ASTAMP(1);
        // ---- the workgroup's sums (block_finish_sums' arithmetic) as one record of four tagged parts
        float acc = 0.0f; Sums3 sm;
        if (live) {
            acc += pp.x * ap.x + pp.y * ap.y + pp.z * ap.z + pa.x * aa.x + pa.y * aa.y + pa.z * aa.z;
            sm.add(mp.x, rp.x, ap.x); sm.add(mp.y, rp.y, ap.y); sm.add(mp.z, rp.z, ap.z);
            sm.add(ma.x, ra.x, aa.x); sm.add(ma.y, ra.y, aa.y); sm.add(ma.z, ra.z, aa.z);
        }
        {
            const float wa = wave_sum_all(acc);
            const double w0 = wave_sum_all_f64(sm.n), w1 = wave_sum_all_f64(sm.s1), w2 = wave_sum_all_f64(sm.s2);
            if (lane == 0) { red[wave] = wa; redd[3 * wave] = w0; redd[3 * wave + 1] = w1; redd[3 * wave + 2] = w2; }
            lds_barrier();
            if (tid == 0) {
                float sa = 0.0f; double b0 = 0.0, b1 = 0.0, b2 = 0.0;
                for (int w = 0; w < BLOCK / 64; ++w) { sa += red[w]; b0 += redd[3 * w]; b1 += redd[3 * w + 1]; b2 += redd[3 * w + 2]; }
                const u64r q0 = (u64r)__double_as_longlong(b0), q1 = (u64r)__double_as_longlong(b1), q2 = (u64r)__double_as_longlong(b2);
                const unsigned off = 16u * (unsigned)(4u * par * (unsigned)a.nwg + (unsigned)wg), ps = 16u * (unsigned)a.nwg;
                u32x4r d;
                d.x = __float_as_uint(sa); d.y = tag; d.z = (unsigned)(q0 >> 32); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, RREC, off, 0, 16);
                d.x = (unsigned)q0; d.y = tag; d.z = (unsigned)(q1 >> 32); d.w = tag;        __builtin_amdgcn_raw_buffer_store_b128(d, RREC, off + ps, 0, 16);
                d.x = (unsigned)q1; d.y = tag; d.z = (unsigned)(q2 >> 32); d.w = tag;        __builtin_amdgcn_raw_buffer_store_b128(d, RREC, off + 2u * ps, 0, 16);
                d.x = (unsigned)q2; d.y = tag; d.z = 0u; d.w = tag;                          __builtin_amdgcn_raw_buffer_store_b128(d, RREC, off + 3u * ps, 0, 16);
            }
        }
        // ---- A p_k out: tagged granules for the workgroups that have my vertex as a ghost -- BEHIND the record in the CU's store queue: the record is what the whole
        //      chip waits for, the granules are looked at a tree's latency later
        if (live) {
            const unsigned o0 = 24u * (unsigned)n + 48u * par * (unsigned)N, o1 = o0 + 24u * (unsigned)N;
            u32x4r d; u32x2r e;
            d.x = __float_as_uint(ap.x); d.y = tag; d.z = __float_as_uint(ap.y); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, RAG, o0, 0, 16);
            e.x = __float_as_uint(ap.z); e.y = tag;                                        __builtin_amdgcn_raw_buffer_store_b64(e, RAG, o0 + 16u, 0, 16);
            d.x = __float_as_uint(aa.x); d.y = tag; d.z = __float_as_uint(aa.y); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, RAG, o1, 0, 16);
            e.x = __float_as_uint(aa.z); e.y = tag;                                        __builtin_amdgcn_raw_buffer_store_b64(e, RAG, o1 + 16u, 0, 16);
        }
        ASTAMP(2);
        // ---- the sums' tree.  A record / lane record is four 16-byte parts {alphaD, tag, N.hi, tag} {N.lo, tag, S1.hi, tag} {S1.lo, tag, S2.hi, tag} {S2.lo, tag, 0, tag}.
        auto poll4 = [&](const rsrc_r& R, unsigned off, unsigned ps, unsigned what, unsigned idx, float& v_ad, double& v0, double& v1, double& v2) {
            for (;;) {
                const u32x4r g0 = __builtin_amdgcn_raw_buffer_load_b128(R, off, 0, 16), g1 = __builtin_amdgcn_raw_buffer_load_b128(R, off + ps, 0, 16);
                const u32x4r g2 = __builtin_amdgcn_raw_buffer_load_b128(R, off + 2u * ps, 0, 16), g3 = __builtin_amdgcn_raw_buffer_load_b128(R, off + 3u * ps, 0, 16);
                if (g0.y == tag && g0.w == tag && g1.y == tag && g1.w == tag && g2.y == tag && g2.w == tag && g3.y == tag) {
                    v_ad = __uint_as_float(g0.x);
                    v0 = __longlong_as_double((long long)(((u64r)g0.z << 32) | (u64r)g1.x));
                    v1 = __longlong_as_double((long long)(((u64r)g1.z << 32) | (u64r)g2.x));
                    v2 = __longlong_as_double((long long)(((u64r)g2.z << 32) | (u64r)g3.x));
                    return;
                }
                if (ares_spin_fail(sp, a.ctl, what, idx, tag)) { dead = true; return; }
            }
        };
        auto store4 = [&](const rsrc_r& R, unsigned off, unsigned ps, float sa, double b0, double b1, double b2) {
            const u64r q0 = (u64r)__double_as_longlong(b0), q1 = (u64r)__double_as_longlong(b1), q2 = (u64r)__double_as_longlong(b2);
            u32x4r d;
            d.x = __float_as_uint(sa); d.y = tag; d.z = (unsigned)(q0 >> 32); d.w = tag; __builtin_amdgcn_raw_buffer_store_b128(d, R, off, 0, 16);
            d.x = (unsigned)q0; d.y = tag; d.z = (unsigned)(q1 >> 32); d.w = tag;        __builtin_amdgcn_raw_buffer_store_b128(d, R, off + ps, 0, 16);
            d.x = (unsigned)q1; d.y = tag; d.z = (unsigned)(q2 >> 32); d.w = tag;        __builtin_amdgcn_raw_buffer_store_b128(d, R, off + 2u * ps, 0, 16);
            d.x = (unsigned)q2; d.y = tag; d.z = 0u; d.w = tag;                          __builtin_amdgcn_raw_buffer_store_b128(d, R, off + 3u * ps, 0, 16);
        };
        if (wave == 0) {
            // lane records: [parity][copy][part][64]; totals (behind them): [parity][copy][part].  Up to ARAP_RES_ROOT workgroups the last workgroup reads the lane records
            // and everybody else polls its four totals (fewest pollers: 6.3 vs 6.9 us per iteration at 25 workgroups, 8.9 vs 9.2 at 100); beyond, one more level costs more
            // than it saves (12.8 vs 12.2 us at 400) and every workgroup reads the lane records itself, from its XCD's copy
            const bool rooted = a.nwg <= ARAP_RES_ROOT;
            const unsigned lps = 16u * 64u;
            auto lbase = [&](unsigned copy) { return 16u * 4u * 64u * (par * ARAP_RES_COPIES + copy); };
            const unsigned fbase = 16u * 4u * 64u * 2u * ARAP_RES_COPIES + 16u * 4u * ARAP_RES_COPIES * par;
            // level 1 (workgroups 0 .. 63 of more than 64): lane j takes record wg + 64 j; lane 0 adds them up ascending from zero, as lane `wg` of load_iteration_sums does
            if (a.nwg > 64 && wg < 64) {
                float v_ad = 0.0f; double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                const int i = wg + 64 * lane;
                if (lane < 8 && i < a.nwg && !dead) poll4(RREC, 16u * (unsigned)(4u * par * (unsigned)a.nwg + (unsigned)i), 16u * (unsigned)a.nwg, 2u, (unsigned)i, v_ad, v0, v1, v2);
                float t = 0.0f; double x = 0.0, y = 0.0, z = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float fj = __shfl(v_ad, j); const double x0 = __shfl(v0, j), x1 = __shfl(v1, j), x2 = __shfl(v2, j);
                    if (wg + 64 * j < a.nwg) { t += fj; x += x0; y += x1; z += x2; }
                }
                if (lane < (rooted ? 1 : ARAP_RES_COPIES)) store4(RLREC, lbase((unsigned)lane) + 16u * (unsigned)wg, lps, t, x, y, z);
            }
            float ad_ = 0.0f; double n_ = 0.0, s1_ = 0.0, s2_ = 0.0;
            if (!rooted || wg == a.nwg - 1) {
                // level 2: lane l takes lane record l -- with at most 64 workgroups record l itself, added to zero like a lane's only slot -- and the wave runs the
                // butterflies of load_iteration_sums; the root publishes the four totals, one copy per XCD
                float t = 0.0f; double x = 0.0, y = 0.0, z = 0.0;
                if (lane < nlane && !dead) {
                    if (a.nwg > 64) poll4(RLREC, lbase(rooted ? 0u : (unsigned)(wg % ARAP_RES_COPIES)) + 16u * (unsigned)lane, lps, 3u, (unsigned)lane, t, x, y, z);
                    else {
                        float v_ad = 0.0f; double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                        poll4(RREC, 16u * (unsigned)(4u * par * (unsigned)a.nwg + (unsigned)lane), 16u * (unsigned)a.nwg, 2u, (unsigned)lane, v_ad, v0, v1, v2);
                        t += v_ad; x += v0; y += v1; z += v2;
                    }
                }
                ad_ = wave_sum_all(t); n_ = wave_sum_all_f64(x); s1_ = wave_sum_all_f64(y); s2_ = wave_sum_all_f64(z);
                if (rooted && lane < ARAP_RES_COPIES) store4(RLREC, fbase + 64u * (unsigned)lane, 16u, ad_, n_, s1_, s2_);
            } else {
                // level 3 (rooted, everybody else): the totals, from my XCD's copy
                if (lane == 0 && !dead) poll4(RLREC, fbase + 64u * (unsigned)(wg % ARAP_RES_COPIES), 16u, 4u, 0u, ad_, n_, s1_, s2_);
            }
            if (lane == 0) { s_tot_f = ad_; s_tot[0] = n_; s_tot[1] = s1_; s_tot[2] = s2_; }
        }
        ASTAMP(3);
        __syncthreads();
        ASTAMP(4);
        // ---- my ghosts' A p_k: the loads go out now -- every workgroup has published (an earlier first look is mostly stale and doubles the fabric traffic, which is
        //      what this loop is bound by: 9 us of waiting at 400 workgroups) -- and are looked at after my own vertex's update
        u32x4r xa[GH], ya[GH]; u32x2r xb[GH], yb[GH]; unsigned of[GH];
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            const int g = tid + h * BLOCK;
            const int v = g < nghost ? s_gl[g] : 0;
            of[h] = (g >= nghost || dead) ? 0xffffffffu : 24u * (unsigned)v + 48u * par * (unsigned)N;
            if (of[h] != 0xffffffffu) {
                xa[h] = __builtin_amdgcn_raw_buffer_load_b128(RAG, of[h], 0, 16); xb[h] = __builtin_amdgcn_raw_buffer_load_b64(RAG, of[h] + 16u, 0, 16);
                ya[h] = __builtin_amdgcn_raw_buffer_load_b128(RAG, of[h] + 24u * (unsigned)N, 0, 16); yb[h] = __builtin_amdgcn_raw_buffer_load_b64(RAG, of[h] + 24u * (unsigned)N + 16u, 0, 16);
            }
        }
        const float ad = s_tot_f; const double qn_ = s_tot[0], q1 = s_tot[1], q2 = s_tot[2];
        const float al = safe_div<false>(an, ad);
        double bn = qn_ - 2.0 * (double)al * q1 + (double)al * (double)al * q2;
        if (!(bn > 0.0)) bn = 0.0;
        const float bnf = (float)bn;
        if (wg == 0 && tid == 0) { a.words[2 * k] = ad; a.words[2 * k + 1] = bnf; }
        const float beta = safe_div<false>(bnf, an);
        an = bnf;
        ASTAMP(5);
        // ---- PCGUpdate for iteration k + 1 (pcg_kernels.hip: k_pcg_update, expression for expression): r -= alpha A p ; delta += alpha p ; p = M^-1 r + beta p --
        //      at my vertex, and at my ghosts from their owners' A p_k
        if (k + 1 < a.L) {
            rp.x = __builtin_fmaf(-al, ap.x, rp.x); rp.y = __builtin_fmaf(-al, ap.y, rp.y); rp.z = __builtin_fmaf(-al, ap.z, rp.z);
            ra.x = __builtin_fmaf(-al, aa.x, ra.x); ra.y = __builtin_fmaf(-al, aa.y, ra.y); ra.z = __builtin_fmaf(-al, aa.z, ra.z);
            dp.x = __builtin_fmaf(al, pp.x, dp.x); dp.y = __builtin_fmaf(al, pp.y, dp.y); dp.z = __builtin_fmaf(al, pp.z, dp.z);
            da.x = __builtin_fmaf(al, pa.x, da.x); da.y = __builtin_fmaf(al, pa.y, da.y); da.z = __builtin_fmaf(al, pa.z, da.z);
            const float zx = rp.x * mp.x, zy = rp.y * mp.y, zz = rp.z * mp.z, wx = ra.x * ma.x, wy = ra.y * ma.y, wz = ra.z * ma.z;
            pp.x = __builtin_fmaf(beta, pp.x, zx); pp.y = __builtin_fmaf(beta, pp.y, zy); pp.z = __builtin_fmaf(beta, pp.z, zz);
            pa.x = __builtin_fmaf(beta, pa.x, wx); pa.y = __builtin_fmaf(beta, pa.y, wy); pa.z = __builtin_fmaf(beta, pa.z, wz);
            if (live) { lp[qn] = make_float4(pp.x, pp.y, pp.z, 0.0f); lp[ARAP_RES_SPAN + qn] = make_float4(pa.x, pa.y, pa.z, 0.0f); }
#pragma unroll
            for (int h = 0; h < GH; ++h) {
                if (of[h] == 0xffffffffu) continue;
                while (!(xa[h].y == tag && xa[h].w == tag && xb[h].y == tag && ya[h].y == tag && ya[h].w == tag && yb[h].y == tag)) {
                    if (ares_spin_fail(sp, a.ctl, 1u, (of[h] - 48u * par * (unsigned)N) / 24u, tag)) { dead = true; break; }
                    xa[h] = __builtin_amdgcn_raw_buffer_load_b128(RAG, of[h], 0, 16); xb[h] = __builtin_amdgcn_raw_buffer_load_b64(RAG, of[h] + 16u, 0, 16);
                    ya[h] = __builtin_amdgcn_raw_buffer_load_b128(RAG, of[h] + 24u * (unsigned)N, 0, 16); yb[h] = __builtin_amdgcn_raw_buffer_load_b64(RAG, of[h] + 24u * (unsigned)N + 16u, 0, 16);
                }
                const int g = tid + h * BLOCK, u = g < own_s ? g : g + nown;
                const float av[6] = { __uint_as_float(xa[h].x), __uint_as_float(xa[h].z), __uint_as_float(xb[h].x), __uint_as_float(ya[h].x), __uint_as_float(ya[h].z), __uint_as_float(yb[h].x) };
                const float4 p0 = lp[u], p1 = lp[ARAP_RES_SPAN + u];
                float pv[6] = { p0.x, p0.y, p0.z, p1.x, p1.y, p1.z };
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float rn = __builtin_fmaf(-al, av[c], gr[c * ARAP_RES_GHOSTS + g]);
                    gr[c * ARAP_RES_GHOSTS + g] = rn;
                    pv[c] = __builtin_fmaf(beta, pv[c], rn * gm[c * ARAP_RES_GHOSTS + g]);
                }
                lp[u] = make_float4(pv[0], pv[1], pv[2], 0.0f); lp[ARAP_RES_SPAN + u] = make_float4(pv[3], pv[4], pv[5], 0.0f);
            }
        }
        dead = dead || __hip_atomic_load(a.ctl + ARES_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        ASTAMP(6);
        __syncthreads();                                                // (p_{k+1} of every staged vertex is in LDS; the staging arrays are free again)
        ASTAMP(7);
    }
    if (live) {
        float* pl = (a.L & 1) ? a.p1 : a.p0;                            // p_{L-1}: where L launches of the flat update leave it
        st3(pl, n, pp); st3(pl, (long)N + n, pa);
        st3(a.r, n, rp); st3(a.r, (long)N + n, ra); st3(a.Ap, n, ap); st3(a.Ap, (long)N + n, aa); st3(a.delta, n, dp); st3(a.delta, (long)N + n, da);
    }
}

int g_arap_unrolled = 1;       // tools / tests: 0 = the loop form for every layout
int g_arap_recompute = 1;      // tools / tests: 0 = read the stored G planes (round 2's kernels)

// launches the unrolled ELL form when the layout allows it (maxdeg = S / N <= 8), the loop form otherwise
template <typename... A>
void launch_arap_apply(int grid, hipStream_t stream, ELay L, A... a)
{
    const long maxdeg = L.S && L.N > 0 ? L.S / L.N : 0;
    if (g_arap_unrolled && maxdeg >= 1 && maxdeg <= 6) hipLaunchKernelGGL(k_arap_apply_ell<6>, dim3(grid), dim3(BLOCK), 0, stream, a...);
    else if (g_arap_unrolled && maxdeg >= 1 && maxdeg <= 8) hipLaunchKernelGGL(k_arap_apply_ell<8>, dim3(grid), dim3(BLOCK), 0, stream, a...);
    else hipLaunchKernelGGL(k_arap_apply, dim3(grid), dim3(BLOCK), 0, stream, a...);
}

}  // namespace

extern "C" {

void thallo_hip_arap_debug_set(int what, int value) { if (what == 0) g_arap_unrolled = value; if (what == 1) g_arap_recompute = value; }

int thallo_hip_lapgraph_cost(int N, const int* out_ptr, const int* out_v1, const float* X, const float* A, float w_fit,
                             float* cost_out, thallo_stream_t stream)
{
    const int grid = vgrid(N);
    hipLaunchKernelGGL(k_lapg_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, out_ptr, out_v1, X, A, w_fit, cost_out);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_lapgraph_pcg_init(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                 const float* X, const float* A, float w_fit, float* r, float* z, float* p_prev, float* delta,
                                 float* diag_out, float* aN_out, thallo_stream_t stream)
{
    const int grid = vgrid(N);
    hipLaunchKernelGGL(k_lapg_init, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, out_ptr, out_v1, in_ptr, in_src, X, A, w_fit, r, z, p_prev, delta, diag_out, aN_out);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_lapgraph_apply_jtj(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                  float w_fit, const float* p, float* Ap, float* aD_out, thallo_stream_t stream)
{
    const int grid = vgrid(N);
    hipLaunchKernelGGL(k_lapg_apply, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, out_ptr, out_v1, in_ptr, in_src, w_fit, p, Ap, aD_out);
    int e = check_launch(); return e ? e : grid;
}

int thallo_hip_arap_cost(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                         const float* original, const float* constraints, float w_fit, float w_reg, float* cost_out, long ell_stride, thallo_stream_t stream)
{
    if (n0 < 0 || n1 > N || n0 >= n1 || ell_stride < 0) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(n1 - n0);
    hipLaunchKernelGGL(k_arap_cost, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, n0, n1, out_ptr, out_v1, position, angle, original, constraints, w_fit, w_reg, cost_out, L);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_arap_precompute2(int N, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                                const float* original, float w_reg, float* F, float* G, float* SC, long ell_stride, thallo_stream_t stream)
{
    if (ell_stride < 0) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(N);
    hipLaunchKernelGGL(k_arap_precompute, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, out_ptr, out_v1, position, angle, original, w_reg, F, G, SC, L);
    return check_launch();
}
int thallo_hip_arap_precompute(int N, const int* out_ptr, const int* out_v1, const float* position, const float* angle,
                               const float* original, float w_reg, float* F, float* G, long ell_stride, thallo_stream_t stream)
{
    return thallo_hip_arap_precompute2(N, out_ptr, out_v1, position, angle, original, w_reg, F, G, nullptr, ell_stride, stream);
}
int thallo_hip_arap_recompute_supported(int N, long ell_stride)
{
    return g_arap_recompute && N > 0 && ell_stride > 0 && ell_stride % N == 0 && ell_stride / N <= 8 ? 1 : 0;
}
int thallo_hip_arap_apply_jtj_rc(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                 const float* constraints, const float* original, const float* SC, float w_fit, float w_reg,
                                 const float* p, float* Ap, float* aD_out, long ell_stride, const float* r, const float* pre, double* s3_out,
                                 thallo_fin_t fin, thallo_stream_t stream)
{
    if (n0 < 0 || n1 > N || n0 >= n1 || !original || !SC || !p || !Ap || !aD_out) return -(int)hipErrorInvalidValue;
    if (ell_stride <= 0 || ell_stride % N != 0 || ell_stride / N > 8) return -(int)hipErrorNotSupported;
    if (s3_out && (!r || !pre)) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!s3_out || !fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(n1 - n0);
    const FinArgs f = { fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, grid };
    if (ell_stride / N <= 6) hipLaunchKernelGGL(k_arap_apply_rc<6>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, n0, n1, out_ptr, out_v1, in_ptr, in_src, constraints, original, SC, w_fit, w_reg, p, Ap, aD_out, L, r, pre, s3_out, f);
    else                     hipLaunchKernelGGL(k_arap_apply_rc<8>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, n0, n1, out_ptr, out_v1, in_ptr, in_src, constraints, original, SC, w_fit, w_reg, p, Ap, aD_out, L, r, pre, s3_out, f);
    int e = check_launch(); return e ? e : grid;
}
/* ---- the resident PCG loop (k_arap_resident): exchange memory layout [control words 256 B | lane records and totals 8192 x ARAP_RES_COPIES B | records 2 x 4 x nwg x 16 B | ghost lists nwg x ARAP_RES_LISTW ints | A p granules 2 x 2 x N x 3 x 8 B] */
static inline int ares_nwg(int N) { return (N + BLOCK - 1) / BLOCK; }
static inline long ares_off_lrec() { return 256; }
static inline long ares_off_rec() { return 256 + 8192L * ARAP_RES_COPIES + 1024; }
static inline long ares_off_range(int nwg) { return ares_off_rec() + 128L * nwg; }
static inline long ares_off_pg(int nwg) { return (ares_off_range(nwg) + 4L * ARAP_RES_LISTW * nwg + 255) / 256 * 256; }
long thallo_hip_arap_resident_bytes(int N) { if (N < 1) return 0; const int nwg = ares_nwg(N); return ares_off_pg(nwg) + 96L * N + 256 + 8192; }      // (+ 8 KB: the research build's per-workgroup stamps)
/* floats of the overflow buffer for ELL lists of out_slots / in_slots edge slots (0: none needed) */
long thallo_hip_arap_resident_overflow_floats(int N, int out_slots, int in_slots) { const long x = (long)std::max(0, out_slots - 6) + std::max(0, in_slots - 6); return N < 1 ? 0 : 9L * N * x; }
long thallo_hip_arap_resident_lists_offset(int N) { return N < 1 ? -1 : ares_off_range(ares_nwg(N)); }
int thallo_hip_arap_resident_max_ghosts(void) { return ARAP_RES_GHOSTS; }
/* 1: the shape can run the resident loop -- the ELL layout (at most 32 edge slots; the first 6 of a vertex live in registers, the rest go through memory), and every workgroup resident at once
 * (the staged sets are the caller's to check: at most thallo_hip_arap_resident_max_ghosts() vertices of other workgroups per workgroup) */
int thallo_hip_arap_resident_fits(int N, long ell_stride)
{
    if (N < 1 || ell_stride <= 0 || ell_stride % N != 0 || ell_stride / N > 32) return 0;      // (the ELL layout; slots beyond 6 through memory)
    const int nwg = ares_nwg(N);
    if (nwg > 512 || nwg > THALLO_MAX_PARTIALS || 96.0 * N >= 4294967296.0) return 0;         // (a lane record adds up to 8 records; granule offsets are 32-bit)
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_arap_resident<6, true>, BLOCK, 0);
        per_cu = (e == hipSuccess && n >= 1) ? n : -1;
        if (e != hipSuccess) (void)hipGetLastError();
    }
    return per_cu > 0 && (long)per_cu * thallo_hip_device_cu_count() >= nwg ? 1 : 0;
}
/* L iterations from what thallo_hip_arap_pcg_init left (r_0 in r, M^-1 in pre, zeros in p0 and delta): leaves r_{L-1}, A p_{L-1}, p_{L-1} (in p0 / p1 by the parity of
 * L, like L launches of the flat update), delta = sum_{k < L-1} alpha_k p_k and words[2k] = alphaD_k, words[2k + 1] = betaN_k -- what PCGUpdate + applyJTJ per iteration
 * leave, bit for bit.  xbuf: thallo_hip_arap_resident_bytes(N) zero-filled bytes with the workgroups' ghost lists filled in.  Returns the grid size. */
int thallo_hip_arap_pcg_resident(int N, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_src,
                                 const float* constraints, const float* original, const float* SC, float w_fit, float w_reg, long ell_stride,
                                 float* r, float* Ap, const float* pre, float* p0, float* p1, float* delta, thallo_sum_t alphaN0, float* words,
                                 void* xbuf, float* overflow, int in_slots, int L, thallo_stream_t stream)
{
    if (in_slots < 0 || in_slots > 32 || ((ell_stride / (N > 0 ? N : 1) > 6 || in_slots > 6) && !overflow)) return -(int)hipErrorInvalidValue;
    if (N < 1 || L < 1 || !out_ptr || !out_v1 || !in_ptr || !in_src || !constraints || !original || !SC || !r || !Ap || !pre || !p0 || !p1 || !delta || !words || !xbuf || !alphaN0.partials)
        return -(int)hipErrorInvalidValue;
    if (!thallo_hip_arap_resident_fits(N, ell_stride)) return -(int)hipErrorNotSupported;
    ArapResArgs a; memset(&a, 0, sizeof(a));
    a.N = N; a.nwg = ares_nwg(N); a.L = L; a.ell = ell_stride;
    a.out_ptr = out_ptr; a.out_v1 = out_v1; a.in_ptr = in_ptr; a.in_src = in_src; a.Cn = constraints; a.O = original; a.SC = SC; a.wf = w_fit; a.wr = w_reg;
    a.r = r; a.Ap = Ap; a.pre = pre; a.p0 = p0; a.p1 = p1; a.delta = delta; a.aN0 = alphaN0; a.words = words;
    a.ovf = overflow; a.maxo = std::max(6, (int)(ell_stride / N)); a.maxi = std::max(6, in_slots);
    char* base = (char*)xbuf;
#ifdef ARAP_STAMPS
    g_arap_dbg_xbuf = xbuf;
#endif
    a.ctl = (unsigned*)base; a.rec = (u64r*)(base + ares_off_rec()); a.lrec = (u64r*)(base + ares_off_lrec()); a.wlist = (const int*)(base + ares_off_range(a.nwg)); a.ag = (u64r*)(base + ares_off_pg(a.nwg)); a.stamps = (unsigned*)(base + ares_off_pg(a.nwg) + 96L * N + 256);
    hipLaunchKernelGGL(k_arap_res_begin, dim3(1), dim3(64), 0, (hipStream_t)stream, a.ctl, (unsigned)L);
    if (a.maxo > 6 || a.maxi > 6) hipLaunchKernelGGL((k_arap_resident<6, true>), dim3(a.nwg), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else                          hipLaunchKernelGGL((k_arap_resident<6, false>), dim3(a.nwg), dim3(BLOCK), 0, (hipStream_t)stream, a);
    int e = check_launch(); return e ? e : a.nwg;
}
/* 1: a bounded wait inside the resident loop ran out since the words were cleared (pm: what, workgroup, thread, index, tag); clear != 0 resets the error word */
int thallo_hip_arap_resident_status(void* xbuf, int clear, unsigned* pm, thallo_stream_t stream)
{
    if (!xbuf) return 0;
    unsigned w[ARES_CTL_WORDS];
    if (hipMemcpyAsync(w, xbuf, sizeof(w), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
    if (pm) for (int i = 0; i < 5; ++i) pm[i] = w[ARES_PM + i];
    if (clear && w[ARES_ERR]) { const unsigned z = 0; (void)hipMemcpyAsync((unsigned*)xbuf + ARES_ERR, &z, sizeof(unsigned), hipMemcpyHostToDevice, (hipStream_t)stream); (void)hipStreamSynchronize((hipStream_t)stream); }
    return w[ARES_ERR] ? 1 : 0;
}
/* an order-sensitive checksum of n ints (a plan asks whether the sparse maps behind unchanged pointers are still the ones its incidence lists were built from) */
__global__ __launch_bounds__(BLOCK) void k_checksum_i32(long n, const int* __restrict__ v, unsigned long long* __restrict__ out)
{
    unsigned long long h = 0;
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
        h += ((unsigned long long)(unsigned)v[i] + 0x9e3779b97f4a7c15ull) * (2ull * (unsigned long long)i + 1ull);
    for (int o = 32; o >= 1; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}
int thallo_hip_checksum_i32(long n, const int* v, unsigned long long* out_device, thallo_stream_t stream)
{
    if (n < 1 || !v || !out_device) return -(int)hipErrorInvalidValue;
    int grid = (int)std::min<long>((n + BLOCK - 1) / BLOCK, 512);
    hipLaunchKernelGGL(k_checksum_i32, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, n, v, out_device);
    return check_launch();
}
/* dst[i] = src[idx[i]] (scatter == 0) or dst[idx[i]] = src[i] (scatter != 0) for N float3: a vertex array between the caller's numbering and the plan's (plugins.cpp: ArapPlugin) */
__global__ __launch_bounds__(BLOCK) void k_permute3(int N, const int* __restrict__ idx, const float* __restrict__ src, float* __restrict__ dst, int scatter)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const long a = scatter ? i : idx[i], b = scatter ? idx[i] : i;
    dst[3 * b] = src[3 * a]; dst[3 * b + 1] = src[3 * a + 1]; dst[3 * b + 2] = src[3 * a + 2];
}
int thallo_hip_permute3(int N, const int* idx, const float* src, float* dst, int scatter, thallo_stream_t stream)
{
    if (N < 1 || !idx || !src || !dst || src == dst) return -(int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_permute3, dim3((N + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t)stream, N, idx, src, dst, scatter);
    return check_launch();
}
int thallo_hip_arap_pcg_init(int N, int n0, int n1, const int* out_ptr, const int* in_ptr, const int* in_edge, const float* position,
                             const float* constraints, const float* F, const float* G, float w_fit, float w_reg,
                             float* r, float* pre, float* z, float* p_prev, float* delta, float* diag_out, float* aN_out, long ell_stride, thallo_stream_t stream)
{
    if (n0 < 0 || n1 > N || n0 >= n1 || ell_stride < 0) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(n1 - n0);
    hipLaunchKernelGGL(k_arap_init, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, N, n0, n1, out_ptr, in_ptr, in_edge, position, constraints, F, G, w_fit, w_reg,
                       r, pre, z, p_prev, delta, diag_out, aN_out, L);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_arap_apply_jtj(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                              const float* constraints, const float* G, float w_fit, float w_reg,
                              const float* p, float* Ap, float* aD_out, long ell_stride, thallo_stream_t stream)
{
    if (n0 < 0 || n1 > N || n0 >= n1 || ell_stride < 0) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(n1 - n0);
    launch_arap_apply(grid, (hipStream_t)stream, L, N, n0, n1, out_ptr, out_v1, in_ptr, in_edge, in_src, constraints, G, w_fit, w_reg, p, Ap, aD_out, L, (const float*)nullptr, (const float*)nullptr, (double*)nullptr, FinArgs{});
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_arap_apply_jtj_sums_fin(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                                       const float* constraints, const float* G, float w_fit, float w_reg,
                                       const float* p, float* Ap, float* aD_out, long ell_stride, const float* r, const float* pre, double* s3_out,
                                       thallo_fin_t fin, thallo_stream_t stream)
{
    if (n0 < 0 || n1 > N || n0 >= n1 || ell_stride < 0 || !r || !pre || !s3_out) return -(int)hipErrorInvalidValue;
    if (fin.tickets && (!fin.alphaD_word || !fin.betaN_word || !fin.alphaN.partials)) return -(int)hipErrorInvalidValue;
    const ELay L = { ell_stride, N };
    const int grid = vgrid(n1 - n0);
    const FinArgs f = { fin.alphaN, fin.tickets, fin.alphaD_word, fin.betaN_word, 0, grid };
    launch_arap_apply(grid, (hipStream_t)stream, L, N, n0, n1, out_ptr, out_v1, in_ptr, in_edge, in_src, constraints, G, w_fit, w_reg, p, Ap, aD_out, L, r, pre, s3_out, f);
    int e = check_launch(); return e ? e : grid;
}
int thallo_hip_arap_apply_jtj_sums(int N, int n0, int n1, const int* out_ptr, const int* out_v1, const int* in_ptr, const int* in_edge, const int* in_src,
                                   const float* constraints, const float* G, float w_fit, float w_reg,
                                   const float* p, float* Ap, float* aD_out, long ell_stride, const float* r, const float* pre, double* s3_out, thallo_stream_t stream)
{
    const thallo_fin_t none = { { nullptr, 0 }, nullptr, nullptr, nullptr };
    return thallo_hip_arap_apply_jtj_sums_fin(N, n0, n1, out_ptr, out_v1, in_ptr, in_edge, in_src, constraints, G, w_fit, w_reg, p, Ap, aD_out, ell_stride, r, pre, s3_out, none, stream);
}

}  // extern "C"
