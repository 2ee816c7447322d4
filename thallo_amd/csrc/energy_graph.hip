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
