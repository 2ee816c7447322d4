/*
 * thallo_oracle.c -- see thallo_oracle.h.  TEST INFRASTRUCTURE ONLY.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, so the
 * float arithmetic below is the plain IEEE sequence the reference's CPU mode runs).
 */
#include "thallo_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------ utilities */

void orc_default_params(OrcSolverParams* sp)
{   /* gauss_newton.t:41-55 */
    sp->residual_reset_period   = 10;
    sp->min_relative_decrease   = 1e-3f;
    sp->min_trust_region_radius = 1e-32f;
    sp->max_trust_region_radius = 1e16f;
    sp->q_tolerance             = 0.0001f;
    sp->function_tolerance      = 0.000001f;
    sp->trust_region_radius     = 1e4f;
    sp->radius_decrease_factor  = 2.0f;
    sp->min_lm_diagonal         = 1e-6f;
    sp->max_lm_diagonal         = 1e32f;
    sp->nIterations             = 10;
    sp->lIterations             = 10;
    sp->use_lm                  = 0;
    sp->float_sums              = 0;
}

void orc_msvc_rand_fill(float* out, long n, unsigned seed)
{   /* tests/minimal/main.cpp:52-54 with the MSVC C runtime: RAND_MAX = 32767 */
    unsigned s = seed;
    for (long i = 0; i < n; ++i) {
        s = s * 214013u + 2531011u;
        int v = (int)((s >> 16) & 0x7fff);
        out[i] = (float)((double)v / 32767.0);
    }
}

/* summation helper: double accumulator, or float serial (cpu_cuda.t order) */
typedef struct { double d; float f; int fl; } Acc;
static inline void acc_init(Acc* a, int fl) { a->d = 0.0; a->f = 0.0f; a->fl = fl; }
static inline void acc_add(Acc* a, float v) { if (a->fl) a->f = a->f + v; else a->d += (double)v; }
static inline double acc_get(const Acc* a) { return a->fl ? (double)a->f : a->d; }

static inline const float* img(const OrcEnergy* e, int param) { return (const float*)e->params[param]; }

/* ------------------------------------------------------------------ energies */

/* E5: tests/minimal/laplacian.t:1-14.  params: 0 X (unknown), 1 A.  dims W,H.
 *   fit = w_fit*(X(x,y)-A(x,y))
 *   reg = { Select(G, X(x,y)-X(x+1,y), 0), Select(InBounds(x,y+1), X(x,y)-X(x,y+1), 0) }
 * G = InBounds(x+1,y+1) as shipped (iconst[0]=0) or InBounds(x+1,y) (iconst[0]=1: the
 * variant gold.png was produced with). */
static int rows_lap_image(const OrcEnergy* e, long elem, OrcRow* out)
{
    const long W = e->dims[0], H = e->dims[1];
    const long x = elem % W, y = elem / W;
    const float* X = img(e, 0); const float* A = img(e, 1);
    const float w = e->fconst[0];
    OrcRow* r = out;
    r->nnz = 1; r->col[0] = (int)elem; r->val[0] = w; r->r = w * (X[elem] - A[elem]); ++r;
    int gx = e->iconst[0] ? (x + 1 < W) : (x + 1 < W && y + 1 < H);
    if (gx) { r->nnz = 2; r->col[0] = (int)elem; r->val[0] = 1.0f; r->col[1] = (int)(elem + 1); r->val[1] = -1.0f;
              r->r = X[elem] - X[elem + 1]; }
    else    { r->nnz = 0; r->r = 0.0f; }
    ++r;
    if (y + 1 < H) { r->nnz = 2; r->col[0] = (int)elem; r->val[0] = 1.0f; r->col[1] = (int)(elem + W); r->val[1] = -1.0f;
                     r->r = X[elem] - X[elem + W]; }
    else           { r->nnz = 0; r->r = 0.0f; }
    return 3;
}

/* E6: tests/minimal_graph/laplacian.t:1-17.  params: 0 X, 1 A, 2 v0, 3 v1.  dims N,E.
 * groups in name order: fit (N elements) then reg (E elements). */
static int rows_lap_graph(const OrcEnergy* e, long elem, OrcRow* out)
{
    const long N = e->dims[0];
    const float* X = img(e, 0); const float* A = img(e, 1);
    if (elem < N) {
        const float w = e->fconst[0];
        out->nnz = 1; out->col[0] = (int)elem; out->val[0] = w; out->r = w * (X[elem] - A[elem]);
        return 1;
    }
    const int* v0 = (const int*)e->params[2]; const int* v1 = (const int*)e->params[3];
    const long k = elem - N;
    const int a = v0[k], b = v1[k];
    out->nnz = 2; out->col[0] = a; out->val[0] = 1.0f; out->col[1] = b; out->val[1] = -1.0f;
    out->r = X[a] - X[b];
    return 1;
}

/* E1: examples/image_warping/image_warping.t:1-31.
 * params: 0 Offset float2 (unknown), 1 Angle float (unknown), 2 UrShape float2, 3 Constraints float2,
 *         4 Mask float, 5 &w_fitSqrt, 6 &w_regSqrt.  dims W,H.
 * flat unknown layout (thallo.t:1104-1125): [Offset: 2*pix+c][Angle: 2*N+pix]. */
static int excl_image_warping(const OrcEnergy* e, long flat)
{   /* image_warping.t:14-15: both unknowns excluded where Mask != 0 */
    const long N = (long)e->dims[0] * e->dims[1];
    long pix = flat < 2 * N ? flat / 2 : flat - 2 * N;
    return img(e, 4)[pix] != 0.0f;
}
static int rows_image_warping(const OrcEnergy* e, long elem, OrcRow* out)
{
    const long W = e->dims[0], H = e->dims[1], N = W * H;
    const long x = elem % W, y = elem / W;
    const float* O = img(e, 0); const float* Ang = img(e, 1); const float* U = img(e, 2);
    const float* C = img(e, 3); const float* M = img(e, 4);
    const float wf = *(const float*)e->params[5];
    const float wr = *(const float*)e->params[6];
    static const int DX[4] = { 1, -1, 0, 0 }, DY[4] = { 0, 0, 1, -1 };   /* image_warping.t:18 */
    const float a = Ang[elem];
    const float ca = cosf(a), sa = sinf(a);
    OrcRow* r = out;
    for (int d = 0; d < 4; ++d) {
        const long xn = x + DX[d], yn = y + DY[d];
        const int inb = (xn >= 0 && xn < W && yn >= 0 && yn < H);
        const long j = inb ? yn * W + xn : 0;
        const int valid = inb && M[elem] == 0.0f && M[j] == 0.0f;
        if (!valid) { r[0].nnz = 0; r[0].r = 0.0f; r[1].nnz = 0; r[1].r = 0.0f; r += 2; continue; }
        const float dux = U[2 * elem] - U[2 * j], duy = U[2 * elem + 1] - U[2 * j + 1];
        const float dox = O[2 * elem] - O[2 * j], doy = O[2 * elem + 1] - O[2 * j + 1];
        /* Rotate2D (lib.t:138-142) and its angle derivative */
        const float rx = ca * dux + (-sa) * duy, ry = sa * dux + ca * duy;
        const float gx = (-sa) * dux - ca * duy, gy = ca * dux - sa * duy;
        r[0].nnz = 3; r[0].col[0] = (int)(2 * elem);     r[0].val[0] = wr;
                      r[0].col[1] = (int)(2 * j);        r[0].val[1] = -wr;
                      r[0].col[2] = (int)(2 * N + elem); r[0].val[2] = -wr * gx;
        r[0].r = wr * (dox - rx);
        r[1].nnz = 3; r[1].col[0] = (int)(2 * elem + 1); r[1].val[0] = wr;
                      r[1].col[1] = (int)(2 * j + 1);    r[1].val[1] = -wr;
                      r[1].col[2] = (int)(2 * N + elem); r[1].val[2] = -wr * gy;
        r[1].r = wr * (doy - ry);
        r += 2;
    }
    const int vfit = (C[2 * elem] >= 0.0f) && (C[2 * elem + 1] >= 0.0f) && (M[elem] == 0.0f);
    for (int c = 0; c < 2; ++c, ++r) {
        if (vfit) { r->nnz = 1; r->col[0] = (int)(2 * elem + c); r->val[0] = wf; r->r = wf * (O[2 * elem + c] - C[2 * elem + c]); }
        else      { r->nnz = 0; r->r = 0.0f; }
    }
    return 10;
}

/* E2: examples/arap_mesh_deformation/arap_mesh_deformation.t:1-21.
 * params: 0 &w_fitSqrt, 1 &w_regSqrt, 2 Position float3 (unknown), 3 Angle float3 (unknown),
 *         4 Original float3, 5 Constraints float3, 6 V0 int[E], 7 V1 int[E].   dims N,E.
 * flat layout: [Position 3*n+c][Angle 3*N+3*n+c].  groups: fit (N elems), reg (E elems). */
static void rot3_zyx(const float* a, float R[9])
{   /* lib.t:123-137 */
    const float ca = cosf(a[0]), cb = cosf(a[1]), cg = cosf(a[2]);
    const float sa = sinf(a[0]), sb = sinf(a[1]), sg = sinf(a[2]);
    R[0] = cg * cb;  R[1] = -sg * ca + cg * sb * sa;  R[2] = sg * sa + cg * sb * ca;
    R[3] = sg * cb;  R[4] = cg * ca + sg * sb * sa;   R[5] = -cg * sa + sg * sb * ca;
    R[6] = -sb;      R[7] = cb * sa;                  R[8] = cb * ca;
}
static void drot3_zyx(const float* a, float dA[9], float dB[9], float dG[9])
{
    const float ca = cosf(a[0]), cb = cosf(a[1]), cg = cosf(a[2]);
    const float sa = sinf(a[0]), sb = sinf(a[1]), sg = sinf(a[2]);
    /* d/d alpha */
    dA[0] = 0.0f;     dA[1] = sg * sa + cg * sb * ca;   dA[2] = sg * ca - cg * sb * sa;
    dA[3] = 0.0f;     dA[4] = -cg * sa + sg * sb * ca;  dA[5] = -cg * ca - sg * sb * sa;
    dA[6] = 0.0f;     dA[7] = cb * ca;                  dA[8] = -cb * sa;
    /* d/d beta */
    dB[0] = -cg * sb; dB[1] = cg * cb * sa;             dB[2] = cg * cb * ca;
    dB[3] = -sg * sb; dB[4] = sg * cb * sa;             dB[5] = sg * cb * ca;
    dB[6] = -cb;      dB[7] = -sb * sa;                 dB[8] = -sb * ca;
    /* d/d gamma */
    dG[0] = -sg * cb; dG[1] = -cg * ca - sg * sb * sa;  dG[2] = cg * sa - sg * sb * ca;
    dG[3] = cg * cb;  dG[4] = -sg * ca + cg * sb * sa;  dG[5] = sg * sa + cg * sb * ca;
    dG[6] = 0.0f;     dG[7] = 0.0f;                     dG[8] = 0.0f;
}
static inline void mv3(const float M[9], const float v[3], float o[3])
{
    o[0] = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    o[1] = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    o[2] = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
}
static int rows_arap(const OrcEnergy* e, long elem, OrcRow* out)
{
    const long N = e->dims[0];
    const float wf = *(const float*)e->params[0];
    const float wr = *(const float*)e->params[1];
    const float* P = img(e, 2); const float* A = img(e, 3); const float* Or = img(e, 4); const float* C = img(e, 5);
    if (elem < N) {
        const int valid = C[3 * elem] >= -999999.9f;
        for (int c = 0; c < 3; ++c) {
            if (valid) { out[c].nnz = 1; out[c].col[0] = (int)(3 * elem + c); out[c].val[0] = wf;
                         out[c].r = wf * (P[3 * elem + c] - C[3 * elem + c]); }
            else       { out[c].nnz = 0; out[c].r = 0.0f; }
        }
        return 3;
    }
    const long k = elem - N;
    const int v0 = ((const int*)e->params[6])[k], v1 = ((const int*)e->params[7])[k];
    float R[9], dA[9], dB[9], dG[9], dv[3], Rv[3], ga[3], gb[3], gg[3];
    for (int c = 0; c < 3; ++c) dv[c] = Or[3 * v0 + c] - Or[3 * v1 + c];
    rot3_zyx(&A[3 * v0], R); drot3_zyx(&A[3 * v0], dA, dB, dG);
    mv3(R, dv, Rv); mv3(dA, dv, ga); mv3(dB, dv, gb); mv3(dG, dv, gg);
    for (int c = 0; c < 3; ++c) {
        OrcRow* r = &out[c];
        r->nnz = 5;
        r->col[0] = 3 * v0 + c;               r->val[0] = wr;
        r->col[1] = 3 * v1 + c;               r->val[1] = -wr;
        r->col[2] = (int)(3 * N + 3 * v0);     r->val[2] = -wr * ga[c];
        r->col[3] = (int)(3 * N + 3 * v0 + 1); r->val[3] = -wr * gb[c];
        r->col[4] = (int)(3 * N + 3 * v0 + 2); r->val[4] = -wr * gg[c];
        r->r = wr * ((P[3 * v0 + c] - P[3 * v1 + c]) - Rv[c]);
    }
    return 3;
}

/* E4: examples/bundle_adjustment/bundle_adjustment.t:1-34 (Snavely reprojection error).
 * params: 0 cameras float9 (unknown: angle-axis 0-2, translation 3-5, focal 6, l1 7, l2 8), 1 points float3 (unknown),
 *         2 observations float2, 3 oToC int[O], 4 oToP int[O].  dims C,P,O.
 * flat layout: [cameras 9*c+k | points 9*C + 3*p + k].  One element = one observation = 2 rows x 12 nonzeros.
 * Derivatives by forward-mode AD over the expression of lib.t:514-555 (AngleAxisRotatePoint; the Select on
 * theta2 > 1e-8 differentiates the chosen branch, ad.t:800-809) -- the same partials the reference's symbolic
 * AD produces. */
typedef struct { float v; float d[12]; } Jet;
static inline Jet jc(float c) { Jet r; r.v = c; memset(r.d, 0, sizeof(r.d)); return r; }
static inline Jet jvar(float c, int k) { Jet r = jc(c); r.d[k] = 1.0f; return r; }
static inline Jet jadd(Jet a, Jet b) { Jet r; r.v = a.v + b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline Jet jsub(Jet a, Jet b) { Jet r; r.v = a.v - b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline Jet jmul(Jet a, Jet b) { Jet r; r.v = a.v * b.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static inline Jet jdiv(Jet a, Jet b) { Jet r; const float ib = 1.0f / b.v; r.v = a.v * ib; for (int i = 0; i < 12; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib; return r; }
static inline Jet jneg(Jet a) { Jet r; r.v = -a.v; for (int i = 0; i < 12; ++i) r.d[i] = -a.d[i]; return r; }
static inline Jet jsqrt(Jet a) { Jet r; r.v = sqrtf(a.v); const float k = 0.5f / r.v; for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * k; return r; }
static inline Jet jsin(Jet a) { Jet r; r.v = sinf(a.v); const float k = cosf(a.v); for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * k; return r; }
static inline Jet jcos(Jet a) { Jet r; r.v = cosf(a.v); const float k = -sinf(a.v); for (int i = 0; i < 12; ++i) r.d[i] = a.d[i] * k; return r; }

static void ba_residual_jets(const float* cam, const float* pt, const float* obs, Jet out[2])
{
    Jet c[9], X[3];
    for (int k = 0; k < 9; ++k) c[k] = jvar(cam[k], k);
    for (int k = 0; k < 3; ++k) X[k] = jvar(pt[k], 9 + k);
    Jet theta2 = jadd(jadd(jmul(c[0], c[0]), jmul(c[1], c[1])), jmul(c[2], c[2]));
    Jet p[3];
    if (theta2.v > 1e-8f) {
        Jet theta = jsqrt(theta2), ct = jcos(theta), st = jsin(theta), ti = jdiv(jc(1.0f), theta);
        Jet w[3] = { jmul(c[0], ti), jmul(c[1], ti), jmul(c[2], ti) };
        Jet wx[3] = { jsub(jmul(w[1], X[2]), jmul(w[2], X[1])), jsub(jmul(w[2], X[0]), jmul(w[0], X[2])), jsub(jmul(w[0], X[1]), jmul(w[1], X[0])) };
        Jet tmp = jmul(jadd(jadd(jmul(w[0], X[0]), jmul(w[1], X[1])), jmul(w[2], X[2])), jsub(jc(1.0f), ct));
        for (int k = 0; k < 3; ++k) p[k] = jadd(jadd(jmul(X[k], ct), jmul(wx[k], st)), jmul(w[k], tmp));
    } else {
        Jet wx[3] = { jsub(jmul(c[1], X[2]), jmul(c[2], X[1])), jsub(jmul(c[2], X[0]), jmul(c[0], X[2])), jsub(jmul(c[0], X[1]), jmul(c[1], X[0])) };
        for (int k = 0; k < 3; ++k) p[k] = jadd(X[k], wx[k]);
    }
    for (int k = 0; k < 3; ++k) p[k] = jadd(p[k], c[3 + k]);
    Jet cx = jdiv(jneg(p[0]), p[2]), cy = jdiv(jneg(p[1]), p[2]);
    Jet r2 = jadd(jmul(cx, cx), jmul(cy, cy));
    Jet dist = jadd(jc(1.0f), jmul(r2, jadd(c[7], jmul(c[8], r2))));
    Jet fd = jmul(c[6], dist);
    out[0] = jsub(jc(obs[0]), jmul(cx, fd));
    out[1] = jsub(jc(obs[1]), jmul(cy, fd));
}
static int rows_bundle(const OrcEnergy* e, long elem, OrcRow* out)
{
    const long C = e->dims[0];
    const float* cams = img(e, 0); const float* pts = img(e, 1); const float* obs = img(e, 2);
    const int ci = ((const int*)e->params[3])[elem], pi = ((const int*)e->params[4])[elem];
    Jet r[2];
    ba_residual_jets(&cams[9 * ci], &pts[3 * pi], &obs[2 * elem], r);
    for (int q = 0; q < 2; ++q) {
        out[q].nnz = 12; out[q].r = r[q].v;
        for (int k = 0; k < 9; ++k) { out[q].col[k] = 9 * ci + k; out[q].val[k] = r[q].d[k]; }
        for (int k = 0; k < 3; ++k) { out[q].col[9 + k] = (int)(9 * C + 3 * pi + k); out[q].val[9 + k] = r[q].d[9 + k]; }
    }
    return 2;
}

/* E3: examples/shape_from_shading/shape_from_shading.t:1-112.
 * params 0-15: host floats w_p, w_s, w_g (squared weights: the .t takes sqrt, :27), f_x, f_y, u_x, u_y, L_1..L_9;
 * 16 X float (unknown), 17 D_i float, 18 Im float, 19 edgeMaskR uint8, 20 edgeMaskC uint8.   dims W,H.
 * Guarded loads return 0 outside the image (thallo.t:876-883).  One element = one pixel = 6 rows:
 *   fit, shading_h, shading_v, reg.x, reg.y, reg.z.
 * B_I(c) = [D(c-ex)>0 & D(c)>0 & D(c-ey)>0] * (B(n(c)) - I(c)) depends on X(c), X(c-ex), X(c-ey); its three partials come
 * from 3-wide forward-mode AD (what the reference's gradient images of the computed array B_I_comp hold, :79-80).
 * `valid` of the reg term contains comparisons on X: zero derivative (ad.t:824-829). */
typedef struct { float v, d[3]; } J3;
static inline J3 k3(float c) { J3 r = { c, { 0, 0, 0 } }; return r; }
static inline J3 v3(float c, int k) { J3 r = k3(c); r.d[k] = 1.0f; return r; }
static inline J3 a3(J3 a, J3 b) { J3 r; r.v = a.v + b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline J3 s3(J3 a, J3 b) { J3 r; r.v = a.v - b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline J3 m3(J3 a, J3 b) { J3 r; r.v = a.v * b.v; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
static inline J3 c3(J3 a, float c) { J3 r; r.v = a.v * c; for (int i = 0; i < 3; ++i) r.d[i] = a.d[i] * c; return r; }

typedef struct { int W, H; const float *X, *D, *Im; const unsigned char *mR, *mC; float wp, ws, wg, fx, fy, ux, uy, L[9]; } Sfs;
static inline float sfs_at(const Sfs* q, const float* a, long x, long y) { return (x >= 0 && x < q->W && y >= 0 && y < q->H) ? a[y * q->W + x] : 0.0f; }
static void sfs_bind(const OrcEnergy* e, Sfs* q)
{
    q->W = (int)e->dims[0]; q->H = (int)e->dims[1];
    const float wp = *(const float*)e->params[0], ws = *(const float*)e->params[1], wg = *(const float*)e->params[2];
    q->wp = sqrtf(wp); q->ws = sqrtf(ws); q->wg = sqrtf(wg);
    q->fx = *(const float*)e->params[3]; q->fy = *(const float*)e->params[4]; q->ux = *(const float*)e->params[5]; q->uy = *(const float*)e->params[6];
    for (int k = 0; k < 9; ++k) q->L[k] = *(const float*)e->params[7 + k];
    q->X = img(e, 16); q->D = img(e, 17); q->Im = img(e, 18);
    q->mR = (const unsigned char*)e->params[19]; q->mC = (const unsigned char*)e->params[20];
}
/* B_I at pixel (x,y) with derivatives w.r.t. X(x,y) [0], X(x-1,y) [1], X(x,y-1) [2] */
static J3 sfs_BI(const Sfs* q, long x, long y)
{
    if (!(sfs_at(q, q->D, x - 1, y) > 0.0f && sfs_at(q, q->D, x, y) > 0.0f && sfs_at(q, q->D, x, y - 1) > 0.0f)) return k3(0.0f);
    const J3 c = v3(sfs_at(q, q->X, x, y), 0), l = v3(sfs_at(q, q->X, x - 1, y), 1), u = v3(sfs_at(q, q->X, x, y - 1), 2);
    const float i = (float)x, j = (float)y;
    const J3 nx = c3(m3(u, s3(c, l)), 1.0f / q->fy);                       /* shape_from_shading.t:47 */
    const J3 ny = c3(m3(l, s3(c, u)), 1.0f / q->fx);                       /* :48 */
    const J3 nz = s3(a3(c3(nx, (q->ux - i) / q->fx), c3(ny, (q->uy - j) / q->fy)), c3(m3(l, u), 1.0f / (q->fx * q->fy)));   /* :49 */
    const J3 sq = a3(a3(m3(nx, nx), m3(ny, ny)), m3(nz, nz));
    J3 inv;
    if (sq.v > 0.0f) { inv.v = 1.0f / sqrtf(sq.v); const float k = -0.5f * inv.v / sq.v; for (int t = 0; t < 3; ++t) inv.d[t] = k * sq.d[t]; }
    else inv = k3(1.0f);
    const J3 n0 = m3(inv, nx), n1 = m3(inv, ny), n2 = m3(inv, nz);
    const float* L = q->L;
    J3 B = k3(L[0]);
    B = a3(B, c3(n1, L[1])); B = a3(B, c3(n2, L[2])); B = a3(B, c3(n0, L[3]));
    B = a3(B, c3(m3(n0, n1), L[4])); B = a3(B, c3(m3(n1, n2), L[5]));
    B = a3(B, c3(a3(s3(c3(m3(n0, n0), -1.0f), m3(n1, n1)), c3(m3(n2, n2), 2.0f)), L[6]));
    B = a3(B, c3(m3(n2, n0), L[7])); B = a3(B, c3(s3(m3(n0, n0), m3(n1, n1)), L[8]));
    const float I = sfs_at(q, q->Im, x, y) * 0.5f + 0.25f * (sfs_at(q, q->Im, x - 1, y) + sfs_at(q, q->Im, x, y - 1));   /* :69 */
    return s3(B, k3(I));
}
static inline void row_add(OrcRow* r, long W, long H, long x, long y, float v)
{
    if (x < 0 || x >= W || y < 0 || y >= H || v == 0.0f) return;
    const int col = (int)(y * W + x);
    for (int k = 0; k < r->nnz; ++k) if (r->col[k] == col) { r->val[k] += v; return; }
    r->col[r->nnz] = col; r->val[r->nnz] = v; r->nnz++;
}
static int rows_sfs(const OrcEnergy* e, long elem, OrcRow* out)
{
    Sfs q; sfs_bind(e, &q);
    const long W = q.W, H = q.H, x = elem % W, y = elem / W;
    for (int k = 0; k < 6; ++k) { out[k].nnz = 0; out[k].r = 0.0f; }
    const float Xc = q.X[elem];
    /* fit (:86-87) */
    if (q.D[elem] > 0.0f) { out[0].r = q.wp * (Xc - q.D[elem]); row_add(&out[0], W, H, x, y, q.wp); }
    /* shading (:90-93): InBoundsExpanded(x,y,1) */
    if (x >= 1 && x + 1 < W && y >= 1 && y + 1 < H) {
        const J3 b0 = sfs_BI(&q, x, y), bx = sfs_BI(&q, x + 1, y), by = sfs_BI(&q, x, y + 1);
        const float h = q.wg * (float)q.mR[elem], v = q.wg * (float)q.mC[elem];
        out[1].r = h * (b0.v - bx.v);
        row_add(&out[1], W, H, x, y, h * b0.d[0]); row_add(&out[1], W, H, x - 1, y, h * b0.d[1]); row_add(&out[1], W, H, x, y - 1, h * b0.d[2]);
        row_add(&out[1], W, H, x + 1, y, -h * bx.d[0]); row_add(&out[1], W, H, x, y, -h * bx.d[1]); row_add(&out[1], W, H, x + 1, y - 1, -h * bx.d[2]);
        out[2].r = v * (b0.v - by.v);
        row_add(&out[2], W, H, x, y, v * b0.d[0]); row_add(&out[2], W, H, x - 1, y, v * b0.d[1]); row_add(&out[2], W, H, x, y - 1, v * b0.d[2]);
        row_add(&out[2], W, H, x, y + 1, -v * by.d[0]); row_add(&out[2], W, H, x - 1, y + 1, -v * by.d[1]); row_add(&out[2], W, H, x, y, -v * by.d[2]);
    }
    /* regularisation (:96-104) */
    static const int DX[4] = { -1, 0, 1, 0 }, DY[4] = { 0, -1, 0, 1 };
    int valid = q.D[elem] > 0.0f;
    for (int d = 0; d < 4 && valid; ++d) {
        if (!(sfs_at(&q, q.D, x + DX[d], y + DY[d]) > 0.0f)) valid = 0;
        else if (!(fabsf(Xc - sfs_at(&q, q.X, x + DX[d], y + DY[d])) < 0.01f)) valid = 0;
    }
    if (valid) {
        for (int c = 0; c < 3; ++c) {
            #define COEF(ix, iy) (c == 0 ? ((float)(ix) - q.ux) / q.fx : c == 1 ? ((float)(iy) - q.uy) / q.fy : 1.0f)
            float acc = 4.0f * (COEF(x, y) * Xc);
            row_add(&out[3 + c], W, H, x, y, q.ws * 4.0f * COEF(x, y));
            for (int d = 0; d < 4; ++d) {
                const long xn = x + DX[d], yn = y + DY[d];
                acc -= COEF(xn, yn) * sfs_at(&q, q.X, xn, yn);
                row_add(&out[3 + c], W, H, xn, yn, -q.ws * COEF(xn, yn));
            }
            #undef COEF
            out[3 + c].r = q.ws * acc;
        }
    }
    return 6;
}

int orc_energy_init(OrcEnergy* e, int kind, const unsigned* dims, void** params,
                    const float* fconst, const int* iconst)
{
    memset(e, 0, sizeof(*e));
    e->kind = kind; e->params = params;
    if (fconst) memcpy(e->fconst, fconst, sizeof(e->fconst));
    if (iconst) memcpy(e->iconst, iconst, sizeof(e->iconst));
    switch (kind) {
    case ORC_LAPLACIAN_IMAGE: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1];
        long N = (long)dims[0] * dims[1];
        e->n_img = 1; e->img_param[0] = 0; e->img_chan[0] = 1; e->img_count[0] = N;
        e->n_elems = N; e->rows = rows_lap_image; e->use_precond = 0;
        break; }
    case ORC_LAPLACIAN_GRAPH: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1];
        e->n_img = 1; e->img_param[0] = 0; e->img_chan[0] = 1; e->img_count[0] = dims[0];
        e->n_elems = (long)dims[0] + dims[1]; e->rows = rows_lap_graph; e->use_precond = 0;
        break; }
    case ORC_IMAGE_WARPING: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1];
        long N = (long)dims[0] * dims[1];
        e->n_img = 2;
        e->img_param[0] = 0; e->img_chan[0] = 2; e->img_count[0] = N;
        e->img_param[1] = 1; e->img_chan[1] = 1; e->img_count[1] = N;
        e->n_elems = N; e->rows = rows_image_warping; e->excluded = excl_image_warping;
        e->use_precond = 1;   /* image_warping.t:11 */
        break; }
    case ORC_ARAP_MESH: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1];
        e->n_img = 2;
        e->img_param[0] = 2; e->img_chan[0] = 3; e->img_count[0] = dims[0];
        e->img_param[1] = 3; e->img_chan[1] = 3; e->img_count[1] = dims[0];
        e->n_elems = (long)dims[0] + dims[1]; e->rows = rows_arap;
        e->use_precond = 1;   /* arap_mesh_deformation.t:13 */
        break; }
    case ORC_SFS: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1];
        e->n_img = 1; e->img_param[0] = 16; e->img_chan[0] = 1; e->img_count[0] = (long)dims[0] * dims[1];
        e->n_elems = (long)dims[0] * dims[1]; e->rows = rows_sfs;
        e->use_precond = 0;   /* shape_from_shading.t has no UsePreconditioner() */
        break; }
    case ORC_BUNDLE_ADJUST: {
        e->dims[0] = dims[0]; e->dims[1] = dims[1]; e->dims[2] = dims[2];
        e->n_img = 2;
        e->img_param[0] = 0; e->img_chan[0] = 9; e->img_count[0] = dims[0];
        e->img_param[1] = 1; e->img_chan[1] = 3; e->img_count[1] = dims[1];
        e->n_elems = dims[2]; e->rows = rows_bundle;
        e->use_precond = 1;   /* bundle_adjustment.t:9 */
        break; }
    default:
        return -1;
    }
    e->img_off[0] = 0;
    for (int i = 0; i < e->n_img; ++i) e->img_off[i + 1] = e->img_off[i] + e->img_count[i] * e->img_chan[i];
    e->n_unknowns = e->img_off[e->n_img];
    return 0;
}

/* ------------------------------------------------------------------ generic fmap */

/* Threaded mode (orc_set_threads(n > 1), double-accumulator sums only): the five row loops below run under OpenMP -- same rows,
 * same arithmetic per row, scatter-adds as float atomics (order-free like the reference's own atomics, util.t:40-50) and per-thread
 * double sums.  It exists so that the full-size configurations (2048^2, ladybug-1723) can be checked on the GPU box's host cores in
 * seconds; tests/test_oracle_golden.py pins it against the serial loops (which stay the bit-exact known-answer path). */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

double orc_cost(const OrcEnergy* e, int fl)
{   /* thallo.t:3939-3949: per element 0.5*sum r_k^2 ; gauss_newton.t:1067-1079 reduces over elements */
    if (g_threads > 1 && !fl) {
        double total = 0.0;
#pragma omp parallel num_threads(g_threads) reduction(+:total)
        {
            OrcRow rows[ORC_MAX_ROWS];
#pragma omp for schedule(static)
            for (long el = 0; el < e->n_elems; ++el) {
                int n = e->rows(e, el, rows);
                float s = 0.0f;
                for (int k = 0; k < n; ++k) s = s + rows[k].r * rows[k].r;
                total += (double)(0.5f * s);
            }
        }
        return total;
    }
    OrcRow rows[ORC_MAX_ROWS]; Acc a; acc_init(&a, fl);
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        float s = 0.0f;
        for (int k = 0; k < n; ++k) s = s + rows[k].r * rows[k].r;
        acc_add(&a, 0.5f * s);
    }
    return acc_get(&a);
}

void orc_eval_jtf(const OrcEnergy* e, float* r, float* pre)
{   /* thallo.t:3898-3902: R[u] += -1*partial*F ; Pre[u] += partial*partial */
    if (g_threads > 1) {
#pragma omp parallel num_threads(g_threads)
        {
            OrcRow rows[ORC_MAX_ROWS];
#pragma omp for schedule(static)
            for (long el = 0; el < e->n_elems; ++el) {
                int n = e->rows(e, el, rows);
                for (int k = 0; k < n; ++k)
                    for (int j = 0; j < rows[k].nnz; ++j) {
                        const float v = rows[k].val[j];
                        const float dr = -1.0f * v * rows[k].r, dp = v * v;
#pragma omp atomic
                        r[rows[k].col[j]] += dr;
#pragma omp atomic
                        pre[rows[k].col[j]] += dp;
                    }
            }
        }
        return;
    }
    OrcRow rows[ORC_MAX_ROWS];
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        for (int k = 0; k < n; ++k)
            for (int j = 0; j < rows[k].nnz; ++j) {
                const float v = rows[k].val[j];
                r[rows[k].col[j]]   += -1.0f * v * rows[k].r;
                pre[rows[k].col[j]] += v * v;
            }
    }
}

double orc_apply_jtj(const OrcEnergy* e, const float* p, float* Ap, int fl)
{   /* thallo.t:3551-3566: Jp = sum partial*P[u]; Ap_X[u] += Jp*partial; result += P[u]*Jp*partial */
    if (g_threads > 1 && !fl) {
        double total = 0.0;
#pragma omp parallel num_threads(g_threads) reduction(+:total)
        {
            OrcRow rows[ORC_MAX_ROWS];
#pragma omp for schedule(static)
            for (long el = 0; el < e->n_elems; ++el) {
                int n = e->rows(e, el, rows);
                float d = 0.0f;
                for (int k = 0; k < n; ++k) {
                    float Jp = 0.0f;
                    for (int j = 0; j < rows[k].nnz; ++j) Jp = Jp + rows[k].val[j] * p[rows[k].col[j]];
                    for (int j = 0; j < rows[k].nnz; ++j) {
                        const float jtjp = Jp * rows[k].val[j];
#pragma omp atomic
                        Ap[rows[k].col[j]] += jtjp;
                        d = d + p[rows[k].col[j]] * jtjp;
                    }
                }
                total += (double)d;
            }
        }
        return total;
    }
    OrcRow rows[ORC_MAX_ROWS]; Acc a; acc_init(&a, fl);
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        float d = 0.0f;
        for (int k = 0; k < n; ++k) {
            float Jp = 0.0f;
            for (int j = 0; j < rows[k].nnz; ++j) Jp = Jp + rows[k].val[j] * p[rows[k].col[j]];
            for (int j = 0; j < rows[k].nnz; ++j) {
                const float jtjp = Jp * rows[k].val[j];
                Ap[rows[k].col[j]] += jtjp;
                d = d + p[rows[k].col[j]] * jtjp;
            }
        }
        acc_add(&a, d);
    }
    return acc_get(&a);
}

void orc_compute_ctc(const OrcEnergy* e, float inv_radius, float* ctc)
{   /* thallo.t:3929-3933 */
    if (g_threads > 1) {
#pragma omp parallel num_threads(g_threads)
        {
            OrcRow rows[ORC_MAX_ROWS];
#pragma omp for schedule(static)
            for (long el = 0; el < e->n_elems; ++el) {
                int n = e->rows(e, el, rows);
                for (int k = 0; k < n; ++k)
                    for (int j = 0; j < rows[k].nnz; ++j) {
                        const float c = rows[k].val[j] * rows[k].val[j] * inv_radius;
#pragma omp atomic
                        ctc[rows[k].col[j]] += c;
                    }
            }
        }
        return;
    }
    OrcRow rows[ORC_MAX_ROWS];
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        for (int k = 0; k < n; ++k)
            for (int j = 0; j < rows[k].nnz; ++j)
                ctc[rows[k].col[j]] += rows[k].val[j] * rows[k].val[j] * inv_radius;
    }
}

double orc_model_cost(const OrcEnergy* e, const float* delta, int fl)
{   /* thallo.t:3848-3863: 0.5 * sum (F + J delta)^2 */
    if (g_threads > 1 && !fl) {
        double total = 0.0;
#pragma omp parallel num_threads(g_threads) reduction(+:total)
        {
            OrcRow rows[ORC_MAX_ROWS];
#pragma omp for schedule(static)
            for (long el = 0; el < e->n_elems; ++el) {
                int n = e->rows(e, el, rows);
                float s = 0.0f;
                for (int k = 0; k < n; ++k) {
                    float jd = 0.0f;
                    for (int j = 0; j < rows[k].nnz; ++j) jd = jd + rows[k].val[j] * delta[rows[k].col[j]];
                    const float m = rows[k].r + jd;
                    s = s + m * m;
                }
                total += (double)(0.5f * s);
            }
        }
        return total;
    }
    OrcRow rows[ORC_MAX_ROWS]; Acc a; acc_init(&a, fl);
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        float s = 0.0f;
        for (int k = 0; k < n; ++k) {
            float jd = 0.0f;
            for (int j = 0; j < rows[k].nnz; ++j) jd = jd + rows[k].val[j] * delta[rows[k].col[j]];
            const float m = rows[k].r + jd;
            s = s + m * m;
        }
        acc_add(&a, 0.5f * s);
    }
    return acc_get(&a);
}

long orc_count_rows(const OrcEnergy* e, long* nnz_out)
{
    OrcRow rows[ORC_MAX_ROWS]; long nr = 0, nnz = 0;
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        nr += n;
        for (int k = 0; k < n; ++k) nnz += rows[k].nnz;
    }
    if (nnz_out) *nnz_out = nnz;
    return nr;
}

long orc_export_csr(const OrcEnergy* e, int* rowptr, int* colind, float* vals, float* resid)
{   /* the matrix gauss_newton.t:327-487 (generateDumpJ) materialises, rows in element order */
    OrcRow rows[ORC_MAX_ROWS]; long nr = 0, nnz = 0;
    for (long el = 0; el < e->n_elems; ++el) {
        int n = e->rows(e, el, rows);
        for (int k = 0; k < n; ++k) {
            rowptr[nr] = (int)nnz;
            if (resid) resid[nr] = rows[k].r;
            for (int j = 0; j < rows[k].nnz; ++j) { colind[nnz] = rows[k].col[j]; vals[nnz] = rows[k].val[j]; ++nnz; }
            ++nr;
        }
    }
    rowptr[nr] = (int)nnz;
    return nnz;
}

/* ------------------------------------------------------------------ driver */

static inline int is_excl(const OrcEnergy* e, long i) { return e->excluded ? e->excluded(e, i) : 0; }

static void linear_update(OrcEnergy* e, const float* delta)
{   /* gauss_newton.t:901-906 */
    for (int k = 0; k < e->n_img; ++k) {
        float* X = (float*)e->params[e->img_param[k]];
        const long off = e->img_off[k], len = e->img_off[k + 1] - off;
        for (long i = 0; i < len; ++i)
            if (!is_excl(e, off + i)) X[i] = X[i] + delta[off + i];
    }
}
static void copy_unknowns(OrcEnergy* e, float* dst_flat, int to_flat)
{
    for (int k = 0; k < e->n_img; ++k) {
        float* X = (float*)e->params[e->img_param[k]];
        const long off = e->img_off[k], len = e->img_off[k + 1] - off;
        for (long i = 0; i < len; ++i)
            if (!is_excl(e, off + i)) { if (to_flat) dst_flat[off + i] = X[i]; else X[i] = dst_flat[off + i]; }
    }
}

/* PCG iterations each Gauss-Newton / LM step of the last orc_solve ran (the zeta test of :1666-1686 ends the LM loop early): what the tests compare the
 * device-side early exit with */
static int g_pcg_counts[1024], g_pcg_steps = 0;
/* the LM trust region as the last solve left it (radius, radius_decrease_factor): what a caller needs to CONTINUE a trajectory step by step from the state the oracle reached
 * (tools/single_step_parity.py: every step of the device solver started from the oracle's state) */
static float g_last_radius = 0.0f, g_last_decrease = 0.0f;
void orc_last_trust_region(float* radius, float* decrease_factor) { if (radius) *radius = g_last_radius; if (decrease_factor) *decrease_factor = g_last_decrease; }
int orc_last_pcg_counts(int* out, int cap)
{
    const int n = g_pcg_steps < cap ? g_pcg_steps : cap;
    for (int i = 0; i < n; ++i) out[i] = g_pcg_counts[i];
    return g_pcg_steps;
}

int orc_solve(OrcEnergy* e, const OrcSolverParams* sp, double* costs, int costs_cap, float* trace, int trace_cap)
{
    g_pcg_steps = 0;
    const long n = e->n_unknowns;
    const int fl = sp->float_sums;
    const int lm = sp->use_lm;
    float* delta = calloc(n, 4); float* r = calloc(n, 4); float* z = calloc(n, 4); float* p = calloc(n, 4);
    float* Ap = calloc(n, 4); float* pre = calloc(n, 4);
    float* b = calloc(n, 4); float* Adelta = calloc(n, 4); float* CtC = calloc(n, 4); float* SSq = calloc(n, 4);
    float* prevX = calloc(n, 4);
    int ntrace = 0, ncost = 0, nIter = 0;
    float radius = sp->trust_region_radius, decrease_factor = sp->radius_decrease_factor;

    float prevCost = (float)orc_cost(e, fl);                   /* init, gauss_newton.t:1193 */
    if (ncost < costs_cap) costs[ncost++] = prevCost;

    while (nIter < sp->nIterations) {
        /* ---- Nonlinear Setup (gauss_newton.t:1566-1606) */
        memset(delta, 0, n * 4); memset(Ap, 0, n * 4); memset(r, 0, n * 4); memset(pre, 0, n * 4);
        orc_eval_jtf(e, r, pre);                                /* PCGInit1 residual-wise :998-1003 */
        Acc aN; acc_init(&aN, fl);
        for (long i = 0; i < n; ++i) {                          /* PCGInit1_Finish :712-731 */
            if (is_excl(e, i)) continue;
            float m;
            if (e->use_precond) { float s = 1.0f + sqrtf(pre[i]); m = 1.0f / (s * s); }   /* guardedInvert CERES :638-648 */
            else m = 1.0f;
            pre[i] = m; p[i] = m * r[i];
            acc_add(&aN, r[i] * p[i]);
        }
        float alphaN = (float)acc_get(&aN);
        float Q0 = 0.0f, Q1 = 0.0f;
        if (lm) {                                               /* :1595-1606 */
            if (nIter == 0) for (long i = 0; i < n; ++i) if (!is_excl(e, i)) SSq[i] = pre[i];   /* PCGSaveSSq */
            memset(CtC, 0, n * 4);
            orc_compute_ctc(e, 1.0f / radius, CtC);
            Acc aQ; acc_init(&aN, fl); acc_init(&aQ, fl);
            for (long i = 0; i < n; ++i) {                      /* PCGFinalizeDiagonal :936-969 */
                if (is_excl(e, i)) continue;
                const float unclamped = CtC[i];
                const float invS = 1.0f / SSq[i];
                const float cm = invS / radius;
                const float lo = sp->min_lm_diagonal * cm, hi = sp->max_lm_diagonal * cm;
                const float c = fminf(fmaxf(unclamped, lo), hi);
                CtC[i] = c;
                const float m = 1.0f / (c + radius * unclamped);
                pre[i] = m; b[i] = r[i]; p[i] = m * r[i];
                acc_add(&aN, r[i] * p[i]);
                acc_add(&aQ, 0.5f * (delta[i] * (r[i] + r[i])));
            }
            alphaN = (float)acc_get(&aN); Q0 = (float)acc_get(&aQ);
        }
        /* ---- Linear Solve (:1615-1687) */
        int pcg_done = 0;
        for (int lIter = 0; lIter < sp->lIterations; ++lIter) {
            pcg_done = lIter + 1;
            memset(Ap, 0, n * 4);
            double dd = orc_apply_jtj(e, p, Ap, fl);            /* PCGStep1 residual-wise :1006-1015 */
            float alphaD;
            if (lm) {                                           /* PCGStep1_Finish :777-787 */
                Acc aD; acc_init(&aD, fl);
                for (long i = 0; i < n; ++i) { if (is_excl(e, i)) continue; Ap[i] = Ap[i] + CtC[i] * p[i]; acc_add(&aD, p[i] * Ap[i]); }
                alphaD = (float)acc_get(&aD);
            } else alphaD = (float)dd;
            float alpha = 0.0f;                                  /* safeDivideIfNotLM :226-234 */
            if (lm) alpha = alphaN / alphaD; else if (alphaD != 0.0f) alpha = alphaN / alphaD;
            Acc aB, aQ; acc_init(&aB, fl); acc_init(&aQ, fl);
            if (lm && ((lIter + 1) % sp->residual_reset_period) == 0) {   /* :1653-1657 */
                for (long i = 0; i < n; ++i) if (!is_excl(e, i)) delta[i] = delta[i] + alpha * p[i];
                memset(Adelta, 0, n * 4);
                orc_apply_jtj(e, delta, Adelta, fl);
                for (long i = 0; i < n; ++i) if (!is_excl(e, i)) Adelta[i] += delta[i] * CtC[i];
                for (long i = 0; i < n; ++i) {
                    if (is_excl(e, i)) continue;
                    r[i] = b[i] - Adelta[i];
                    z[i] = pre[i] * r[i];
                    acc_add(&aB, z[i] * r[i]);
                    acc_add(&aQ, 0.5f * (delta[i] * (r[i] + b[i])));
                }
            } else {
                for (long i = 0; i < n; ++i) {                  /* PCGStep2 :801-843 */
                    if (is_excl(e, i)) continue;
                    delta[i] = delta[i] + alpha * p[i];
                    r[i] = r[i] - alpha * Ap[i];
                    z[i] = pre[i] * r[i];
                    acc_add(&aB, z[i] * r[i]);
                    if (lm) acc_add(&aQ, 0.5f * (delta[i] * (r[i] + b[i])));
                }
            }
            const float betaN = (float)acc_get(&aB);
            float beta = 0.0f;                                   /* PCGStep3 :889-899 */
            if (lm) beta = betaN / alphaN; else if (alphaN != 0.0f) beta = betaN / alphaN;
            for (long i = 0; i < n; ++i) if (!is_excl(e, i)) p[i] = z[i] + beta * p[i];
            if (trace && ntrace < trace_cap) { trace[2 * ntrace] = alpha; trace[2 * ntrace + 1] = beta; ++ntrace; }
            alphaN = betaN;                                      /* :1665 */
            if (lm) {                                            /* :1666-1686 */
                Q1 = (float)acc_get(&aQ);
                if (!isfinite(Q1)) break;
                const float zeta = (float)(lIter + 1) * (Q1 - Q0) / Q1;
                if (!isfinite(zeta)) break;
                if (zeta < sp->q_tolerance) break;
                Q0 = Q1;
            }
        }
        if (g_pcg_steps < 1024) g_pcg_counts[g_pcg_steps++] = pcg_done;
        /* ---- Nonlinear Finish (:1690-1765) */
        float model_cost_change = 0.0f;
        if (lm) {
            const float mc = (float)orc_model_cost(e, delta, fl);
            model_cost_change = prevCost - mc;
            copy_unknowns(e, prevX, 1);
        }
        linear_update(e, delta);
        if (lm) {
            const float newCost = (float)orc_cost(e, fl);
            const float cost_change = prevCost - newCost;
            const float rel = cost_change / model_cost_change;
            if (cost_change >= 0 && rel > sp->min_relative_decrease) {
                if (cost_change <= prevCost * sp->function_tolerance) {
                    if (ncost < costs_cap) costs[ncost++] = newCost;
                    ++nIter;  /* counted as a taken step in the returned trajectory */
                    goto done;
                }
                const double tmp = 1.0 - pow(2.0 * (double)rel - 1.0, 3.0);
                radius = (float)((double)radius / fmax(1.0 / 3.0, tmp));
                radius = fminf(radius, sp->max_trust_region_radius);
                decrease_factor = 2.0f;
                prevCost = newCost;
            } else {
                copy_unknowns(e, prevX, 0);                      /* revertUpdate */
                radius = radius / decrease_factor;
                decrease_factor = 2.0f * decrease_factor;
                if (radius < sp->min_trust_region_radius) {
                    if (ncost < costs_cap) costs[ncost++] = prevCost;
                    ++nIter;
                    goto done;
                }
            }
            if (ncost < costs_cap) costs[ncost++] = (float)orc_cost(e, fl);
        } else {
            /* GN: cost after the step, as Thallo_ProblemCurrentCost reports it (:1787-1793) */
            if (ncost < costs_cap) costs[ncost++] = (float)orc_cost(e, fl);
        }
        ++nIter;
    }
done:
    g_last_radius = radius; g_last_decrease = decrease_factor;
    free(delta); free(r); free(z); free(p); free(Ap); free(pre);
    free(b); free(Adelta); free(CtC); free(SSq); free(prevX);
    return nIter;
}

int orc_solve_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst,
                   const OrcSolverParams* sp, double* costs, int costs_cap, float* trace, int trace_cap)
{
    OrcEnergy e;
    if (orc_energy_init(&e, kind, dims, params, fconst, iconst)) return -1;
    return orc_solve(&e, sp, costs, costs_cap, trace, trace_cap);
}

/* thin wrappers so ctypes callers need not mirror the OrcEnergy struct */
#define WRAP_PROLOGUE OrcEnergy e; if (orc_energy_init(&e, kind, dims, params, fconst, iconst)) return -1;
double orc_cost_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst, int fl)
{ WRAP_PROLOGUE return orc_cost(&e, fl); }
long orc_n_unknowns_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst)
{ WRAP_PROLOGUE return e.n_unknowns; }
int orc_eval_jtf_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst, float* r, float* pre)
{ WRAP_PROLOGUE orc_eval_jtf(&e, r, pre); return 0; }
double orc_apply_jtj_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst,
                          const float* p, float* Ap, int fl)
{ WRAP_PROLOGUE return orc_apply_jtj(&e, p, Ap, fl); }
long orc_count_rows_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst, long* nnz)
{ WRAP_PROLOGUE return orc_count_rows(&e, nnz); }
long orc_export_csr_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst,
                         int* rowptr, int* colind, float* vals, float* resid)
{ WRAP_PROLOGUE return orc_export_csr(&e, rowptr, colind, vals, resid); }
int orc_excluded_mask_kind(int kind, const unsigned* dims, void** params, const float* fconst, const int* iconst, unsigned char* mask)
{ WRAP_PROLOGUE for (long i = 0; i < e.n_unknowns; ++i) mask[i] = (unsigned char)is_excl(&e, i); return 0; }
