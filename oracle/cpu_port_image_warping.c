/*
 * cpu_port_image_warping.c -- OpenMP CPU port of the GN + PCG loop for image_warping, used ONLY as
 * bench.py's `cpu_baseline` ("kind": "port") and checked against the row-form oracle in tests.
 * TEST / BASELINE INFRASTRUCTURE: the product never links this.
 *
 * Why a port: the reference ships no runnable CPU path in this snapshot (no Ceres source, cpuOnly
 * needs Terra), so the baseline is the same algorithm -- reference recurrences
 * gauss_newton.t:678-752,801-843,889-906 with the energy of image_warping.t:17-31 in gather
 * (unknown-wise) form, float32, dot products accumulated in double -- threaded over image rows.
 */
#define _GNU_SOURCE
#include <math.h>
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
    int W, H; long N;
    const float *U, *C, *M;
    float wf, wr;
    float *cs; unsigned char* fl;
} IW;

static inline int act(const IW* q, long x, long y) { return x >= 0 && x < q->W && y >= 0 && y < q->H && (q->fl[y * q->W + x] & 1); }

static double iw_cost(const IW* q, const float* O, const float* A)
{
    double acc = 0.0;
    static const int DX[4] = { 1, -1, 0, 0 }, DY[4] = { 0, 0, 1, -1 };
#pragma omp parallel for reduction(+:acc) schedule(static)
    for (long y = 0; y < q->H; ++y)
        for (long x = 0; x < q->W; ++x) {
            const long i = y * q->W + x;
            if (q->M[i] != 0.0f) continue;
            const float ci = cosf(A[i]), si = sinf(A[i]);
            float s = 0.0f;
            for (int d = 0; d < 4; ++d) {
                const long xn = x + DX[d], yn = y + DY[d];
                if (xn < 0 || xn >= q->W || yn < 0 || yn >= q->H) continue;
                const long j = yn * q->W + xn;
                if (q->M[j] != 0.0f) continue;
                const float dux = q->U[2 * i] - q->U[2 * j], duy = q->U[2 * i + 1] - q->U[2 * j + 1];
                const float ex = q->wr * ((O[2 * i] - O[2 * j]) - (ci * dux - si * duy));
                const float ey = q->wr * ((O[2 * i + 1] - O[2 * j + 1]) - (si * dux + ci * duy));
                s += ex * ex + ey * ey;
            }
            if (q->C[2 * i] >= 0.0f && q->C[2 * i + 1] >= 0.0f) {
                const float fx = q->wf * (O[2 * i] - q->C[2 * i]), fy = q->wf * (O[2 * i + 1] - q->C[2 * i + 1]);
                s += fx * fx + fy * fy;
            }
            acc += 0.5f * s;
        }
    return acc;
}

/* evalJTF + PCGInit1_Finish: r, pre(inverted), p = pre*r ; returns alphaN */
static double iw_init(IW* q, const float* O, const float* A, float* r, float* pre, float* p, float* delta)
{
    const long N = q->N; const int W = q->W;
    const float wr2 = q->wr * q->wr, wf2 = q->wf * q->wf;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) { q->cs[2 * i] = cosf(A[i]); q->cs[2 * i + 1] = sinf(A[i]); q->fl[i] = q->M[i] == 0.0f; }
    double aN = 0.0;
    static const int DX[4] = { 1, -1, 0, 0 }, DY[4] = { 0, 0, 1, -1 };
#pragma omp parallel for reduction(+:aN) schedule(static)
    for (long y = 0; y < q->H; ++y)
        for (long x = 0; x < W; ++x) {
            const long i = y * W + x;
            float rx = 0, ry = 0, ra = 0, mx = 0, ma = 0;
            if (q->fl[i] & 1) {
                const float ci = q->cs[2 * i], si = q->cs[2 * i + 1];
                float jx = 0, jy = 0, ja = 0, dgo = 0, dga = 0;
                for (int d = 0; d < 4; ++d) {
                    if (!act(q, x + DX[d], y + DY[d])) continue;
                    const long j = (y + DY[d]) * W + x + DX[d];
                    const float dux = q->U[2 * i] - q->U[2 * j], duy = q->U[2 * i + 1] - q->U[2 * j + 1];
                    const float dox = O[2 * i] - O[2 * j], doy = O[2 * i + 1] - O[2 * j + 1];
                    const float eix = dox - (ci * dux - si * duy), eiy = doy - (si * dux + ci * duy);
                    const float cj = q->cs[2 * j], sj = q->cs[2 * j + 1];
                    const float ejx = -dox + (cj * dux - sj * duy), ejy = -doy + (sj * dux + cj * duy);
                    const float gix = -si * dux - ci * duy, giy = ci * dux - si * duy;
                    jx += eix - ejx; jy += eiy - ejy; ja -= gix * eix + giy * eiy;
                    dgo += 2.0f; dga += gix * gix + giy * giy;
                }
                jx *= wr2; jy *= wr2; ja *= wr2; dgo *= wr2; dga *= wr2;
                if (q->C[2 * i] >= 0.0f && q->C[2 * i + 1] >= 0.0f) {
                    q->fl[i] |= 2; jx += wf2 * (O[2 * i] - q->C[2 * i]); jy += wf2 * (O[2 * i + 1] - q->C[2 * i + 1]); dgo += wf2;
                }
                rx = -jx; ry = -jy; ra = -ja;
                float s = 1.0f + sqrtf(dgo); mx = 1.0f / (s * s);
                s = 1.0f + sqrtf(dga); ma = 1.0f / (s * s);
            }
            r[2 * i] = rx; r[2 * i + 1] = ry; r[2 * N + i] = ra;
            pre[2 * i] = mx; pre[2 * i + 1] = mx; pre[2 * N + i] = ma;
            p[2 * i] = mx * rx; p[2 * i + 1] = mx * ry; p[2 * N + i] = ma * ra;
            delta[2 * i] = delta[2 * i + 1] = delta[2 * N + i] = 0.0f;
            aN += (double)(rx * p[2 * i] + ry * p[2 * i + 1] + ra * p[2 * N + i]);
        }
    return aN;
}

static double iw_apply(const IW* q, const float* p, float* Ap)
{
    const long N = q->N; const int W = q->W;
    const float wr2 = q->wr * q->wr, wf2 = q->wf * q->wf;
    double aD = 0.0;
    static const int DX[4] = { 1, -1, 0, 0 }, DY[4] = { 0, 0, 1, -1 };
#pragma omp parallel for reduction(+:aD) schedule(static)
    for (long y = 0; y < q->H; ++y)
        for (long x = 0; x < W; ++x) {
            const long i = y * W + x;
            float ax = 0, ay = 0, aa = 0;
            const float pxi = p[2 * i], pyi = p[2 * i + 1], pai = p[2 * N + i];
            if (q->fl[i] & 1) {
                const float ci = q->cs[2 * i], si = q->cs[2 * i + 1];
                for (int d = 0; d < 4; ++d) {
                    if (!act(q, x + DX[d], y + DY[d])) continue;
                    const long j = (y + DY[d]) * W + x + DX[d];
                    const float dux = q->U[2 * i] - q->U[2 * j], duy = q->U[2 * i + 1] - q->U[2 * j + 1];
                    const float gix = -si * dux - ci * duy, giy = ci * dux - si * duy;
                    const float cj = q->cs[2 * j], sj = q->cs[2 * j + 1], paj = p[2 * N + j];
                    const float gjx = sj * dux + cj * duy, gjy = -cj * dux + sj * duy;
                    const float dpx = pxi - p[2 * j], dpy = pyi - p[2 * j + 1];
                    const float ex = dpx - gix * pai, ey = dpy - giy * pai;
                    ax += dpx + ex + gjx * paj; ay += dpy + ey + gjy * paj; aa -= gix * ex + giy * ey;
                }
                ax *= wr2; ay *= wr2; aa *= wr2;
                if (q->fl[i] & 2) { ax += wf2 * pxi; ay += wf2 * pyi; }
            }
            Ap[2 * i] = ax; Ap[2 * i + 1] = ay; Ap[2 * N + i] = aa;
            aD += (double)(pxi * ax + pyi * ay + pai * aa);
        }
    return aD;
}

/* Runs nIterations GN steps of lIterations PCG iterations each, in place.  costs[0..nIterations].
 * seconds_pcg (optional) receives the wall time spent inside the PCG loops only.
 * Returns the number of threads used. */
int orc_cpu_port_image_warping2(int W, int H, float* O, float* A, const float* U, const float* C, const float* M,
                                float w_fit, float w_reg, int nIterations, int lIterations,
                                double* costs, double* seconds_pcg, double* seconds_total, float* ab_trace, int trace_cap);
int orc_cpu_port_image_warping(int W, int H, float* O, float* A, const float* U, const float* C, const float* M,
                               float w_fit, float w_reg, int nIterations, int lIterations,
                               double* costs, double* seconds_pcg, double* seconds_total)
{
    return orc_cpu_port_image_warping2(W, H, O, A, U, C, M, w_fit, w_reg, nIterations, lIterations, costs, seconds_pcg, seconds_total, 0, 0);
}

/* Thread placement for the baseline measurement: thread t of the team is pinned to the t-th CPU this process may run on (explicit
 * sched_setaffinity -- OMP_PROC_BIND is read when libgomp is loaded, which in bench.py happened long before), and every vector is FIRST TOUCHED by
 * the thread that will stream it (static schedule, same partition as the PCG loops), so that on a multi-socket host the pages sit next to their
 * threads.  pin = 0 undoes the pinning. */
static int g_pinned = 0;
static cpu_set_t g_all;            /* the mask the process had before the first pinning: what pin = 0 restores (NOT the caller's current mask -- the
                                      caller is thread 0 of the team and is itself pinned to one CPU at that point) */
static int g_have_all = 0;
static void pin_threads(int pin)
{
    if (!g_have_all) {
        CPU_ZERO(&g_all);
        if (sched_getaffinity(0, sizeof(g_all), &g_all) != 0) return;
        g_have_all = 1;
    }
    int cpus[CPU_SETSIZE], nc = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &g_all)) cpus[nc++] = c;
    if (nc == 0) return;
    int ok = 1;
#pragma omp parallel reduction(&&:ok)
    {
        cpu_set_t one;
        if (pin) { CPU_ZERO(&one); CPU_SET(cpus[omp_get_thread_num() % nc], &one); } else one = g_all;
        ok = sched_setaffinity(0, sizeof(one), &one) == 0;
    }
    g_pinned = pin && ok;
}
int orc_cpu_port_threads_pinned(void) { return g_pinned; }
static float* alloc_touch(long n, const float* src)
{
    float* v = malloc((size_t)n * sizeof(float));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) v[i] = src ? src[i] : 0.0f;
    return v;
}

/* ... and (ab_trace != NULL) alpha_k, beta_k of the first trace_cap PCG iterations overall: ab_trace[2j], ab_trace[2j+1] */
int orc_cpu_port_image_warping2(int W, int H, float* O_io, float* A_io, const float* U_in, const float* C_in, const float* M_in,
                                float w_fit, float w_reg, int nIterations, int lIterations,
                                double* costs, double* seconds_pcg, double* seconds_total, float* ab_trace, int trace_cap)
{
    const long N = (long)W * H, n = 3 * N;
    pin_threads(1);
    float *O = alloc_touch(2 * N, O_io), *A = alloc_touch(N, A_io);
    float *U = alloc_touch(2 * N, U_in), *C = alloc_touch(2 * N, C_in), *M = alloc_touch(N, M_in);
    IW q; q.W = W; q.H = H; q.N = N; q.U = U; q.C = C; q.M = M; q.wf = w_fit; q.wr = w_reg;
    q.cs = alloc_touch(2 * N, 0); q.fl = malloc(N);
    float *r = alloc_touch(n, 0), *pre = alloc_touch(n, 0), *p = alloc_touch(n, 0), *delta = alloc_touch(n, 0), *Ap = alloc_touch(n, 0), *z = alloc_touch(n, 0);
    double t_pcg = 0.0; const double t0 = omp_get_wtime();
    if (costs) costs[0] = (float)iw_cost(&q, O, A);
    for (int it = 0; it < nIterations; ++it) {
        float aN = (float)iw_init(&q, O, A, r, pre, p, delta);
        const double tp = omp_get_wtime();
        for (int k = 0; k < lIterations; ++k) {
            const float aD = (float)iw_apply(&q, p, Ap);
            const float alpha = aD != 0.0f ? aN / aD : 0.0f;
            double bN = 0.0;
#pragma omp parallel for reduction(+:bN) schedule(static)
            for (long i = 0; i < n; ++i) {
                delta[i] += alpha * p[i];
                const float rr = r[i] - alpha * Ap[i];
                r[i] = rr; const float zz = pre[i] * rr; z[i] = zz;
                bN += (double)(zz * rr);
            }
            const float beta = aN != 0.0f ? (float)bN / aN : 0.0f;
            { const long j = (long)it * lIterations + k; if (ab_trace && j < trace_cap) { ab_trace[2 * j] = alpha; ab_trace[2 * j + 1] = beta; } }
#pragma omp parallel for schedule(static)
            for (long i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
            aN = (float)bN;
        }
        t_pcg += omp_get_wtime() - tp;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N; ++i) {
            O[2 * i] += delta[2 * i]; O[2 * i + 1] += delta[2 * i + 1]; A[i] += delta[2 * N + i];
        }
        if (costs) costs[it + 1] = (float)iw_cost(&q, O, A);
    }
    if (seconds_pcg) *seconds_pcg = t_pcg;
    if (seconds_total) *seconds_total = omp_get_wtime() - t0;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) { O_io[2 * i] = O[2 * i]; O_io[2 * i + 1] = O[2 * i + 1]; A_io[i] = A[i]; }
    free(q.cs); free(q.fl); free(r); free(pre); free(p); free(delta); free(Ap); free(z); free(O); free(A); free(U); free(C); free(M);
    const int pinned = g_pinned;
    pin_threads(0);
    g_pinned = pinned;                    /* what the run that just ended had */
    return omp_get_max_threads();
}
