// sfs_pair_device.hpp -- what shape_from_shading's pixel-pair kernels share (energy_sfs_pair.hip: one launch per PCG iteration; energy_sfs_resident.hip: the whole PCG
// loop of a GN step in one launch): raw-buffer access, the x neighbours of a pixel pair, the flags of a pair, the camera / weights block.  Internal to csrc/.
#pragma once
#include <stdlib.h>
#include <stdint.h>
#include "iw_march.hpp"

namespace {

using namespace thallo;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffffu, 0x00020000); }
__device__ __forceinline__ u32x2 bld2(rsrc_t r, unsigned vo, unsigned so) { return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0); }
__device__ __forceinline__ void bst2(rsrc_t r, unsigned vo, unsigned so, v2f a) { u32x2 v; v.x = __float_as_uint(a.x); v.y = __float_as_uint(a.y); __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, 0); }
__device__ __forceinline__ void bst2u(rsrc_t r, unsigned vo, unsigned so, unsigned a, unsigned b) { u32x2 v; v.x = a; v.y = b; __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, 0); }
__device__ __forceinline__ v2f f2(u32x2 u) { return v2f{ __uint_as_float(u.x), __uint_as_float(u.y) }; }
__device__ __forceinline__ void take2u(u32x2& d, const u32x2& s) { unsigned long long a; take_pair(a, __builtin_bit_cast(unsigned long long, s)); d = __builtin_bit_cast(u32x2, a); }
inline int check_launch() { hipError_t e = hipGetLastError(); return e == hipSuccess ? 0 : -(int)e; }

// x neighbours of a pixel pair (every lane active; lanes 0 / 63 read 0 from outside the wave: they produce no output)
__device__ __forceinline__ v2f nbL(v2f v) { return v2f{ from_left(v.y), v.x }; }
__device__ __forceinline__ v2f nbR(v2f v) { return v2f{ v.y, from_right(v.x) }; }
struct M2 { bool x, y; };
__device__ __forceinline__ v2f sel(M2 m, v2f a, v2f b) { return v2f{ m.x ? a.x : b.x, m.y ? a.y : b.y }; }
__device__ __forceinline__ v2f sel(bool m, v2f a, v2f b) { return v2f{ m ? a.x : b.x, m ? a.y : b.y }; }
__device__ __forceinline__ v2f splat(float a) { return v2f{ a, a }; }
// the flags of a pair: pixel 0 in bits 0-7, pixel 1 in bits 8-15
__device__ __forceinline__ unsigned fl_left(unsigned f)  { return (((unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x138, 0xf, 0xf, true) >> 8) & 0xffu) | ((f & 0xffu) << 8); }
__device__ __forceinline__ unsigned fl_right(unsigned f) { return ((f >> 8) & 0xffu) | (((unsigned)__builtin_amdgcn_mov_dpp((int)f, 0x130, 0xf, 0xf, true) & 0xffu) << 8); }
__device__ __forceinline__ M2 bit(unsigned f, unsigned b) { return M2{ (f & b) != 0u, (f & (b << 8)) != 0u }; }

constexpr int PM_USE = 124, PM_NT = 256;
struct PCam { float wp, ws, wg, fx, fy, ux, uy; float L[9]; };
__device__ __forceinline__ float coef0(const PCam& cm, int x) { return ((float)x - cm.ux) / cm.fx; }
__device__ __forceinline__ float coef1(const PCam& cm, int y) { return ((float)y - cm.uy) / cm.fy; }
inline PCam cam_of(const float* hp)
{
    PCam c; c.wp = sqrtf(hp[0]); c.ws = sqrtf(hp[1]); c.wg = sqrtf(hp[2]); c.fx = hp[3]; c.fy = hp[4]; c.ux = hp[5]; c.uy = hp[6];
    for (int k = 0; k < 9; ++k) c.L[k] = hp[7 + k];
    return c;
}

// ------------------------------------------------------------------------------------------ precompute on pixel pairs, closed-form partials
// BI(c) = B(n^(X(c), X(c-ex), X(c-ey))) - I(c) and its three partials (shape_from_shading.t:40-80).  With c, l, u the three depths,
//   n = ( u (c-l) / f_y,  l (c-u) / f_x,  n_x a_x + n_y a_y - l u / (f_x f_y) ),   a_x = (u_x - x) / f_x, a_y = (u_y - y) / f_y,   n^ = n / |n|,
//   B = L1 + L2 n^_y + L3 n^_z + L4 n^_x + L5 n^_x n^_y + L6 n^_y n^_z + L7 (-n^_x^2 - n^_y^2 + 2 n^_z^2) + L8 n^_z n^_x + L9 (n^_x^2 - n^_y^2)
// the chain rule collapses to ONE 3-vector:  h = |n|^-1 (g - n^ (g . n^)),  g = grad_n^ B;  dB/dq = h . dn/dq  for q in {c, l, u}, and dn/dq are the products above with
// one factor removed.  Values go through eval_BI_vals' operations in its order (energy_sfs.hip); the partials agree with its forward-mode duals to rounding.
struct BIv { v2f b, dc, dl, du; };
__device__ __forceinline__ BIv eval_BI_pair(const PCam& cm, v2f Dl, v2f Dc, v2f Du, v2f c, v2f l, v2f u, v2f Ic, v2f Il, v2f Iu, v2f ax, float ay)
{
    const v2f Z2 = { 0.f, 0.f };
    const M2 on = { Dl.x > 0.0f && Dc.x > 0.0f && Du.x > 0.0f, Dl.y > 0.0f && Dc.y > 0.0f && Du.y > 0.0f };
    const float ify = 1.0f / cm.fy, ifx = 1.0f / cm.fx, kxy = 1.0f / (cm.fx * cm.fy);
    const v2f cl = c - l, cu = c - u;
    const v2f nx = (u * cl) * ify;
    const v2f ny = (l * cu) * ifx;
    const v2f nz = (nx * ax + ny * ay) - (l * u) * kxy;
    const v2f sq = nx * nx + ny * ny + nz * nz;
    const M2 pos = { sq.x > 0.0f, sq.y > 0.0f };
    const v2f inv = sel(pos, v2f{ 1.0f / sqrtf(sq.x), 1.0f / sqrtf(sq.y) }, splat(1.0f));
    const v2f n0 = inv * nx, n1 = inv * ny, n2 = inv * nz;
    const float* L = cm.L;
    v2f B = splat(L[0]);
    B = B + n1 * L[1]; B = B + n2 * L[2]; B = B + n0 * L[3];
    B = B + (n0 * n1) * L[4]; B = B + (n1 * n2) * L[5];
    B = B + (((n0 * n0) * -1.0f - n1 * n1) + (n2 * n2) * 2.0f) * L[6];
    B = B + (n2 * n0) * L[7]; B = B + (n0 * n0 - n1 * n1) * L[8];
    const v2f I = Ic * 0.5f + 0.25f * (Il + Iu);
    // gradient of the SH polynomial in n^
    const v2f g0 = L[3] + n1 * L[4] - (2.0f * L[6]) * n0 + n2 * L[7] + (2.0f * L[8]) * n0;
    const v2f g1 = L[1] + n0 * L[4] + n2 * L[5] - (2.0f * L[6]) * n1 - (2.0f * L[8]) * n1;
    const v2f g2 = L[2] + n1 * L[5] + (4.0f * L[6]) * n2 + n0 * L[7];
    const v2f gn = g0 * n0 + g1 * n1 + g2 * n2;
    const v2f h0 = sel(pos, inv * (g0 - n0 * gn), g0), h1 = sel(pos, inv * (g1 - n1 * gn), g1), h2 = sel(pos, inv * (g2 - n2 * gn), g2);
    const v2f A = h0 + h2 * ax, Bq = h1 + h2 * ay, Cq = h2 * kxy;
    const v2f uf = u * ify, lf = l * ifx;
    BIv r;
    r.b  = sel(on, B - I, Z2);
    r.dc = sel(on, A * uf + Bq * lf, Z2);
    r.dl = sel(on, Bq * (cu * ifx) - A * uf - Cq * u, Z2);
    r.du = sel(on, A * (cl * ify) - Bq * lf - Cq * l, Z2);
    return r;
}

}  // namespace
