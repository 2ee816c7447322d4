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

}  // namespace
