// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/lane_xor_check tools/lane_xor_check.hip  (run on the GPU box)
// prints, for each exchange primitive of device_common.hpp's butterfly, which lane's value every lane receives
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../thallo_amd/csrc/device_common.hpp"
using namespace thallo;
__global__ void k(unsigned* out, float* sums, double* dsums)
{
    const unsigned l = threadIdx.x;
    out[0 * 64 + l] = lane_xor_u32<1>(l);
    out[1 * 64 + l] = lane_xor_u32<2>(l);
    out[2 * 64 + l] = lane_xor_u32<4>(l);
    out[3 * 64 + l] = lane_xor_u32<8>(l);
    swap_halves_u32<16>(l, out[4 * 64 + l], out[5 * 64 + l]);
    swap_halves_u32<32>(l, out[6 * 64 + l], out[7 * 64 + l]);
    {
        const float f = (float)l;
        float a = f; a += __builtin_bit_cast(float, lane_xor_u32<1>(__builtin_bit_cast(unsigned, a))); dsums[64 + l] = a;
        a = f; a += __builtin_bit_cast(float, lane_xor_u32<2>(__builtin_bit_cast(unsigned, a))); dsums[128 + l] = a;
        a = f; a += __builtin_bit_cast(float, lane_xor_u32<4>(__builtin_bit_cast(unsigned, a))); dsums[192 + l] = a;
        a = f; a += __builtin_bit_cast(float, lane_xor_u32<8>(__builtin_bit_cast(unsigned, a))); dsums[256 + l] = a;
        { unsigned x, y; swap_halves_u32<16>(__builtin_bit_cast(unsigned, f), x, y); dsums[320 + l] = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y); }
        { unsigned x, y; swap_halves_u32<32>(__builtin_bit_cast(unsigned, f), x, y); dsums[384 + l] = __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y); }
    }
    sums[l] = wave_sum_all((float)(1u << (l % 20)) + 0.001f * l);
    dsums[l] = wave_sum_all_f64((double)l * 1.000001 + 1e-9 * l * l);
}
int main()
{
    unsigned* d; float* s; double* ds; hipMalloc(&d, 8 * 64 * 4); hipMalloc(&s, 64 * 4); hipMalloc(&ds, 7 * 64 * 8);
    k<<<1, 64>>>(d, s, ds);
    unsigned h[8 * 64]; float hs[64]; double hd[7 * 64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof hs, hipMemcpyDeviceToHost); hipMemcpy(hd, ds, sizeof hd, hipMemcpyDeviceToHost);
    const char* nm[8] = { "xor1", "xor2", "xor4", "xor8", "swap16.a", "swap16.b", "swap32.a", "swap32.b" };
    for (int q = 0; q < 8; ++q) { printf("%-9s", nm[q]); for (int l = 0; l < 64; ++l) printf(" %2u", h[q * 64 + l]); printf("\n"); }
    // the butterfly on the host, same order
    float v[64]; double w[64];
    for (int l = 0; l < 64; ++l) { v[l] = (float)(1u << (l % 20)) + 0.001f * l; w[l] = (double)l * 1.000001 + 1e-9 * l * l; }
    for (int m = 32; m >= 1; m >>= 1) { float t[64]; double u[64]; for (int l = 0; l < 64; ++l) { t[l] = v[l] + v[l ^ m]; u[l] = w[l] + w[l ^ m]; } for (int l = 0; l < 64; ++l) { v[l] = t[l]; w[l] = u[l]; } }
    { const int Ms[6] = { 1, 2, 4, 8, 16, 32 };
      for (int q = 0; q < 6; ++q) { int nb = 0; for (int l = 0; l < 64; ++l) nb += hd[64 * (q + 1) + l] != (double)(l + (l ^ Ms[q])); printf("float step xor %d: %d wrong lanes (lane 5: %g, want %d)\n", Ms[q], nb, hd[64 * (q + 1) + 5], 5 + (5 ^ Ms[q])); } }
    int bad = 0; for (int l = 0; l < 64; ++l) bad += (hs[l] != v[l]) + (hd[l] != w[l]);
    printf("sum lane0 gpu %.9g host %.9g ; double gpu %.17g host %.17g ; mismatching lanes %d\n", hs[0], v[0], hd[0], w[0], bad);
    return bad != 0;
}
