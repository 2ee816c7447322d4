// pk_rate.hip -- issue rate of v_pk_fma_f32 / v_fma_f32 / v_fma_f64 / v_cvt_f64_f32 / s_nop / v_mov_dpp on gfx950, at 1, 2, 4 waves per SIMD (tools only).
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    v2f x0 = {a, b}, x1 = {b, a}, x2 = {a + 1, b}, x3 = {a, b + 1}, x4 = {a + 2, b}, x5 = {a, b + 2}, x6 = {a + 3, b}, x7 = {a, b + 3};
    const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    double d0 = a, d1 = b, d2 = a + b, d3 = a - b;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 16 v_pk_fma_f32 per trip
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(c));
                  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(c));)
        } else if (MODE == 1) { // 64 v_fma_f32 per trip (scalar halves)
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0.x) : "v"(m.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1.x) : "v"(m.x), "v"(c.x));
                  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2.x) : "v"(m.x), "v"(c.x)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 2) { // v_fma_f64
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d2), "v"(d3)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d2), "v"(d3));
                  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(d2), "v"(d3)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(d2), "v"(d3));)
        } else if (MODE == 3) { // v_cvt_f64_f32
            REP16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(x0.x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(x1.x));
                  asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(x2.x)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(x3.x));)
        } else if (MODE == 4) { // s_nop 0
            REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");)
        } else if (MODE == 5) { // v_mov_b32_dpp
            REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x0.x) : "v"(x4.x)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x1.x) : "v"(x5.x));
                  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x2.x) : "v"(x6.x)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x3.x) : "v"(x7.x));)
        } else if (MODE == 6) { // s_mul_i32 / s_add mix (SALU)
            int s = i;
            REP16(asm volatile("s_add_i32 %0, %0, 3\n s_lshl_b32 %0, %0, 1\n s_add_i32 %0, %0, 5\n s_lshl_b32 %0, %0, 1" : "+s"(s));)
            x0.x += (float)s * 1e-30f;
        } else if (MODE == 7) { // v_pk_mul_f32 + v_pk_add_f32 alternating
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x1) : "v"(c));
                  asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(m)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x3) : "v"(c));)
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = x0.x + x0.y + x1.x + x1.y + x2.x + x2.y + x3.x + x3.y + x4.x + x5.x + x6.x + x7.x + (float)(d0 + d1 + d2 + d3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[gridDim.x * blockDim.x] = (float)(t1 - t0);
}
template <int MODE> void run(const char* name, int per_trip)
{
    float* out; hipMalloc(&out, sizeof(float) * (256 * 1024 * 4 + 16));
    const int iters = 2000;
    for (int wg_per_cu : {1, 2, 4}) {
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float cyc; hipMemcpy(&cyc, out + grid * 256, 4, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD %d: %.2f shader cycles per instruction per wave (s_memtime), %.3f ms\n", name, wg_per_cu, cyc / ((double)iters * per_trip), ms);
    }
    hipFree(out);
}
int main()
{
    run<0>("v_pk_fma_f32", 64); run<1>("v_fma_f32", 64); run<7>("v_pk_mul/add_f32", 64); run<2>("v_fma_f64", 64); run<3>("v_cvt_f64_f32", 64);
    run<4>("s_nop 0", 64); run<5>("v_mov_b32_dpp", 64); run<6>("s_add/s_lshl", 64);
    return 0;
}
