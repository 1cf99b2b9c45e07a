// valu_rate — microbenchmark behind DESIGN.md's roofline: sustained wave64 issue rate of the unfused
// f32 instructions the hot loop is made of (v_mul_f32 / v_add_f32 vs v_pk_mul_f32 / v_pk_add_f32,
// v_fma_f32, v_rcp_f32) on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, float a, float b)
{
    float x[8]; float2v y[8];
    for (int i = 0; i < 8; i++) { x[i] = a + threadIdx.x * 1e-6f + i; y[i] = float2v{x[i], x[i] + 0.5f}; }
    float2v bb = float2v{b, b * 1.0001f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) x[i] = x[i] * b;                                  // v_mul_f32
            if (MODE == 1) { x[i] = x[i] * b; x[i] = x[i] + a; }             // v_mul_f32 + v_add_f32 (unfused)
            if (MODE == 2) y[i] = y[i] * bb;                                 // v_pk_mul_f32
            if (MODE == 3) { y[i] = y[i] * bb; y[i] = y[i] + bb; }           // v_pk_mul_f32 + v_pk_add_f32
            if (MODE == 4) x[i] = __builtin_fmaf(x[i], b, a);                // v_fma_f32
            if (MODE == 5) x[i] = __builtin_amdgcn_rcpf(x[i]);               // v_rcp_f32
            if (MODE == 6) y[i] = __builtin_elementwise_fma(y[i], bb, bb);   // v_pk_fma_f32
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> int run(const char *name, int ops_per_iter_per_lane, int blocks_per_cu, float *d)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0f, 0.9999f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0f, 0.9999f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double lane_ops = (double)grid * 256 * ITERS * 8 * ops_per_iter_per_lane;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.2f T lane-ops/s  (%.1f lane-ops/clk/CU @2.4GHz)\n", name, blocks_per_cu, ms, lane_ops / ms * 1e-9,
           lane_ops / (ms * 1e-3) / 2.4e9 / 256);
    return 0;
}

int main()
{
    float *d; CHECK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 2, 4, 8}) {
        if (run<0>("v_mul_f32", 1, w, d)) return 1;
        if (run<1>("v_mul_f32+v_add_f32", 2, w, d)) return 1;
        if (run<2>("v_pk_mul_f32", 2, w, d)) return 1;
        if (run<3>("v_pk_mul_f32+v_pk_add_f32", 4, w, d)) return 1;
        if (run<4>("v_fma_f32 (1 op)", 1, w, d)) return 1;
        if (run<6>("v_pk_fma_f32 (2 ops)", 2, w, d)) return 1;
        if (run<5>("v_rcp_f32", 1, w, d)) return 1;
    }
    CHECK(hipFree(d));
    return 0;
}
