// tools/half_wave.hip -- does gfx950 skip the idle half of a wave64 VALU instruction?  A wave64 instruction runs as two
// passes over a 32-lane SIMD; if a pass whose 32 lanes are all masked off is skipped, compacting the active lanes of a
// divergent phase into one half of the wave would double its rate.  Times a dependent-free v_fma chain with all 64 lanes
// active, lanes 0-31, lanes 32-63, every other lane (32 lanes spread over both halves), and one lane.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/half_wave tools/half_wave.hip && tools/bin/half_wave
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) chain(float *out, int mode, int iters)
{
    const int lane = threadIdx.x & 63;
    bool on = true;
    if (mode == 1) on = lane < 32;
    if (mode == 2) on = lane >= 32;
    if (mode == 3) on = (lane & 1) == 0;
    if (mode == 4) on = lane == 0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 2.0f, g = 3.0f, h = 4.0f;
    if (on) {
        for (int i = 0; i < iters; i++) {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}

int main()
{
    const int blocks = 256 * 8, iters = 20000;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    float *out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[5] = {"all 64 lanes", "lanes 0-31", "lanes 32-63", "even lanes", "lane 0 only"};
    for (int mode = 0; mode < 5; mode++) {
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, mode, 100);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, mode, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)blocks * 4 * iters * 8;          // wave instructions
        printf("%-14s %8.3f ms   %.2f SIMD cycles per wave64 v_fma at 2.4 GHz (1024 SIMDs)\n", names[mode], ms, ms * 1e-3 * 2.4e9 * 1024 / inst);
    }
    return 0;
}
