// tools/nd_exact.hip -- the cut-down Box-Muller factors of pt_math.hpp (nd_cos_of_draw, nd_rho_of_draw) against the
// generic compositions kTwoPi * u -> pt_cosf and u -> pt_logf -> sqrtf, for EVERY 32-bit draw (bitwise).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -o tools/bin/nd_exact tools/nd_exact.hip && tools/bin/nd_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../path_tracer_amd/csrc/pt_math.hpp"

__global__ void check(unsigned long long *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;          // 2^24 threads x 256 draws
    unsigned long long bad_cos = 0, bad_rho = 0;
    for (uint32_t j = 0; j < 256; j++) {
        const uint32_t r = tid * 256u + j;
        const float u = (float)r / 4294967296.0f;
        const float c_ref = pt::pt_cosf(pt::kTwoPi * u);
        const float rho_ref = sqrtf(-2.0f * pt::pt_logf(u));
        if (__float_as_uint(pt::nd_cos_of_draw(r)) != __float_as_uint(c_ref)) bad_cos++;
        if (__float_as_uint(pt::nd_rho_of_draw(r)) != __float_as_uint(rho_ref)) bad_rho++;
    }
    atomicAdd(&out[0], bad_cos); atomicAdd(&out[1], bad_rho);
}

int main()
{
    unsigned long long *d, h[2] = {0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess) { std::fprintf(stderr, "no GPU\n"); return 1; }
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3((1u << 24) / 256), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("all 2^32 draws: nd_cos_of_draw differs from pt_cosf(2 pi u) on %llu, nd_rho_of_draw differs from sqrtf(-2 pt_logf(u)) on %llu\n", h[0], h[1]);
    return 0;
}
