// tools/sqrt_exact.hip -- exhaustive checks (all 2^32 float bit patterns) of short exact sequences for
//   (1) sqrtf(x)                : v_sqrt_f32 + one fused residual step with h = 0.5 * v_rsq_f32(x), for 2^-64 <= x < inf
//                                 (below ~2^-103 the residual x - s*s underflows and the step goes wrong)
//   (2) a / b, b = 1 .. 256     : v_rcp_f32(b) + one residual correction (the per-sample `incoming / bounces`, rt.comp:322)
// against the correctly rounded results the compiler's IEEE expansions give.  A sequence is usable where its result is a
// normal number (the kernel checks the class and falls back to the IEEE expansion otherwise), so a mismatch counts only
// when the fast result is normal ("bad"); "fallback" counts the inputs that would take the slow path.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/bin/sqrt_exact tools/sqrt_exact.hip && tools/bin/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ bool is_normal(float v) { return __builtin_amdgcn_classf(v, 0x108); }     // +-normal

__global__ void check_sqrt(unsigned long long *out)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;          // 2^24 threads x 256 values
    unsigned long long bad = 0, fb = 0, bad2 = 0;
    for (uint32_t j = 0; j < 256; j++) {
        const uint32_t bits = tid * 256u + j;
        const float x = __uint_as_float(bits);
        const float ref = sqrtf(x);
        const float s = __builtin_amdgcn_sqrtf(x);
        const float h = 0.5f * __builtin_amdgcn_rsqf(x);
        const float e = fmaf(-s, s, x);
        const float s1 = fmaf(e, h, s);
        // usable for 2^-64 <= x < inf: one unsigned compare on the bits, which also rejects 0, denormals, inf, NaN, negatives
        const bool ok_in = (bits - 0x1f800000u) < 0x60000000u;
        if (!ok_in) fb++;
        else if (__float_as_uint(s1) != __float_as_uint(ref)) bad++;
        // variant without v_sqrt: s = x * rsq(x), same correction
        const float rs = __builtin_amdgcn_rsqf(x);
        const float sb = x * rs;
        const float s2 = fmaf(fmaf(-sb, sb, x), 0.5f * rs, sb);
        if (ok_in && __float_as_uint(s2) != __float_as_uint(ref)) bad2++;
    }
    atomicAdd(&out[0], bad); atomicAdd(&out[1], fb); atomicAdd(&out[2], bad2);
}

__global__ void check_div(unsigned long long *out, int b_lo, int b_hi)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;          // 2^24 threads x 256 values
    unsigned long long bad = 0, fb = 0;
    for (int bi = b_lo; bi <= b_hi; bi++) {
        const float b = (float)bi;
        const float r = __builtin_amdgcn_rcpf(b);
        for (uint32_t j = 0; j < 256; j++) {
            const float a = __uint_as_float(tid * 256u + j);
            const float ref = a / b;
            const float q0 = a * r;
            const float q = fmaf(fmaf(-b, q0, a), r, q0);
            if (!is_normal(q)) fb++;
            else if (__float_as_uint(q) != __float_as_uint(ref)) bad++;
        }
    }
    atomicAdd(&out[3], bad); atomicAdd(&out[4], fb);
}

int main()
{
    unsigned long long *d, h[5] = {0, 0, 0, 0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess) { std::fprintf(stderr, "no GPU\n"); return 1; }
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check_sqrt, dim3((1u << 24) / 256), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    for (int b = 1; b <= 256; b += 32) {                                  // 8 launches of a few seconds each
        hipLaunchKernelGGL(check_div, dim3((1u << 24) / 256), dim3(256), 0, 0, d, b, b + 31);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        std::printf("div: b <= %d done\n", b + 31); std::fflush(stdout);
    }
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("sqrtf(x), all 2^32 x: v_sqrt + residual * 0.5*v_rsq : wrong on %llu inputs in [2^-64, inf) (other inputs, slow path: %llu)\n", h[0], h[1]);
    std::printf("                      x*v_rsq + residual * 0.5*v_rsq: wrong on %llu inputs in [2^-64, inf)\n", h[2]);
    std::printf("a / b, all 2^32 a x b = 1..256: v_rcp + one correction : wrong on %llu normal results (non-normal results, slow path: %llu)\n", h[3], h[4]);
    return 0;
}
