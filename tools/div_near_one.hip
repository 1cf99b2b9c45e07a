// tools/div_near_one.hip -- exhaustive check of a short division for the sphere test's two quotients (-hb -+ sqrt(disc)) / a
// (rt.comp:136-137), where a = dot(dir, dir) of a NORMALISED direction: a few float values next to 1.0.  For every divisor a
// whose bit pattern lies within K of 1.0f's (K = 64: |a - 1| < 8e-6, far more than normalisation noise) and every one of the
// 2^32 numerators x, compare   q = fma(fma(-a, x*r, x), r, x*r),  r = v_rcp_f32(a)   with the IEEE quotient x / a the compiler emits.
// Prints, per variant, the number of (x, a) pairs whose bit patterns differ, split into pairs the kernel's acceptance rule
// lets through (|x| >= 2^-100 and the short form's result is a normal number) and pairs it sends to the IEEE divide.  The kernel may use
// the short form only where the count is 0.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/div_near_one tools/div_near_one.hip && /tmp/div_near_one
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int K = 64;

__global__ void check(unsigned long long *bad, uint32_t a_bits)
{
    const float a = __uint_as_float(a_bits);
    const float r = __builtin_amdgcn_rcpf(a);
    const float r1 = fmaf(fmaf(-a, r, 1.0f), r, r);          // refined reciprocal (== 1/a for normal results, tools/rcp_exact.hip)
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) << 8;      // 2^24 threads x 256 numerators
    for (uint32_t j = 0; j < 256; j++) {
        const float x = __uint_as_float(base | j);
        const float ref = x / a;
        // the acceptance rule a kernel would apply: the numerator is at least 2^-100 in magnitude (below that
        // the residual x - a * q0, ~2^-24 |x|, falls under the denormal range and corrects nothing) and the SHORT form's own
        // result is a normal number (an overflowing x * r turns it into NaN, which the rule rejects); everything else takes
        // the IEEE divide
        const uint32_t ex = ((base | j) >> 23) & 0xffu;
        // A: raw reciprocal + one residual correction
        float q0 = x * r;
        const float qa = fmaf(fmaf(-a, q0, x), r, q0);
        // B: refined reciprocal + one residual correction
        q0 = x * r1;
        const float qb = fmaf(fmaf(-a, q0, x), r1, q0);
        if (__float_as_uint(qa) != __float_as_uint(ref)) {
            const bool plain = ex >= 27u && __builtin_amdgcn_classf(qa, 0x108);
            if (plain) { b0++; if (atomicAdd(&bad[4], 1ull) < 24) printf("  differ: x %08x a %08x ieee %08x short %08x\n", base | j, a_bits, __float_as_uint(ref), __float_as_uint(qa)); }
            else b1++;
        }
        if (__float_as_uint(qb) != __float_as_uint(ref)) { if (ex >= 27u && __builtin_amdgcn_classf(qb, 0x108)) b2++; else b3++; }
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
    if (b2) atomicAdd(&bad[2], b2);
    if (b3) atomicAdd(&bad[3], b3);
}

int main()
{
    unsigned long long *d, h[5] = {0, 0, 0, 0, 0};
    if (hipMalloc(&d, sizeof h) != hipSuccess) { std::fprintf(stderr, "no GPU\n"); return 1; }
    hipMemset(d, 0, sizeof h);
    for (int k = -K; k <= K; k++) {
        hipLaunchKernelGGL(check, dim3((1u << 24) / 256), dim3(256), 0, 0, d, (uint32_t)(0x3f800000 + k));
        if ((k & 15) == 0) { if (hipDeviceSynchronize() != hipSuccess) return 1; std::printf("  a = 1.0f %+d ulp-steps done\n", k); std::fflush(stdout); }
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    std::printf("x / a for all 2^32 x and the %d divisors within %d bit patterns of 1.0f: pairs that differ from the IEEE divide\n", 2 * K + 1, K);
    std::printf("  A rcp + 1 correction         : %llu accepted by the kernel's rule (|x| >= 2^-100, short result normal), %llu rejected by it\n", h[0], h[1]);
    std::printf("  B refined rcp + 1 correction : %llu accepted by the kernel's rule (|x| >= 2^-100, short result normal), %llu rejected by it\n", h[2], h[3]);
    return 0;
}
