// rcp_exact -- exhaustive check (all 2^32 float32 bit patterns) of a cheap correctly-rounded
// reciprocal candidate against IEEE 1.0f/x, as used by the hot loop's `inv = 1.0 / det`.
// candidate: pt::exact_rcp (pt_math.hpp): r0 = v_rcp_f32(x); e = fma(-x, r0, 1); r = fma(e, r0, r0), IEEE divide
// when r is not a normal number.  -DFAST_ONLY=1 checks the bare Newton step instead.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#include "../path_tracer_amd/csrc/pt_math.hpp"
#if FAST_ONLY
__device__ __forceinline__ float cand(float x)
{
    float r0 = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
#else
__device__ __forceinline__ float cand(float x) { return pt::exact_rcp(x); }   // the function the kernels use
#endif

// bins: by exponent field of x (0..255) -> mismatch count; plus first few examples
__global__ void sweep(unsigned long long *bins, unsigned long long *total, uint32_t *examples)
{
    const uint64_t n = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float x = __uint_as_float((uint32_t)i);
        float want = 1.0f / x, got = cand(x);
        uint32_t a = __float_as_uint(want), b = __float_as_uint(got);
        bool nan_both = (want != want) && (got != got);
        if (a != b && !nan_both) {
            atomicAdd(&bins[(i >> 23) & 0xff], 1ull);
            unsigned long long k = atomicAdd(total, 1ull);
            if (k < 16) { examples[3 * k] = (uint32_t)i; examples[3 * k + 1] = a; examples[3 * k + 2] = b; }
        }
    }
}

int main()
{
    unsigned long long *bins, *total; uint32_t *ex;
    CHECK(hipMalloc(&bins, 256 * 8)); CHECK(hipMalloc(&total, 8)); CHECK(hipMalloc(&ex, 48 * 4));
    CHECK(hipMemset(bins, 0, 256 * 8)); CHECK(hipMemset(total, 0, 8)); CHECK(hipMemset(ex, 0, 48 * 4));
    hipLaunchKernelGGL(sweep, dim3(256 * 16), dim3(256), 0, 0, bins, total, ex);
    CHECK(hipDeviceSynchronize());
    unsigned long long hb[256], ht; uint32_t hex[48];
    CHECK(hipMemcpy(hb, bins, sizeof hb, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&ht, total, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex, ex, sizeof hex, hipMemcpyDeviceToHost));
    printf("total mismatches: %llu of 2^32\n", ht);
    for (int e = 0; e < 256; e++) if (hb[e]) printf("  exponent field %3d: %llu\n", e, hb[e]);
    for (int k = 0; k < 16 && k < (int)ht; k++) printf("  x=%08x want=%08x got=%08x\n", hex[3 * k], hex[3 * k + 1], hex[3 * k + 2]);
    return 0;
}
