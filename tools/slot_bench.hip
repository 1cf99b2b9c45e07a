// slot_bench -- isolates the hot loop (pt::test_slots: one broadcast ray against the 11 register-resident
// triangles of each lane) and reports SIMD cycles per ray-triangle wave-test for 1..3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize [-DPT_SLOT_GROUP=n] -o slot_bench slot_bench.hip
#include "../path_tracer_amd/csrc/pt_kernels.hpp"
#include <cstdio>
#ifndef RAY_IN_VGPR
#define RAY_IN_VGPR 0
#endif
#include <vector>
using namespace pt;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64, 3) k(const float *soa, const float *rays, int n_rays, int reps, float *out, unsigned long long *cyc)
{
    TriRegs<kSlots> T;
    load_chunk<kSlots>(T, soa, 0, threadIdx.x);
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < n_rays; i++) {
#if RAY_IN_VGPR
            const float *p = rays + 6 * (i + (threadIdx.x >> 6));      // same ray in every lane, but kept in VGPRs
            v3 ro = V(p[0], p[1], p[2]), rd = V(p[3], p[4], p[5]);
#else
            const float *p = rays + 6 * i;
            v3 ro = V(uniform_load((const float4 *)p, 0).x, p[1], p[2]), rd = V(p[3], p[4], p[5]);
            ro.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ro.x)));
            ro.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ro.y)));
            ro.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ro.z)));
            rd.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rd.x)));
            rd.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rd.y)));
            rd.z = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rd.z)));
#endif
            float lt; int ls; bool bad = false;
            test_slots<kSlots, true>(T, ro, rd, lt, ls, bad);
            acc += lt + ls + (bad ? 1.0f : 0.0f);
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int n_rays = 64, reps = 200;
    std::vector<float> soa(9 * kChunk), rays(6 * n_rays);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : soa) v = rnd() - 0.5f;
    for (int i = 0; i < n_rays; i++) { rays[6 * i] = 1.5f; rays[6 * i + 1] = 0.4f; rays[6 * i + 2] = 0; rays[6 * i + 3] = -1; rays[6 * i + 4] = rnd() - 0.5f; rays[6 * i + 5] = rnd() - 0.5f; }
    float *d_soa, *d_rays, *d_out; unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_soa, soa.size() * 4)); CHECK(hipMalloc(&d_rays, rays.size() * 4)); CHECK(hipMalloc(&d_out, 256 * 16 * 64 * 4)); CHECK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CHECK(hipMemcpy(d_soa, soa.data(), soa.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
    for (int wps : {1, 2, 3}) {
        int grid = 256 * 4 * wps;
        for (int it = 0; it < 2; it++) { hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d_soa, d_rays, n_rays, reps, d_out, d_cyc); CHECK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> c(grid);
        CHECK(hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : c) avg += v; avg /= grid;
        double tests = (double)n_rays * reps * kSlots;
        printf("ray_in_vgpr=%d slot_group=%d waves/SIMD=%d : s_memtime ticks per wave-test (one wave) %.1f, per SIMD %.1f\n", RAY_IN_VGPR, PT_SLOT_GROUP, wps, avg / tests, avg / tests / wps);
    }
    return 0;
}
