// loop_bench -- the hot loop in isolation with the LDS triangle store: every wave tests `n_rays*reps` broadcast rays against
// the 11 triangles per lane (pt::test_slots + the DPP reduction), at 1..8 waves per SIMD.  Reports VALU-issue efficiency
// = (VALU instructions x 2 cycles) / elapsed SIMD cycles, given the instruction count per ray from the ISA (INSTR_PER_RAY).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o loop_bench loop_bench.hip
#include "../path_tracer_amd/csrc/pt_kernels.hpp"
#include <cstdio>
#include <vector>
using namespace pt;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WPS>
__global__ void __launch_bounds__(256, WPS) k(const float *soa, const float *rays, int n_rays, int reps, float *out, unsigned long long *cyc)
{
    __shared__ float4 ldsA[kSlots * kWave];
    __shared__ float4 ldsB[kSlots * kWave];
    __shared__ float ldsC[kSlots * kWave];
    for (int i = threadIdx.x; i < kSlots * kWave; i += 256) {
        const float *g = soa + i;
        const int n = kSlots * kWave;
        ldsA[i] = make_float4(g[0], g[n], g[2 * n], g[3 * n]);
        ldsB[i] = make_float4(g[4 * n], g[5 * n], g[6 * n], g[7 * n]);
        ldsC[i] = g[8 * n];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    LdsStore<kSlots> st;
    st.A = ldsA + lane; st.B = ldsB + lane; st.C = ldsC + lane;
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < n_rays; i++) {
            const float4 a = uniform_load((const float4 *)rays, 2 * i), b = uniform_load((const float4 *)rays, 2 * i + 1);
            v3 ro = V(to_vgpr(a.x), to_vgpr(a.y), to_vgpr(a.z)), rd = V(to_vgpr(b.x), to_vgpr(b.y), to_vgpr(b.z));
            float lt; int ls; bool bad = false;
            test_slots<kSlots, true>(st, ro, rd, lt, ls, bad);
            const float tmin = readlane_f(wave_min_to_lane63(lt), 63);
            acc += tmin + ls + (bad ? 1.0f : 0.0f);
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    unsigned long long w1 = wall_clock64();
    if (lane == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; cyc[256 * 8 * 4 + blockIdx.x * 4 + (threadIdx.x >> 6)] = w1 - w0; }
}

template <int WPS>
__global__ void __launch_bounds__(64, WPS) kreg(const float *soa, const float *rays, int n_rays, int reps, float *out, unsigned long long *cyc)
{
    RegStore<kSlots> st;
    SceneView sc; sc.mesh_soa = soa; sc.n_chunks = 1;
    st.init(sc, threadIdx.x);
    float acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int r = 0; r < reps; r++)
        for (int i = 0; i < n_rays; i++) {
            const float4 a = uniform_load((const float4 *)rays, 2 * i), b = uniform_load((const float4 *)rays, 2 * i + 1);
            v3 ro = V(to_vgpr(a.x), to_vgpr(a.y), to_vgpr(a.z)), rd = V(to_vgpr(b.x), to_vgpr(b.y), to_vgpr(b.z));
            float lt; int ls; bool bad = false;
            test_slots<kSlots, true>(st, ro, rd, lt, ls, bad);
            const float tmin = readlane_f(wave_min_to_lane63(lt), 63);
            acc += tmin + ls + (bad ? 1.0f : 0.0f);
        }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 * 8 * 4 + blockIdx.x] = w1 - w0; }
}

template <int WPS>
int runreg(const float *d_soa, const float *d_rays, int n_rays, int reps, float *d_out, unsigned long long *d_cyc)
{
    const int grid = 1024 * WPS;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 2; it++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kreg<WPS>, dim3(grid), dim3(64), 0, 0, d_soa, d_rays, n_rays, reps, d_out, d_cyc);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> c(grid), w(grid);
    CHECK(hipMemcpy(c.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(w.data(), d_cyc + 256 * 8 * 4, grid * 8, hipMemcpyDeviceToHost));
    double avg = 0, wavg = 0; for (auto v : c) avg += v; for (auto v : w) wavg += v;
    const double ghz = avg / (wavg * 10);
    const double rays_per_wave = (double)n_rays * reps;
    printf("REG waves/SIMD=%d : %.3f ms, clock %.3f GHz, %.0f cycles per wave-ray per SIMD (kernel time based)\n", WPS, ms, ghz, ms * 1e6 / (rays_per_wave * WPS) * ghz);
    return 0;
}

template <int WPS>
int run(const float *d_soa, const float *d_rays, int n_rays, int reps, float *d_out, unsigned long long *d_cyc)
{
    const int grid = 256 * WPS;     // blocks of 4 waves: WPS per CU
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 2; it++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<WPS>, dim3(grid), dim3(256), 0, 0, d_soa, d_rays, n_rays, reps, d_out, d_cyc);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> c(grid * 4), w(grid * 4);
    CHECK(hipMemcpy(c.data(), d_cyc, grid * 4 * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(w.data(), d_cyc + 256 * 8 * 4, grid * 4 * 8, hipMemcpyDeviceToHost));
    double avg = 0, wavg = 0, wmax = 0; for (auto v : c) avg += v; avg /= c.size();
    for (auto v : w) { wavg += v; if (v > wmax) wmax = v; } wavg /= w.size();

    const double rays_per_wave = (double)n_rays * reps;
    const double wave_rays = rays_per_wave * grid * 4;
    const double ghz = avg / (wavg * 10);
    printf("LDS waves/SIMD=%d : %.3f ms, clock %.3f GHz, %.0f cycles per wave-ray per SIMD (kernel time based)  (%.2f G wave-rays/s chip)\n",
           WPS, ms, ghz, ms * 1e6 / (rays_per_wave * WPS) * ghz, wave_rays / ms / 1e6);
    return 0;
}

int main()
{
    const int n_rays = 64, reps = 100;
    std::vector<float> soa(9 * kChunk), rays(8 * n_rays);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : soa) v = rnd() - 0.5f;
    for (int i = 0; i < n_rays; i++) { float *r = &rays[8 * i]; r[0] = 1.5f; r[1] = 0.4f; r[2] = 0; r[3] = 0; r[4] = -1; r[5] = rnd() - 0.5f; r[6] = rnd() - 0.5f; r[7] = 0; }
    float *d_soa, *d_rays, *d_out; unsigned long long *d_cyc;
    CHECK(hipMalloc(&d_soa, soa.size() * 4)); CHECK(hipMalloc(&d_rays, rays.size() * 4)); CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * 2)); CHECK(hipMalloc(&d_cyc, 2 * 256 * 8 * 4 * 8));
    CHECK(hipMemcpy(d_soa, soa.data(), soa.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
    if (runreg<1>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (runreg<2>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (runreg<3>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (run<3>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (run<4>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (run<5>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    if (run<6>(d_soa, d_rays, n_rays, reps, d_out, d_cyc)) return 1;
    return 0;
}
