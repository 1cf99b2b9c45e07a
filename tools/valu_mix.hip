// valu_mix -- SIMD cycles per wave64 instruction for the instruction forms the hot loop uses, 8 waves per SIMD,
// independent instructions with hand-picked registers (VGPR bank = index mod 4).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","vcc","s20","s21","s22","s23","s24","s25","s26","s27"

// 8 instructions per group, 16 groups per asm block = 128 instructions
#define BODY(name, i0, i1, i2, i3, i4, i5, i6, i7)                                                                   \
    __global__ void __launch_bounds__(64) name(int reps, unsigned long long *cyc)                                    \
    {                                                                                                                \
        asm volatile("v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n" \
                     "v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n" \
                     "v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n" \
                     "s_mov_b32 s20, 1.0\n s_mov_b32 s21, 1.0\n s_mov_b32 s22, 1.0\n s_mov_b32 s23, 1.0\n" ::: CLOB);     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();                                   \
        for (int r = 0; r < reps; r++)                                                                               \
            asm volatile(R16(i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n") ::: CLOB);             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();                                   \
        if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = w1 - w0; }                  \
    }

// dst banks rotate; sources chosen per case
BODY(k_mul_diffbank, "v_mul_f32 v10, v21, v22", "v_mul_f32 v11, v22, v23", "v_mul_f32 v12, v23, v24", "v_mul_f32 v13, v24, v25", "v_mul_f32 v14, v25, v26", "v_mul_f32 v15, v26, v27", "v_mul_f32 v16, v27, v28", "v_mul_f32 v17, v28, v29")
BODY(k_mul_samebank, "v_mul_f32 v10, v21, v25", "v_mul_f32 v11, v22, v26", "v_mul_f32 v12, v23, v27", "v_mul_f32 v13, v24, v28", "v_mul_f32 v14, v25, v29", "v_mul_f32 v15, v26, v30", "v_mul_f32 v16, v27, v31", "v_mul_f32 v17, v28, v32")
BODY(k_mul_sgpr, "v_mul_f32 v10, s20, v22", "v_mul_f32 v11, s21, v23", "v_mul_f32 v12, s22, v24", "v_mul_f32 v13, s23, v25", "v_mul_f32 v14, s20, v26", "v_mul_f32 v15, s21, v27", "v_mul_f32 v16, s22, v28", "v_mul_f32 v17, s23, v29")
BODY(k_sub_diffbank, "v_sub_f32 v10, v21, v22", "v_sub_f32 v11, v22, v23", "v_sub_f32 v12, v23, v24", "v_sub_f32 v13, v24, v25", "v_sub_f32 v14, v25, v26", "v_sub_f32 v15, v26, v27", "v_sub_f32 v16, v27, v28", "v_sub_f32 v17, v28, v29")
BODY(k_fma_diffbank, "v_fma_f32 v10, v21, v22, v23", "v_fma_f32 v11, v22, v23, v24", "v_fma_f32 v12, v23, v24, v25", "v_fma_f32 v13, v24, v25, v26", "v_fma_f32 v14, v25, v26, v27", "v_fma_f32 v15, v26, v27, v28", "v_fma_f32 v16, v27, v28, v29", "v_fma_f32 v17, v28, v29, v30")
BODY(k_fma_samebank, "v_fma_f32 v10, v21, v25, v29", "v_fma_f32 v11, v22, v26, v30", "v_fma_f32 v12, v23, v27, v31", "v_fma_f32 v13, v24, v28, v32", "v_fma_f32 v14, v21, v25, v29", "v_fma_f32 v15, v22, v26, v30", "v_fma_f32 v16, v23, v27, v31", "v_fma_f32 v17, v24, v28, v32")
BODY(k_cmp_vcc, "v_cmp_lt_f32 vcc, v21, v22", "v_cmp_lt_f32 vcc, v22, v23", "v_cmp_lt_f32 vcc, v23, v24", "v_cmp_lt_f32 vcc, v24, v25", "v_cmp_lt_f32 vcc, v25, v26", "v_cmp_lt_f32 vcc, v26, v27", "v_cmp_lt_f32 vcc, v27, v28", "v_cmp_lt_f32 vcc, v28, v29")
BODY(k_cmp_sgpr, "v_cmp_lt_f32 s[24:25], v21, v22", "v_cmp_lt_f32 s[26:27], v22, v23", "v_cmp_lt_f32 s[24:25], v23, v24", "v_cmp_lt_f32 s[26:27], v24, v25", "v_cmp_lt_f32 s[24:25], v25, v26", "v_cmp_lt_f32 s[26:27], v26, v27", "v_cmp_lt_f32 s[24:25], v27, v28", "v_cmp_lt_f32 s[26:27], v28, v29")
BODY(k_cndmask, "v_cndmask_b32 v10, v21, v22, vcc", "v_cndmask_b32 v11, v22, v23, vcc", "v_cndmask_b32 v12, v23, v24, vcc", "v_cndmask_b32 v13, v24, v25, vcc", "v_cndmask_b32 v14, v25, v26, vcc", "v_cndmask_b32 v15, v26, v27, vcc", "v_cndmask_b32 v16, v27, v28, vcc", "v_cndmask_b32 v17, v28, v29, vcc")
BODY(k_minmax, "v_min_f32 v10, v21, v22", "v_max_f32 v11, v22, v23", "v_min_f32 v12, v23, v24", "v_max_f32 v13, v24, v25", "v_min_f32 v14, v25, v26", "v_max_f32 v15, v26, v27", "v_min_f32 v16, v27, v28", "v_max_f32 v17, v28, v29")
BODY(k_rcp, "v_rcp_f32 v10, v21", "v_rcp_f32 v11, v22", "v_rcp_f32 v12, v23", "v_rcp_f32 v13, v24", "v_rcp_f32 v14, v25", "v_rcp_f32 v15, v26", "v_rcp_f32 v16, v27", "v_rcp_f32 v17, v28")
BODY(k_class, "v_cmp_class_f32 vcc, v21, v22", "v_cmp_class_f32 vcc, v22, v23", "v_cmp_class_f32 vcc, v23, v24", "v_cmp_class_f32 vcc, v24, v25", "v_cmp_class_f32 vcc, v25, v26", "v_cmp_class_f32 vcc, v26, v27", "v_cmp_class_f32 vcc, v27, v28", "v_cmp_class_f32 vcc, v28, v29")
BODY(k_mul_dep, "v_mul_f32 v10, v10, v22", "v_mul_f32 v11, v11, v23", "v_mul_f32 v12, v12, v24", "v_mul_f32 v13, v13, v25", "v_mul_f32 v14, v14, v26", "v_mul_f32 v15, v15, v27", "v_mul_f32 v16, v16, v28", "v_mul_f32 v17, v17, v29")
BODY(k_mul_chain, "v_mul_f32 v10, v10, v22", "v_mul_f32 v10, v10, v23", "v_mul_f32 v10, v10, v24", "v_mul_f32 v10, v10, v25", "v_mul_f32 v10, v10, v26", "v_mul_f32 v10, v10, v27", "v_mul_f32 v10, v10, v28", "v_mul_f32 v10, v10, v29")
BODY(k_mul_e64, "v_mul_f32_e64 v10, v21, v22", "v_mul_f32_e64 v11, v22, v23", "v_mul_f32_e64 v12, v23, v24", "v_mul_f32_e64 v13, v24, v25", "v_mul_f32_e64 v14, v25, v26", "v_mul_f32_e64 v15, v26, v27", "v_mul_f32_e64 v16, v27, v28", "v_mul_f32_e64 v17, v28, v29")
BODY(k_mov_dpp, "v_mov_b32_dpp v10, v21 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v11, v22 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v12, v23 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v13, v24 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v14, v25 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v15, v26 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v16, v27 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v17, v28 row_shr:1 row_mask:0xf bank_mask:0xf")

typedef void (*kern_t)(int, unsigned long long *);
int run(const char *name, kern_t k, int wps, unsigned long long *d_cyc)
{
    const int reps = 2000, grid = 1024 * wps;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int it = 0; it < 2; it++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, reps, d_cyc);
        CHECK(hipEventRecord(e1, 0)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> c(2 * grid);
    CHECK(hipMemcpy(c.data(), d_cyc, 2 * grid * 8, hipMemcpyDeviceToHost));
    double t = 0, w = 0; for (int i = 0; i < grid; i++) { t += c[2 * i]; w += c[2 * i + 1]; }
    const double ghz = t / (w * 10);
    const double instr_per_simd = 128.0 * reps * wps;
    printf("%-16s waves/SIMD=%d : %.3f ms, clock %.3f GHz, %.2f SIMD cycles per instruction\n", name, wps, ms, ghz, ms * 1e6 * ghz / instr_per_simd);
    return 0;
}

int main()
{
    unsigned long long *d_cyc; CHECK(hipMalloc(&d_cyc, 2 * 1024 * 8 * 8));
    struct { const char *n; kern_t k; } ks[] = {{"mul diffbank", k_mul_diffbank}, {"mul samebank", k_mul_samebank}, {"mul sgpr", k_mul_sgpr}, {"mul e64", k_mul_e64}, {"sub diffbank", k_sub_diffbank},
        {"fma diffbank", k_fma_diffbank}, {"fma samebank", k_fma_samebank}, {"cmp vcc", k_cmp_vcc}, {"cmp sgpr", k_cmp_sgpr}, {"cndmask", k_cndmask}, {"min/max", k_minmax}, {"rcp", k_rcp}, {"cmp_class", k_class},
        {"mul dep8", k_mul_dep}, {"mul chain", k_mul_chain}, {"mov dpp", k_mov_dpp}};
    for (int wps : {1, 2, 4, 8})
        for (auto &e : ks) if (run(e.n, e.k, wps, d_cyc)) return 1;
    return 0;
}
