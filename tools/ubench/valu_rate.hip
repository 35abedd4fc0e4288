// valu_rate.hip — issue-rate microbenchmark for the VALU forms the fused kernel leans on (gfx950).
// Prints cycles per wave-instruction per SIMD at 1, 2 and 4 waves/SIMD, alone and beside f32 MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP 64
#define ITERS 200

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, float seed, int with_mfma) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f32x2 q = {seed * 0.5f, seed * 0.25f};
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float b = seed * 0.999f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (with_mfma) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc0, 0, 0, 0);
            }
            if constexpr (OP == 0) {  // v_add_f32 x8
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 1) {  // v_max_f32
                asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 2) {  // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 3) {  // v_pk_add_f32 x8 (4 regs pairs, twice)
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
            } else if constexpr (OP == 4) {  // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
            } else if constexpr (OP == 5) {  // v_pk_add_f32 clamp
                asm volatile("v_pk_add_f32 %0, %0, %4 clamp\n v_pk_add_f32 %1, %1, %4 clamp\n v_pk_add_f32 %2, %2, %4 clamp\n v_pk_add_f32 %3, %3, %4 clamp\n"
                             "v_pk_add_f32 %0, %0, %4 clamp\n v_pk_add_f32 %1, %1, %4 clamp\n v_pk_add_f32 %2, %2, %4 clamp\n v_pk_add_f32 %3, %3, %4 clamp\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
            } else if constexpr (OP == 6) {  // v_exp_f32
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (OP == 7) {  // v_cmp + v_cndmask pairs (4 pairs)
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            } else if constexpr (OP == 8) {  // v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n"
                             "v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 9) {  // v_permlane32_swap x8 (4 pairs twice)
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if constexpr (OP == 10) {  // v_mul_f32 clamp (e64)
                asm volatile("v_mul_f32_e64 %0, %0, %8 clamp\n v_mul_f32_e64 %1, %1, %8 clamp\n v_mul_f32_e64 %2, %2, %8 clamp\n v_mul_f32_e64 %3, %3, %8 clamp\n"
                             "v_mul_f32_e64 %4, %4, %8 clamp\n v_mul_f32_e64 %5, %5, %8 clamp\n v_mul_f32_e64 %6, %6, %8 clamp\n v_mul_f32_e64 %7, %7, %8 clamp\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 11) {  // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
            } else if constexpr (OP == 12) {  // v_mov_b32
                asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                             "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if constexpr (OP == 13) {  // nothing but the MFMA (second one to keep the pipe full)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc1, 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] +
              acc0[0] + acc0[1] + acc1[0] + acc1[3];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int n_valu_per_rep8) {
    float* out; long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    for (int mf = 0; mf < 2; ++mf) {
        printf("%-26s %s:", name, mf ? "+1 MFMA/8" : "alone    ");
        for (int wps : {1, 2, 4}) {
            // wps blocks of 256 threads per CU -> wps waves per SIMD
            k<OP><<<256 * wps, 256>>>(out, cyc, 1.0f, mf);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<OP><<<256 * wps, 256>>>(out, cyc, 1.0f, mf);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double n_inst = (double)ITERS * (REP / 8) * n_valu_per_rep8;   // per wave
            // wall-clock cycles per SIMD per instruction at an assumed 2.4 GHz
            const double cyc_wall = ms * 1e-3 * 2.4e9 / (n_inst * wps);
            printf("  wps=%d %6.2f cyc/inst/SIMD (wave clk %6.2f/inst)", wps, cyc_wall, (double)c / n_inst);
        }
        printf("\n");
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_add_f32", 8);
    run<1>("v_max_f32", 8);
    run<2>("v_fma_f32", 8);
    run<3>("v_pk_add_f32", 8);
    run<4>("v_pk_fma_f32", 8);
    run<5>("v_pk_add_f32 clamp", 8);
    run<11>("v_pk_mul_f32", 8);
    run<10>("v_mul_f32 clamp", 8);
    run<6>("v_exp_f32", 8);
    run<7>("v_cmp+v_cndmask", 8);
    run<8>("v_max3_f32", 8);
    run<9>("v_permlane32/16_swap", 8);
    run<12>("v_mov_b32", 8);
    run<13>("mfma_16x16x4_f32 only", 1);
    return 0;
}
