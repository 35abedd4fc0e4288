// mfma_dep_dst.hip — reproducer for the round-6 finding (tools/experiments/README.md): on gfx950 a dependent MFMA whose destination
// differs from its source C, issued right behind the MFMA that produces that C, reads a partly stale C.
//   D1 = A1 (16x32 fp16) . B1 + 0          v_mfma_f32_16x16x32_f16 acc, a1, b1, 0
//   D2 = A2 (16x16 fp16) . B2 + D1         v_mfma_f32_16x16x16_f16 dst, a2, b2, acc        dst == acc (in place)  |  dst != acc, n wait states between
// Small integer inputs: every product and sum is exact in fp32, so any difference from the host's result is the hardware's.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NOPS>      // MODE 0: in place; 1: different destination with NOPS x `s_nop 0` between the two; 2..7 below
__global__ void k(const f16x8* a1, const f16x8* b1, const f16x4* a2, const f16x4* b2, f32x4* out) {
    const int l = threadIdx.x;
    const f16x8 A1 = a1[l], B1 = b1[l];
    const f16x4 A2 = a2[l], B2 = b2[l];
    f32x4 acc, dst;
    if constexpr (MODE == 0) {
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc) : "v"(A1), "v"(B1), "v"(A2), "v"(B2));
        out[l] = acc;
    } else if constexpr (MODE == 5) {          // K32 -> K16, IN PLACE, NOPS wait states between
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                     ".rept %5\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc) : "v"(A1), "v"(B1), "v"(A2), "v"(B2), "n"(NOPS));
        out[l] = acc;
    } else if constexpr (MODE == 2 || MODE == 6) {   // K32 -> K32 (the second one on the same operands): in place (2) / other destination (6)
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n\t"
                     ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x32_f16 %1, %2, %3, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc), "=&v"(dst) : "v"(A1), "v"(B1), "n"(NOPS));
        if (MODE == 2) asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                     "s_nop 15\n\ts_nop 15" : "=&v"(dst) : "v"(A1), "v"(B1));
        out[l] = dst;
    } else if constexpr (MODE == 3 || MODE == 7) {   // K16 -> K16: in place (3) / other destination with NOPS (7)
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %2, %3, 0\n\t"
                     ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x16_f16 %1, %2, %3, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc), "=&v"(dst) : "v"(A2), "v"(B2), "n"(NOPS));
        if (MODE == 3) asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %1, %2, 0\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n\t"
                     "s_nop 15\n\ts_nop 15" : "=&v"(dst) : "v"(A2), "v"(B2));
        out[l] = dst;
    } else if constexpr (MODE == 4) {          // K16 -> K32, in place, NOPS between
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x16_f16 %0, %3, %4, 0\n\t"
                     ".rept %5\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc) : "v"(A1), "v"(B1), "v"(A2), "v"(B2), "n"(NOPS));
        out[l] = acc;
    } else {
        asm volatile("s_nop 7\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n\t"
                     ".rept %6\n\ts_nop 0\n\t.endr\n\t"
                     "v_mfma_f32_16x16x16_f16 %1, %4, %5, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "=&v"(acc), "=&v"(dst) : "v"(A1), "v"(B1), "v"(A2), "v"(B2), "n"(NOPS));
        out[l] = dst;
    }
}

int main() {
    std::vector<_Float16> a1(64 * 8), b1(64 * 8), a2(64 * 4), b2(64 * 4);
    srand(3);
    for (auto& v : a1) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : b1) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : a2) v = (_Float16)(float)(rand() % 7 - 3);
    for (auto& v : b2) v = (_Float16)(float)(rand() % 7 - 3);
    // host: lane l = (g = l / 16, c = l % 16) holds A[row c][k = (g, i)], B[k = (g, i)][col c]; D[row 4g' + i'][col c'] in lane (g', c'), register i'
    std::vector<float> want(64 * 4), want32(64 * 4), want16(64 * 4);
    for (int gp = 0; gp < 4; ++gp)
        for (int cp = 0; cp < 16; ++cp)
            for (int ip = 0; ip < 4; ++ip) {
                const int row = 4 * gp + ip, col = cp;
                float s = 0.f;
                for (int g = 0; g < 4; ++g) {
                    for (int i = 0; i < 8; ++i) s += (float)a1[(g * 16 + row) * 8 + i] * (float)b1[(g * 16 + col) * 8 + i];
                    for (int i = 0; i < 4; ++i) s += (float)a2[(g * 16 + row) * 4 + i] * (float)b2[(g * 16 + col) * 4 + i];
                }
                want[(gp * 16 + cp) * 4 + ip] = s;
                float s32 = 0.f, s16 = 0.f;
                for (int g = 0; g < 4; ++g) {
                    for (int i = 0; i < 8; ++i) s32 += (float)a1[(g * 16 + row) * 8 + i] * (float)b1[(g * 16 + col) * 8 + i];
                    for (int i = 0; i < 4; ++i) s16 += (float)a2[(g * 16 + row) * 4 + i] * (float)b2[(g * 16 + col) * 4 + i];
                }
                want32[(gp * 16 + cp) * 4 + ip] = 2.f * s32;
                want16[(gp * 16 + cp) * 4 + ip] = 2.f * s16;
            }
    f16x8 *da1, *db1; f16x4 *da2, *db2; f32x4* dout;
    hipMalloc(&da1, 1024); hipMalloc(&db1, 1024); hipMalloc(&da2, 512); hipMalloc(&db2, 512); hipMalloc(&dout, 1024);
    hipMemcpy(da1, a1.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db1, b1.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(da2, a2.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), 512, hipMemcpyHostToDevice);
    std::vector<float> got(64 * 4);
    auto check = [&](const char* name, const std::vector<float>& w = std::vector<float>()) {
        const std::vector<float>& ref = w.empty() ? want : w;
        hipDeviceSynchronize();
        hipMemcpy(got.data(), dout, 1024, hipMemcpyDeviceToHost);
        int bad[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) bad[i] += got[l * 4 + i] != ref[l * 4 + i];
        printf("%-58s wrong lanes per accumulator register: %2d %2d %2d %2d\n", name, bad[0], bad[1], bad[2], bad[3]);
    };
#define RUN1(N) do { k<1, N><<<1, 64>>>(da1, db1, da2, db2, dout); char nm[96]; snprintf(nm, sizeof nm, "destination != source C, %2d wait state(s) between", N); check(nm); } while (0)
    k<0, 0><<<1, 64>>>(da1, db1, da2, db2, dout); check("destination == source C (in place), back to back");
    RUN1(0); RUN1(1); RUN1(2); RUN1(3); RUN1(4); RUN1(5); RUN1(6); RUN1(7); RUN1(8); RUN1(9); RUN1(10); RUN1(11); RUN1(12); RUN1(14); RUN1(16); RUN1(20);
#define RUNM(M, N, W, TXT) do { k<M, N><<<1, 64>>>(da1, db1, da2, db2, dout); char nm[96]; snprintf(nm, sizeof nm, TXT ", %2d wait state(s) between", N); check(nm, W); } while (0)
    RUNM(5, 0, want, "K32 -> K16 in place"); RUNM(5, 2, want, "K32 -> K16 in place"); RUNM(5, 4, want, "K32 -> K16 in place"); RUNM(5, 5, want, "K32 -> K16 in place");
    RUNM(4, 0, want, "K16 -> K32 in place"); RUNM(4, 2, want, "K16 -> K32 in place"); RUNM(4, 4, want, "K16 -> K32 in place");
    RUNM(2, 0, want32, "K32 -> K32 in place"); RUNM(6, 0, want32, "K32 -> K32, other destination"); RUNM(6, 2, want32, "K32 -> K32, other destination");
    RUNM(6, 4, want32, "K32 -> K32, other destination"); RUNM(6, 6, want32, "K32 -> K32, other destination"); RUNM(6, 8, want32, "K32 -> K32, other destination");
    RUNM(3, 0, want16, "K16 -> K16 in place"); RUNM(7, 0, want16, "K16 -> K16, other destination"); RUNM(7, 2, want16, "K16 -> K16, other destination");
    RUNM(7, 4, want16, "K16 -> K16, other destination");
    return 0;
}
