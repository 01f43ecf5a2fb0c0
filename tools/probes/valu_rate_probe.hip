// Issue rate of the vector instructions the alignment kernel's blend could be built from (gfx950): cycles per wave-instruction
// with 1, 2 and 4 waves per SIMD, 8 independent chains per wave.   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void probe(unsigned* out, unsigned long long* cyc, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;
    unsigned b = 0x3f803f80u + threadIdx.x, c = 0x40004000u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) {
            REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 1) {
            REP8(asm volatile("v_dot2c_f32_bf16 %0, %8, %9\n v_dot2c_f32_bf16 %1, %8, %9\n v_dot2c_f32_bf16 %2, %8, %9\n v_dot2c_f32_bf16 %3, %8, %9\n"
                              "v_dot2c_f32_bf16 %4, %8, %9\n v_dot2c_f32_bf16 %5, %8, %9\n v_dot2c_f32_bf16 %6, %8, %9\n v_dot2c_f32_bf16 %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 2) {
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                              "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                              : "+v"(*(unsigned long long*)&a0), "+v"(*(unsigned long long*)&a2), "+v"(*(unsigned long long*)&a4), "+v"(*(unsigned long long*)&a6)
                              : "v"(*(unsigned long long*)&b), "v"(*(unsigned long long*)&c));)
        } else if constexpr (OP == 3) {
            REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n"
                              "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 4) {
            REP8(asm volatile("v_lshlrev_b32 %0, 16, %0\n v_and_b32 %1, %8, %1\n v_lshlrev_b32 %2, 16, %2\n v_and_b32 %3, %8, %3\n"
                              "v_lshlrev_b32 %4, 16, %4\n v_and_b32 %5, %8, %5\n v_lshlrev_b32 %6, 16, %6\n v_and_b32 %7, %8, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 5) {
            REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n"
                              "v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (OP == 6) {
            REP8(asm volatile("v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %8\n v_mul_i32_i24 %3, %3, %8\n"
                              "v_mul_i32_i24 %4, %4, %8\n v_mul_i32_i24 %5, %5, %8\n v_mul_i32_i24 %6, %6, %8\n v_mul_i32_i24 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else {
            REP8(asm volatile("v_dot2_f32_bf16 %0, %8, %9, %0\n v_dot2_f32_bf16 %1, %8, %9, %1\n v_dot2_f32_bf16 %2, %8, %9, %2\n v_dot2_f32_bf16 %3, %8, %9, %3\n"
                              "v_dot2_f32_bf16 %4, %8, %9, %4\n v_dot2_f32_bf16 %5, %8, %9, %5\n v_dot2_f32_bf16 %6, %8, %9, %6\n v_dot2_f32_bf16 %7, %8, %9, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int instrPerRep) {
    unsigned* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 200;
    printf("%-22s", name);
    for (int wavesPerSimd : {1, 2, 4}) {
        const int threads = 64 * 4 * wavesPerSimd;     // one workgroup per CU, 4 SIMDs
        probe<OP><<<256, threads>>>(out, cyc, iters);
        probe<OP><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        unsigned long long s = 0;
        for (auto v : h) s += v;
        const double perWaveInstr = (double)s / 256 / (iters * 8.0 * instrPerRep);
        printf("  %d wave(s)/SIMD: %5.2f cyc per wave-instruction (%5.2f per SIMD-instruction)", wavesPerSimd, perWaveInstr, perWaveInstr / wavesPerSimd);
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32", 8);
    run<1>("v_dot2c_f32_bf16", 8);
    run<7>("v_dot2_f32_bf16", 8);
    run<2>("v_pk_fma_f32", 8);
    run<3>("v_perm_b32", 8);
    run<4>("v_lshlrev/v_and", 8);
    run<5>("v_cvt_pk_bf16_f32", 8);
    run<6>("v_mul_i32_i24", 8);
    return 0;
}
