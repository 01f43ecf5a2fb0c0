// How many wait states does gfx950 need between v_dot2c_f32_bf16 and an instruction that reads its result?  (hipcc 7.2 inserts none and
// the hardware does not interlock: a dependent v_cvt_pk_bf16_f32 two instructions later read the OLD accumulator.)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NOPS, int KIND>
__global__ void k(const unsigned* in, float* o) {
    unsigned a = in[threadIdx.x & 1], b = in[2];
    float acc = 100.f, r;
    if constexpr (KIND == 0) {        // dot2c -> v_mov
        if constexpr (NOPS == 0) asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n v_mov_b32 %1, %0" : "+v"(acc), "=v"(r) : "v"(a), "v"(b));
        else asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n s_nop %4\n v_mov_b32 %1, %0" : "+v"(acc), "=v"(r) : "v"(a), "v"(b), "n"(NOPS - 1));
    } else if constexpr (KIND == 1) { // dot2c -> dot2c (accumulate chain)
        if constexpr (NOPS == 0) asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n v_dot2c_f32_bf16 %0, %2, %3\n s_nop 7\n s_nop 7\n v_mov_b32 %1, %0" : "+v"(acc), "=v"(r) : "v"(a), "v"(b));
        else asm volatile("v_dot2c_f32_bf16 %0, %2, %3\n s_nop %4\n v_dot2c_f32_bf16 %0, %2, %3\n s_nop 7\n s_nop 7\n v_mov_b32 %1, %0" : "+v"(acc), "=v"(r) : "v"(a), "v"(b), "n"(NOPS - 1));
    } else {                          // v_perm_b32 (VALU write of a source) -> dot2c
        unsigned a2;
        if constexpr (NOPS == 0) asm volatile("v_perm_b32 %4, %2, %2, %5\n v_dot2c_f32_bf16 %0, %4, %3\n s_nop 7\n s_nop 7\n v_mov_b32 %1, %0" : "+v"(acc), "=&v"(r), "+v"(a), "+v"(b), "=&v"(a2) : "v"(0x07060100u));
        else asm volatile("v_perm_b32 %4, %2, %2, %5\n s_nop %6\n v_dot2c_f32_bf16 %0, %4, %3\n s_nop 7\n s_nop 7\n v_mov_b32 %1, %0" : "+v"(acc), "=&v"(r), "+v"(a), "+v"(b), "=&v"(a2) : "v"(0x07060100u), "n"(NOPS - 1));
    }
    o[threadIdx.x] = r;
}
template <int NOPS, int KIND>
void run(const unsigned* di, float* d, float want) {
    k<NOPS, KIND><<<1, 64>>>(di, d);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("  %d wait states: %8.3f%s", NOPS, h[0], h[0] == want ? " ok " : " BAD");
}
int main() {
    unsigned h[3] = {0x40003f80u, 0x40003f80u, 0x3e803f00u};   // (1, 2) . (0.5, 0.25) = 1.0
    unsigned* di; float* d; hipMalloc(&di, 64); hipMalloc(&d, 256);
    hipMemcpy(di, h, 12, hipMemcpyHostToDevice);
    printf("dot2c -> v_mov (want 101):\n"); run<0, 0>(di, d, 101); run<1, 0>(di, d, 101); run<2, 0>(di, d, 101); run<3, 0>(di, d, 101); run<4, 0>(di, d, 101); run<6, 0>(di, d, 101); run<8, 0>(di, d, 101); printf("\n");
    printf("dot2c -> dot2c same accumulator (want 102):\n"); run<0, 1>(di, d, 102); run<1, 1>(di, d, 102); run<2, 1>(di, d, 102); run<3, 1>(di, d, 102); run<4, 1>(di, d, 102); run<6, 1>(di, d, 102); run<8, 1>(di, d, 102); printf("\n");
    printf("v_perm_b32 -> dot2c source (want 101):\n"); run<0, 2>(di, d, 101); run<1, 2>(di, d, 101); run<2, 2>(di, d, 101); run<4, 2>(di, d, 101); printf("\n");
    return 0;
}
