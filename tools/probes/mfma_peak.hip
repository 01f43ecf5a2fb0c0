// What the matrix pipe of this box sustains with nothing else in the way: every wave runs independent v_mfma_f32_32x32x16_bf16 chains out of
// registers (no LDS, no memory), 1 / 2 / 4 waves per SIMD on all CUs, for launches of ~50 us .. ~20 ms.  Prints TFLOP/s against wall time
// (HIP events) and the s_memtime ticks per microsecond.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x + 2 * e)); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, (size_t)cus * 4 * 1024 * 4);
    hipMalloc(&ticks, (size_t)cus * 4 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {             // waves per SIMD: workgroups of 4 waves, wps workgroups per CU
        for (int iters : {250, 2500, 25000, 250000}) {
            const int grid = cus * wps;
            mfma_loop<4><<<grid, 256>>>(out, ticks, iters);      // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<4><<<grid, 256>>>(out, ticks, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            const double flop = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 16);
            printf("waves/SIMD %d  iters %6d  %9.1f us  %7.1f TFLOP/s  (%.3f of 2500)  s_memtime %.1f ticks/us  MFMA every %.1f ticks per SIMD\n", wps, iters,
                   ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15, t / (ms * 1e3), (double)t / ((double)iters * 16 * wps));
        }
    }
    return 0;
}
