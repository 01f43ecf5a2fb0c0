// Probe: how fast can one CU run the halo-conv inner loop (9 taps x 32 channels out of LDS) for
// different per-wave register tiles?  No global traffic inside the loop: LDS fragment reads + MFMA only.
//   hipcc -O3 --offload-arch=gfx950 lds_mfma_probe.hip -o lds_mfma_probe && ./lds_mfma_probe
// Shapes (couts x rows per wave, K split over wave halves or not):
//   0: 2x1 full K, 8 waves   (throughput kernel)      1: 2x1 K-split, 16 waves (per-frame kernel)
//   2: 2x2 K-split, 8 waves                            3: 2x2 full K, 4 waves
//   4: 2x4 K-split, 4 waves                            5: 2x4 full K, 2 waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int HW_ = 34, PITCH = 80, TH = 8;
constexpr int HALO_BYTES = (TH + 2) * HW_ * PITCH, W_BYTES = 64 * 9 * PITCH;

template <int ROWS, bool KSPLIT, int NW>
__global__ __launch_bounds__(64 * NW) void probe(float* out, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    for (int i = tid; i < (HALO_BYTES + W_BYTES) / 4; i += blockDim.x)
        reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20);   // small bf16-ish values
    __syncthreads();
    constexpr int NRG = TH / ROWS;
    const int rg = wave % NRG, kh2 = KSPLIT ? wave / NRG : 0;
    const char* sh = smem;
    const char* sw = smem + HALO_BYTES;
    f32x16 acc[ROWS][2];
    for (int j = 0; j < ROWS; ++j)
        for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    constexpr int NK = KSPLIT ? 1 : 2;     // k-halves this wave multiplies
    for (int c = 0; c < nchunks; ++c) {
        const char* hb = sh + ((rg * ROWS * HW_) + lr) * PITCH;
        const char* wb0 = sw + (lr * 9) * PITCH;
        const char* wb1 = sw + ((lr + 32) * 9) * PITCH;
        uint4 fa[2][2][NK], fb[2][ROWS][NK];
        auto load_tap = [&](int set, int tap9) {
            const int kh = tap9 / 3, kw = tap9 % 3;
#pragma unroll
            for (int h = 0; h < NK; ++h) {
                const int ck = 16 * (2 * (KSPLIT ? kh2 : h) + lh);
                fa[set][0][h] = *reinterpret_cast<const uint4*>(wb0 + tap9 * PITCH + ck);
                fa[set][1][h] = *reinterpret_cast<const uint4*>(wb1 + tap9 * PITCH + ck);
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                    fb[set][j][h] = *reinterpret_cast<const uint4*>(hb + ((j + kh) * HW_ + kw) * PITCH + ck);
            }
        };
        load_tap(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (2 + ROWS) * NK, 0);
#pragma unroll
        for (int tap9 = 0; tap9 < 9; ++tap9) {
            const int set = tap9 & 1;
            if (tap9 < 8) load_tap(set ^ 1, tap9 + 1);
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < NK; ++h)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[set][i][h]), __builtin_bit_cast(bf16x8, fb[set][j][h]),
                            acc[j][i], 0, 0, 0);
            if (tap9 < 8) __builtin_amdgcn_sched_group_barrier(0x100, (2 + ROWS) * NK, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2 * ROWS * NK, 0);
        }
        __syncthreads();      // the real kernels have (at least) one barrier per chunk
    }
    float s = 0.f;
    for (int j = 0; j < ROWS; ++j)
        for (int i = 0; i < 2; ++i)
            for (int r = 0; r < 16; ++r) s += acc[j][i][r];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

template <int ROWS, bool KSPLIT, int NW>
void run(const char* name, float* out, int grid) {
    const int nchunks = 200;
    const size_t lds = HALO_BYTES + W_BYTES;
    hipFuncSetAttribute((const void*)probe<ROWS, KSPLIT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<ROWS, KSPLIT, NW>), dim3(grid), dim3(64 * NW), lds, 0, out, nchunks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double blocksPerCu = grid / 256.0;
    const double usPerChunkPerCu = ms * 1e3 / nchunks;              // all co-resident blocks advance together
    const double flopPerChunk = 2.0 * 256 * 64 * 288 * blocksPerCu;  // 256 px x 64 couts x (9 taps x 32 ch)
    printf("%-28s grid %4d  %.3f us per chunk-round per CU  -> %.2f TFLOP/s per CU (peak 9.77), chip %.0f TFLOP/s\n", name,
           grid, usPerChunkPerCu, flopPerChunk / usPerChunkPerCu * 1e-6, flopPerChunk / usPerChunkPerCu * 1e-6 * 256);
}

int main() {
    float* out;
    hipMalloc(&out, 64 << 20);
    for (int grid : {256, 512}) {
        run<1, false, 8>("0: 2x1 full K, 8 waves", out, grid);
        run<1, true, 16>("1: 2x1 K-split, 16 waves", out, grid);
        run<2, true, 8>("2: 2x2 K-split, 8 waves", out, grid);
        run<2, false, 4>("3: 2x2 full K, 4 waves", out, grid);
        run<4, true, 4>("4: 2x4 K-split, 4 waves", out, grid);
        run<4, false, 2>("5: 2x4 full K, 2 waves", out, grid);
    }
    hipDeviceSynchronize();
    return 0;
}
