// Semantics probe for buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950: where do the 64 x 16 bytes of one wave
// instruction land (M0 base + lane * 16 + immediate), and what do lanes whose offset fails the buffer range check write?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/dma_probe.hip -o tools/probes/dma_probe && tools/probes/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(const unsigned* src, unsigned bytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;   // 16 KB poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)bytes, 0x00020000);
    // lane l reads 16 bytes at element (63 - l) * 4 (reversed, to see the lane -> LDS mapping); lanes 8..15 are out of range
    unsigned off = (unsigned)(63 - lane) * 16 + wave * 1024;
    if (lane >= 8 && lane < 16) off = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 2048), 16, off, 0, 32, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += blockDim.x) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}

int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 16384);
    hipMalloc(&o, 16384);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(128), 16384, 0, d, 2048u * 4, o);
    std::vector<unsigned> r(4096);
    hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("wave %d (LDS base %d, imm 32):\n", w, w * 2048);
        for (int slot = 0; slot < 68; ++slot) {
            const unsigned* p = &r[(w * 2048 + slot * 16) / 4];
            printf("  slot %2d @%5d: %08x %08x %08x %08x\n", slot, w * 2048 + slot * 16, p[0], p[1], p[2], p[3]);
        }
    }
    return 0;
}
