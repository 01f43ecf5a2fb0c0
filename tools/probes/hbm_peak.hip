// What the memory system of this box sustains for plain streaming kernels: read-only (sum), write-only (fill), copy and read-read-write (a + b -> c)
// over buffers far larger than the 256 MB Infinity Cache, 16-byte accesses, grid-stride.   hipcc --offload-arch=gfx950 -O3 hbm_peak.hip -o hbm_peak
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, size_t n, unsigned* out) {
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = a[i];
        s ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        a[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_add(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        c[i] = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    uint4 *a, *b, *c;
    unsigned* out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&out, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 8192, 32768}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_read<<<grid, 256>>>(a, n, out);
                else if (mode == 1) k_write<<<grid, 256>>>(a, n);
                else if (mode == 2) k_copy<<<grid, 256>>>(a, b, n);
                else k_add<<<grid, 256>>>(a, b, c, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = bytes * (mode == 0 || mode == 1 ? 1.0 : mode == 2 ? 2.0 : 3.0);
            printf("grid %6d  %-28s %8.1f us  %6.2f TB/s  (%.3f of 8)\n", grid,
                   mode == 0 ? "read 1 GiB" : mode == 1 ? "write 1 GiB" : mode == 2 ? "copy 1 GiB -> 1 GiB" : "a + b -> c (1 GiB each)", best * 1e3, moved / (best * 1e-3) / 1e12,
                   moved / (best * 1e-3) / 8e12);
        }
    }
    return 0;
}
