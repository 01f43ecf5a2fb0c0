// Probe: start-to-end duration of an EMPTY kernel as a function of grid, workgroup size and LDS request
// (rocprofv3 --kernel-trace gives the durations):  hipcc -O3 --offload-arch=gfx950 launch_probe.hip -o launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Args { const void* p[8]; int v[40]; };      // a ConvArgs-sized kernarg block
__global__ void k_empty(Args a) { extern __shared__ char s[]; if (a.v[0] == 12345) s[threadIdx.x] = 1; }
int main() {
    Args a{};
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int cfg[][3] = {{256, 64, 0}, {256, 128, 0}, {256, 256, 0}, {256, 512, 0}, {256, 1024, 0}, {256, 512, 73 * 1024},
                          {256, 1024, 146 * 1024}, {512, 256, 0}, {1024, 256, 0}, {2048, 256, 0}, {512, 512, 73 * 1024},
                          {1792, 512, 73 * 1024}, {4096, 512, 73 * 1024}};
    for (int it = 0; it < 12; ++it)
        for (auto& c : cfg) hipLaunchKernelGGL(k_empty, dim3(c[0]), dim3(c[1]), c[2], 0, a);
    hipDeviceSynchronize();
    printf("ok\n");
    return 0;
}
