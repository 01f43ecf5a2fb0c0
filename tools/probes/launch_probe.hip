// Probe: cost of one EMPTY kernel inside a hipGraph of back-to-back dependent launches (what a replayed forward pays per
// launch boundary) as a function of workgroup size, LDS request and kernarg size; grid = 256 workgroups.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/launch_probe.hip -o tools/probes/launch_probe && tools/probes/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { const void* p[8]; int v[60]; };      // a ConvArgs-sized kernarg block (304 bytes)
struct Small { const void* p; int v[2]; };
__global__ void k_big(Big a) { extern __shared__ char s[]; if (a.v[0] == 12345) s[threadIdx.x] = 1; }
__global__ void k_small(Small a) { extern __shared__ char s[]; if (a.v[0] == 12345) s[threadIdx.x] = 1; }

template <typename K, typename A>
float time_graph(K kern, A args, int grid, int threads, int lds, hipStream_t st) {
    const int N = 200;
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, args);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return best * 1e3f / N;
}

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{};
    Small s{};
    const int thr[] = {64, 256, 512, 1024};
    const int lds[] = {0, 32 * 1024, 73 * 1024, 116 * 1024, 146 * 1024, 158 * 1024};
    printf("us per empty launch in a 200-launch graph, grid 256 (rows: threads; columns: LDS bytes)\n            ");
    for (int l : lds) printf("%9d", l);
    printf("\n");
    for (int t : thr) {
        printf("big   %4d: ", t);
        for (int l : lds) printf("%9.2f", time_graph(k_big, b, 256, t, l, st));
        printf("\nsmall %4d: ", t);
        for (int l : lds) printf("%9.2f", time_graph(k_small, s, 256, t, l, st));
        printf("\n");
    }
    printf("grid sweep (512 threads, 73 KB, big args): ");
    for (int g : {64, 128, 256, 512, 1024, 2048}) printf(" %d:%.2f", g, time_graph(k_big, b, g, 512, 73 * 1024, st));
    printf("\n");
    return 0;
}
