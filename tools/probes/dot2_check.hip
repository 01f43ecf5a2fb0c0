#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__global__ void k(const uint4* c, const float4* w, uint4* o1, float* o2) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const uint4 q0 = c[4 * t], q1 = c[4 * t + 1], q2 = c[4 * t + 2], q3 = c[4 * t + 3];
    const float4 wt = w[t];
    const bf16x2 w01 = __builtin_bit_cast(bf16x2, cvt_pk_bf16(wt.x, wt.y)), w23 = __builtin_bit_cast(bf16x2, cvt_pk_bf16(wt.z, wt.w));
    const unsigned c0[4] = {q0.x, q0.y, q0.z, q0.w}, c1[4] = {q1.x, q1.y, q1.z, q1.w}, c2[4] = {q2.x, q2.y, q2.z, q2.w}, c3[4] = {q3.x, q3.y, q3.z, q3.w};
    unsigned o[4];
    for (int d = 0; d < 4; ++d) {
        const bf16x2 tlo = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(c1[d], c0[d], 0x05040100u));
        const bf16x2 thi = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(c1[d], c0[d], 0x07060302u));
        const bf16x2 blo = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(c3[d], c2[d], 0x05040100u));
        const bf16x2 bhi = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(c3[d], c2[d], 0x07060302u));
        float lo = __builtin_amdgcn_fdot2_f32_bf16(tlo, w01, 0.f, false);
        float hi = __builtin_amdgcn_fdot2_f32_bf16(thi, w01, 0.f, false);
        lo = __builtin_amdgcn_fdot2_f32_bf16(blo, w23, lo, false);
        hi = __builtin_amdgcn_fdot2_f32_bf16(bhi, w23, hi, false);
        o[d] = cvt_pk_bf16(lo, hi);
        // f32 reference of the same
        const float ws[4] = {wt.x, wt.y, wt.z, wt.w};
        const unsigned cs[4] = {c0[d], c1[d], c2[d], c3[d]};
        float rl = 0.f, rh = 0.f;
        for (int i = 0; i < 4; ++i) { rl += ws[i] * __uint_as_float(cs[i] << 16); rh += ws[i] * __uint_as_float(cs[i] & 0xffff0000u); }
        o2[8 * t + 2 * d] = rl; o2[8 * t + 2 * d + 1] = rh;
    }
    o1[t] = make_uint4(o[0], o[1], o[2], o[3]);
}
int main() {
    const int N = 256;
    unsigned* hc = (unsigned*)malloc(N * 16 * 4); float* hw = (float*)malloc(N * 16);
    srand(1);
    for (int i = 0; i < N * 16; ++i) { float a = (rand() % 2001 - 1000) / 500.f, b = (rand() % 2001 - 1000) / 500.f; unsigned ua = *(unsigned*)&a >> 16, ub = *(unsigned*)&b >> 16; hc[i] = ua | (ub << 16); }
    for (int i = 0; i < N * 4; ++i) hw[i] = (rand() % 1000) / 1000.f;
    uint4 *dc, *do1; float4* dw; float* do2;
    hipMalloc(&dc, N * 64); hipMalloc(&dw, N * 16); hipMalloc(&do1, N * 16); hipMalloc(&do2, N * 32);
    hipMemcpy(dc, hc, N * 64, hipMemcpyHostToDevice); hipMemcpy(dw, hw, N * 16, hipMemcpyHostToDevice);
    k<<<N / 64, 64>>>(dc, dw, do1, do2);
    unsigned* h1 = (unsigned*)malloc(N * 16); float* h2 = (float*)malloc(N * 32);
    hipMemcpy(h1, do1, N * 16, hipMemcpyDeviceToHost); hipMemcpy(h2, do2, N * 32, hipMemcpyDeviceToHost);
    double maxe = 0, maxr = 0;
    for (int t = 0; t < N; ++t) for (int d = 0; d < 4; ++d) {
        unsigned v = h1[4 * t + d];
        unsigned lo = v << 16, hi = v & 0xffff0000u;
        double e0 = fabs(*(float*)&lo - h2[8 * t + 2 * d]), e1 = fabs(*(float*)&hi - h2[8 * t + 2 * d + 1]);
        if (e0 > maxe) maxe = e0; if (e1 > maxe) maxe = e1;
        if (fabs(h2[8 * t + 2 * d]) > maxr) maxr = fabs(h2[8 * t + 2 * d]);
    }
    printf("dot2 blend vs f32 blend: max |err| %g (max |ref| %g)\n", maxe, maxr);
    int bad = 0;
    for (int t = 0; t < N; ++t) for (int d = 0; d < 4; ++d) {
        unsigned v = h1[4 * t + d]; unsigned lo = v << 16, hi = v & 0xffff0000u;
        double e0 = fabs(*(float*)&lo - h2[8 * t + 2 * d]), e1 = fabs(*(float*)&hi - h2[8 * t + 2 * d + 1]);
        if ((e0 > 0.05 || e1 > 0.05) && bad++ < 6)
            printf("t%d d%d: corners %08x %08x %08x %08x  w %f %f %f %f  dot2 (%f, %f)  f32 (%f, %f)\n", t, d, hc[16 * t + d], hc[16 * t + 4 + d], hc[16 * t + 8 + d],
                   hc[16 * t + 12 + d], hw[4 * t], hw[4 * t + 1], hw[4 * t + 2], hw[4 * t + 3], *(float*)&lo, *(float*)&hi, h2[8 * t + 2 * d], h2[8 * t + 2 * d + 1]);
    }
    printf("bad values: %d of %d\n", bad, N * 4);
    for (int d = 0; d < 0; ++d) {
        unsigned v = h1[d]; unsigned lo = v << 16, hi = v & 0xffff0000u;
        printf("t0 d%d: corners %08x %08x %08x %08x  w %f %f %f %f  dot2 (%f, %f)  f32 (%f, %f)\n", d, hc[d], hc[4 + d], hc[8 + d], hc[12 + d],
               hw[0], hw[1], hw[2], hw[3], *(float*)&lo, *(float*)&hi, h2[2 * d], h2[2 * d + 1]);
    }
    return 0;
}
