// Fused second-order deformable alignment: offset/mask activation + bilinear gather +
// modulated deformable 3x3 convolution on the matrix cores, never materialising the
// im2col "columns" buffer.
//
// Replaces SecondOrderDeformableAlignment.forward after its conv_offset stack
// (guided_diffusion/unet_new.py:877-898: chunk, 10*tanh, + flow.flip(1).repeat, sigmoid,
// torchvision.ops.deform_conv2d) -- the same operator as the reference's own CUDA
// modulated_deformable_im2col_gpu_kernel + addmm (dcn/src/deform_conv_cuda_kernel.cu:571-633,
// dcn/src/deform_conv_cuda.cpp:540-560).
//
//   raw  : [F][H][W][rawLd]  conv_offset output, 27*G channels: o1 | o2 | mask
//   dy,dx(g,k) = M*tanh(raw[2*(g*9+k)+{0,1}]) + flow_g.{y,x};  flow_g = flow1 (g < G/2) else flow2
//   m(g,k)     = sigmoid(raw[18*G + g*9 + k])
//   Y[p][co]   = bias[co] + sum_{k,ci} Wt[co][k][ci] * m(g(ci),k) * bilinear(X[ci], p + pk + d(g(ci),k))
//
// Same MFMA tiling as conv.hip (weights = A operand, gathered pixels = B operand, 64-byte
// swizzled LDS rows); the activation staging is replaced by a 4-corner NHWC gather of
// 16-byte channel chunks (one deformable group = 8 or 16 contiguous channels).
#include "common.h"

namespace {

struct DcnArgs {
    const void* x[2];  // two input segments (feat_prop | feat_n2), C/2 channels each
    int xLd[2];
    int Cin;           // total input channels (2c)
    const void* raw; int rawLd;
    const float* flow1;  // [F][H][W][2] (dx,dy) or null (zero)
    const float* flow2;
    const void* w;       // [Cout][9][Cin]
    const float* bias;
    void* y; int yLd;
    int F, H, W, Cout, G;
    float maxMag;
    long P;
    int nCoTiles;
};

template <typename E> struct MmaD;
template <> struct MmaD<bf16_t> {
    static constexpr int BKE = 32;
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * i + half; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]),
                                                      __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]),
                                                      __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
    }
};
template <> struct MmaD<float> {
    static constexpr int BKE = 16;
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * half + i; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        const float* af = reinterpret_cast<const float*>(a);
        const float* bf = reinterpret_cast<const float*>(b);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// block tile: 64 couts x 64 pixels, 4 waves as 2 x 2 (one 32x32 accumulator each)
template <typename E>
__global__ __launch_bounds__(256) void dcn_kernel(DcnArgs a) {
    constexpr int BKE = MmaD<E>::BKE;
    constexpr int VEC = ET<E>::VEC;
    constexpr int TC = 64, TP = 64;
    __shared__ __attribute__((aligned(16))) char smem[2 * (TC + TP) * 64];
    constexpr int BUF = (TC + TP) * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave >> 1, wp = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int coTile = bid % a.nCoTiles;
    const long p0 = (long)(bid / a.nCoTiles) * TP;
    const int co0 = coTile * TC;

    const int chunk = tid & 3, srow = tid >> 2;
    long p = p0 + srow;
    const bool pvalid = p < a.P;
    if (!pvalid) p = 0;
    const int pw = (int)(p % a.W);
    const int ph = (int)((p / a.W) % a.H);
    const long pf = p / ((long)a.W * a.H);
    float2 fl1 = make_float2(0.f, 0.f), fl2 = make_float2(0.f, 0.f);
    if (a.flow1) fl1 = *reinterpret_cast<const float2*>(a.flow1 + p * 2);
    if (a.flow2) fl2 = *reinterpret_cast<const float2*>(a.flow2 + p * 2);
    const E* rawp = reinterpret_cast<const E*>(a.raw) + p * a.rawLd;
    const int cpg = a.Cin / a.G;
    const int halfC = a.Cin / 2;

    uint4 xreg, wreg;
    int tap = 0, cb = 0;
    const int cbPerTap = a.Cin / BKE;

    auto issue = [&]() {
        const int c = cb * BKE + chunk * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        if (pvalid) {
            const int g = c / cpg;
            const int o = g * 9 + tap;
            const float ry = ET<E>::ld(rawp + 2 * o), rx = ET<E>::ld(rawp + 2 * o + 1);
            const float rm = ET<E>::ld(rawp + 18 * a.G + o);
            const float2 fl = g < a.G / 2 ? fl1 : fl2;
            const float dy = a.maxMag * tanhf(ry) + fl.y;
            const float dx = a.maxMag * tanhf(rx) + fl.x;
            const float mk = 1.f / (1.f + __expf(-rm));
            const float sy = (float)(ph - 1 + tap / 3) + dy;
            const float sx = (float)(pw - 1 + tap % 3) + dx;
            if (sy > -1.f && sx > -1.f && sy < (float)a.H && sx < (float)a.W) {
                const float fy = floorf(sy), fx = floorf(sx);
                const int y0 = (int)fy, x0 = (int)fx;
                const float ay = sy - fy, ax = sx - fx;
                const float wgt[4] = {(1.f - ay) * (1.f - ax), (1.f - ay) * ax, ay * (1.f - ax), ay * ax};
                const int seg = c >= halfC;
                const E* xb = reinterpret_cast<const E*>(a.x[seg]) + (c - seg * halfC);
                const int ld = a.xLd[seg];
                const long fb = pf * a.H * a.W;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
                    if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
                        float v[VEC];
                        Vec16<E>::load(xb + (fb + (long)yy * a.W + xx) * ld, v);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wgt[q], v[k], acc[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] *= mk;
            }
        }
        if constexpr (sizeof(E) == 4) {
            xreg = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                              __float_as_uint(acc[3]));
        } else {
            xreg = make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]),
                              pack2bf(acc[6], acc[7]));
        }
        const int n = co0 + srow;
        if (n < a.Cout)
            wreg = *reinterpret_cast<const uint4*>(reinterpret_cast<const E*>(a.w) + ((long)n * 9 + tap) * a.Cin + c);
        else
            wreg = make_uint4(0, 0, 0, 0);
    };
    auto advance = [&]() {
        if (++cb == cbPerTap) {
            cb = 0;
            ++tap;
        }
    };
    auto write_lds = [&](int buf) {
        char* base = smem + buf * BUF;
        *reinterpret_cast<uint4*>(base + lds_off(srow, chunk)) = wreg;
        *reinterpret_cast<uint4*>(base + TC * 64 + lds_off(srow, chunk)) = xreg;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = 9 * cbPerTap;
    const int lr = lane & 31, lh = lane >> 5;
    issue();
    advance();
    write_lds(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) {
            issue();
            advance();
        }
        const char* wb = smem + cur * BUF;
        const char* xb = wb + TC * 64;
        uint4 af[2], bf[2];
        af[0] = *reinterpret_cast<const uint4*>(wb + lds_off(wc * 32 + lr, MmaD<E>::chunk(0, lh)));
        af[1] = *reinterpret_cast<const uint4*>(wb + lds_off(wc * 32 + lr, MmaD<E>::chunk(1, lh)));
        bf[0] = *reinterpret_cast<const uint4*>(xb + lds_off(wp * 32 + lr, MmaD<E>::chunk(0, lh)));
        bf[1] = *reinterpret_cast<const uint4*>(xb + lds_off(wp * 32 + lr, MmaD<E>::chunk(1, lh)));
        MmaD<E>::run(af, bf, acc);
        if (more) write_lds(cur ^ 1);
        __syncthreads();
    }
    const long po = p0 + wp * 32 + lr;
    if (po >= a.P) return;
    E* y = reinterpret_cast<E*>(a.y);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int co = co0 + wc * 32 + 8 * g + 4 * lh;
        if (co >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[4 * g + e] + (a.bias ? a.bias[co + e] : 0.f);
        E* dst = y + po * a.yLd + co;
        if constexpr (sizeof(E) == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 pk;
            pk.x = pack2bf(v[0], v[1]);
            pk.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = pk;
        }
    }
}

}  // namespace

extern "C" int flair_dcn_align(const flair_dcn_params* p, const void* x0, const void* x1, const void* raw,
                               const float* flow1, const float* flow2, const void* w, const float* bias, void* y,
                               hipStream_t stream) {
    FLAIR_CHECK(p && x0 && x1 && raw && w && y, "flair_dcn_align: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_dcn_align: bad dtype");
    const int vec = p->dtype == FLAIR_BF16 ? 8 : 4, bke = p->dtype == FLAIR_BF16 ? 32 : 16;
    FLAIR_CHECK(p->G > 0 && p->G % 2 == 0 && p->Cin % p->G == 0 && (p->Cin / p->G) % vec == 0 &&
                    (p->Cin / 2) % bke == 0,
                "flair_dcn_align: Cin=%d G=%d not supported", p->Cin, p->G);
    FLAIR_CHECK(p->Cout % 4 == 0 && p->raw_ld >= 27 * p->G, "flair_dcn_align: Cout / raw_ld");
    DcnArgs a;
    a.x[0] = x0; a.x[1] = x1; a.xLd[0] = p->x_ld[0]; a.xLd[1] = p->x_ld[1];
    a.Cin = p->Cin; a.raw = raw; a.rawLd = p->raw_ld; a.flow1 = flow1; a.flow2 = flow2;
    a.w = w; a.bias = bias; a.y = y; a.yLd = p->y_ld;
    a.F = p->F; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.G = p->G; a.maxMag = p->max_residue_magnitude;
    a.P = (long)p->F * p->H * p->W;
    a.nCoTiles = cdiv(p->Cout, 64);
    const int grid = cdiv(a.P, 64) * a.nCoTiles;
    if (p->dtype == FLAIR_BF16)
        hipLaunchKernelGGL(dcn_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(dcn_kernel<float>, dim3(grid), dim3(256), 0, stream, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
