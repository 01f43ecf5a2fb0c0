// Implicit-GEMM convolution for NHWC clips on gfx950 MFMA.
//
// Replaces every F.conv2d / F.conv3d / F.conv1d(1x1) / nn.Linear-on-pixels call of the
// reference's UNet (guided_diffusion/unet_new.py:240-295 ResBlock convs, :359,:367
// qkv/proj, :455-459 temporal-attention projections, :659-668 BasicVSR++ trunks,
// :859-867 offset stack; mmedit SPyNet 7x7 convs).
//
// Problem shape:  Y[p][co] = act( sum_{tap,ci} X[p + tap][ci] * Wt[co][tap][ci] + bias[co] )
//                            + R0[p][co] + R1[p][co],   then * out_scale
// with p running over the T*H*W pixels of a clip stored [T][H][W][C] (channels
// innermost), zero "same" padding, stride 1, taps = KT*KH*KW.
//
// Mapping to the matrix core: D^T (Cout x pixels) = Wt (Cout x K) * im2col(X)^T (K x pixels),
// i.e. the weights are the MFMA "A" operand and the gathered activation rows the
// "B" operand, so that each lane of the 32x32 accumulator holds 4 consecutive output
// CHANNELS of one pixel per register quad -> 8/16-byte coalesced NHWC stores.
// K is walked tap by tap, input segment by segment (up to 4 separately stored
// channel groups replace torch.cat), 64 bytes of channels per step; both operand
// tiles are staged through LDS in 64-byte rows with a 16-byte-chunk XOR swizzle
// (chunk ^= (row>>2)&3) that makes every ds_read_b128 fragment read conflict-free.
// Global loads for step k+1 are issued before the MFMAs of step k and written to the
// other LDS buffer afterwards (one barrier per step).
#include "common.h"

namespace {

struct ConvArgs {
    const void* x[4];
    int segC[4];   // channels of each input segment (multiple of BKE)
    int segLd[4];  // elements between consecutive pixels of that segment
    int nseg;
    const void* w;      // [Cout][taps][CinTot]
    const float* bias;  // [Cout] or null
    const void* res0;
    const void* res1;
    int res0Ld, res1Ld;
    void* y;
    int yLd;
    int T, H, W;
    int KT, KH, KW;
    int Cout, CinTot;
    int act;
    float outScale;
    long P;  // T*H*W
    int nPixTiles, nCoTiles;
};

template <typename E> struct Mma;

// bf16: one K-step = 32 channels = two 32x32x16 MFMAs per (co-tile, pixel-tile) pair.
template <> struct Mma<bf16_t> {
    static constexpr int BKE = 32;
    // fr[0], fr[1]: the two 16-byte chunks this lane needs from its 64-byte row.
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * i + half; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]),
                                                      __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]),
                                                      __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
    }
};

// f32: one K-step = 16 channels = eight 32x32x2 MFMAs.  Lane half h consumes channels
// 8h..8h+7 of the step (the same permutation on both operands, so the sum is complete).
template <> struct Mma<float> {
    static constexpr int BKE = 16;
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * half + i; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        const float* af = reinterpret_cast<const float*>(a);
        const float* bf = reinterpret_cast<const float*>(b);
#pragma unroll
        for (int s = 0; s < 8; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// TC x TP block tile (couts x pixels), 4 waves arranged WC x WP.
template <typename E, int TC, int TP, int WC, int WP>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int BKE = Mma<E>::BKE;
    constexpr int VEC = ET<E>::VEC;
    constexpr int FC = TC / WC / 32;  // 32x32 fragments per wave along cout
    constexpr int FP = TP / WP / 32;  // ... along pixels
    constexpr int XR = TP / 64;       // activation rows staged per thread
    constexpr int WR = TC / 64;       // weight rows staged per thread
    static_assert(WC * WP == 4 && FC >= 1 && FP >= 1 && XR >= 1 && WR >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf][ W tile (TC rows) | X tile (TP rows) ], 64-byte rows
    constexpr int BUF = (TC + TP) * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int coTile = bid % a.nCoTiles;
    const int pixTile = bid / a.nCoTiles;
    const long p0 = (long)pixTile * TP;
    const int co0 = coTile * TC;

    // ---- per-thread staging coordinates (fixed over the K loop) --------------
    const int chunk = tid & 3;
    const int srow = tid >> 2;  // 0..63
    int xt[XR], xh[XR], xw[XR];
    bool xvalid[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        long p = p0 + srow + 64 * i;
        xvalid[i] = p < a.P;
        if (!xvalid[i]) p = 0;
        int w_ = (int)(p % a.W);
        long q = p / a.W;
        xw[i] = w_;
        xh[i] = (int)(q % a.H);
        xt[i] = (int)(q / a.H);
    }
    const int taps = a.KT * a.KH * a.KW;
    const int pt = a.KT / 2, ph = a.KH / 2, pw = a.KW / 2;

    uint4 xreg[XR], wreg[WR];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    // K-loop state (block uniform)
    int tap = 0, seg = 0, cb = 0, segOff = 0;
    int dt = -pt, dh = -ph, dw = -pw;

    auto issue_loads = [&]() {
        const E* xs = reinterpret_cast<const E*>(a.x[seg]);
        const int ld = a.segLd[seg];
        const int coff = cb * BKE + chunk * VEC;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int t2 = xt[i] + dt, h2 = xh[i] + dh, w2 = xw[i] + dw;
            const bool ok = xvalid[i] && (unsigned)t2 < (unsigned)a.T && (unsigned)h2 < (unsigned)a.H &&
                            (unsigned)w2 < (unsigned)a.W;
            if (ok) {
                const long pix = ((long)t2 * a.H + h2) * a.W + w2;
                xreg[i] = *reinterpret_cast<const uint4*>(xs + pix * ld + coff);
            } else {
                xreg[i] = zero4;
            }
        }
        const E* ws = reinterpret_cast<const E*>(a.w);
        const long kofs = (long)tap * a.CinTot + segOff + coff;
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            const int n = co0 + srow + 64 * j;
            if (n < a.Cout)
                wreg[j] = *reinterpret_cast<const uint4*>(ws + (long)n * taps * a.CinTot + kofs);
            else
                wreg[j] = zero4;
        }
    };
    auto advance = [&]() {
        ++cb;
        if (cb * BKE >= a.segC[seg]) {
            cb = 0;
            segOff += a.segC[seg];
            ++seg;
            if (seg >= a.nseg) {
                seg = 0;
                segOff = 0;
                ++tap;
                ++dw;
                if (dw > pw) {
                    dw = -pw;
                    ++dh;
                    if (dh > ph) {
                        dh = -ph;
                        ++dt;
                    }
                }
            }
        }
    };
    auto write_lds = [&](int buf) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < WR; ++j)
            *reinterpret_cast<uint4*>(base + lds_off(srow + 64 * j, chunk)) = wreg[j];
#pragma unroll
        for (int i = 0; i < XR; ++i)
            *reinterpret_cast<uint4*>(base + TC * 64 + lds_off(srow + 64 * i, chunk)) = xreg[i];
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = taps * (a.CinTot / BKE);
    const int lr = lane & 31, lh = lane >> 5;

    issue_loads();
    advance();
    write_lds(0);
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) {
            issue_loads();
            advance();
        }
        const char* wb = smem + cur * BUF;
        const char* xb = wb + TC * 64;
        uint4 af[FC][2], bfr[FP][2];
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const int row = wc * (TC / WC) + i * 32 + lr;
            af[i][0] = *reinterpret_cast<const uint4*>(wb + lds_off(row, Mma<E>::chunk(0, lh)));
            af[i][1] = *reinterpret_cast<const uint4*>(wb + lds_off(row, Mma<E>::chunk(1, lh)));
        }
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int row = wp * (TP / WP) + j * 32 + lr;
            bfr[j][0] = *reinterpret_cast<const uint4*>(xb + lds_off(row, Mma<E>::chunk(0, lh)));
            bfr[j][1] = *reinterpret_cast<const uint4*>(xb + lds_off(row, Mma<E>::chunk(1, lh)));
        }
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j) Mma<E>::run(af[i], bfr[j], acc[i][j]);
        if (more) write_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane = pixel column, register quad = 4 consecutive couts ----
    E* y = reinterpret_cast<E*>(a.y);
    const E* r0 = reinterpret_cast<const E*>(a.res0);
    const E* r1 = reinterpret_cast<const E*>(a.res1);
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const long p = p0 + wp * (TP / WP) + j * 32 + lr;
        if (p >= a.P) continue;
#pragma unroll
        for (int i = 0; i < FC; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + wc * (TC / WC) + i * 32 + 8 * g + 4 * lh;
                if (co >= a.Cout) continue;  // Cout is a multiple of 4
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
                if (a.bias) {
                    const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
                if (r0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += ET<E>::ld(r0 + p * a.res0Ld + co + e);
                }
                if (r1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += ET<E>::ld(r1 + p * a.res1Ld + co + e);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= a.outScale;
                E* dst = y + p * a.yLd + co;
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
    }
}

template <typename E, int TC, int TP, int WC, int WP>
int launch(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.nPixTiles = cdiv(a.P, TP);
    a.nCoTiles = cdiv(a.Cout, TC);
    const int grid = a.nPixTiles * a.nCoTiles;
    const size_t lds = 2 * (TC + TP) * 64;
    hipLaunchKernelGGL((conv_igemm_kernel<E, TC, TP, WC, WP>), dim3(grid), dim3(256), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// Tile choice: keep >= ~2 workgroups per CU when the problem allows it.
// 0: 128 couts x 128 pixels, 1: 64 x 128, 2: 64 x 64.
int choose_variant(long P, int Cout) {
    const long tiles128 = (long)cdiv(P, 128) * cdiv(Cout, 128);
    if (Cout > 64 && tiles128 >= 512) return 0;
    const long tiles64x128 = (long)cdiv(P, 128) * cdiv(Cout, 64);
    if (tiles64x128 >= 512) return 1;
    return 2;
}

template <typename E>
int dispatch(const ConvArgs& a, hipStream_t s) {
    switch (choose_variant(a.P, a.Cout)) {
        case 0: return launch<E, 128, 128, 2, 2>(a, s);
        case 1: return launch<E, 64, 128, 1, 4>(a, s);
        default: return launch<E, 64, 64, 2, 2>(a, s);
    }
}

}  // namespace

extern "C" int flair_conv_variant(const flair_conv_params* p) {
    if (!p) return -1;
    return choose_variant((long)p->T * p->H * p->W, p->Cout);
}

extern "C" int flair_conv_nhwc(const flair_conv_params* p, const void* const* x, const void* w,
                               const float* bias, const void* res0, const void* res1, void* y,
                               hipStream_t stream) {
    FLAIR_CHECK(p && x && w && y, "flair_conv_nhwc: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_conv_nhwc: bad dtype %d", p->dtype);
    FLAIR_CHECK(p->nseg >= 1 && p->nseg <= 4, "flair_conv_nhwc: nseg %d not in 1..4", p->nseg);
    FLAIR_CHECK(p->T > 0 && p->H > 0 && p->W > 0 && p->Cout > 0, "flair_conv_nhwc: empty shape");
    FLAIR_CHECK((p->KT & 1) && (p->KH & 1) && (p->KW & 1), "flair_conv_nhwc: kernel taps must be odd");
    FLAIR_CHECK(p->Cout % 4 == 0, "flair_conv_nhwc: Cout %d must be a multiple of 4", p->Cout);
    const int bke = p->dtype == FLAIR_BF16 ? 32 : 16;
    const int esz = p->dtype == FLAIR_BF16 ? 2 : 4;
    ConvArgs a{};
    a.CinTot = 0;
    for (int i = 0; i < p->nseg; ++i) {
        FLAIR_CHECK(x[i], "flair_conv_nhwc: segment %d is null", i);
        FLAIR_CHECK(p->seg_c[i] > 0 && p->seg_c[i] % bke == 0,
                    "flair_conv_nhwc: segment %d has %d channels; must be a multiple of %d", i, p->seg_c[i], bke);
        FLAIR_CHECK(p->seg_ld[i] >= p->seg_c[i] && (p->seg_ld[i] * esz) % 16 == 0 &&
                        ((uintptr_t)x[i]) % 16 == 0,
                    "flair_conv_nhwc: segment %d stride/alignment", i);
        a.x[i] = x[i];
        a.segC[i] = p->seg_c[i];
        a.segLd[i] = p->seg_ld[i];
        a.CinTot += p->seg_c[i];
    }
    FLAIR_CHECK(p->y_ld >= p->Cout && (p->y_ld * esz) % 8 == 0 && ((uintptr_t)y) % 16 == 0,
                "flair_conv_nhwc: output stride/alignment");
    FLAIR_CHECK(((uintptr_t)w) % 16 == 0, "flair_conv_nhwc: weight alignment");
    FLAIR_CHECK(!bias || ((uintptr_t)bias) % 16 == 0, "flair_conv_nhwc: bias alignment");
    a.nseg = p->nseg;
    a.w = w;
    a.bias = bias;
    a.res0 = res0;
    a.res1 = res1;
    a.res0Ld = p->res_ld[0];
    a.res1Ld = p->res_ld[1];
    a.y = y;
    a.yLd = p->y_ld;
    a.T = p->T; a.H = p->H; a.W = p->W;
    a.KT = p->KT; a.KH = p->KH; a.KW = p->KW;
    a.Cout = p->Cout;
    a.act = p->act;
    a.outScale = p->out_scale;
    a.P = (long)p->T * p->H * p->W;
    if (p->dtype == FLAIR_BF16) return dispatch<bf16_t>(a, stream);
    return dispatch<float>(a, stream);
}
