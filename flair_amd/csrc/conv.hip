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
#include <stdlib.h>

#include <type_traits>

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
    int tFast;             // DMA kernel: frame index fastest in the tile order (launches with temporal taps)
    int resPrefetch;       // DMA kernel: 0 no residual prefetch, 1 LDS-DMA touch of the residual lines one chunk ahead
    unsigned segBytes[4];  // addressable bytes of each input segment / of the weights
    unsigned wBytes;
    float* part;           // split-K: f32 partial sums [splitK][P][Cout] (null when splitK == 1)
    int splitK;
    const float* fbias;    // optional per-(frame, channel) bias [T][fbiasLd] (emb added to h)
    int fbiasLd;
    int debug;             // phase-timing switches, compiled in only with -DFLAIR_TIMING_SWITCHES (see FLAIR_DBG)
    int stride;            // spatial stride (1 or 2; im2col path only)
    int tapShift;          // asym_pad: K/2 added to every spatial tap offset (taps start at stride*i)
    int reflect;           // reflect_pad: out-of-frame taps read the mirrored pixel (im2col path)
    int ldsSwz;            // halo kernels: conflict-free lane -> staged-row order of the ds_write_b128 (staged_row)
    float actParam;        // FLAIR_ACT_DCN_OFFSETS: max residue magnitude
    int actPeriod;         // FLAIR_ACT_DCN_OFFSETS: 3 * deform groups
    int Hin, Win;          // input frame size (== H, W when stride == 1); H, W, P describe the OUTPUT
    int esz;               // element size in bytes (2: bf16, 4: f32)
};

// Phase-timing switches of the halo kernel (how profiles/README.md's dissection was measured):
// 1 skip the MFMA phase, 2 skip the in-loop reloads, 3/4/5 return before the first fetch / after
// the first staged chunk / before the epilogue.  A build without -DFLAIR_TIMING_SWITCHES (the
// product build) compiles them out: FLAIR_DBG is the constant 0 and the environment is not read.
#ifdef FLAIR_TIMING_SWITCHES
#define FLAIR_DBG(a) ((a).debug)
#else
#define FLAIR_DBG(a) 0
#endif

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
    // one half of the K-step (16 channels)
    static __device__ __forceinline__ void run1(const uint4& a, const uint4& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
    static constexpr int MFMA_PER_HALF = 1;
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
    static __device__ __forceinline__ void run1(const uint4& a, const uint4& b, f32x16& acc) {
        const float* af = reinterpret_cast<const float*>(&a);
        const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
    static constexpr int MFMA_PER_HALF = 4;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

// 16-byte piece `id` of a staged [rows][64 B] LDS image at 80-byte pitch -> its row (piece = id & 3).  ds_write_b128 is
// banked (address / 4) % 32 over groups of 8 consecutive lanes (MI355X_MICROARCH.md, LDS): two rows r, r+1 of one group
// start 20 banks apart and collide on 4 banks (one extra LDS cycle per group: SQ_LDS_BANK_CONFLICT ~ 1 per LDS instruction
// of the halo kernels, profiles/r02y_conv_pmc_sq.txt), rows r and r+4 start 16 banks apart and fill the 32 banks exactly.
// So a lane group of 8 stages rows (8k + j, 8k + j + 4).  The switch keeps the linear order for A/B runs.
__device__ __forceinline__ int staged_row(int id, bool swizzled) {
    if (!swizzled) return id >> 2;
    const int g = id >> 3;
    return (g >> 2) * 8 + (g & 3) + 4 * ((id >> 2) & 1);
}

// Epilogue for one register quad: 4 consecutive output channels of one pixel.
template <typename E>
__device__ __forceinline__ void store_quad(const ConvArgs& a, long p, int co, float v0, float v1, float v2, float v3) {
    if (co >= a.Cout) return;  // Cout is a multiple of 4
    float v[4] = {v0, v1, v2, v3};
    if (a.bias) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (a.fbias) {
        const float* fb = a.fbias + (p / ((long)a.H * a.W)) * a.fbiasLd + co;
        v[0] += fb[0]; v[1] += fb[1]; v[2] += fb[2]; v[3] += fb[3];
    }
    if (a.act == FLAIR_ACT_DCN_OFFSETS) {
        dcn_offset_act<4>(v, co, a.actParam, a.actPeriod);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
    }
    const E* r0 = reinterpret_cast<const E*>(a.res0);
    const E* r1 = reinterpret_cast<const E*>(a.res1);
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
    E* dst = reinterpret_cast<E*>(a.y) + p * a.yLd + co;
    if constexpr (sizeof(E) == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 pk;
        pk.x = pack2bf(v[0], v[1]);
        pk.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(dst) = pk;
    }
}

// Epilogue of one wave's FC x FP accumulator fragments (32 couts x 32 pixels each), Cout % 8 == 0: after v_permlane32_swap a lane
// holds 8 consecutive couts of its pixel per 16-cout half, so bias / residuals / output move as 16-byte (bf16) or 2 x 16-byte
// (f32) pieces, and the activation is compiled for ONE class (ACT: 0 max(v, slope v) = none / ReLU / LeakyReLU, 1 DCN offsets and
// masks, 2 SiLU, 3 exact GELU), chosen by one wave-uniform branch at the call site (store_quad above -- 8-byte pieces, six
// activation variants inlined per quad -- made conv_igemm_kernel<128,128> 77 KB of code, more than the instruction cache that
// two CUs share).  Branch-free memory access: buffer descriptors based at the wave's first pixel / first cout, an out-of-range
// offset on lanes past P or Cout; per fragment, the bias, frame-bias and residual pieces of both halves are requested in one batch.  With the loads and stores under `if (pvalid && co < Cout)` hipcc put `s_waitcnt vmcnt(0)` behind every
// one: each 16-cout half was a bias round trip that also waited for the previous half's store -- on the 1x1 128 -> 64 convolution
// of 16 x 256^2 the epilogue was 60 us of 110 (profiles/r03_dma_switches.txt, r3k).  All lanes must call this (the swaps cross
// the half-waves).
template <typename E, int ACT, int FC, int FP>
__device__ __forceinline__ void igemm_epilogue(const ConvArgs& a, const f32x16 (&acc)[FC][FP], long pbase, int cobase, int lr, int lh) {
    constexpr unsigned ESZ = sizeof(E);
    constexpr int VEC = ET<E>::VEC, NV = 8 / VEC;               // 16-byte pieces per 8 couts: 1 (bf16) or 2 (f32)
    const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : a.act == FLAIR_ACT_LRELU01 ? 0.1f : 0.2f;
    const unsigned yLdB = (unsigned)a.yLd * ESZ, r0LdB = (unsigned)a.res0Ld * ESZ, r1LdB = (unsigned)a.res1Ld * ESZ;
    constexpr unsigned SPAN = FP * 32;                           // pixels the wave's offsets stay below
    const __amdgpu_buffer_rsrc_t yd = make_rsrc(reinterpret_cast<char*>(a.y) + ((size_t)pbase * a.yLd + cobase) * ESZ, SPAN * yLdB);
    const __amdgpu_buffer_rsrc_t r0d = make_rsrc(reinterpret_cast<const char*>(a.res0) + (a.res0 ? ((size_t)pbase * a.res0Ld + cobase) * ESZ : 0),
                                                 a.res0 ? SPAN * r0LdB : 0u);
    const __amdgpu_buffer_rsrc_t r1d = make_rsrc(reinterpret_cast<const char*>(a.res1) + (a.res1 ? ((size_t)pbase * a.res1Ld + cobase) * ESZ : 0),
                                                 a.res1 ? SPAN * r1LdB : 0u);
    const int coLeft = a.Cout - cobase;                          // couts of the tensor from the wave's first one on (may be <= 0)
    const __amdgpu_buffer_rsrc_t bd = make_rsrc(a.bias ? a.bias + cobase : nullptr, a.bias && coLeft > 0 ? (unsigned)coLeft * 4u : 0u);
    const long hw = (long)a.H * a.W;
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const long p = pbase + j * 32 + lr;
        const bool pvalid = p < a.P;
        const unsigned pl = (unsigned)(j * 32 + lr);
        const float* fbrow = a.fbias ? a.fbias + (pvalid ? p / hw : 0) * a.fbiasLd + cobase : nullptr;
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            // every load of this fragment (two 16-cout halves) is requested before the first use: one round trip per fragment
            uint4 bq[2][2], r0v[2][NV], r1v[2][NV], fq[2][2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int cw = i * 32 + 16 * jj + 8 * lh;
                const bool ok = pvalid && cw < coLeft;
                bq[jj][0] = buf_load16(bd, 4u * (unsigned)cw);              // (past Cout: outside the descriptor, reads 0)
                bq[jj][1] = buf_load16(bd, 4u * (unsigned)cw + 16u);
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    r0v[jj][q] = buf_load16(r0d, ok ? pl * r0LdB + (unsigned)(cw + q * VEC) * ESZ : FLAIR_OOB);
                    r1v[jj][q] = buf_load16(r1d, ok ? pl * r1LdB + (unsigned)(cw + q * VEC) * ESZ : FLAIR_OOB);
                }
                fq[jj][0] = fq[jj][1] = make_uint4(0u, 0u, 0u, 0u);
            }
            if (a.fbias) {                                       // wave-uniform; clamped in-range address on padding lanes
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int cw = i * 32 + 16 * jj + 8 * lh;
                    const float* fp = fbrow + (cw < coLeft ? cw : 0);
                    fq[jj][0] = *reinterpret_cast<const uint4*>(fp);
                    fq[jj][1] = *reinterpret_cast<const uint4*>(fp + 4);
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x16& c = acc[i][j];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[8 * jj + e]), __float_as_uint(c[8 * jj + 4 + e]), false, false);
                    v[e] = __uint_as_float(sw2[0]);
                    v[4 + e] = __uint_as_float(sw2[1]);
                }
                const int cw = i * 32 + 16 * jj + 8 * lh;
                const bool ok = pvalid && cw < coLeft;
                {
                    const uint4 b0 = bq[jj][0], b1 = bq[jj][1], f0 = fq[jj][0], f1 = fq[jj][1];
                    v[0] += __uint_as_float(b0.x) + __uint_as_float(f0.x); v[1] += __uint_as_float(b0.y) + __uint_as_float(f0.y);
                    v[2] += __uint_as_float(b0.z) + __uint_as_float(f0.z); v[3] += __uint_as_float(b0.w) + __uint_as_float(f0.w);
                    v[4] += __uint_as_float(b1.x) + __uint_as_float(f1.x); v[5] += __uint_as_float(b1.y) + __uint_as_float(f1.y);
                    v[6] += __uint_as_float(b1.z) + __uint_as_float(f1.z); v[7] += __uint_as_float(b1.w) + __uint_as_float(f1.w);
                }
                if constexpr (ACT == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
                } else if constexpr (ACT == 1) {
                    dcn_offset_act<8>(v, cobase + cw, a.actParam, a.actPeriod);
                } else if constexpr (ACT == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f));
                }
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    float t[VEC];
                    Vec16<E>::load(reinterpret_cast<const E*>(&r0v[jj][q]), t);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[q * VEC + e] += t[e];
                    Vec16<E>::load(reinterpret_cast<const E*>(&r1v[jj][q]), t);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[q * VEC + e] += t[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= a.outScale;
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    alignas(16) E out[VEC];
                    Vec16<E>::store(out, v + q * VEC);
                    const uint4 ov = *reinterpret_cast<const uint4*>(out);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd,
                                                           (int)(ok ? pl * yLdB + (unsigned)(cw + q * VEC) * ESZ : FLAIR_OOB), 0, 0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 3x3 (x KT) convolution with an LDS-staged HALO tile.
//
// The im2col kernel above re-reads every activation 9 (27) times from L2.  Here a workgroup
// owns TH image rows x 32 columns of one frame and 64 output channels; per K chunk (64 bytes
// of input channels, per temporal tap) it stages the (TH+2) x 34 pixel halo ONCE plus the 9
// spatial taps of the weights, then runs all 9 taps x 2 MFMA k-steps out of LDS:
// 18x fewer barriers per MFMA and ~7x less L2->LDS traffic.  One wavefront per image row:
// lanes = 32 consecutive pixels (MFMA B operand, read at the tap-shifted halo address),
// 64 couts = two A fragments.  Pixel pitch 80 B and weight row pitch 80 B make every
// ds_read_b128 conflict-free (odd multiples of 16 B).  Loads of chunk k+1 are issued before
// the MFMAs of chunk k (register staging), written after a barrier.
template <typename E, int TH, int RPW, int PF>
__global__ __launch_bounds__(64 * TH / RPW, (2 * 64 * TH / RPW + 255) / 256)  // two workgroups per CU
void conv3x3_halo_kernel(ConvArgs a) {
    prefetch_kernargs<sizeof(ConvArgs)>();
    constexpr int BKE = Mma<E>::BKE;
    constexpr int VEC = ET<E>::VEC;
    constexpr int NT = 64 * TH / RPW;              // one wavefront per RPW image rows
    constexpr int HW_ = 34;                        // halo width (32 + 2)
    constexpr int PITCH = 80;                      // bytes per staged pixel / weight row
    constexpr int HALO_ROWS = (TH + 2) * HW_;
    constexpr int HALO_PIECES = (HALO_ROWS + 7) / 8 * 8 * 4;   // whole groups of 8 rows (see staged_row)
    constexpr int W_PIECES = 64 * 9 * 4;
    constexpr int HI = (HALO_PIECES + NT - 1) / NT;
    constexpr int WI = (W_PIECES + NT - 1) / NT;
    constexpr int HALO_BYTES = (TH + 2) * HW_ * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sh = smem;                // halo
    char* sw = smem + HALO_BYTES;   // weights [64][9] rows of PITCH

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int coTile = bid % a.nCoTiles;
    int rest = bid / a.nCoTiles;
    const int tilesW = a.W / 32, tilesH = (a.H + TH - 1) / TH;
    const int tw = rest % tilesW;
    rest /= tilesW;
    const int th = rest % tilesH;
    const int t = rest / tilesH;
    const int h0 = th * TH, w0 = tw * 32, co0 = coTile * 64;
    const int taps = a.KT * 9;
    const int pt = a.KT / 2;

    uint4 hreg[PF][HI], wreg[PF][WI];              // PF register sets = prefetch depth in chunks
    int dt = -pt, seg = 0, cb = 0, segOff = 0;
    // skip temporal taps that fall outside the clip (block-uniform)
    auto dt_valid = [&](int d) { return (unsigned)(t + d) < (unsigned)a.T; };
    while (dt <= pt && !dt_valid(dt)) ++dt;

    // per-thread piece coordinates, fixed over the K loop: pixel index inside the frame (or -1
    // outside the image -> zero padding through the buffer range check) and weight row offset
    constexpr unsigned ESZ = sizeof(E);
    int hpix[HI];
    unsigned hq[HI], woff[WI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int id = i * NT + tid;
        const int pix = staged_row(id, a.ldsSwz);
        const int r = pix / HW_, c = pix % HW_;
        const int hh = h0 + r - 1, ww = w0 + c - 1;
        const bool ok = pix < HALO_ROWS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        hpix[i] = ok ? hh * a.W + ww : -1;
        hq[i] = (id & 3) * VEC * ESZ;
    }
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w, a.wBytes);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int id = i * NT + tid;
        const int row = staged_row(id, a.ldsSwz);
        const int co = row / 9, tap9 = row % 9;
        const bool ok = id < W_PIECES && co0 + co < a.Cout;
        woff[i] = ok ? (unsigned)(((co0 + co) * taps + tap9) * a.CinTot + (id & 3) * VEC) * ESZ : FLAIR_OOB;
    }

    auto issue = [&](uint4 (&hr)[HI], uint4 (&wr)[WI]) {
        // buffer resource = ONE input frame (clips may exceed the 2 GiB a resource can address)
        const unsigned ld = (unsigned)a.segLd[seg] * ESZ;
        const char* frame = reinterpret_cast<const char*>(a.x[seg]) + (size_t)(t + dt) * a.H * a.W * ld;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(frame, a.segBytes[seg]);
        const unsigned cofs = (unsigned)(cb * BKE) * ESZ;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            const unsigned off = hpix[i] >= 0 ? (unsigned)hpix[i] * ld + cofs + hq[i] : FLAIR_OOB;
            hr[i] = buf_load16(xr, off);
        }
        const unsigned kofs = (unsigned)((dt + pt) * 9 * a.CinTot + segOff + cb * BKE) * ESZ;
#pragma unroll
        for (int i = 0; i < WI; ++i) wr[i] = buf_load16(wrs, woff[i] == FLAIR_OOB ? FLAIR_OOB : woff[i] + kofs);
    };
    auto advance = [&]() {   // the K loop is exhausted when dt > pt
        ++cb;
        if (cb * BKE >= a.segC[seg]) {
            cb = 0;
            segOff += a.segC[seg];
            if (++seg >= a.nseg) {
                seg = 0;
                segOff = 0;
                ++dt;
                while (dt <= pt && !dt_valid(dt)) ++dt;
            }
        }
    };
    auto write_lds = [&](const uint4 (&hr)[HI], const uint4 (&wr)[WI]) {
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            const int id = i * NT + tid;
            const int row = staged_row(id, a.ldsSwz);
            if (row < HALO_ROWS) *reinterpret_cast<uint4*>(sh + row * PITCH + (id & 3) * 16) = hr[i];
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int id = i * NT + tid;
            if (id < W_PIECES) *reinterpret_cast<uint4*>(sw + staged_row(id, a.ldsSwz) * PITCH + (id & 3) * 16) = wr[i];
        }
    };
    // number of K chunks this workgroup walks (temporal taps outside the clip are skipped)
    int nValidDt = 0;
    for (int d = -pt; d <= pt; ++d) nValidDt += dt_valid(d) ? 1 : 0;
    const int nch = nValidDt * (a.CinTot / BKE);

    f32x16 acc[RPW][2];
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    auto compute = [&]() {
        // ---- 9 taps x 2 k-steps out of LDS.  Fragments of tap t+1 are requested before the
        // MFMAs of tap t are issued (two fragment sets), so LDS latency hides under the matrix pipe.
        const char* hb = sh + ((wave * RPW * HW_) + lr) * PITCH;
        const char* wb0 = sw + (lr * 9) * PITCH;
        const char* wb1 = sw + ((lr + 32) * 9) * PITCH;
        uint4 fa0[2][2], fa1[2][2], fb[2][RPW][2];
        auto load_tap = [&](int set, int tap9) {
            const int kh = tap9 / 3, kw = tap9 % 3;
            fa0[set][0] = *reinterpret_cast<const uint4*>(wb0 + tap9 * PITCH + 16 * Mma<E>::chunk(0, lh));
            fa0[set][1] = *reinterpret_cast<const uint4*>(wb0 + tap9 * PITCH + 16 * Mma<E>::chunk(1, lh));
            fa1[set][0] = *reinterpret_cast<const uint4*>(wb1 + tap9 * PITCH + 16 * Mma<E>::chunk(0, lh));
            fa1[set][1] = *reinterpret_cast<const uint4*>(wb1 + tap9 * PITCH + 16 * Mma<E>::chunk(1, lh));
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const char* hp = hb + ((j + kh) * HW_ + kw) * PITCH;
                fb[set][j][0] = *reinterpret_cast<const uint4*>(hp + 16 * Mma<E>::chunk(0, lh));
                fb[set][j][1] = *reinterpret_cast<const uint4*>(hp + 16 * Mma<E>::chunk(1, lh));
            }
        };
        load_tap(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * RPW, 0);
#pragma unroll
        for (int tap9 = 0; tap9 < 9; ++tap9) {
            const int set = tap9 & 1;
            if (tap9 < 8) load_tap(set ^ 1, tap9 + 1);
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                Mma<E>::run(fa0[set], fb[set][j], acc[j][0]);
                Mma<E>::run(fa1[set], fb[set][j], acc[j][1]);
            }
            // pin the interleave: next tap's LDS reads are issued ahead of this tap's MFMAs
            if (tap9 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * RPW, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, (sizeof(E) == 2 ? 4 : 16) * RPW, 0);
        }
    };

    // software pipeline: chunk kk is multiplied out of LDS while chunks kk+1 .. kk+PF are in
    // flight in registers (set kk % PF holds chunk kk until it has been written to LDS)
    int issued = 0;
    if (FLAIR_DBG(a) == 3) return;
#pragma unroll
    for (int s_ = 0; s_ < PF; ++s_)
        if (issued < nch) {
            issue(hreg[s_], wreg[s_]);
            advance();
            ++issued;
        }
    write_lds(hreg[0], wreg[0]);
    __syncthreads();
    if (FLAIR_DBG(a) == 4) return;
    for (int k = 0; k < nch; k += PF) {
#pragma unroll
        for (int par = 0; par < PF; ++par) {
            const int kk = k + par;
            if (kk < nch) {
                if (issued < nch) {       // set `par` went to LDS already: refill it
                    if (FLAIR_DBG(a) != 2) issue(hreg[par], wreg[par]);
                    advance();
                    ++issued;
                }
                if (FLAIR_DBG(a) != 1) compute();
                if (kk + 1 < nch) {
                    __syncthreads();      // everyone is done reading the staged chunk
                    write_lds(hreg[(par + 1) % PF], wreg[(par + 1) % PF]);
                    __syncthreads();
                }
            }
        }
    }

    if (FLAIR_DBG(a) == 5) return;
    // ---- epilogue, straight from the accumulators.  A lane of the 32x32 MFMA result holds 4 consecutive couts of its
    // pixel per register quad (quad g: couts 8g + 4*lh ..); v_permlane32_swap between the two half-waves turns quads
    // 2j, 2j+1 into 8 consecutive couts per lane (lower half 16j .. 16j+7, upper half 16j+8 .. 16j+15): one 16-byte bf16
    // store and 16-byte residual loads per lane, 32 contiguous bytes per pixel and instruction (f32: a quad already is
    // 16 bytes).  Measured against the LDS-transposed epilogue below on the clip-level shapes (tools/bench_conv.py, one
    // box): 117.5 vs 121.6 us (64->64 2-D), 290 vs 268 us (3-D), equal elsewhere -- two workgroups per CU already hide
    // either epilogue, so the LDS form stays the default and this one is selected with FLAIR_CONV_SWAP_EPILOGUE=1.
    if ((a.Cout & 7) == 0 && a.debug == 7) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int h = h0 + wave * RPW + j, w = w0 + lr;
            const long p = ((long)t * a.H + h) * a.W + w;
            const bool rowok = h < a.H;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (sizeof(E) == 4) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = co0 + i * 32 + 8 * g + 4 * lh;
                        if (rowok) store_quad<E>(a, p, co, acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2],
                                                 acc[j][i][4 * g + 3]);
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int co = co0 + i * 32 + 16 * jj + 8 * lh;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto sw2 = __builtin_amdgcn_permlane32_swap(
                                __float_as_uint(acc[j][i][8 * jj + e]), __float_as_uint(acc[j][i][8 * jj + 4 + e]), false, false);
                            v[e] = __uint_as_float(sw2[0]);
                            v[4 + e] = __uint_as_float(sw2[1]);
                        }
                        if (!rowok || co >= a.Cout) continue;
                        if (a.bias) {
                            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + co);
                            const float4 b1 = *reinterpret_cast<const float4*>(a.bias + co + 4);
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        }
                        if (a.fbias) {
                            const float* fb = a.fbias + (long)t * a.fbiasLd + co;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += fb[e];
                        }
                        if (a.act == FLAIR_ACT_DCN_OFFSETS) {
                            dcn_offset_act<8>(v, co, a.actParam, a.actPeriod);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], a.act);
                        }
                        if (a.res0) {
                            float r[8];
                            Vec16<E>::load(reinterpret_cast<const E*>(a.res0) + p * a.res0Ld + co, r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e];
                        }
                        if (a.res1) {
                            float r[8];
                            Vec16<E>::load(reinterpret_cast<const E*>(a.res1) + p * a.res1Ld + co, r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= a.outScale;
                        Vec16<E>::store(reinterpret_cast<E*>(a.y) + p * a.yLd + co, v);
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue.  The accumulator layout gives each lane 4 channels of one pixel (8-byte
    // pieces scattered over 32 pixels per store).  Transpose each wave's 32 x 64 tile through LDS
    // (staged as f32, so bias / activation / residuals stay exact) and let consecutive lanes
    // write consecutive 16-byte pieces: whole 128-byte lines, vectorised residual loads.
    if ((a.Cout & 7) == 0) {
        constexpr int VEC = ET<E>::VEC;
        constexpr int CHUNKS = 64 / VEC;                        // 16-byte output pieces per pixel
        __syncthreads();                                        // LDS is free: all MFMA reads are done
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            char* tile = smem + (wave * RPW + j) * 32 * (64 * 4 + 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(tile + lr * (64 * 4 + 16) + (i * 32 + 8 * g + 4 * lh) * 4) =
                        make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2],
                                    acc[j][i][4 * g + 3]);
        }
        __syncthreads();
        constexpr int FPITCH = 64 * 4 + 16;                     // f32 staging pitch (both dtypes)
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int h = h0 + wave * RPW + j;
            if (h >= a.H) continue;
            const char* tile = smem + (wave * RPW + j) * 32 * FPITCH;
            const long prow = ((long)t * a.H + h) * a.W + w0;
#pragma unroll
            for (int it = 0; it < 32 * CHUNKS / 64; ++it) {
                const int id = it * 64 + lane;
                const int px = id / CHUNKS, ch = id % CHUNKS;
                const int co = co0 + ch * VEC;
                if (co >= a.Cout) continue;
                const long p = prow + px;
                float v[VEC];
                const float* src = reinterpret_cast<const float*>(tile + px * FPITCH) + ch * VEC;
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(src + e);
                    v[e] = q.x; v[e + 1] = q.y; v[e + 2] = q.z; v[e + 3] = q.w;
                }
                if (a.bias) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += a.bias[co + e];
                }
                if (a.fbias) {
                    const float* fb = a.fbias + (long)t * a.fbiasLd + co;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += fb[e];
                }
                if (a.act == FLAIR_ACT_DCN_OFFSETS) {
                    dcn_offset_act<VEC>(v, co, a.actParam, a.actPeriod);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] = apply_act(v[e], a.act);
                }
                if (a.res0) {
                    float r[VEC];
                    Vec16<E>::load(reinterpret_cast<const E*>(a.res0) + p * a.res0Ld + co, r);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += r[e];
                }
                if (a.res1) {
                    float r[VEC];
                    Vec16<E>::load(reinterpret_cast<const E*>(a.res1) + p * a.res1Ld + co, r);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += r[e];
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] *= a.outScale;
                Vec16<E>::store(reinterpret_cast<E*>(a.y) + p * a.yLd + co, v);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int h = h0 + wave * RPW + j, w = w0 + lr;
        if (h >= a.H) continue;
        const long p = ((long)t * a.H + h) * a.W + w;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store_quad<E>(a, p, co0 + i * 32 + 8 * g + 4 * lh, acc[j][i][4 * g], acc[j][i][4 * g + 1],
                              acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
    }
}

template <typename E, int TH, int RPW, int PF>
int launch_halo(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.nCoTiles = cdiv(a.Cout, 64);
    const int grid = a.T * cdiv(a.H, TH) * (a.W / 32) * a.nCoTiles;
    const size_t lds = (size_t)(TH + 2) * 34 * 80 + 64 * 9 * 80;
    hipLaunchKernelGGL((conv3x3_halo_kernel<E, TH, RPW, PF>), dim3(grid), dim3(64 * TH / RPW), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// ---------------------------------------------------------------------------------------
// K-split halo kernel for launches of at most one workgroup per CU (the per-frame convolutions of
// the BasicVSR++ recurrence): same TH x 32 px x 64 cout tile and staging as above, but a wavefront
// owns RPW image rows x 64 couts x ONE 16-channel half of the K step (wave = row group + k-half *
// TH/RPW), so the tile is spread over twice the wavefronts per row group and the serial MFMA chain
// of a chunk halves: with one workgroup per CU nothing else hides the staging phases, and more
// resident waves do (measured: -10..-20 % per launch; the register tile shape itself does not
// matter, tools/probes/lds_mfma_probe.hip).  The two k-halves of a tile are summed through the
// LDS staging tile of the epilogue.  Two LDS stages (one barrier per chunk), one workgroup per CU (up to 146 KB of LDS,
// 256 VGPRs).  Needs Cout % 8 == 0.
template <typename E, int TH, int RPW, int CF>
__global__ __launch_bounds__(256 * TH / RPW / CF, (256 * TH / RPW / CF) / 256) void conv3x3_halo_ks_kernel(ConvArgs a) {
    prefetch_kernargs<sizeof(ConvArgs)>();
    constexpr int NRG = TH / RPW;                  // row groups
    constexpr int NCG = 2 / CF;                    // cout-fragment groups (CF fragments of 32 couts per wave)
    constexpr int NW = 2 * NRG * NCG, NT = 64 * NW;
    constexpr int BKE = Mma<E>::BKE;
    constexpr int VEC = ET<E>::VEC;
    constexpr unsigned ESZ = sizeof(E);
    constexpr int HW_ = 34, PITCH = 80;
    constexpr int HALO_ROWS = (TH + 2) * HW_;
    constexpr int HALO_PIECES = (HALO_ROWS + 7) / 8 * 8 * 4;   // whole groups of 8 rows (see staged_row)
    constexpr int W_PIECES = 64 * 9 * 4;
    constexpr int HI = (HALO_PIECES + NT - 1) / NT;
    constexpr int WI = (W_PIECES + NT - 1) / NT;
    constexpr int HALO_BYTES = (TH + 2) * HW_ * PITCH;
    constexpr int STAGE_BYTES = HALO_BYTES + 64 * 9 * PITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int rp = wave % NRG, kh2 = (wave / NRG) & 1, cg = wave / (2 * NRG);   // row group, k-half, cout group
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int coTile = bid % a.nCoTiles;
    int rest = bid / a.nCoTiles;
    const int tilesW = a.W / 32, tilesH = (a.H + TH - 1) / TH;
    const int tw = rest % tilesW;
    rest /= tilesW;
    const int th = rest % tilesH;
    const int t = rest / tilesH;
    const int h0 = th * TH, w0 = tw * 32, co0 = coTile * 64;
    const int taps = a.KT * 9;
    const int pt = a.KT / 2;

    // the bias values of this lane's epilogue piece (see the epilogue), requested before anything else
    float biasv[VEC];
    {
        const int bco = co0 + (lane % (64 / VEC)) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) biasv[e] = (a.bias && bco < a.Cout) ? a.bias[bco + e] : 0.f;
    }
    uint4 hreg[HI], wreg[WI];
    int dt = -pt, seg = 0, cb = 0, segOff = 0;
    auto dt_valid = [&](int d) { return (unsigned)(t + d) < (unsigned)a.T; };
    while (dt <= pt && !dt_valid(dt)) ++dt;

    int hpix[HI];
    unsigned hq[HI], woff[WI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int id = i * NT + tid;
        const int pix = staged_row(id, a.ldsSwz);
        const int r = pix / HW_, c = pix % HW_;
        const int hh = h0 + r - 1, ww = w0 + c - 1;
        const bool ok = pix < HALO_ROWS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        hpix[i] = ok ? hh * a.W + ww : -1;
        hq[i] = (id & 3) * VEC * ESZ;
    }
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w, a.wBytes);
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int id = i * NT + tid;
        const int row = staged_row(id, a.ldsSwz);
        const int co = row / 9, tap9 = row % 9;
        const bool ok = id < W_PIECES && co0 + co < a.Cout;
        woff[i] = ok ? (unsigned)(((co0 + co) * taps + tap9) * a.CinTot + (id & 3) * VEC) * ESZ : FLAIR_OOB;
    }
    auto issue = [&]() {
        const unsigned ld = (unsigned)a.segLd[seg] * ESZ;
        const char* frame = reinterpret_cast<const char*>(a.x[seg]) + (size_t)(t + dt) * a.H * a.W * ld;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(frame, a.segBytes[seg]);   // one input frame
        const unsigned cofs = (unsigned)(cb * BKE) * ESZ;
#pragma unroll
        for (int i = 0; i < HI; ++i)
            hreg[i] = buf_load16(xr, hpix[i] >= 0 ? (unsigned)hpix[i] * ld + cofs + hq[i] : FLAIR_OOB);
        const unsigned kofs = (unsigned)((dt + pt) * 9 * a.CinTot + segOff + cb * BKE) * ESZ;
#pragma unroll
        for (int i = 0; i < WI; ++i) wreg[i] = buf_load16(wrs, woff[i] == FLAIR_OOB ? FLAIR_OOB : woff[i] + kofs);
        // advance the K walk (exhausted when dt > pt)
        ++cb;
        if (cb * BKE >= a.segC[seg]) {
            cb = 0;
            segOff += a.segC[seg];
            if (++seg >= a.nseg) {
                seg = 0;
                segOff = 0;
                ++dt;
                while (dt <= pt && !dt_valid(dt)) ++dt;
            }
        }
    };
    auto write_lds = [&](int stage) {
        char* sh = smem + stage * STAGE_BYTES;
        char* sw = sh + HALO_BYTES;
#pragma unroll
        for (int i = 0; i < HI; ++i) {
            const int id = i * NT + tid;
            const int row = staged_row(id, a.ldsSwz);
            if (row < HALO_ROWS) *reinterpret_cast<uint4*>(sh + row * PITCH + (id & 3) * 16) = hreg[i];
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int id = i * NT + tid;
            if (id < W_PIECES) *reinterpret_cast<uint4*>(sw + staged_row(id, a.ldsSwz) * PITCH + (id & 3) * 16) = wreg[i];
        }
    };
    int nValidDt = 0;
    for (int d = -pt; d <= pt; ++d) nValidDt += dt_valid(d) ? 1 : 0;
    const int nch = nValidDt * (a.CinTot / BKE);

    f32x16 acc[RPW][CF];   // [row of the group][cout fragment]
#pragma unroll
    for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int i = 0; i < CF; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    const int ck = 16 * Mma<E>::chunk(kh2, lh);      // this wave's 16-byte piece of every staged row
    auto compute = [&](int stage) {
        const char* sh = smem + stage * STAGE_BYTES;
        const char* hb = sh + ((RPW * rp * HW_) + lr) * PITCH + ck;
        const char* wb0 = sh + HALO_BYTES + ((cg * CF * 32 + lr) * 9) * PITCH + ck;
        uint4 fa[2][CF], fb[2][RPW];
        auto load_tap = [&](int set, int tap9) {
            const int kh = tap9 / 3, kw = tap9 % 3;
#pragma unroll
            for (int i = 0; i < CF; ++i)
                fa[set][i] = *reinterpret_cast<const uint4*>(wb0 + (i * 32 * 9 + tap9) * PITCH);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                fb[set][j] = *reinterpret_cast<const uint4*>(hb + ((kh + j) * HW_ + kw) * PITCH);
        };
        load_tap(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, CF + RPW, 0);
#pragma unroll
        for (int tap9 = 0; tap9 < 9; ++tap9) {
            const int set = tap9 & 1;
            if (tap9 < 8) load_tap(set ^ 1, tap9 + 1);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
#pragma unroll
                for (int i = 0; i < CF; ++i) Mma<E>::run1(fa[set][i], fb[set][j], acc[j][i]);
            // pin the interleave: next tap's LDS reads are issued ahead of this tap's MFMAs
            if (tap9 < 8) __builtin_amdgcn_sched_group_barrier(0x100, CF + RPW, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, CF * RPW * Mma<E>::MFMA_PER_HALF, 0);
        }
    };

    // two-stage pipeline: chunk k is multiplied out of stage k&1 while chunk k+1 is written to the
    // other stage and chunk k+2 is in flight from L2 (one register set, one barrier per chunk)
    // (timing switches of the diagnostic build: 21 no MFMA phase, 22 return before the epilogue, 23 no in-loop staging,
    //  24 return at once)
    if (FLAIR_DBG(a) == 24) return;
    issue();
    write_lds(0);
    if (nch > 1) issue();
    __syncthreads();
    for (int k = 0; k < nch; ++k) {
        if (FLAIR_DBG(a) != 21) compute(k & 1);
        if (k + 1 < nch) {
            if (FLAIR_DBG(a) != 23) write_lds((k + 1) & 1);      // stage (k+1)&1 was last read before the previous barrier
            if (k + 2 < nch && FLAIR_DBG(a) != 23) issue();
            __syncthreads();
        }
    }
    if (FLAIR_DBG(a) == 22) return;

    // ---- epilogue: sum the two k-halves in the f32 staging tile, then transposed, coalesced stores.
    // Each lane finishes the SAME 16-byte channel piece (ch = lane % CHUNKS) of different pixels in every pass, so its VEC
    // bias values were requested at kernel entry (biasv) and the residual pieces are requested here, ahead of the three
    // barriers of the k-half reduction: a per-frame launch has nothing else to hide those round trips behind (they were
    // issued after the staging tile was read: ~1 us of exposed latency per launch, 1 264 launches per step).
    constexpr int FPITCH = 64 * 4 + 16;
    constexpr int CHUNKS = 64 / VEC;                         // 16-byte output pieces per pixel
    constexpr int PARTS = NW / TH;                           // wavefronts sharing one output row
    constexpr int NIT = 32 * CHUNKS / 64 / PARTS;            // passes per wave
    const int orow = wave % TH, part = wave / TH;
    const int h = h0 + orow;
    const long prow = ((long)t * a.H + h) * a.W + w0;
    const int ech = lane % CHUNKS;                           // this lane's piece in every pass
    const int eco = co0 + ech * VEC;
    const bool eok = h < a.H && eco < a.Cout;
    uint4 r0v[NIT], r1v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int px = ((k * PARTS + part) * 64 + lane) / CHUNKS;
        const long p = prow + px;
        r0v[k] = r1v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (eok && a.res0) r0v[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const E*>(a.res0) + p * a.res0Ld + eco);
        if (eok && a.res1) r1v[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const E*>(a.res1) + p * a.res1Ld + eco);
    }
    __syncthreads();                                         // LDS is free: all fragment reads are done
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (kh2 == pass) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                char* tile = smem + (RPW * rp + j) * 32 * FPITCH + lr * FPITCH;
#pragma unroll
                for (int i = 0; i < CF; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4* dst = reinterpret_cast<float4*>(tile + ((cg * CF + i) * 32 + 8 * g + 4 * lh) * 4);
                        float4 v = make_float4(acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2],
                                               acc[j][i][4 * g + 3]);
                        if (pass == 1) {
                            const float4 o = *dst;
                            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                        }
                        *dst = v;
                    }
            }
        }
        __syncthreads();
    }
    if (!eok) return;
    const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : a.act == FLAIR_ACT_LRELU01 ? 0.1f : 0.2f;
    const bool leaky = a.act == FLAIR_ACT_NONE || a.act == FLAIR_ACT_RELU || a.act == FLAIR_ACT_LRELU01 || a.act == FLAIR_ACT_LRELU02;
    const char* tile = smem + orow * 32 * FPITCH;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int px = ((k * PARTS + part) * 64 + lane) / CHUNKS;
        const long p = prow + px;
        float v[VEC];
        const float* src = reinterpret_cast<const float*>(tile + px * FPITCH) + ech * VEC;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            const float4 q = *reinterpret_cast<const float4*>(src + e);
            v[e] = q.x; v[e + 1] = q.y; v[e + 2] = q.z; v[e + 3] = q.w;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += biasv[e];
        if (a.fbias) {
            const float* fb = a.fbias + (long)t * a.fbiasLd + eco;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += fb[e];
        }
        if (leaky) {                       // none / ReLU / LeakyReLU: max(v, slope v); one class per branch keeps the code short
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], v[e] * slope);
        } else if (a.act == FLAIR_ACT_DCN_OFFSETS) {
            dcn_offset_act<VEC>(v, eco, a.actParam, a.actPeriod);
        } else if (a.act == FLAIR_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = silu_f(v[e]);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = apply_act(v[e], a.act);
        }
        if (a.res0) {
            float r[VEC];
            Vec16<E>::load(reinterpret_cast<const E*>(&r0v[k]), r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += r[e];
        }
        if (a.res1) {
            float r[VEC];
            Vec16<E>::load(reinterpret_cast<const E*>(&r1v[k]), r);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] *= a.outScale;
        Vec16<E>::store(reinterpret_cast<E*>(a.y) + p * a.yLd + eco, v);
    }
}

template <typename E, int TH, int RPW, int CF>
int launch_halo_ks(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.nCoTiles = cdiv(a.Cout, 64);
    const int grid = a.T * cdiv(a.H, TH) * (a.W / 32) * a.nCoTiles;
    const size_t lds = 2 * ((size_t)(TH + 2) * 34 * 80 + 64 * 9 * 80);
    static LdsAttrOnce attr;
    {
        const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&conv3x3_halo_ks_kernel<E, TH, RPW, CF>));
        FLAIR_CHECK(e == hipSuccess, "flair_conv_nhwc: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((conv3x3_halo_ks_kernel<E, TH, RPW, CF>), dim3(grid), dim3(256 * TH / RPW / CF), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// ---------------------------------------------------------------------------------------
// Persistent LDS-DMA 3x3 (x KT) convolution (round 3; bf16; variant 8).
//
// What the counters said about conv3x3_halo_kernel<8> (profiles/r02z_conv_pmc_sq.txt): per K chunk a
// workgroup moves 73 KB through the VGPRs into LDS (ds_write_b128: 13 cycles of register transfer per instruction), every
// wave re-reads 1.5 fragments per MFMA, and each of the 4096 workgroups of a clip-level launch pays its own first-fetch
// latency and epilogue with only one other workgroup per CU to hide them: 2 660 LDS cycles against 2 304 MFMA cycles per
// chunk, 0.30 of the matrix peak.  This kernel changes all three:
//   * staging is `buffer_load_dwordx4 ... lds` (LDS-DMA): no VGPR round trip, no ds_write; the LDS image is 64-byte rows
//     (one pixel / one (tap, cout) of a 32-channel chunk) written linearly by the DMA, with the 16-byte chunk XOR-swizzled
//     on the SOURCE address (pixel rows by (column >> 2) & 3, weight rows by (cout >> 2) & 3) so that every ds_read_b128
//     fragment read is conflict-free (16 lanes of a read group always cover 16 distinct 16-byte slots);
//   * a wave owns TWO image rows x 32 pixels x 64 couts (four 32x32 accumulators): for one column tap kw the four halo
//     rows it touches are read once and serve all three row taps: 20 fragment reads per 24 MFMAs (0.83 instead of 1.5);
//   * workgroups are persistent: 8 waves = a 16-row x 32-pixel x 64-cout tile per step, one workgroup per CU walking its
//     share of the tiles; the (tile, chunk) sequence is one software pipeline -- chunk n+1 (of this tile or the next) is in
//     flight by DMA into the other LDS stage while chunk n is multiplied, so first-fetch latency and the epilogue stores of
//     a tile hide behind the neighbouring tiles' matrix work.  One barrier per chunk.
// Tiles are dealt so that the workgroups of one XCD (blockIdx & 7) work on neighbouring tiles of one frame at the same
// time (shared halo lines and weights hit in that XCD's L2).  Epilogue straight from the accumulators (v_permlane32_swap
// pairs give each lane 8 consecutive couts of its pixel: 16-byte stores / residual loads).
// Needs bf16, stride 1, 3x3 spatial taps, W % 32 == 0, H % 16 == 0, Cout % 8 == 0.
struct DmaTile {
    int t, h0, w0, co0;
};

// NW wavefronts per workgroup, RPW image rows per wavefront (tile = NW * RPW rows x 32 pixels x 64 couts):
//   <8, 2> clip-level launches (the shape described above);
//   <8, 1> / <4, 1> single-round launches of 256 tiles of 8 / 4 rows = the per-frame convolutions of the BasicVSR++
//   recurrence at 256^2 (c = 64) and 128^2 (c = 128): one tile per workgroup, where the gain is the short prologue (DMA instead
//   of load -> ds_write -> barrier) and the register epilogue (no LDS transposition, no k-half reduction barriers).
//   NSTAGE = 3 (one-tile launches only): three LDS stages, chunks n + 1 and n + 2 in flight while chunk n is multiplied (counted
//   vmcnt): a per-frame launch at 128^2 multiplies a chunk in ~0.5 us but needs ~1.5 us to fetch one, so depth hides what
//   a single chunk in flight cannot.
//   KS = 2 (round 4, <8, 1, 3, 2>): K split -- waves w and w + NW / 2 share an image row, each multiplies ONE 16-channel k-step of every
//   32-channel chunk (the decomposition of conv3x3_halo_ks_kernel on LDS-DMA staging with three stages): 4-row tiles with eight waves, for
//   the per-frame convolutions of the 128^2 level, where four waves of one row each (one per SIMD) cannot hide their fragment reads.
//   After the last chunk the pair exchanges one 32-cout accumulator through LDS and each wave finishes 32 couts of its row.
template <int NW, int RPW, int NSTAGE, int KS>
__device__ __forceinline__ void conv3x3_dma_body(const ConvArgs& a, int nTiles, int tilesPerXcd) {
    using E = bf16_t;
    static_assert(KS == 1 || (KS == 2 && RPW == 1 && NSTAGE == 3 && NW % 2 == 0), "K split: one-tile form with one row per wave");
    constexpr int ROWW = NW / KS;                          // waves along the rows of the tile
    constexpr int KSTEPS = 2 / KS;                         // 16-channel k-steps of a chunk this wave multiplies
    constexpr int TH = ROWW * RPW, HWP = 34;               // tile rows, halo pitch in pixels
    constexpr int HALO_ROWS = (TH + 2) * HWP;              // staged pixels
    constexpr int HALO_INSTR = (HALO_ROWS + 15) / 16;      // DMA wave-instructions of 16 rows x 64 B
    constexpr int W_INSTR = 64 * 9 / 16;                   // 36
    constexpr int HALO_BYTES = HALO_INSTR * 1024;
    constexpr int STAGE_BYTES = HALO_BYTES + W_INSTR * 1024;
    constexpr int NSLOT = (HALO_INSTR + W_INSTR + NW - 1) / NW;   // DMA instructions per wave and chunk
    constexpr int BIAS_OFF = NSTAGE * STAGE_BYTES + NW * 1024;  // two slots of 64 f32 biases (tile parity) behind the DMA scratch
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = KS == 1 ? wave : wave % ROWW, kpart = KS == 1 ? 0 : wave / ROWW;      // row group / k-step of this wave
    const int lr = lane & 31, lh = lane >> 5;
    const int taps = a.KT * 9, pt = a.KT / 2;
    const int chunksPerTap = a.CinTot / 32;
    const int H = a.H, W = a.W, T = a.T;

    // ---- this workgroup's tiles: XCD x (= blockIdx & 7) owns the list entries [x * tilesPerXcd, (x + 1) * tilesPerXcd),
    // its workgroups (slot = blockIdx >> 3) take entries slot, slot + slotsPerXcd, ...
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slotsPerXcd = gridDim.x >> 3;
    const int tilesW = W / 32, tilesH = H / TH, nCoTiles = a.nCoTiles;
    auto tile_at = [&](int i, DmaTile& tl) -> bool {       // i-th tile of this workgroup
        const int inXcd = slot + i * slotsPerXcd;
        const int id = xcd * tilesPerXcd + inXcd;
        if (inXcd >= tilesPerXcd || id >= nTiles) return false;
        int rest = id;
        tl.co0 = (rest % nCoTiles) * 64;
        rest /= nCoTiles;
        if (a.tFast) {
            // temporal taps: the frame index runs fastest, so the workgroups of an XCD work on ALL frames of a few spatial tiles at
            // the same time and the three output frames that read one input tile find it in that XCD's L2 together
            tl.t = rest % T;
            rest /= T;
            tl.w0 = (rest % tilesW) * 32;
            tl.h0 = (rest / tilesW) * TH;
            return true;
        }
        tl.w0 = (rest % tilesW) * 32;
        rest /= tilesW;
        tl.h0 = (rest % tilesH) * TH;
        tl.t = rest / tilesH;
        return true;
    };

    // ---- DMA slots of this wave: instruction ids wave * NSLOT + i over [halo instructions | weight instructions | idle].
    // Lane l of an instruction fills LDS row 16 * id + (l >> 2), physical 16-byte chunk l & 3, with LOGICAL chunk
    // (l & 3) ^ swz(row) of that row's pixel / weight row.  Idle slots and out-of-image pixels use an out-of-range offset
    // (the DMA writes zeros: zero padding for free, no branches).
    const int lrow = lane >> 2, lchunk = lane & 3;
    // A slot is a halo OR a weight instruction (wave-uniformly), so one word serves both: halo slots hold the pixel index
    // inside the frame (0xffffffff outside the image), weight slots the byte offset of the lane's row / chunk inside the
    // packed weights (FLAIR_OOB past the last cout); hchunk: 16 * logical chunk of a halo slot (tile independent).
    unsigned slotv[NSLOT], hchunk[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int id = wave * NSLOT + i;
        const int c = (id * 16 + lrow) % HWP;
        hchunk[i] = (unsigned)((lchunk ^ ((c >> 2) & 3)) << 4);
    }
    auto setup_tile = [&](const DmaTile& tl) {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int id = wave * NSLOT + i;
            const int R = id * 16 + lrow;
            const int hr = R / HWP, c = R - hr * HWP;
            const int hh = tl.h0 - 1 + hr, ww = tl.w0 - 1 + c;
            const bool okh = R < HALO_ROWS && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            const unsigned hv = okh ? (unsigned)(hh * W + ww) : 0xffffffffu;
            const int row = R - HALO_INSTR * 16;               // tap9 * 64 + co (weight slots)
            const int tap9 = row >> 6, co = row & 63;
            const bool okw = row >= 0 && row < 576 && tl.co0 + co < a.Cout;
            const unsigned wv = okw ? (unsigned)(((tl.co0 + co) * taps + tap9) * a.CinTot) * 2u + ((lchunk ^ ((co >> 2) & 3)) << 4) : FLAIR_OOB;
            slotv[i] = id >= HALO_INSTR ? wv : hv;
        }
    };

    const u32x4_t wdesc = make_desc(a.w, a.wBytes);
    // K walk state of the ISSUE side (block uniform)
    struct Walk {
        int dt, seg, cb, segOff, j;
    };
    // tFast: the temporal taps are walked in an order rotated by the output frame, dt_j(t) = ((j - t) mod KT) - pt, so that the three
    // workgroups that read one input tile (output frames t - 1, t, t + 1 of one spatial tile, side by side on one XCD) ask for it at
    // the same position of their walks instead of two / four chunk times apart
    auto dt_at = [&](int t, int j) { return a.tFast ? (j + a.KT * T - t) % a.KT - pt : j - pt; };
    auto walk_first = [&](int t, Walk& wk) {
        wk.j = 0; wk.seg = 0; wk.cb = 0; wk.segOff = 0;
        while (wk.j < a.KT && (unsigned)(t + dt_at(t, wk.j)) >= (unsigned)T) ++wk.j;
        wk.dt = dt_at(t, wk.j);
    };
    auto walk_next = [&](int t, Walk& wk) {
        ++wk.cb;
        if (wk.cb * 32 >= a.segC[wk.seg]) {
            wk.cb = 0;
            wk.segOff += a.segC[wk.seg];
            if (++wk.seg >= a.nseg) {
                wk.seg = 0;
                wk.segOff = 0;
                ++wk.j;
                while (wk.j < a.KT && (unsigned)(t + dt_at(t, wk.j)) >= (unsigned)T) ++wk.j;
                wk.dt = dt_at(t, wk.j);        // (past the last tap: never issued, remIssue ends the walk)
            }
        }
    };
    auto chunks_of = [&](int t) {
        int n = 0;
        for (int d = -pt; d <= pt; ++d) n += (unsigned)(t + d) < (unsigned)T ? 1 : 0;
        return n * chunksPerTap;
    };
    auto issue = [&](const DmaTile& tl, const Walk& wk, int stage) {
        const unsigned ld = (unsigned)a.segLd[wk.seg] * 2u;
        const char* frame = reinterpret_cast<const char*>(a.x[wk.seg]) + (size_t)(tl.t + wk.dt) * H * W * ld;
        const u32x4_t xdesc = make_desc(frame, a.segBytes[wk.seg]);      // one input frame
        const unsigned cofs = (unsigned)(wk.cb * 64);
        const unsigned kofs = (unsigned)((wk.dt + pt) * 9 * a.CinTot + wk.segOff + wk.cb * 32) * 2u;
        const unsigned sbase = (unsigned)(stage * STAGE_BYTES + wave * NSLOT * 1024);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const bool isw = wave * NSLOT + i >= HALO_INSTR;        // wave-uniform
            const unsigned sv = slotv[i];
            const unsigned offh = sv == 0xffffffffu ? FLAIR_OOB : sv * ld + cofs + hchunk[i];
            const unsigned offw = sv == FLAIR_OOB ? FLAIR_OOB : sv + kofs;
            u32x4_t d;
            d.x = isw ? wdesc.x : xdesc.x;
            d.y = isw ? wdesc.y : xdesc.y;
            d.z = isw ? wdesc.z : xdesc.z;
            d.w = 0x00020000u;
            // instruction ids past the last weight row (wave 7's tail slots) write their zeros to a scratch area behind the
            // two stages instead of the first bytes of the next stage
            const bool idle = wave * NSLOT + i >= HALO_INSTR + W_INSTR;
            dma16(d, isw ? offw : offh, idle ? (unsigned)(NSTAGE * STAGE_BYTES + ((wave * NSLOT + i - HALO_INSTR - W_INSTR) % NW) * 1024) : sbase + i * 1024);
        }
    };

    // ---- fragment read offsets (per lane, fixed).  A: weight row tap9 * 64 + cf * 32 + lr, chunk 2 s + lh;
    // B: halo row (2 wave + j + kh) * 34 + kw + lr, same chunk; XOR terms as written by the DMA.
    unsigned aoff[KSTEPS], boff[3][KSTEPS];
#pragma unroll
    for (int si = 0; si < KSTEPS; ++si) {
        const int s_ = KS == 1 ? si : kpart;               // K split: the wave's own k-step
        aoff[si] = (unsigned)(HALO_BYTES + lr * 64 + (((2 * s_ + lh) ^ ((lr >> 2) & 3)) << 4));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            boff[kw][si] = (unsigned)((RPW * wr * HWP + kw + lr) * 64 + (((2 * s_ + lh) ^ (((kw + lr) >> 2) & 3)) << 4));
    }

    f32x16 acc[RPW][2];            // [row j][cout fragment]
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };
    // One chunk = 3 column taps x 3 row taps x 8 MFMAs.  Fragments of step n + 1 (A: 4 reads; B: 8 more reads when the
    // column tap changes) are requested before the 8 MFMAs of step n are issued (two fragment sets).
    auto compute_as = [&](int stage, auto revTag) {
        constexpr bool REV = decltype(revTag)::value;      // walk the column taps right to left (experiment: de-phase wave pairs)
        const char* sb = smem + stage * STAGE_BYTES;
        uint4 fb[2][RPW + 2][KSTEPS];   // [set][halo row RPW wave + h][k-step]
        uint4 fa[2][2][KSTEPS];         // [set][cout fragment][k-step]
        auto load_b = [&](int set, int kw) {
#pragma unroll
            for (int h = 0; h < RPW + 2; ++h)
#pragma unroll
                for (int s_ = 0; s_ < KSTEPS; ++s_)
                    fb[set][h][s_] = *reinterpret_cast<const uint4*>(sb + h * (HWP * 64) + boff[kw][s_]);
        };
        auto load_a = [&](int set, int kh, int kw) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int s_ = 0; s_ < KSTEPS; ++s_)
                    fa[set][cf][s_] = *reinterpret_cast<const uint4*>(sb + (kh * 3 + kw) * 4096 + cf * 2048 + aoff[s_]);
        };
        auto mma = [&](const uint4 (&fa_)[KSTEPS], const uint4 (&fb_)[KSTEPS], f32x16& c) {
            if constexpr (KS == 1) Mma<E>::run(fa_, fb_, c);
            else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa_[0]), __builtin_bit_cast(bf16x8, fb_[0]), c, 0, 0, 0);
        };
        auto KW = [](int q) { return REV ? 2 - q : q; };
        load_b(0, KW(0));
        load_a(0, 0, KW(0));
        __builtin_amdgcn_sched_group_barrier(0x100, KSTEPS * (RPW + 2) + 2 * KSTEPS, 0);
#pragma unroll
        for (int step = 0; step < 9; ++step) {
            const int kq = step / 3, kh = step % 3;
            const int nkq = (step + 1) / 3, nkh = (step + 1) % 3;
            if (step < 8) {
                if (nkh == 0) load_b(nkq & 1, KW(nkq));
                load_a((step + 1) & 1, nkh, KW(nkq));
            }
            if (FLAIR_DBG(a) == 17) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < RPW; ++j)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) mma(fa[step & 1][cf], fb[kq & 1][j + kh], acc[j][cf]);
            if (FLAIR_DBG(a) == 17) __builtin_amdgcn_s_setprio(0);
            if (step < 8) {
                if (nkh == 0) __builtin_amdgcn_sched_group_barrier(0x100, KSTEPS * (RPW + 2) + 2 * KSTEPS, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 2 * KSTEPS, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x8, 2 * KSTEPS * RPW, 0);
        }
    };
    auto compute = [&](int stage) {
#ifdef FLAIR_TIMING_SWITCHES
        if (FLAIR_DBG(a) == 18 && wave >= NW / 2) {
            compute_as(stage, std::true_type{});
            return;
        }
#endif
        compute_as(stage, std::false_type{});
    };
    // Epilogue, written for a low instruction count: it runs once per tile on all eight waves at once, so nothing hides its
    // VALU issue time (timing switches, profiles/r03_dma_switches.txt: 74 us of a 156 us two-chunk convolution in the first
    // version -- 22 us stores, 12 us bias round trips, the rest instruction issue and instruction-cache misses of eight
    // inlined copies x six activation variants = 14 700 instructions).  Now: one unrolled pass per activation CLASS (chosen
    // by one wave-uniform branch outside), biases from LDS, one 64-bit address per tile and wave, pointer bumps after that.
    auto epilogue_as = [&](const DmaTile& tl, int biasSlot, auto actTag, auto resTag) {
        constexpr int ACT = decltype(actTag)::value;       // 0: max(v, slope v)   1: DCN offsets / masks   2: SiLU
        constexpr bool HASRES = decltype(resTag)::value;   // a residual input exists (its own copy of the code: see below)
        const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : a.act == FLAIR_ACT_LRELU01 ? 0.1f : 0.2f;
        // Branch-free: every access is a buffer instruction on a wave-uniform descriptor (the wave's first pixel, this tile's
        // first cout); lanes whose couts are padding use an out-of-range offset (loads deliver 0, stores are dropped).  With
        // EXEC-masked branches around the loads / stores, hipcc merged its counters at every join into `s_waitcnt vmcnt(0)`:
        // each residual piece was a serial round trip to memory AND waited for the stores before it (43 us of a 152 us 64 -> 64
        // convolution with residual).  Now all residual pieces of the wave are requested in one batch up front
        // (the staging fragments are dead, the registers exist) and the waits are counted; the frame bias is part of the tile's
        // bias slot in LDS (bias_request).
        const long p0w = ((long)tl.t * H + tl.h0 + RPW * wr) * W + tl.w0;                // first pixel of the wave's rows (wave-uniform)
        const int cl = 8 * lh;                                                           // this lane's cout offset inside a 16-cout half
        const unsigned yLdB = (unsigned)a.yLd * 2u, rLdB = (unsigned)a.res0Ld * 2u;
        const unsigned span = (unsigned)(RPW * W);                                       // pixels the wave's offsets stay below
        const __amdgpu_buffer_rsrc_t yd = make_rsrc(reinterpret_cast<char*>(a.y) + (p0w * a.yLd + tl.co0) * 2, span * yLdB);
        const __amdgpu_buffer_rsrc_t r0d = make_rsrc(reinterpret_cast<const char*>(a.res0) + (a.res0 ? (p0w * a.res0Ld + tl.co0) * 2 : 0),
                                                     a.res0 ? span * rLdB : 0u);
        const E* r1b = a.res1 ? reinterpret_cast<const E*>(a.res1) + (p0w + lr) * a.res1Ld + tl.co0 + cl : nullptr;
        const float* bl = reinterpret_cast<const float*>(smem + BIAS_OFF + (biasSlot & 1) * 256) + cl;
        const float scale = a.outScale;
        const unsigned yLane = (unsigned)lr * yLdB + 2u * cl, rLane = (unsigned)lr * rLdB + 2u * cl;
        // Without a residual no load is issued at all: a load here is younger than the next chunk's DMA pieces, so waiting
        // for it also waits for those, which otherwise have the whole epilogue left to land (+9 us on a 105 us convolution).
        constexpr bool ALLUP = RPW <= 2;           // four rows per wave: one batch per 32-cout fragment (the accumulators leave no room for all 16 pieces)
        uint4 r0v[ALLUP ? 2 : 1][2][RPW];
        auto res_request = [&](int i) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int cofs = i * 32 + 16 * jj;
                const bool lane_ok = tl.co0 + cofs + cl < a.Cout;
#pragma unroll
                for (int j = 0; j < RPW; ++j)
                    r0v[ALLUP ? i : 0][jj][j] = buf_load16(r0d, lane_ok ? rLane + (unsigned)(j * W) * rLdB + 2u * cofs : FLAIR_OOB);
            }
        };
        if constexpr (HASRES && ALLUP) {
            res_request(0);
            res_request(1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (HASRES && !ALLUP) res_request(i);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int cofs = i * 32 + 16 * jj;                                       // cout offset of this group inside the tile
                // (Cout % 8 == 0) lanes whose 8 couts are padding still take part in the v_permlane32_swap below (a swap
                // under a divergent branch hands the active half garbage); their store offset is out of range
                const bool lane_ok = tl.co0 + cofs + cl < a.Cout;
                float bv[8];
                {
                    const float4 b0 = *reinterpret_cast<const float4*>(bl + cofs);
                    const float4 b1 = *reinterpret_cast<const float4*>(bl + cofs + 4);
                    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
                }
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i][8 * jj + e]),
                                                                          __float_as_uint(acc[j][i][8 * jj + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw2[0]) + bv[e];
                        v[4 + e] = __uint_as_float(sw2[1]) + bv[4 + e];
                    }
                    if constexpr (ACT == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
                    } else if constexpr (ACT == 1) {
                        dcn_offset_act_fast<8>(v, tl.co0 + cofs + cl, a.actParam, a.actPeriod, 1.f);
                    } else if constexpr (ACT == 2) {
                        // v_rcp_f32 (1 ulp) instead of the IEEE divide of silu_f: ten instructions less per element, and the
                        // result is rounded to bf16 right below
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= __builtin_amdgcn_rcpf(1.f + __expf(-v[e]));
                    }                                      // ACT == 3: no activation and out_scale == 1 (the ResBlock convolutions)
                    if constexpr (HASRES) {
                        float r[8];
                        Vec16<E>::load(reinterpret_cast<const E*>(&r0v[ALLUP ? i : 0][jj][j]), r);      // zeros without res0 / on padding lanes
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                        if (r1b && lane_ok) {
                            Vec16<E>::load(r1b + (long)j * W * a.res1Ld + cofs, r);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += r[e];
                        }
                    }
                    if constexpr (ACT != 3) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= scale;
                    }
                    alignas(16) E out[8];
                    Vec16<E>::store(out, v);
                    const uint4 ov = *reinterpret_cast<const uint4*>(out);
                    const unsigned yo = lane_ok ? yLane + (unsigned)(j * W) * yLdB + 2u * cofs : FLAIR_OOB;
                    if (FLAIR_DBG(a) != 16)
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd, (int)yo, 0, 0);
                    else asm volatile("" ::"v"(ov.x), "v"(ov.y), "v"(ov.z), "v"(ov.w));
                }
            }
        }
    };
    // One-tile forms (RPW == 1, the per-frame launches): the plain epilogue -- global loads / stores under EXEC-masked
    // branches.  The buffer-descriptor form below costs these launches ~0.15 us per K chunk (its descriptors raise the
    // SGPR pressure of the chunk loop: 224 -> 64 at 256^2 26.0 -> 27.0 us, same box), and a per-frame convolution with a
    // residual is rare on this path (the recurrence's residual blocks are fused chains).
    auto epilogue_plain_as = [&](const DmaTile& tl, int biasSlot, auto actTag) {
        constexpr int ACT = decltype(actTag)::value;       // 0: max(v, slope v)   1: DCN offsets / masks   2: SiLU
        const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : a.act == FLAIR_ACT_LRELU01 ? 0.1f : 0.2f;
        const long p0 = ((long)tl.t * H + tl.h0 + RPW * wr) * W + tl.w0 + lr;            // this lane's pixel in row j = 0
        const int cl = 8 * lh;                                                           // this lane's cout offset inside a 16-cout half
        E* yb = reinterpret_cast<E*>(a.y) + p0 * a.yLd + tl.co0 + cl;
        const E* r0b = a.res0 ? reinterpret_cast<const E*>(a.res0) + p0 * a.res0Ld + tl.co0 + cl : nullptr;
        const E* r1b = a.res1 ? reinterpret_cast<const E*>(a.res1) + p0 * a.res1Ld + tl.co0 + cl : nullptr;
        const float* bl = reinterpret_cast<const float*>(smem + BIAS_OFF + (biasSlot & 1) * 256) + cl;
        const float scale = a.outScale;
#pragma unroll
        for (int i = 0; i < 2 / KS; ++i)                                                 // K split: the wave's sum sits in acc[.][0], its couts start at 32 * kpart
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int cofs = (KS == 1 ? i : kpart) * 32 + 16 * jj;                   // cout offset of this group inside the tile
                if (tl.co0 + cofs >= a.Cout) continue;                                   // wave-uniform: the whole 16-cout group is padding
                // (Cout % 8 == 0) the upper 8 couts of the group may be padding: those lanes still take part in the
                // v_permlane32_swap below (a swap under a divergent branch hands the active half garbage), only their
                // loads / stores are predicated
                const bool lane_ok = tl.co0 + cofs + cl < a.Cout;
                float bv[8];
                {
                    const float4 b0 = *reinterpret_cast<const float4*>(bl + cofs);
                    const float4 b1 = *reinterpret_cast<const float4*>(bl + cofs + 4);
                    bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
                }
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i][8 * jj + e]),
                                                                          __float_as_uint(acc[j][i][8 * jj + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw2[0]) + bv[e];
                        v[4 + e] = __uint_as_float(sw2[1]) + bv[4 + e];
                    }
                    if constexpr (ACT == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
                    } else if constexpr (ACT == 1) {
                        dcn_offset_act<8>(v, tl.co0 + cofs + cl, a.actParam, a.actPeriod);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                    }
                    const long ro = (long)j * W;                                         // row j of the wave's pair
                    if (r0b && lane_ok) {
                        float r[8];
                        Vec16<E>::load(r0b + ro * a.res0Ld + cofs, r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
                    if (r1b && lane_ok) {
                        float r[8];
                        Vec16<E>::load(r1b + ro * a.res1Ld + cofs, r);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += r[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= scale;
                    if (!lane_ok) continue;
                    if (FLAIR_DBG(a) != 16) Vec16<E>::store(yb + ro * a.yLd + cofs, v);
                    else asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
                }
            }
    };
    auto epilogue = [&](const DmaTile& tl, int biasSlot) {
        const bool res = a.res0 || a.res1;
        auto go = [&](auto actTag) {
            if constexpr (RPW == 1) {
                epilogue_plain_as(tl, biasSlot, actTag);
            } else {
                if (res) epilogue_as(tl, biasSlot, actTag, std::true_type{});
                else epilogue_as(tl, biasSlot, actTag, std::false_type{});
            }
        };
        if (a.act == FLAIR_ACT_DCN_OFFSETS) go(std::integral_constant<int, 1>{});
        else if (a.act == FLAIR_ACT_SILU) go(std::integral_constant<int, 2>{});
        else if (RPW >= 2 && a.act == FLAIR_ACT_NONE && a.outScale == 1.f) go(std::integral_constant<int, 3>{});   // pack + store only
        else go(std::integral_constant<int, 0>{});      // NONE / RELU / LeakyReLU (GELU: refused on the host)
    };
    // Residual prefetch: one 4-byte LDS-DMA per output pixel of the wave (64 couts x 2 bytes = the pixel's 128-byte line),
    // landing in the DMA scratch area, requested BEFORE the DMA pieces of the tile's last chunk: the epilogue's real loads
    // then find their lines in the L2 instead of paying a trip to HBM with every wave of the CU waiting.  No register
    // receives data, so nothing can be clobbered; the pieces are older than that chunk's DMA and retire with it.
    auto prefetch_res = [&](const DmaTile& tl) {
        const long p0w = ((long)tl.t * H + tl.h0 + RPW * wr) * W + tl.w0;
        const u32x4_t rd = make_desc(reinterpret_cast<const char*>(a.res0) + (p0w * a.res0Ld + tl.co0) * 2, (unsigned)(RPW * W * a.res0Ld) * 2u);
#pragma unroll
        for (int jj = 0; jj < (RPW + 1) / 2; ++jj) {
            const int j = 2 * jj + (lane >> 5);
            const unsigned voff = j < RPW ? (unsigned)((j * W + lr) * a.res0Ld) * 2u : FLAIR_OOB;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"((unsigned)(NSTAGE * STAGE_BYTES + wave * 1024)), "s"(rd)
                         : "memory");
        }
    };

    // ---- the (tile, chunk) pipeline.  `cur` is the tile being multiplied, `nxt` the tile whose chunks are being issued.
    DmaTile cur, nxt;
    int iCur = 0, iNxt = 0;
    if (!tile_at(0, cur)) return;
    nxt = cur;
    Walk wk;
    walk_first(nxt.t, wk);
    setup_tile(nxt);
    // biases of a tile: requested when its first chunk is issued (threads 0..15, one float4 each), written to the LDS slot
    // of the tile's parity after the MFMA phase that follows, read in its epilogue at least one barrier later
    float4 biasReg = make_float4(0.f, 0.f, 0.f, 0.f);
    bool biasPending = false;
    auto bias_request = [&](const DmaTile& tl) {
        if (tid < 16) {
            const int bco = tl.co0 + 4 * tid;
            biasReg = (a.bias && bco < a.Cout) ? *reinterpret_cast<const float4*>(a.bias + bco) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.fbias && bco < a.Cout) {            // per-(frame, cout) bias of this tile's frame: folded into the same LDS slot
                const float* fb = a.fbias + (long)tl.t * a.fbiasLd + bco;
                biasReg.x += fb[0]; biasReg.y += fb[1]; biasReg.z += fb[2]; biasReg.w += fb[3];
            }
        }
        biasPending = true;
    };
    auto bias_commit = [&](int slot_) {
        if (tid < 16) *reinterpret_cast<float4*>(smem + BIAS_OFF + (slot_ & 1) * 256 + 16 * tid) = biasReg;
        biasPending = false;
    };
    bias_request(nxt);
    bias_commit(0);
    if constexpr (NSTAGE == 3) {
        // ---- one tile, three stages: wait(n) -> barrier -> issue(n + 2) -> multiply(n).  When chunk n is waited for, the only
        // younger DMA in flight is chunk n + 1 (NSLOT instructions of this wave): vmcnt(NSLOT) retires chunk n and leaves it.
        const int nch = chunks_of(cur.t);
        issue(cur, wk, 0);
        walk_next(cur.t, wk);
        if (nch > 1) {
            issue(cur, wk, 1);
            walk_next(cur.t, wk);
        }
        zero_acc();
        for (int n = 0; n < nch; ++n) {
            if (n + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                    // chunk n is in LDS for everybody; stage (n + 2) % 3 = (n - 1) % 3 is free
            if (n + 2 < nch) {
                if (FLAIR_DBG(a) != 12) issue(cur, wk, (n + 2) % 3);
                walk_next(cur.t, wk);
            }
            if (FLAIR_DBG(a) != 11) compute(n % 3);
        }
        if constexpr (KS == 2) {
            // ---- sum the two k-steps of every row: the wave of k-step 0 finishes cout fragment 0, its partner fragment 1; each parks the
            // fragment it gives away in LDS (stage 0: every wave is past its last fragment read after the barrier)
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);
            float* mine = red + (wave * 64 + lane) * 16;
            const float* theirs = red + ((wave ^ ROWW) * 64 + lane) * 16;                // partner: same row, other k-step (ROWW is a power of two)
            static_assert((ROWW & (ROWW - 1)) == 0, "partner by xor");
            if (kpart == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 4) *reinterpret_cast<float4*>(mine + r) = make_float4(acc[0][1][r], acc[0][1][r + 1], acc[0][1][r + 2], acc[0][1][r + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 4) *reinterpret_cast<float4*>(mine + r) = make_float4(acc[0][0][r], acc[0][0][r + 1], acc[0][0][r + 2], acc[0][0][r + 3]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][1][r];                 // the kept fragment moves to slot 0 (the epilogue reads slot 0)
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(theirs + r);
                acc[0][0][r] += v.x; acc[0][0][r + 1] += v.y; acc[0][0][r + 2] += v.z; acc[0][0][r + 3] += v.w;
            }
        }
        if (FLAIR_DBG(a) != 13) epilogue(cur, 0);
        return;
    }
    const bool res0any = a.res0 || a.res1;
    int remIssue = chunks_of(nxt.t);           // chunks of `nxt` not yet issued
    int remCompute = remIssue;                 // chunks of `cur` not yet multiplied
    bool more = true;                          // is there a chunk left to issue
    int stage = 0;
    issue(nxt, wk, 0);
    walk_next(nxt.t, wk);
    --remIssue;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    zero_acc();
    while (true) {
        // issue the next chunk (of this tile, or the first chunk of the next tile) into the other stage: every wave has
        // passed the barrier that ended the last read of that stage
        if (more && remIssue == 0) {
            ++iNxt;
            more = tile_at(iNxt, nxt);
            if (more) {
                walk_first(nxt.t, wk);
                setup_tile(nxt);
                remIssue = chunks_of(nxt.t);
                bias_request(nxt);
            }
        }
        if constexpr (RPW >= 2)
            if (remCompute == 1 && a.res0 && FLAIR_DBG(a) != 19) {
                if (a.resPrefetch == 1) prefetch_res(cur);
            }
        if (more) {
            if (FLAIR_DBG(a) != 12) issue(nxt, wk, stage ^ 1);     // (timing switches: 11 no MFMA phase, 12 no DMA, 13 no epilogue)
            walk_next(nxt.t, wk);
            --remIssue;
        }
        if (FLAIR_DBG(a) != 11) compute(stage);
        if (biasPending) bias_commit(iNxt);
        bool storesInFlight = false;
        if (--remCompute == 0) {
            if (FLAIR_DBG(a) != 13) {
                epilogue(cur, iCur);
                // every tile issues exactly 4 * RPW store instructions per wave as its youngest memory operations (a second
                // residual is loaded under a branch, where hipcc places its own waits: not counted on)
                // (RPW == 2: padding lanes' stores are issued too, out of range; RPW == 1: full tiles without a residual branch)
                storesInFlight = (RPW >= 2 ? !a.res1 : cur.co0 + 64 <= a.Cout && !res0any) && FLAIR_DBG(a) != 14 && FLAIR_DBG(a) != 16;
            }
            ++iCur;
            if (!tile_at(iCur, cur)) break;
            remCompute = chunks_of(cur.t);
            zero_acc();
        }
        // This wave's DMA pieces of the next stage have landed: vmcnt counts loads, LDS-DMA and stores in issue order, so
        // with the 8 epilogue stores as the youngest operations vmcnt(8) retires every DMA and leaves the stores in flight
        // (waiting for their acknowledgement from memory cost 40 % of a two-chunk convolution: profiles/r03_dma_switches.txt)
        if (storesInFlight) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * RPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                      // ... and so have everybody else's
        stage ^= 1;
    }
}

template <int NW, int RPW, int NSTAGE>
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void conv3x3_dma_kernel(ConvArgs a, int nTiles, int tilesPerXcd) {
    prefetch_kernargs<sizeof(ConvArgs) + 8>();
    conv3x3_dma_body<NW, RPW, NSTAGE, 1>(a, nTiles, tilesPerXcd);
}
template <int NW, int RPW, int NSTAGE>          // the K-split form under its own name (rocprofv3 rows of the other forms keep theirs)
__global__ __launch_bounds__(64 * NW, (NW + 3) / 4) void conv3x3_dma_ks_kernel(ConvArgs a, int nTiles, int tilesPerXcd) {
    prefetch_kernargs<sizeof(ConvArgs) + 8>();
    conv3x3_dma_body<NW, RPW, NSTAGE, 2>(a, nTiles, tilesPerXcd);
}

template <int NW, int RPW, int NSTAGE, int KS = 1>
int launch_dma(const ConvArgs& a0, hipStream_t s) {
    constexpr int TH = NW / KS * RPW;
    ConvArgs a = a0;
    a.nCoTiles = cdiv(a.Cout, 64);
    static const int tfast = getenv("FLAIR_DMA_TFAST") ? atoi(getenv("FLAIR_DMA_TFAST")) : 1;
    a.tFast = tfast && a.KT == 3 && a.T > 1;
    // round 4: the touch no longer pays (clip-level family 16.62 / 16.71 ms without, 16.73 / 16.74 with, same box) and the counters show
    // the touched lines fetched a second time by the real loads (profiles/r04_dma_read_traffic_by_shape.txt): off by default
    static const int resPf = getenv("FLAIR_DMA_RES_PREFETCH") ? atoi(getenv("FLAIR_DMA_RES_PREFETCH")) : 0;
    a.resPrefetch = resPf;
    const int nTiles = a.T * (a.H / TH) * (a.W / 32) * a.nCoTiles;
    const int nCu = flair_cu_count();                                  // of the CURRENT device
    int grid = nTiles < nCu ? (nTiles + 7) / 8 * 8 : nCu / 8 * 8;      // a multiple of 8: every XCD gets the same number of slots
    if (grid < 8) grid = 8;
    const int tilesPerXcd = (nTiles + 7) / 8;
    constexpr int HALO_INSTR = ((TH + 2) * 34 + 15) / 16;
    const size_t lds = NSTAGE * (size_t)(HALO_INSTR + 36) * 1024 + NW * 1024 + 512;  // stages + idle DMA slots' scratch + two bias slots
    if (NSTAGE == 3) FLAIR_CHECK(nTiles <= grid, "flair_conv_nhwc: the three-stage form runs one tile per workgroup");
    static LdsAttrOnce attr;
    {
        const void* fn;
        if constexpr (KS == 1) fn = reinterpret_cast<const void*>(&conv3x3_dma_kernel<NW, RPW, NSTAGE>);
        else fn = reinterpret_cast<const void*>(&conv3x3_dma_ks_kernel<NW, RPW, NSTAGE>);
        const hipError_t e = flair_max_lds_once(attr, fn);
        FLAIR_CHECK(e == hipSuccess, "flair_conv_nhwc: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if constexpr (KS == 1) hipLaunchKernelGGL((conv3x3_dma_kernel<NW, RPW, NSTAGE>), dim3(grid), dim3(64 * NW), lds, s, a, nTiles, tilesPerXcd);
    else hipLaunchKernelGGL((conv3x3_dma_ks_kernel<NW, RPW, NSTAGE>), dim3(grid), dim3(64 * NW), lds, s, a, nTiles, tilesPerXcd);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// One-tile 3x3 convolution with a THREE-deep weight ring (round 4): the per-frame convolutions of the 256^2 level (Cout <= 64, one
// 8 x 32 tile per workgroup, 4-7 K chunks).  conv3x3_dma_kernel<8, 1, 2> stages halo + weights of a chunk together in two 58 KB stages:
// chunk n + 1 is requested when chunk n starts to be multiplied and must have landed 2 304 MFMA cycles later -- it has not (measured
// ~2.2 us per chunk against 1.15 us of matrix work).  Three such stages do not fit the LDS; here the 36 KB weight images go through a
// ring of three (requested TWO chunks ahead) and the 22 KB halo images through a ring of two, 152 KB together, every wave issuing
// the same number of DMA instructions per phase so that one immediate vmcnt serves all waves (the recipe of conv_resident_kernel /
// conv_pair_kernel in chain.hip).  Epilogue from the accumulators; both residual inputs are requested at the head of the last chunk.
// bf16, stride 1, KT = 1, W % 32 == 0, H % 8 == 0, segments multiples of 32 channels, activations none / ReLU / LeakyReLU(0.1 | 0.2).
template <bool HALO3>       // false: halo ring of two + weight ring of three; true: halo ring of THREE + weight ring of two (138 KB)
__global__ __launch_bounds__(512, 2) void conv_frame_kernel(ConvArgs a) {
    prefetch_kernargs<sizeof(ConvArgs)>();
    using E = bf16_t;
    constexpr int NW = 8, HWP = 34;
    constexpr int HINSTR = 22, HIMG = HINSTR * 1024, WINSTR = 36, SLOT = WINSTR * 1024;
    constexpr int HR = HALO3 ? 3 : 2, WR = HALO3 ? 2 : 3;
    constexpr int RING = HR * HIMG, BIAS = RING + WR * SLOT;
    constexpr int NH = (HINSTR + NW - 1) / NW, NWS = (WINSTR + NW - 1) / NW;   // 3, 5
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int lrow = lane >> 2, lchunk = lane & 3;
    const int tilesW = a.W / 32, perFrame = tilesW * (a.H / 8);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = bid / perFrame, tile = bid - t * perFrame;
    const int h0 = (tile / tilesW) * 8, w0 = (tile % tilesW) * 32;
    const int nS = a.CinTot / 32;

    const u32x4_t wdesc = make_desc(a.w, a.wBytes);
    const u32x4_t bdesc = make_desc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    dma16(bdesc, (unsigned)(lane * 16), (unsigned)BIAS);
    // halo pixel of each of this wave's halo instructions (frame-relative pixel index, or "outside") and its swizzled 16-byte piece
    unsigned hpix[NH], hpiece[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int k = (wave * NH + i) % HINSTR;
        const int R = k * 16 + lrow;
        const int hr = R / HWP, c = R - hr * HWP;
        const int hh = h0 - 1 + hr, ww = w0 - 1 + c;
        const bool ok = R < 10 * HWP && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        hpix[i] = ok ? (unsigned)(hh * a.W + ww) : 0xffffffffu;
        hpiece[i] = (unsigned)((lchunk ^ ((c >> 2) & 3)) << 4);
    }
    unsigned wlane[NWS];
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
        const int id = (wave * NWS + i) % WINSTR;
        const int row = id * 16 + lrow, tap9 = row >> 6, co = row & 63;
        wlane[i] = co < a.Cout ? (unsigned)((co * 9 + tap9) * a.CinTot) * 2u + (unsigned)((lchunk ^ ((co >> 2) & 3)) << 4) : FLAIR_OOB;
    }
    // K walk of the halo issues over (segment, 32-channel chunk of the segment)
    int hseg = 0, hcb = 0;
    auto issue_halo = [&](int slot) {
        const unsigned ld = (unsigned)a.segLd[hseg] * 2u;
        const u32x4_t xdesc = make_desc(reinterpret_cast<const char*>(a.x[hseg]) + (size_t)t * a.H * a.W * ld, a.segBytes[hseg]);
        const unsigned cofs = (unsigned)(hcb * 64);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int k = (wave * NH + i) % HINSTR;
            dma16(xdesc, hpix[i] == 0xffffffffu ? FLAIR_OOB : hpix[i] * ld + cofs + hpiece[i], (unsigned)(slot * HIMG + k * 1024));
        }
        if (++hcb * 32 >= a.segC[hseg]) {
            hcb = 0;
            ++hseg;
        }
    };
    auto issue_w = [&](int s_) {
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int id = (wave * NWS + i) % WINSTR;
            dma16(wdesc, wlane[i] == FLAIR_OOB ? FLAIR_OOB : wlane[i] + (unsigned)(s_ * 64), (unsigned)(RING + (s_ % WR) * SLOT + id * 1024));
        }
    };
    issue_halo(0);
    issue_w(0);
    if (nS > 1) {
        issue_halo(1);
        issue_w(1);
    }
    if (HALO3 && nS > 2) issue_halo(2);

    unsigned aoff[2], boff[3][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = (unsigned)(lr * 64 + (((2 * ks + lh) ^ ((lr >> 2) & 3)) << 4));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            boff[kw][ks] = (unsigned)((wave * HWP + kw + lr) * 64 + (((2 * ks + lh) ^ (((kw + lr) >> 2) & 3)) << 4));
    }
    f32x16 acc[2];
    auto compute = [&](int wslot, int hslot) {
        const char* wb = smem + RING + wslot * SLOT;
        const char* xb = smem + hslot * HIMG;
        uint4 fb[2][3][2], fa[2][2][2];
        auto load_b = [&](int set, int kw) {
#pragma unroll
            for (int h = 0; h < 3; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[set][h][ks] = *reinterpret_cast<const uint4*>(xb + h * (HWP * 64) + boff[kw][ks]);
        };
        auto load_a = [&](int set, int kh, int kw) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[set][cf][ks] = *reinterpret_cast<const uint4*>(wb + (kh * 3 + kw) * 4096 + cf * 2048 + aoff[ks]);
        };
        load_b(0, 0);
        load_a(0, 0, 0);
#pragma unroll
        for (int step = 0; step < 9; ++step) {
            const int kq = step / 3, kh = step % 3;
            const int nkq = (step + 1) / 3, nkh = (step + 1) % 3;
            if (step < 8) {
                if (nkh == 0) load_b(nkq & 1, nkq);
                load_a((step + 1) & 1, nkh, nkq);
            }
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    acc[cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step & 1][cf][ks]), __builtin_bit_cast(bf16x8, fb[kq & 1][kh][ks]), acc[cf], 0, 0, 0);
        }
    };

    const long prow = ((long)t * a.H + h0 + wave) * a.W + w0;
    const unsigned yLdB = (unsigned)a.yLd * 2u, r0LdB = (unsigned)a.res0Ld * 2u, r1LdB = (unsigned)a.res1Ld * 2u;
    const __amdgpu_buffer_rsrc_t yd = make_rsrc(reinterpret_cast<char*>(a.y) + prow * a.yLd * 2, (unsigned)a.W * yLdB);
    const __amdgpu_buffer_rsrc_t r0d = make_rsrc(a.res0 ? reinterpret_cast<const char*>(a.res0) + prow * a.res0Ld * 2 : nullptr, a.res0 ? (unsigned)a.W * r0LdB : 0u);
    const __amdgpu_buffer_rsrc_t r1d = make_rsrc(a.res1 ? reinterpret_cast<const char*>(a.res1) + prow * a.res1Ld * 2 : nullptr, a.res1 ? (unsigned)a.W * r1LdB : 0u);
    uint4 r0v[4], r1v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) r0v[g] = r1v[g] = make_uint4(0u, 0u, 0u, 0u);

    for (int s_ = 0; s_ < nS; ++s_) {
        // halo s_ and weights s_ have landed; younger in flight: at s_ = 0 halo 1 + weights 1 (+ halo 2), later the weights of s_ + 1
        // (HALO3: the halo of s_ + 1, issued behind the weights of s_)
        if constexpr (HALO3) {
            if (s_ + 1 >= nS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (s_ == 0 && nS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NH + NWS) : "memory");
            else if (s_ == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH + NWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");      // the halo of s_ + 1 (prologue / behind the weights of s_)
        } else {
            if (s_ + 1 >= nS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (s_ == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH + NWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWS) : "memory");
        }
        __builtin_amdgcn_s_barrier();                    // ... for everybody; everybody is done with chunk s_ - 1
        if (s_ == 0) {
            const float* sb = reinterpret_cast<const float*>(smem + BIAS);
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bq = *reinterpret_cast<const float4*>(sb + cf * 32 + 8 * g + 4 * lh);
                    acc[cf][4 * g] = bq.x; acc[cf][4 * g + 1] = bq.y; acc[cf][4 * g + 2] = bq.z; acc[cf][4 * g + 3] = bq.w;
                }
        }
        if constexpr (HALO3) {
            if (s_ >= 1 && s_ + 1 < nS) issue_w(s_ + 1);                   // weight slot of chunk s_ - 1
            if (s_ >= 1 && s_ + 2 < nS) issue_halo((s_ + 2) % 3);          // halo slot of chunk s_ - 1
        } else {
            if (s_ >= 1 && s_ + 1 < nS) issue_halo((s_ + 1) & 1);          // halo slot of chunk s_ - 1
            if (s_ + 2 < nS) issue_w(s_ + 2);                              // weight slot of chunk s_ - 1
        }
        if (s_ + 1 == nS) {                                            // last chunk: the residual inputs, used in the epilogue
            if (a.res0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) r0v[g] = buf_load16(r0d, 16 * g + 8 * lh < a.Cout ? (unsigned)lr * r0LdB + (unsigned)(32 * g + 16 * lh) : FLAIR_OOB);
            }
            if (a.res1) {
#pragma unroll
                for (int g = 0; g < 4; ++g) r1v[g] = buf_load16(r1d, 16 * g + 8 * lh < a.Cout ? (unsigned)lr * r1LdB + (unsigned)(32 * g + 16 * lh) : FLAIR_OOB);
            }
        }
        compute(s_ % WR, s_ % HR);
    }

    const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : a.act == FLAIR_ACT_LRELU01 ? 0.1f : 0.2f;
    const float scale = a.outScale;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int g = 2 * cf + jj;                   // couts 16 g + 8 lh .. + 7
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[cf][8 * jj + e]), __float_as_uint(acc[cf][8 * jj + 4 + e]), false, false);
                v[e] = __uint_as_float(sw2[0]);
                v[4 + e] = __uint_as_float(sw2[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
            float r[8];
            Vec16<E>::load(reinterpret_cast<const E*>(&r0v[g]), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
            Vec16<E>::load(reinterpret_cast<const E*>(&r1v[g]), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] + r[e]) * scale;
            alignas(16) E out[8];
            Vec16<E>::store(out, v);
            const uint4 ov = *reinterpret_cast<const uint4*>(out);
            const unsigned yo = 16 * g + 8 * lh < a.Cout ? (unsigned)lr * yLdB + (unsigned)(32 * g + 16 * lh) : FLAIR_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd, (int)yo, 0, 0);
        }
}

static bool frame_kernel_ok(const ConvArgs& a) {
    if (a.esz != 2 || a.KT != 1 || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.fbias || a.Cout > 64 || a.Cout % 8 || a.W % 32 || a.H % 8) return false;
    if (!(a.act == FLAIR_ACT_NONE || a.act == FLAIR_ACT_RELU || a.act == FLAIR_ACT_LRELU01 || a.act == FLAIR_ACT_LRELU02)) return false;
    if (a.tapShift || a.reflect || a.CinTot % 32 || a.CinTot < 32) return false;
    for (int i = 0; i < a.nseg; ++i)
        if (a.segC[i] % 32) return false;
    return true;
}

template <bool HALO3>
static int launch_frame(const ConvArgs& a, hipStream_t s) {
    constexpr size_t lds = (HALO3 ? 3 * 22 + 2 * 36 : 2 * 22 + 3 * 36) * 1024 + 1024;
    static LdsAttrOnce attr;
    const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&conv_frame_kernel<HALO3>));
    FLAIR_CHECK(e == hipSuccess, "flair_conv_nhwc: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(conv_frame_kernel<HALO3>, dim3(a.T * (a.H / 8) * (a.W / 32)), dim3(512), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// TC x TP block tile (couts x pixels), 4 waves arranged WC x WP.
// PD = K steps whose operands are in flight in registers ahead of the step being multiplied.  One K step is 8 MFMAs per wave
// (64 x 64 tile: 2), far less than one round trip to the L2 / HBM, so with PD = 1 (rounds 1-2) every step lasted one load
// latency: 1x1 256 -> 128 on 16 x 128^2 ran at 1.7 TB/s, the deep-K 3x3x3 convolutions of the 16^2 .. 4^2 levels at ~0.7 us per step.
// KWID = 2 (bf16, every input segment a multiple of 64 channels): one K step = 64 channels = 128-byte LDS rows, so a staging load
// instruction covers 8 rows x 128 bytes (whole cache lines) instead of 16 rows x 64 bytes.  The 64-byte form keeps the texture
// addresser busy ~38 cycles per 1 KB instruction where whole lines take about half (profiles/r03_igemm_1x1_pmc.txt: TA busy 75 us
// of the 108 us 1x1 128 -> 64 convolution on 16 x 256^2), and half as many barriers per K.
template <typename E, int TC, int TP, int WC, int WP, int PD, int KWID>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    prefetch_kernargs<sizeof(ConvArgs)>();
    constexpr int BKE = Mma<E>::BKE * KWID;
    constexpr int ROWB = 64 * KWID;                // bytes per LDS row (one K step of one cout / pixel)
    constexpr int CPR = 4 * KWID;                  // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;                 // rows staged per pass of the 256 threads
    constexpr int VEC = ET<E>::VEC;
    constexpr int FC = TC / WC / 32;  // 32x32 fragments per wave along cout
    constexpr int FP = TP / WP / 32;  // ... along pixels
    constexpr int XR = TP / RPP;      // activation rows staged per thread
    constexpr int WR = TC / RPP;      // weight rows staged per thread
    static_assert(WC * WP == 4 && FC >= 1 && FP >= 1 && XR >= 1 && WR >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf][ W tile (TC rows) | X tile (TP rows) ], ROWB-byte rows; 128-byte rows: chunk XOR (row >> 1) & 7 puts the 16 rows
    // of a ds_read_b128 lane group on 16 distinct 16-byte slots of the 256-byte bank window
    constexpr int BUF = (TC + TP) * ROWB;
    auto loff = [](int row, int chunk) { return KWID == 1 ? lds_off(row, chunk) : row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the epilogue builds buffer descriptors from it)
    const int wc = wave / WP, wp = wave % WP;

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int coTile = bid % a.nCoTiles;
    const int pixTile = bid / a.nCoTiles;
    const long p0 = (long)pixTile * TP;
    const int co0 = coTile * TC;

    // ---- per-thread staging coordinates (fixed over the K loop) --------------
    const int chunk = tid % CPR;
    const int srow = tid / CPR;  // 0..RPP-1
    int xt[XR], xh[XR], xw[XR];
    bool xvalid[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        long p = p0 + srow + RPP * i;
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
    const int tFirst = (int)(p0 / ((long)a.H * a.W));                       // frame of the tile's first pixel
    const int tileFrames = (int)((TP + (long)a.H * a.W - 1) / ((long)a.H * a.W)) + 1;   // frames one tile can span

    uint4 xreg[PD][XR], wreg[PD][WR];

    // K-loop state (block uniform); with split-K, blockIdx.y owns steps [ks0, ks1)
    const int nkAll = taps * (a.CinTot / BKE);
    const int ks0 = (int)((long)blockIdx.y * nkAll / a.splitK);
    const int ks1 = (int)((long)(blockIdx.y + 1) * nkAll / a.splitK);
    int tap = ks0 / (a.CinTot / BKE), seg = 0, cb = ks0 % (a.CinTot / BKE), segOff = 0;
    while (cb * BKE >= a.segC[seg]) {
        cb -= a.segC[seg] / BKE;
        segOff += a.segC[seg];
        ++seg;
    }
    int dw = tap % a.KW - pw, dh = (tap / a.KW) % a.KH - ph, dt = tap / (a.KW * a.KH) - pt;

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w, a.wBytes);
    constexpr unsigned ESZ = sizeof(E);
    auto issue_loads = [&](int set) {
        // buffer resource = the frames this pixel tile can touch for the current temporal tap,
        // starting at frame fb (a whole clip may exceed the 2 GiB one resource can address)
        const int ld = a.segLd[seg];
        int fb = tFirst + dt;
        fb = fb < 0 ? 0 : (fb >= a.T ? a.T - 1 : fb);
        int nfr = a.T - fb;
        if (nfr > tileFrames) nfr = tileFrames;
        const size_t frameBytes = (size_t)a.Hin * a.Win * ld * ESZ;
        const __amdgpu_buffer_rsrc_t xr =
            make_rsrc(reinterpret_cast<const char*>(a.x[seg]) + fb * frameBytes, (unsigned)(nfr * frameBytes));
        const int coff = cb * BKE + chunk * VEC;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int t2 = xt[i] + dt;
            int h2 = xh[i] * a.stride + dh + a.tapShift, w2 = xw[i] * a.stride + dw + a.tapShift;
            if (a.reflect) {
                h2 = h2 < 0 ? -h2 : (h2 >= a.Hin ? 2 * a.Hin - 2 - h2 : h2);
                w2 = w2 < 0 ? -w2 : (w2 >= a.Win ? 2 * a.Win - 2 - w2 : w2);
            }
            const bool ok = xvalid[i] && (unsigned)t2 < (unsigned)a.T && (unsigned)h2 < (unsigned)a.Hin &&
                            (unsigned)w2 < (unsigned)a.Win;
            const unsigned off = ok ? (unsigned)((((t2 - fb) * a.Hin + h2) * a.Win + w2) * ld + coff) * ESZ : FLAIR_OOB;
            xreg[set][i] = buf_load16(xr, off);
        }
        const unsigned kofs = (unsigned)(tap * a.CinTot + segOff + coff) * ESZ;
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            const int n = co0 + srow + RPP * j;
            wreg[set][j] = buf_load16(wrs, n < a.Cout ? (unsigned)(n * taps * a.CinTot) * ESZ + kofs : FLAIR_OOB);
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
    auto write_lds = [&](int buf, int set) {
        char* base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < WR; ++j)
            *reinterpret_cast<uint4*>(base + loff(srow + RPP * j, chunk)) = wreg[set][j];
#pragma unroll
        for (int i = 0; i < XR; ++i)
            *reinterpret_cast<uint4*>(base + TC * ROWB + loff(srow + RPP * i, chunk)) = xreg[set][i];
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = ks1 - ks0;
    const int lr = lane & 31, lh = lane >> 5;

    // register set u holds step k with k % PD == u: issued PD steps before it is written to LDS (one step before it is multiplied)
#pragma unroll
    for (int u = 0; u < PD; ++u)
        if (u < nk) {
            issue_loads(u);
            advance();
        }
    write_lds(0, 0);
    __syncthreads();

    for (int ks = 0; ks < nk; ks += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int k = ks + u;
            if (k >= nk) break;                     // block uniform
            const int cur = k & 1;
            if (k + PD < nk) {                      // set u went to LDS in the previous step: free for step k + PD
                if (FLAIR_DBG(a) != 32) issue_loads(u);
                advance();
            }
            const char* wb = smem + cur * BUF;
            const char* xb = wb + TC * ROWB;
#pragma unroll
            for (int sp = 0; sp < KWID; ++sp) {            // 32 channels (two k16 MFMA steps for bf16) per pass
                uint4 af[FC][2], bfr[FP][2];
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const int row = wc * (TC / WC) + i * 32 + lr;
                    af[i][0] = *reinterpret_cast<const uint4*>(wb + loff(row, 4 * sp + Mma<E>::chunk(0, lh)));
                    af[i][1] = *reinterpret_cast<const uint4*>(wb + loff(row, 4 * sp + Mma<E>::chunk(1, lh)));
                }
#pragma unroll
                for (int j = 0; j < FP; ++j) {
                    const int row = wp * (TP / WP) + j * 32 + lr;
                    bfr[j][0] = *reinterpret_cast<const uint4*>(xb + loff(row, 4 * sp + Mma<E>::chunk(0, lh)));
                    bfr[j][1] = *reinterpret_cast<const uint4*>(xb + loff(row, 4 * sp + Mma<E>::chunk(1, lh)));
                }
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FP; ++j) Mma<E>::run(af[i], bfr[j], acc[i][j]);
            }
            if (k + 1 < nk) write_lds(cur ^ 1, (u + 1) % PD);
            __syncthreads();
        }
    }

    if (FLAIR_DBG(a) == 31) {                   // (timing switches of the diagnostic build: 31 no epilogue, 32 no loads past the first steps)
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int j = 0; j < FP; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));     // keep the K loop alive
        return;
    }
    // ---- epilogue: lane = pixel column, register quad = 4 consecutive couts ----
    if (a.splitK > 1) {   // raw f32 partials; bias/act/residual happen in the reduce kernel
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const long p = p0 + wp * (TP / WP) + j * 32 + lr;
            if (p >= a.P) continue;
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co0 + wc * (TC / WC) + i * 32 + 8 * g + 4 * lh;
                    if (co < a.Cout)
                        *reinterpret_cast<float4*>(a.part + ((long)blockIdx.y * a.P + p) * a.Cout + co) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2],
                                        acc[i][j][4 * g + 3]);
                }
        }
        return;
    }
    if ((a.Cout & 7) == 0 && ((uintptr_t)a.y & 15) == 0 && (a.yLd * (int)sizeof(E)) % 16 == 0 &&
        (!a.res0 || (((uintptr_t)a.res0 & 15) == 0 && (a.res0Ld * (int)sizeof(E)) % 16 == 0)) &&
        (!a.res1 || (((uintptr_t)a.res1 & 15) == 0 && (a.res1Ld * (int)sizeof(E)) % 16 == 0))) {
        auto run = [&](auto actTag) {
            igemm_epilogue<E, decltype(actTag)::value, FC, FP>(a, acc, p0 + wp * (TP / WP), co0 + wc * (TC / WC), lr, lh);
        };
        if (a.act == FLAIR_ACT_DCN_OFFSETS) run(std::integral_constant<int, 1>{});
        else if (a.act == FLAIR_ACT_SILU) run(std::integral_constant<int, 2>{});
        else if (a.act == FLAIR_ACT_GELU) run(std::integral_constant<int, 3>{});
        else run(std::integral_constant<int, 0>{});
        return;
    }
#pragma unroll
    for (int j = 0; j < FP; ++j) {       // (a run-time j would index the accumulator array dynamically: scratch memory)
        const long p = p0 + wp * (TP / WP) + j * 32 + lr;
        if (p >= a.P) continue;
#pragma unroll
        for (int i = 0; i < FC; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store_quad<E>(a, p, co0 + wc * (TC / WC) + i * 32 + 8 * g + 4 * lh, acc[i][j][4 * g],
                              acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    }
}

// split-K reduce: sum the f32 partials, then the usual epilogue
template <typename E>
__global__ void conv_splitk_reduce_kernel(ConvArgs a) {
    // (split-K runs on small launches only: P * Cout / 4 < 2^31, so 32-bit index arithmetic -- 64-bit division is a loop
    // here; four partials in flight per thread, summed in slice order so that the result does not depend on the batching)
    const unsigned cq = (unsigned)(a.Cout / 4);
    const unsigned quads = (unsigned)a.P * cq;
    const size_t slice = (size_t)a.P * a.Cout;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += gridDim.x * blockDim.x) {
        const unsigned p = i / cq;
        const int co = (int)(i - p * cq) * 4;
        const float* src = a.part + (size_t)p * a.Cout + co;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= a.splitK; k += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(k + u) * slice);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
            }
        }
        for (; k < a.splitK; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * slice);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        store_quad<E>(a, p, co, s.x, s.y, s.z, s.w);
    }
}

template <typename E, int TC, int TP, int WC, int WP, int PD, int KWID>
int launch_pd(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    a.nPixTiles = cdiv(a.P, TP);
    a.nCoTiles = cdiv(a.Cout, TC);
    const int grid = a.nPixTiles * a.nCoTiles;
    const size_t lds = 2 * (TC + TP) * 64 * KWID;
    hipLaunchKernelGGL((conv_igemm_kernel<E, TC, TP, WC, WP, PD, KWID>), dim3(grid, a.splitK), dim3(256), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    if (a.splitK > 1) {
        long g = (a.P * (a.Cout / 4) + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<E>, dim3((int)g), dim3(256), 0, s, a);
        FLAIR_LAUNCH_CHECK();
    }
    return FLAIR_OK;
}

// FLAIR_IGEMM_PD = 1 selects the one-step-ahead form of rounds 1-2 (A/B switch), default 4 steps in flight;
// FLAIR_IGEMM_WIDE = 0 keeps 32-channel K steps everywhere (default: 64-channel steps, two in flight, for bf16 convolutions whose
// input segments are all multiples of 64 channels -- the same bytes in flight as four 32-channel steps)
template <typename E, int TC, int TP, int WC, int WP>
int launch(const ConvArgs& a, hipStream_t s) {
    static const int pd = getenv("FLAIR_IGEMM_PD") ? atoi(getenv("FLAIR_IGEMM_PD")) : 4;
    static const int wide = getenv("FLAIR_IGEMM_WIDE") ? atoi(getenv("FLAIR_IGEMM_WIDE")) : 1;
    if constexpr (sizeof(E) == 2) {
        bool ok = wide && pd > 1;
        for (int i = 0; i < a.nseg; ++i) ok = ok && a.segC[i] % 64 == 0;
        if (ok) return launch_pd<E, TC, TP, WC, WP, 2, 2>(a, s);
    }
    return pd <= 1 ? launch_pd<E, TC, TP, WC, WP, 1, 1>(a, s) : launch_pd<E, TC, TP, WC, WP, 4, 1>(a, s);
}

// Split-K factor for the im2col path: deep-K convolutions on few pixels (the 16x16 .. 4x4
// levels: K up to 13,824, <= 256 tiles) would otherwise run one long serial K loop per CU.
int choose_split(const ConvArgs& a, int variant) {
    if (variant > 2) return 1;
    const int maxBlocks = variant == 0 ? 512 : 1024, maxSplit = 16;
    const int tc = variant == 0 ? 128 : 64, tp = variant == 2 ? 64 : 128;
    const int bke = 64 / (a.wBytes && a.CinTot ? (int)(a.wBytes / ((unsigned long long)a.Cout * a.KT * a.KH * a.KW * a.CinTot)) : 2);
    const long nk = (long)a.KT * a.KH * a.KW * (a.CinTot / bke);
    const long tiles = (long)cdiv(a.P, tp) * cdiv(a.Cout, tc);
    if (tiles >= 384 || nk < 32) return 1;
    long s = maxBlocks / tiles;
    if (s > maxSplit) s = maxSplit;
    if (s > nk / 8) s = nk / 8;
    return s < 1 ? 1 : (int)s;
}

// Kernel choice.  3..5: halo kernel with 8/4/2 image rows per workgroup (3x3 spatial taps,
// W a multiple of 32); 6/7: its K-split form (8/4 rows) when the launch is exactly one round of
// 256 workgroups; 0..2: im2col tiles 128 couts x 128 pixels, 64 x 128, 64 x 64.
// Either way keep >= ~2 workgroups per CU when the problem allows it.
int choose_variant(const ConvArgs& a) {
    if (!a.reflect && a.stride == 1 && a.KH == 3 && a.KW == 3 && a.W % 32 == 0 && a.H >= 2) {
        const long per = (long)a.T * (a.W / 32) * cdiv(a.Cout, 64);
        // single-round launches (<= one workgroup per CU): K-split kernel, twice the wavefronts
        const bool ks = a.Cout % 8 == 0;
        // persistent LDS-DMA kernel (bf16): launches of more than one round of 8-row tiles whose 16-row tiles fill most CUs
        static const bool useDma = !(getenv("FLAIR_CONV_DMA") && atoi(getenv("FLAIR_CONV_DMA")) == 0);
        if (useDma && a.esz == 2 && ks && a.act != FLAIR_ACT_GELU && a.H % 16 == 0 && per * cdiv(a.H, 8) > 256 && per * (a.H / 16) >= 192) return 8;
        // single-round launches (per-frame convolutions): the one-tile-per-workgroup forms of the LDS-DMA kernel (bf16)
        // FLAIR_CONV_DMA_FRAME: 0 K-split kernels everywhere, 1 (default) <8 rows> form at the 256^2 level, 2 also the <4 rows>
        // form at the 128^2 level.  Measured (profiles/r03_dma_switches.txt): a per-frame launch is launch (1.6 us) + cold first
        // fetch (~4 us: every CU pulls its 58 KB at once after the boundary invalidated the L2s) + epilogue / drain (~3 us)
        // whatever the staging mechanism: <8 rows> 12.6 vs 13.6 us on 64->64, equal on 224->64; <4 rows> 14.3 vs 11.9 us (one
        // wave per SIMD does not hide the per-chunk DMA round trips that 8 K-split waves do), hence off by default.
        // 3 (round 4): the <4 rows> form with EIGHT waves, K split like the K-split kernel (<8, 1, 3, KS = 2>): correct, +0.5 ms per step against
        // the register-staged K-split kernel (77.5 / 77.7 -> 78.1 / 78.1, same box): LDS-DMA staging does not pay on one-tile launches of this size.
        static const int dmaFrame = getenv("FLAIR_CONV_DMA_FRAME") ? atoi(getenv("FLAIR_CONV_DMA_FRAME")) : 1;
        const bool dmaOk = a.esz == 2 && ks && a.act != FLAIR_ACT_GELU;
        if (per * cdiv(a.H, 8) >= 256) return ks && per * cdiv(a.H, 8) <= 256 ? (dmaOk && dmaFrame >= 1 && a.H % 8 == 0 ? 9 : 6) : 3;
        if (per * cdiv(a.H, 4) >= 256) return ks && per * cdiv(a.H, 4) <= 256 ? (dmaOk && dmaFrame >= 3 && a.H % 4 == 0 ? 11 : dmaOk && dmaFrame == 2 && a.H % 4 == 0 ? 10 : 7) : 4;
        return 5;
    }
    const long tiles128 = (long)cdiv(a.P, 128) * cdiv(a.Cout, 128);
    if (a.Cout > 64 && tiles128 >= 512) return 0;
    // deep-K convolutions on few pixels (16x16 / 8x8 levels, K = 2304 .. 13,824): 128x128 tiles with split-K move half
    // the L2->LDS bytes of 64x64 tiles per flop; hipGraph-timed 256->256 3x3x3 at 16x16^2: 57.7 -> 43.4 us,
    // 512->512 at 16x8^2: 58.1 -> 45.2 us (4x4 stays on 64x64: 29.6 vs 34.0 us).  FLAIR_DEEPK_TILE128=0: round-1 choice.
    static const bool deepk128 = !(getenv("FLAIR_DEEPK_TILE128") && atoi(getenv("FLAIR_DEEPK_TILE128")) == 0);
    if (deepk128 && a.Cout >= 128 && a.P >= 1024 && tiles128 <= 64 && (long)a.KT * a.KH * a.KW * a.CinTot >= 2304) return 0;
    const long tiles64x128 = (long)cdiv(a.P, 128) * cdiv(a.Cout, 64);
    if (tiles64x128 >= 512) return 1;
    return 2;
}

template <typename E>
int dispatch(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int variant = choose_variant(a);
    a.splitK = a.part ? choose_split(a, variant) : 1;
    switch (variant) {
        case 0: return launch<E, 128, 128, 2, 2>(a, s);
        case 1: return launch<E, 64, 128, 1, 4>(a, s);
        case 2: return launch<E, 64, 64, 2, 2>(a, s);
        case 3: return launch_halo<E, 8, 1, 1>(a, s);
        case 4: return launch_halo<E, 4, 1, 1>(a, s);
        case 6: {
            // FLAIR_KS_RPW2=1: 8 waves of 2 rows x 64 couts (4 MFMAs per 4 LDS fragment reads) instead of 16 waves of 1 row
            // (2 per 3).  Neutral end to end (90.95 vs 90.84 ms/step, three same-box pairs) and slower per call under
            // rocprofv3 (26.1 vs 22.9 us): the one-row form stays the default.
            static const int rpw2 = getenv("FLAIR_KS_RPW2") ? atoi(getenv("FLAIR_KS_RPW2")) : 0;
            return rpw2 ? launch_halo_ks<E, 8, 2, 2>(a, s) : launch_halo_ks<E, 8, 1, 2>(a, s);
        }
        case 7: {
            // (two rows per wave leave 4 waves per CU here: +3.5 ms/step)
            // FLAIR_KS_CF1=1: one 32-cout fragment per wave, i.e. 16 waves per workgroup on the 4-row tiles of the 128^2 level
            static const int cf1 = getenv("FLAIR_KS_CF1") ? atoi(getenv("FLAIR_KS_CF1")) : 0;
            return cf1 ? launch_halo_ks<E, 4, 1, 1>(a, s) : launch_halo_ks<E, 4, 1, 2>(a, s);
        }
        case 8: {
            // FLAIR_DMA_RPW4=1: four waves of FOUR rows x 32 pixels x 64 couts (eight accumulators per wave, one wave per SIMD):
            // 72 fragment reads per 144 MFMAs instead of 60 per 72
            static const int rpw4 = getenv("FLAIR_DMA_RPW4") ? atoi(getenv("FLAIR_DMA_RPW4")) : 0;
            return rpw4 ? launch_dma<4, 4, 2>(a, s) : launch_dma<8, 2, 2>(a, s);
        }
        case 9: {
            // conv_frame_kernel for the shapes it takes (FLAIR_CONV_FRAME: 2 (default) halo ring of three + weight ring of two, 1 halo ring of
            // two + weight ring of three, 0 conv3x3_dma_kernel<8, 1, 2>).  Same box: per-frame family 19.59 / 19.55 -> 19.03 / 18.95 (1) ->
            // 19.03 / 19.09 (2) ms per step, step 70.89 / 70.92 -> 70.63 / 70.65 -> 70.54 / 70.57 ms.
            static const int frame = getenv("FLAIR_CONV_FRAME") ? atoi(getenv("FLAIR_CONV_FRAME")) : 2;
            if (frame == 1 && frame_kernel_ok(a)) return launch_frame<false>(a, s);
            if (frame == 2 && frame_kernel_ok(a)) return launch_frame<true>(a, s);
            return launch_dma<8, 1, 2>(a, s);
        }
        case 10: return launch_dma<4, 1, 3>(a, s);
        case 11: return launch_dma<8, 1, 3, 2>(a, s);
        default: return launch_halo<E, 2, 1, 1>(a, s);
    }
}

}  // namespace

extern "C" int flair_conv_variant(const flair_conv_params* p);

static void fill_geometry(ConvArgs& a, const flair_conv_params* p) {
    a.stride = p->stride > 1 ? p->stride : 1;
    a.Hin = p->H; a.Win = p->W;
    a.T = p->T; a.KT = p->KT; a.KH = p->KH; a.KW = p->KW; a.Cout = p->Cout;
    a.reflect = p->reflect_pad ? 1 : 0;
    a.act = p->act;
    a.H = (p->H + a.stride - 1) / a.stride;       // "same"-style padding K/2: out = ceil(in / stride)
    a.W = (p->W + a.stride - 1) / a.stride;
    a.P = (long)p->T * a.H * a.W;
    a.CinTot = 0;
    for (int i = 0; i < p->nseg && i < 4; ++i) a.CinTot += p->seg_c[i];
    const int esz = p->dtype == FLAIR_BF16 ? 2 : 4;
    a.esz = esz;
    a.wBytes = (unsigned)((unsigned long long)p->Cout * p->KT * p->KH * p->KW * a.CinTot * esz);
}

extern "C" size_t flair_conv_workspace_bytes(const flair_conv_params* p) {
    if (!p) return 0;
    ConvArgs a{};
    fill_geometry(a, p);
    const int split = choose_split(a, choose_variant(a));
    return split > 1 ? (size_t)split * a.P * a.Cout * sizeof(float) : 0;
}

extern "C" int flair_conv_nhwc(const flair_conv_params* p, const void* const* x, const void* w,
                               const float* bias, const float* frame_bias, const void* res0, const void* res1,
                               void* y, void* workspace, size_t workspace_bytes, hipStream_t stream) {
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
        // buffer resources cover single frames (halo kernels) or the few frames a pixel tile spans
        // (im2col kernels: at most 2 frames of this size), never the whole clip: only ONE FRAME has to
        // stay below 1 GiB
        const unsigned long long bytes = (unsigned long long)p->H * p->W * p->seg_ld[i] * esz;   // one input frame
        FLAIR_CHECK(bytes < 0x40000000ull, "flair_conv_nhwc: one frame of segment %d spans %llu bytes (limit 1 GiB)",
                    i, bytes);
        a.segBytes[i] = (unsigned)bytes;
    }
    {
        const unsigned long long wb = (unsigned long long)p->Cout * p->KT * p->KH * p->KW * a.CinTot * esz;
        FLAIR_CHECK(wb < 0x80000000ull, "flair_conv_nhwc: weights span %llu bytes (limit 2 GiB)", wb);
        a.wBytes = (unsigned)wb;
    }
    FLAIR_CHECK(p->y_ld >= p->Cout && (p->y_ld * esz) % 8 == 0 && ((uintptr_t)y) % 16 == 0,
                "flair_conv_nhwc: output stride/alignment");
    FLAIR_CHECK(((uintptr_t)w) % 16 == 0, "flair_conv_nhwc: weight alignment");
    FLAIR_CHECK(!bias || ((uintptr_t)bias) % 16 == 0, "flair_conv_nhwc: bias alignment");
    a.nseg = p->nseg;
    a.w = w;
    a.bias = bias;
    a.fbias = frame_bias;
    a.fbiasLd = p->frame_bias_ld;
    FLAIR_CHECK(!frame_bias || p->frame_bias_ld >= p->Cout, "flair_conv_nhwc: frame_bias_ld");
    {   // the halo / LDS-DMA epilogues address one frame of y / res0 / res1 through buffer descriptors with 32-bit byte offsets
        const unsigned long long hw = (unsigned long long)p->H * p->W * esz;
        FLAIR_CHECK(hw * p->y_ld < 0x40000000ull && (!res0 || hw * p->res_ld[0] < 0x40000000ull) && (!res1 || hw * p->res_ld[1] < 0x40000000ull),
                    "flair_conv_nhwc: one frame of the output / a residual spans more than 1 GiB");
    }
    a.res0 = res0;
    a.res1 = res1;
    a.res0Ld = p->res_ld[0];
    a.res1Ld = p->res_ld[1];
    a.y = y;
    a.yLd = p->y_ld;
    a.act = p->act;
    a.actParam = p->act_param;
    a.actPeriod = p->act_period;
    FLAIR_CHECK(p->act != FLAIR_ACT_DCN_OFFSETS || (p->act_period > 0 && p->act_period % 24 == 0),
                "flair_conv_nhwc: FLAIR_ACT_DCN_OFFSETS needs act_period = 3 * deform_groups (multiple of 24), got %d",
                p->act_period);
    a.outScale = p->out_scale;
    {
        ConvArgs g{};
        fill_geometry(g, p);
        a.stride = g.stride; a.Hin = g.Hin; a.Win = g.Win; a.T = g.T; a.H = g.H; a.W = g.W; a.P = g.P;
        a.KT = g.KT; a.KH = g.KH; a.KW = g.KW; a.Cout = g.Cout; a.esz = g.esz;
    }
    FLAIR_CHECK(a.stride == 1 || a.stride == 2, "flair_conv_nhwc: stride %d unsupported", p->stride);
    FLAIR_CHECK(!p->asym_pad || (a.stride == 2 && a.KH == a.KW && a.KT == 1 && p->H % 2 == 0 && p->W % 2 == 0),
                "flair_conv_nhwc: asym_pad needs a square 2-D stride-2 kernel on even frames");
    a.tapShift = p->asym_pad ? a.KH / 2 : 0;
    FLAIR_CHECK(!p->reflect_pad || (!p->asym_pad && a.KT == 1 && p->H > a.KH / 2 && p->W > a.KW / 2),
                "flair_conv_nhwc: reflect_pad needs a 2-D kernel smaller than the frame");
    a.reflect = p->reflect_pad ? 1 : 0;
    {
        static const int swz = getenv("FLAIR_CONV_LDS_SWZ") ? atoi(getenv("FLAIR_CONV_LDS_SWZ")) : 1;
        a.ldsSwz = swz;
    }
    a.part = nullptr;
    a.splitK = 1;
    {   // A/B switch: 7 selects the register-transposed (v_permlane32_swap) epilogue of the throughput halo kernel
        static const int swapEpilogue = getenv("FLAIR_CONV_SWAP_EPILOGUE") ? atoi(getenv("FLAIR_CONV_SWAP_EPILOGUE")) : 0;
        a.debug = swapEpilogue ? 7 : 0;
    }
#ifdef FLAIR_TIMING_SWITCHES
    {
        static const int dbg = getenv("FLAIR_CONV_DEBUG") ? atoi(getenv("FLAIR_CONV_DEBUG")) : 0;
        a.debug = dbg;
    }
#endif
    if (workspace && workspace_bytes >= flair_conv_workspace_bytes(p) && flair_conv_workspace_bytes(p) > 0) {
        FLAIR_CHECK(((uintptr_t)workspace) % 16 == 0, "flair_conv_nhwc: workspace alignment");
        a.part = reinterpret_cast<float*>(workspace);
    }
    if (p->dtype == FLAIR_BF16) return dispatch<bf16_t>(a, stream);
    return dispatch<float>(a, stream);
}

extern "C" int flair_conv_variant(const flair_conv_params* p) {
    if (!p) return -1;
    ConvArgs a{};
    fill_geometry(a, p);
    return choose_variant(a);
}
