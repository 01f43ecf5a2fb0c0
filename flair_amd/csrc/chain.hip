// Fused chains of 3x3 convolutions with the intermediate kept in LDS (halo recompute).
//
// The BasicVSR++ recurrence of the FLAIR UNet (guided_diffusion/unet_new.py:700-739,859-898) is a
// chain of ~2 700 DEPENDENT per-frame launches per forward; a 64->64 convolution of one 256x256
// frame is 2.7 us of matrix work inside ~12 us of launch / first fetch / write-back / drain
// (profiles/README.md).  This kernel runs TWO consecutive 3x3 convolutions of such a chain in one
// launch:
//
//   stage A (optional):  M = actA(conv3x3(cat(x[0..nseg)), WA) + biasA)       C channels, kept in LDS
//   stage B:             Y = (actB(conv3x3(M, WB) + biasB) + res0 + res1) * out_scale   CoutB channels
//
// and, without stage A, a convolution whose whole input halo (all C channels) stays resident in LDS
// while the workgroup walks the output channels in blocks of 64 (the c -> 27*G offset convolution:
// the halo is staged once instead of once per 64 output channels).
// Replaces, pairwise: conv_offset[2]+[4] / conv_offset[4]+[6] of SecondOrderDeformableAlignment
// (unet_new.py:859-867) and conv1+conv2 of mmedit's ResidualBlockNoBN inside
// ResidualBlocksWithInputConv (unet_new.py:659-668), including its two residual adds.
//
// A workgroup (8 wavefronts) owns a TH x TW output tile of one frame and ALL channels:
//   * stage A computes the (TH+2) x (TW+2) intermediate tile (1.33x the pixels at 8 x 32) from a
//     (TH+4) x (TW+4) input halo staged per 64-byte channel chunk (register prefetch of the next
//     chunk, unconditional buffer loads -> hardware zero padding), and writes it to LDS in the
//     element type (the same rounding point as the unfused pair), zeroing pixels outside the image
//     (they are stage B's zero padding, not convolution outputs);
//   * stage B reads its B operands straight from that LDS tile.
// MFMA operand roles, the 80-byte LDS pitch (conflict-free ds_read_b128) and the tap loop are those of
// conv3x3_halo_kernel (conv.hip); a work item is 32 pixels x 64 output channels on one wavefront.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct ChainArgs {
    const void* x[4];
    int segC[4], segLd[4], nseg;
    unsigned segBytes[4];
    const void* wA; const float* biasA; int actA; int CinA; unsigned wABytes;
    const void* wB; const float* biasB; int actB; int CoutB; unsigned wBBytes;
    const void* res0; const void* res1; int res0Ld, res1Ld;
    float outScale;
    float actParam; int actPeriod;   // FLAIR_ACT_DCN_OFFSETS on stage B
    void* y; int yLd;
    int T, H, W;
    unsigned long long* dbg;   // probe build only (-DFLAIR_CHAIN_STAMPS): per-workgroup s_memtime stamps
};

// In-kernel phase stamps of the diagnostic build (tools/probes/chain_probe.py); the product build
// compiles them out (no stamp executes, no argument is read).
#ifdef FLAIR_CHAIN_STAMPS
#include <stdlib.h>
#define STAMP(k)                                                                           \
    do {                                                                                   \
        if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <typename E> struct MmaC;
template <> struct MmaC<bf16_t> {
    static constexpr int BKE = 32;
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * i + half; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]),
                                                      __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]),
                                                      __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
    }
    static constexpr int MFMA_PER_STEP = 2;
};
template <> struct MmaC<float> {
    static constexpr int BKE = 16;
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * half + i; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        const float* af = reinterpret_cast<const float*>(a);
        const float* bf = reinterpret_cast<const float*>(b);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
    static constexpr int MFMA_PER_STEP = 8;
};

// activation of N values with ONE (wave-uniform) branch on the runtime code, not one per element
template <int N>
__device__ __forceinline__ void act_vec(float (&v)[N], int act) {
    if (act == FLAIR_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = silu_f(v[e]);
    } else if (act != FLAIR_ACT_NONE) {
        const float slope = act == FLAIR_ACT_RELU ? 0.f : 0.1f;      // max(v, slope * v) for slope in [0, 1)
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], v[e] * slope);
    }
}

constexpr int CHAIN_MAX_COUTB = 512;   // stage-B biases are staged in LDS once

template <typename E, int C, int TH, int TW, bool HASA>
__global__ __launch_bounds__(512, 2) void conv_chain_kernel(ChainArgs a) {
    prefetch_kernargs<sizeof(ChainArgs)>();
    constexpr int NT = 512, NW = 8;
    constexpr int BKE = MmaC<E>::BKE;
    constexpr int VEC = ET<E>::VEC;
    constexpr unsigned ESZ = sizeof(E);
    constexpr int PITCH = 80;
    constexpr int WM = TW + 2, RM = TH + 2, NPM = RM * WM;           // intermediate (stage-B input) region
    constexpr int WA = TW + 4, RA = TH + 4, NPA = RA * WA;           // stage-A input region
    constexpr int NCH = C / BKE;                                     // channel chunks of the intermediate
    constexpr int NB = C / 64;                                       // 64-cout blocks the weight buffer holds
    constexpr int WROWS = 64 * NB;
    constexpr int MID_BYTES = NCH * NPM * PITCH;
    constexpr int W_BYTES = WROWS * 9 * PITCH;
    constexpr int NGA = (NPM + 31) / 32, NITA = NGA * NB, MAXIA = (NITA + NW - 1) / NW;
    constexpr int NGB = TH * TW / 32, NITB = NGB * NB, MAXIB = (NITB + NW - 1) / NW;
    constexpr int IN_PIECES = (HASA ? NPA : NCH * NPM) * 4;          // 16-byte pieces per input staging
    constexpr int HI = (IN_PIECES + NT - 1) / NT;
    constexpr int W_PIECES = WROWS * 9 * 4;
    constexpr int WI = (W_PIECES + NT - 1) / NT;
    static_assert(TH * TW % 32 == 0 && C % 64 == 0, "tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smid = smem;
    char* sw = smem + MID_BYTES;
    char* sin = sw + W_BYTES;                                        // only with stage A
    float* sbias = reinterpret_cast<float*>(sin + (HASA ? NPA * PITCH : 0));   // [C] stage A | [CHAIN_MAX_COUTB] stage B

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int tilesW = a.W / TW, tilesH = (a.H + TH - 1) / TH;
    const int perFrame = tilesW * tilesH;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = bid / perFrame;
    const int tile = bid % perFrame;
    const int h0 = (tile / tilesW) * TH, w0 = (tile % tilesW) * TW;
    const long frameOff = (long)t * a.H * a.W;
    STAMP(0);
    // biases -> LDS once (a per-element global load in the epilogues would serialise on its latency)
    for (int i = C + tid; i < C + CHAIN_MAX_COUTB; i += NT)
        sbias[i] = (a.biasB && i - C < a.CoutB) ? a.biasB[i - C] : 0.f;

    // ---- staging coordinates of the input region (stage-A halo, or the intermediate itself) -------
    // piece id -> (chunk, pixel, 16-byte quarter); pixel -> frame pixel index or -1 (zero padding)
    int ipix[HI], idst[HI];
    unsigned iq[HI];
#pragma unroll
    for (int i = 0; i < HI; ++i) {
        const int id = i * NT + tid;
        constexpr int NP = HASA ? NPA : NPM, WR = HASA ? WA : WM, ORG = HASA ? 2 : 1;
        const int pc = id >> 2;                   // (chunk, pixel) index
        const int ch = HASA ? 0 : pc / NP;
        const int pix = HASA ? pc : pc % NP;
        const int r = pix / WR, c = pix % WR;
        const int hh = h0 - ORG + r, ww = w0 - ORG + c;
        const bool ok = id < IN_PIECES && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        ipix[i] = ok ? hh * a.W + ww : -1;
        iq[i] = (unsigned)(ch * BKE + (id & 3) * VEC) * ESZ;     // byte offset inside the pixel's channels
        idst[i] = id < IN_PIECES ? (ch * NP + pix) * PITCH + (id & 3) * 16 : -1;
    }
    uint4 hreg[2][HI], wreg[2][WI];        // two register sets: stage A keeps the operands of two K chunks in flight

    // K walk of stage A over (segment, chunk)
    int seg = 0, cb = 0, segOff = 0;
    auto issue_in = [&](int set) {
        const unsigned ld = (unsigned)a.segLd[seg] * ESZ;
        const char* frame = reinterpret_cast<const char*>(a.x[seg]) + (size_t)frameOff * ld;
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(frame, a.segBytes[seg]);
        const unsigned cofs = (unsigned)(cb * BKE) * ESZ;
#pragma unroll
        for (int i = 0; i < HI; ++i)
            hreg[set][i] = buf_load16(xr, ipix[i] >= 0 ? (unsigned)ipix[i] * ld + cofs + iq[i] : FLAIR_OOB);
    };
    auto advance_in = [&]() {
        ++cb;
        if (cb * BKE >= a.segC[seg]) {
            cb = 0;
            segOff += a.segC[seg];
            ++seg;
        }
    };
    // weights of `rows` output channels starting at co0, K chunk at element offset kofs of a [Cout][9][cin] pack
    auto issue_w = [&](int set, const __amdgpu_buffer_rsrc_t& wrs, int co0, int coutTot, int cin, int kofs) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int id = i * NT + tid;
            const int row = id >> 2;
            const int co = row / 9, tap9 = row % 9;
            const bool ok = id < W_PIECES && co0 + co < coutTot;
            wreg[set][i] = buf_load16(wrs, ok ? (unsigned)(((co0 + co) * 9 + tap9) * cin + kofs + (id & 3) * VEC) * ESZ
                                              : FLAIR_OOB);
        }
    };
    auto write_w = [&](int set) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int id = i * NT + tid;
            if (id < W_PIECES) *reinterpret_cast<uint4*>(sw + (id >> 2) * PITCH + (id & 3) * 16) = wreg[set][i];
        }
    };
    auto write_in = [&](char* dst, int set) {
#pragma unroll
        for (int i = 0; i < HI; ++i)
            if (idst[i] >= 0) *reinterpret_cast<uint4*>(dst + idst[i]) = hreg[set][i];
    };

    // 9 taps x one K chunk for one work item (32 pixels x 64 couts): B fragments from `pb` (this lane's
    // pixel of the input region, row width inW pixels), A fragments from weight rows wb0 / wb1
    auto compute = [&](const char* pb, int inW, const char* wb0, const char* wb1, f32x16 (&acc)[2]) {
        uint4 fa0[2][2], fa1[2][2], fb[2][2];
        auto load_tap = [&](int set, int tap9) {
            const int kh = tap9 / 3, kw = tap9 % 3;
            fa0[set][0] = *reinterpret_cast<const uint4*>(wb0 + tap9 * PITCH + 16 * MmaC<E>::chunk(0, lh));
            fa0[set][1] = *reinterpret_cast<const uint4*>(wb0 + tap9 * PITCH + 16 * MmaC<E>::chunk(1, lh));
            fa1[set][0] = *reinterpret_cast<const uint4*>(wb1 + tap9 * PITCH + 16 * MmaC<E>::chunk(0, lh));
            fa1[set][1] = *reinterpret_cast<const uint4*>(wb1 + tap9 * PITCH + 16 * MmaC<E>::chunk(1, lh));
            const char* hp = pb + (kh * inW + kw) * PITCH;
            fb[set][0] = *reinterpret_cast<const uint4*>(hp + 16 * MmaC<E>::chunk(0, lh));
            fb[set][1] = *reinterpret_cast<const uint4*>(hp + 16 * MmaC<E>::chunk(1, lh));
        };
        load_tap(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int tap9 = 0; tap9 < 9; ++tap9) {
            const int set = tap9 & 1;
            if (tap9 < 8) load_tap(set ^ 1, tap9 + 1);
            MmaC<E>::run(fa0[set], fb[set], acc[0]);
            MmaC<E>::run(fa1[set], fb[set], acc[1]);
            if (tap9 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 2 * MmaC<E>::MFMA_PER_STEP, 0);
        }
    };

    const __amdgpu_buffer_rsrc_t wBrs = make_rsrc(a.wB, a.wBBytes);
    const int nBatch = (a.CoutB + WROWS - 1) / WROWS;

    if constexpr (HASA) {
        // ================================ stage A ================================
        const __amdgpu_buffer_rsrc_t wArs = make_rsrc(a.wA, a.wABytes);
        const int nkA = a.CinA / BKE;
        // K chunk k lives in register set k & 1: requested two chunks ahead of its multiplication, written to LDS one chunk ahead.
        // With one set (round 2) a chunk was in flight only while the previous one was multiplied -- 0.5-1 us, less than one round
        // trip to the L2 -- and every chunk ended waiting for its successor (stamps: stage-A K loop 12.4 k cycles against an MFMA
        // floor of 6.9 k).  "Chunk nkA" is the first weight chunk of stage B.
        auto issue_step = [&](int k, int set) {
            if (k < nkA) {
                issue_in(set);
                issue_w(set, wArs, 0, C, a.CinA, segOff + cb * BKE);
                advance_in();
            } else if (k == nkA) {
                issue_w(set, wBrs, 0, a.CoutB, C, 0);
            }
        };
        issue_step(0, 0);
        issue_step(1, 1);
        // this lane's pixel of the intermediate region for each of its items; the accumulators start at the
        // bias (lane layout of the 32x32 MFMA result: register r <-> cout 8*(r/4) + 4*lh + r%4 of the fragment)
        f32x16 accA[MAXIA][2];
        int pbOff[MAXIA], wbOff[MAXIA];
#pragma unroll
        for (int ii = 0; ii < MAXIA; ++ii) {
            const int it = wave + ii * NW;
            const int g = it % NGA, blk = (it / NGA) % NB;
            int q = g * 32 + lr;
            if (q >= NPM) q = NPM - 1;                            // tail lanes recompute the last pixel (not stored)
            pbOff[ii] = ((q / WM) * WA + (q % WM)) * PITCH;
            wbOff[ii] = ((blk * 64 + lr) * 9) * PITCH;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.biasA) bq = *reinterpret_cast<const float4*>(a.biasA + blk * 64 + i * 32 + 8 * gq + 4 * lh);
                    accA[ii][i][4 * gq] = bq.x; accA[ii][i][4 * gq + 1] = bq.y;
                    accA[ii][i][4 * gq + 2] = bq.z; accA[ii][i][4 * gq + 3] = bq.w;
                }
        }
        write_in(sin, 0);
        write_w(0);
        __syncthreads();
        STAMP(1);
        auto step = [&](int k, auto setTag) {
            constexpr int SET = decltype(setTag)::value;
            issue_step(k + 2, SET);                                // set SET went to LDS before this chunk's barrier: free again
#pragma unroll
            for (int ii = 0; ii < MAXIA; ++ii)
                if (wave + ii * NW < NITA)
                    compute(sin + pbOff[ii], WA, sw + wbOff[ii], sw + wbOff[ii] + 32 * 9 * PITCH, accA[ii]);
            __syncthreads();                                       // everyone is done with the staged chunk
            if (k + 1 < nkA) write_in(sin, SET ^ 1);
            write_w(SET ^ 1);                                      // chunk k + 1 (k + 1 == nkA: stage B's first weights)
            if (k + 1 < nkA) __syncthreads();
        };
        for (int k = 0; k < nkA; k += 2) {
            step(k, std::integral_constant<int, 0>{});
            if (k + 1 < nkA) step(k + 1, std::integral_constant<int, 1>{});
        }
        STAMP(2);
        // ---- intermediate -> LDS (bias, activation, zero outside the image, element type rounding)
#pragma unroll
        for (int ii = 0; ii < MAXIA; ++ii) {
            const int it = wave + ii * NW;
            if (it >= NITA) continue;
            const int g = it % NGA, blk = (it / NGA) % NB;
            const int q = g * 32 + lr;
            if (q >= NPM) continue;
            const int hh = h0 - 1 + q / WM, ww = w0 - 1 + q % WM;
            const bool inimg = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int co = blk * 64 + i * 32 + 8 * gq + 4 * lh;
                    float v[4] = {accA[ii][i][4 * gq], accA[ii][i][4 * gq + 1], accA[ii][i][4 * gq + 2],
                                  accA[ii][i][4 * gq + 3]};
                    act_vec<4>(v, a.actA);
                    char* dst = smid + ((co / BKE) * NPM + q) * PITCH + (co % BKE) * ESZ;
                    if constexpr (sizeof(E) == 4) {
                        *reinterpret_cast<float4*>(dst) = inimg ? make_float4(v[0], v[1], v[2], v[3])
                                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        uint2 pk;
                        pk.x = inimg ? pack2bf(v[0], v[1]) : 0u;
                        pk.y = inimg ? pack2bf(v[2], v[3]) : 0u;
                        *reinterpret_cast<uint2*>(dst) = pk;
                    }
                }
        }
        __syncthreads();
    } else {
        // the intermediate IS the input: all channel chunks of the (TH+2) x (TW+2) halo, staged once
        issue_in(0);
        issue_w(0, wBrs, 0, a.CoutB, C, 0);
        write_in(smid, 0);
        write_w(0);
        __syncthreads();
    }
    STAMP(3);

    // ================================ stage B ================================
    int pbB[MAXIB], wbB[MAXIB];
#pragma unroll
    for (int ii = 0; ii < MAXIB; ++ii) {
        const int it = wave + ii * NW;
        const int g = it % NGB, blk = it / NGB;
        const int q = g * 32 + lr;
        pbB[ii] = ((q / TW) * WM + (q % TW)) * PITCH;
        wbB[ii] = ((blk * 64 + lr) * 9) * PITCH;
    }
    for (int batch = 0; batch < nBatch; ++batch) {
        f32x16 acc[MAXIB][2];
#pragma unroll
        for (int ii = 0; ii < MAXIB; ++ii) {
            const int blk = ((wave + ii * NW) / NGB) % NB;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bq = *reinterpret_cast<const float4*>(sbias + C + batch * WROWS + blk * 64 + i * 32 +
                                                                       8 * gq + 4 * lh);
                    acc[ii][i][4 * gq] = bq.x; acc[ii][i][4 * gq + 1] = bq.y;
                    acc[ii][i][4 * gq + 2] = bq.z; acc[ii][i][4 * gq + 3] = bq.w;
                }
        }
        for (int ch = 0; ch < NCH; ++ch) {
            // prefetch the next (batch, chunk) weights
            const bool lastChunk = ch + 1 == NCH;
            const bool more = !lastChunk || batch + 1 < nBatch;
            if (more) issue_w(0, wBrs, lastChunk ? (batch + 1) * WROWS : batch * WROWS, a.CoutB, C, lastChunk ? 0 : (ch + 1) * BKE);
#pragma unroll
            for (int ii = 0; ii < MAXIB; ++ii)
                if (wave + ii * NW < NITB)
                    compute(smid + ch * NPM * PITCH + pbB[ii], WM, sw + wbB[ii], sw + wbB[ii] + 32 * 9 * PITCH, acc[ii]);
            __syncthreads();                                       // sw is free
            if (!lastChunk) {
                write_w(0);
                __syncthreads();
            }
        }
        if (batch == 0) STAMP(4);
        // ---- epilogue of this batch, straight from the accumulators (no LDS round trip, no barrier).  A lane of
        // the 32x32 MFMA result holds 4 consecutive couts of its pixel per register quad (quad g: couts 8g + 4*lh ..);
        // v_permlane32_swap between the two half-waves turns quads 2j, 2j+1 into 8 consecutive couts per lane
        // (lower half: 16j .. 16j+7, upper half: 16j+8 .. 16j+15), i.e. one 16-byte bf16 store (f32: the quad is
        // already 16 bytes) and residual loads of the same shape.
#pragma unroll
        for (int ii = 0; ii < MAXIB; ++ii) {
            const int it = wave + ii * NW;
            if (it >= NITB) continue;                               // wave-uniform: every lane of a wave takes part in the swaps
            const int g = it % NGB, blk = it / NGB;
            const int q = g * 32 + lr;
            const int hh = h0 + q / TW, ww = w0 + q % TW;
            const long p = frameOff + (long)hh * a.W + ww;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int cofrag = batch * WROWS + blk * 64 + i * 32;
                if constexpr (sizeof(E) == 4) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int co = cofrag + 8 * gq + 4 * lh;
                        float v[4] = {acc[ii][i][4 * gq], acc[ii][i][4 * gq + 1], acc[ii][i][4 * gq + 2], acc[ii][i][4 * gq + 3]};
                        if (a.actB == FLAIR_ACT_DCN_OFFSETS)
                            dcn_offset_act<4>(v, co, a.actParam, a.actPeriod);
                        else
                            act_vec<4>(v, a.actB);
                        if (co < a.CoutB && hh < a.H) {
                            if (a.res0) {
                                float r[4];
                                Vec16<E>::load(reinterpret_cast<const E*>(a.res0) + p * a.res0Ld + co, r);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r[e];
                            }
                            if (a.res1) {
                                float r[4];
                                Vec16<E>::load(reinterpret_cast<const E*>(a.res1) + p * a.res1Ld + co, r);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += r[e];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= a.outScale;
                            Vec16<E>::store(reinterpret_cast<E*>(a.y) + p * a.yLd + co, v);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int co = cofrag + 16 * j + 8 * lh;    // after the swap
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const auto sw2 = __builtin_amdgcn_permlane32_swap(
                                __float_as_uint(acc[ii][i][8 * j + e]), __float_as_uint(acc[ii][i][8 * j + 4 + e]), false, false);
                            v[e] = __uint_as_float(sw2[0]);
                            v[4 + e] = __uint_as_float(sw2[1]);
                        }
                        if (a.actB == FLAIR_ACT_DCN_OFFSETS)
                            dcn_offset_act<8>(v, co, a.actParam, a.actPeriod);
                        else
                            act_vec<8>(v, a.actB);
                        if (co < a.CoutB && hh < a.H) {
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
        }
        if (batch == 0) STAMP(5);
        if (batch + 1 < nBatch) {
            write_w(0);
            __syncthreads();
        }
    }
    STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------------
// Input-resident 3x3 convolution with many output channels, LDS-DMA weight ring (round 4): the c -> 27*G offset
// convolution of SecondOrderDeformableAlignment at c = 64 (unet_new.py:859-867 conv_offset[6]; 32.6 GFLOP per 256^2 frame,
// 150 launches per step).  The form above (HASA = false) staged each (64-cout block, 32-channel chunk) of the weights through
// registers (load -> ds_write -> barrier -> multiply -> barrier: two barriers and a 46 KB LDS write pass per 36 MFMAs of a
// wave) and ran the tanh / sigmoid epilogue of a block with nothing beside it: 48.7 us per launch = 0.27 of the MFMA peak.
// Here:
//   * the (8+2) x 34 pixel halo of all 64 input channels is brought in ONCE by LDS-DMA (two 64-byte-row images, one per
//     32-channel chunk, 16-byte pieces XOR-swizzled on the source address like conv3x3_dma_kernel);
//   * the weights stream through a ring of three 36 KB stages (one stage = 64 couts x 9 taps x 32 channels), stages s + 1 and
//     s + 2 in flight by LDS-DMA while stage s is multiplied, ONE barrier per stage, counted vmcnt;
//   * two accumulator sets alternate between consecutive 64-cout blocks: the epilogue of block b (bias from LDS, activation,
//     v_permlane32_swap -> 16-byte buffer stores) is issued in four pieces BETWEEN the taps of block b + 1's MFMAs, where the
//     vector ALU is otherwise idle; only the last block's epilogue is exposed.
// A wave owns one image row (32 pixels) of the 8 x 32 tile x the 64 couts of the current block.  bf16, C = 64, W % 32 == 0,
// H % 8 == 0, no residual inputs.
struct ResArgs {
    const void* x; int xLd; unsigned xBytes;      // one input frame
    const void* w; unsigned wBytes;               // [Cout][9][64]
    const float* bias;
    int Cout, act;
    float actParam; int actPeriod;
    float outScale;
    void* y; int yLd;
    int T, H, W;
    unsigned long long* stamps;   // diagnostic build, debug == 20: per-workgroup sums of s_memtime deltas (wait, barrier, compute, total)
    int debug;            // phase switches of the diagnostic build (-DFLAIR_TIMING_SWITCHES): 1 no MFMA phase, 2 no weight DMA after
                          // the prologue, 3 no epilogue, 4 return after the prologue
};
#ifdef FLAIR_TIMING_SWITCHES
#define RES_DBG(a) ((a).debug)
#else
#define RES_DBG(a) 0
#endif

// NW wavefronts per workgroup, each owning RPW = 8 / NW image rows (32 pixels each) x the 64 couts of the current block:
//   <8 waves x 1 row>: 6 fragment reads per 4 MFMAs (1.5 ds_read_b128 per MFMA: three quarters of the LDS read rate at the full
//                      matrix rate), two waves per SIMD;
//   <4 waves x 2 rows>: the (RPW + 2) halo rows of a column tap are read once for its three row taps and a weight fragment
//                      serves both rows: 20 reads per 24 MFMAs (0.83), one wave per SIMD with up to 512 registers.
template <int ACT, int NW>      // ACT 0: max(v, slope v) (none / ReLU / LeakyReLU)   1: DCN offsets / masks   2: SiLU
__global__ __launch_bounds__(64 * NW, NW / 4) void conv_resident_kernel(ResArgs a) {
    prefetch_kernargs<sizeof(ResArgs)>();
    using E = bf16_t;
    constexpr int RPW = 8 / NW;
    constexpr int HWP = 34, HROWS = 10 * HWP, HINSTR = (HROWS + 15) / 16, HBYTES = HINSTR * 1024;    // 22 KB per chunk image
    constexpr int WINSTR = 36, SLOT = WINSTR * 1024, NRING = 3;
    constexpr int RING = 2 * HBYTES, BIAS = RING + NRING * SLOT;                                      // + 2 KB of f32 biases
    constexpr int NH = (HINSTR + NW - 1) / NW, NWS = (WINSTR + NW - 1) / NW;    // DMA instructions per wave: one halo chunk image, one weight stage
    constexpr int NST = 2 * RPW;                    // epilogue stores per wave and stage (4 groups x RPW rows per block)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int lrow = lane >> 2, lchunk = lane & 3;
    const int tilesW = a.W / 32, perFrame = tilesW * (a.H / 8);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = bid / perFrame, tile = bid - t * perFrame;
    const int h0 = (tile / tilesW) * 8, w0 = (tile % tilesW) * 32;
    const int nBlk = (a.Cout + 63) >> 6, nS = 2 * nBlk;

    const unsigned ldB = (unsigned)a.xLd * 2u;
    const u32x4_t xdesc = make_desc(reinterpret_cast<const char*>(a.x) + (size_t)t * a.H * a.W * ldB, a.xBytes);
    const u32x4_t wdesc = make_desc(a.w, a.wBytes);
    const u32x4_t bdesc = make_desc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);

    // ---- prologue DMA, in the order of first use: biases (waves duplicate the two 1 KB pieces), halo chunk 0, weight stage 0,
    // halo chunk 1, weight stage 1 -- stage 0 needs only the first three, so the first wait leaves chunk 1 and stage 1 in flight.
    // Every wave issues the same number of instructions per phase (surplus ids repeat earlier ones: same bytes to the same
    // place), so that one immediate vmcnt serves all waves.
    dma16(bdesc, (unsigned)((wave & 1) * 1024 + lane * 16), (unsigned)(BIAS + (wave & 1) * 1024));
    auto issue_halo = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int k = (wave * NH + i) % HINSTR;
            const int R = k * 16 + lrow;
            const int hr = R / HWP, c = R - hr * HWP;
            const int hh = h0 - 1 + hr, ww = w0 - 1 + c;
            const bool ok = R < HROWS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
            const unsigned voff = ok ? (unsigned)(hh * a.W + ww) * ldB + (unsigned)(ch * 64) + (unsigned)((lchunk ^ ((c >> 2) & 3)) << 4) : FLAIR_OOB;
            dma16(xdesc, voff, (unsigned)(ch * HBYTES + k * 1024));
        }
    };
    issue_halo(0);
    // weight slots of this wave: instruction ids (wave * NWS + i) % 36; lane part of the source offset and its cout inside the block
    unsigned wlane[NWS];
    int wco[NWS];
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
        const int id = (wave * NWS + i) % WINSTR;
        const int row = id * 16 + lrow, tap9 = row >> 6, co = row & 63;
        wlane[i] = (unsigned)((co * 9 + tap9) * 64) * 2u + (unsigned)((lchunk ^ ((co >> 2) & 3)) << 4);
        wco[i] = co;
    }
    auto issue_w = [&](int s_) {                                    // stage s_ = (block, chunk) -> ring slot s_ % 3
        const int blk = s_ >> 1, ch = s_ & 1;
        const unsigned base = (unsigned)(blk * 64 * 9 * 64) * 2u + (unsigned)(ch * 64);
        const unsigned dst = (unsigned)(RING + (s_ % NRING) * SLOT);
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int id = (wave * NWS + i) % WINSTR;
            dma16(wdesc, blk * 64 + wco[i] < a.Cout ? base + wlane[i] : FLAIR_OOB, dst + (unsigned)(id * 1024));
        }
    };
    auto issue_w_one = [&](int s_, int i) {                         // instruction i of stage s_ (the stages after the prologue are
        const int blk = s_ >> 1, ch = s_ & 1;                       // issued one instruction at a time between the taps' MFMAs)
        const unsigned base = (unsigned)(blk * 64 * 9 * 64) * 2u + (unsigned)(ch * 64);
        const int id = (wave * NWS + i) % WINSTR;
        dma16(wdesc, blk * 64 + wco[i] < a.Cout ? base + wlane[i] : FLAIR_OOB, (unsigned)(RING + (s_ % NRING) * SLOT + id * 1024));
    };
    issue_w(0);
    issue_halo(1);
    if (nS > 1) issue_w(1);

    // ---- fragment read offsets (per lane, fixed): A = weight row tap9 * 64 + cf * 32 + lr, B = halo row (RPW wave + h) * 34 + kw + lr
    unsigned aoff[2], boff[3][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = (unsigned)(lr * 64 + (((2 * ks + lh) ^ ((lr >> 2) & 3)) << 4));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
            boff[kw][ks] = (unsigned)((RPW * wave * HWP + kw + lr) * 64 + (((2 * ks + lh) ^ (((kw + lr) >> 2) & 3)) << 4));
    }

    // ---- epilogue of one 64-cout block, one (row, 16-cout group) piece at a time (all lanes take part in the swaps).  The
    // accumulators start at the bias.  DCN offsets / masks: out = A * rcp(1 + 2^(k v)) + B with per-lane constants
    // (residues: mag * tanh(v) = mag - 2 mag / (1 + e^{2v}); masks: 1 / (1 + e^{-v})), three vector and two transcendental
    // instructions per value.
    const float slope = a.act == FLAIR_ACT_NONE ? 1.f : a.act == FLAIR_ACT_RELU ? 0.f : 0.1f;
    const unsigned yLdB = (unsigned)a.yLd * 2u;
    const long prow = ((long)t * a.H + h0 + RPW * wave) * a.W + w0;        // first pixel of the wave's first row
    const __amdgpu_buffer_rsrc_t yd = make_rsrc(reinterpret_cast<char*>(a.y) + prow * a.yLd * 2, (unsigned)(RPW * a.W) * yLdB);
    const unsigned yLane = (unsigned)lr * yLdB + 16u * lh;
    const float* sbias = reinterpret_cast<const float*>(smem + BIAS);
    const float scale = a.outScale;
    auto epi_piece = [&](const f32x16 (&accE)[RPW][2], int blk, int piece) {
        const int j = piece / 4, g = piece % 4;
        const int i = g >> 1, jj = g & 1;
        const int co = blk * 64 + i * 32 + 16 * jj;                  // first cout of the group (wave-uniform); this lane: + 8 * lh
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(accE[j][i][8 * jj + e]), __float_as_uint(accE[j][i][8 * jj + 4 + e]), false, false);
            v[e] = __uint_as_float(sw2[0]);
            v[4 + e] = __uint_as_float(sw2[1]);
        }
        if constexpr (ACT == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope) * scale;
        } else if constexpr (ACT == 1) {
            const bool residue = 3 * ((co + 8 * lh) % a.actPeriod) < 2 * a.actPeriod;
            const float kk = residue ? 2.885390081777927f : -1.4426950408889634f;       // 2 log2(e) | -log2(e)
            const float A = residue ? -2.f * a.actParam * scale : scale, B = residue ? a.actParam * scale : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(kk * v[e])), A, B);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= scale * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[e]));
        }
        alignas(16) E out[8];
        Vec16<E>::store(out, v);
        const uint4 ov = *reinterpret_cast<const uint4*>(out);
        const unsigned yo = co + 8 * lh < a.Cout ? yLane + (unsigned)(j * a.W) * yLdB + 2u * (unsigned)co : FLAIR_OOB;
        // (plain stores: write-through `sc1` / `sc0 sc1` stores measured 176-177 us against 178.6 us on the conv_offset[6] ->
        // alignment pair, non-temporal ones 188 us: profiles/README.md, round 4)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd, (int)yo, 0, 0);
    };

    // ---- one stage = 3 column taps x 3 row taps x (RPW rows x 2 cout fragments x 2 k-steps) MFMAs.  The fragments of step n + 1
    // (A: 4 reads; B: 2 (RPW + 2) more when the column tap changes) are requested before the MFMAs of step n;
    // `hook(step)` runs after the MFMAs of a step have been issued (pieces of the previous block's epilogue).
    auto compute = [&](int slot, int ch, f32x16 (&acc)[RPW][2], auto&& hook) {
        const char* wb = smem + RING + slot * SLOT;
        const char* xb = smem + ch * HBYTES;
        uint4 fb[2][RPW + 2][2];   // [set][halo row RPW wave + h][k-step]
        uint4 fa[2][2][2];         // [set][cout fragment][k-step]
        auto load_b = [&](int set, int kw) {
#pragma unroll
            for (int h = 0; h < RPW + 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fb[set][h][ks] = *reinterpret_cast<const uint4*>(xb + h * (HWP * 64) + boff[kw][ks]);
        };
        auto load_a = [&](int set, int kh, int kw) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fa[set][cf][ks] = *reinterpret_cast<const uint4*>(wb + (kh * 3 + kw) * 4096 + cf * 2048 + aoff[ks]);
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
            for (int j = 0; j < RPW; ++j)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) {
                    acc[j][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step & 1][cf][0]),
                                                                         __builtin_bit_cast(bf16x8, fb[kq & 1][j + kh][0]), acc[j][cf], 0, 0, 0);
                    acc[j][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step & 1][cf][1]),
                                                                         __builtin_bit_cast(bf16x8, fb[kq & 1][j + kh][1]), acc[j][cf], 0, 0, 0);
                }
            hook(step);
        }
    };
    // accumulators of a block start at its biases (register r of a fragment <-> cout 8 (r / 4) + 4 lh + r % 4)
    auto init_acc = [&](f32x16 (&acc)[RPW][2], int blk) {
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(sbias + blk * 64 + cf * 32 + 8 * q + 4 * lh);
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    acc[j][cf][4 * q] = bq.x; acc[j][cf][4 * q + 1] = bq.y; acc[j][cf][4 * q + 2] = bq.z; acc[j][cf][4 * q + 3] = bq.w;
                }
            }
    };
    // wait until weight stage s_ (and everything older: halo, biases) has landed for this wave.  Younger operations at this point:
    // the epilogue stores of the two previous stages (NST each once block 1 has started) and the DMA of stage s_ + 1 (NWS).
    auto wait_stage = [&](int s_) {
        // stage s_ >= 2 was issued DURING stage s_ - 2, its last instruction behind that stage's output stores; younger at this
        // point: the stores and the DMA instructions of stage s_ - 1
        if (s_ + 1 >= nS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s_ == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH + NWS) : "memory");       // halo chunk 1 and stage 1 stay in flight
        else if (s_ <= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWS + NST) : "memory");
    };

#ifdef FLAIR_TIMING_SWITCHES
    unsigned long long stW = 0, stB = 0, stC = 0, stT0 = 0;
    const unsigned long long stStart = __builtin_amdgcn_s_memtime();
#endif
    f32x16 acc0[RPW][2], acc1[RPW][2];
    // one block = two stages; `accP` accumulates block blk, the epilogue of `accQ` (block blk - 1) runs beside it
    auto block = [&](int blk, f32x16 (&accP)[RPW][2], f32x16 (&accQ)[RPW][2]) {
        const bool prev = blk > 0;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int s_ = 2 * blk + ch;
#ifdef FLAIR_TIMING_SWITCHES
            unsigned long long tA = 0, tB = 0, tC = 0;
            if (a.debug == 20) tA = __builtin_amdgcn_s_memtime();
#endif
            wait_stage(s_);
#ifdef FLAIR_TIMING_SWITCHES
            if (a.debug == 20) tB = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_s_barrier();            // stage s_ is in LDS for everybody; everybody is done with stage s_ - 1
#ifdef FLAIR_TIMING_SWITCHES
            if (a.debug == 20) {
                tC = __builtin_amdgcn_s_memtime();
                stW += tB - tA;
                stB += tC - tB;
                stT0 = tC;
            }
#endif
            const bool dma = s_ + 2 < nS && RES_DBG(a) != 2;          // ring slot (s_ + 2) % 3 = (s_ - 1) % 3 is free from here on
            if (RES_DBG(a) == 4) return;
            if (ch == 0) init_acc(accP, blk);        // (the biases landed with the first stage)
            if (RES_DBG(a) == 1) continue;
            // the DMA instructions of stage s_ + 2 go out one per step, the last one BEHIND the stage's output stores (wait_stage
            // counts on that order); issued in one batch ahead of the MFMAs they cost ~100 cycles each with nothing beside them
            auto dma_step = [&](int step) {
                if constexpr (RPW == 1) {
                    constexpr int at[9] = {-1, -1, 0, 1, 2, -1, 3, 4, -1};
                    if (dma && at[step] >= 0) issue_w_one(s_ + 2, at[step]);
                } else {
                    if (dma) issue_w_one(s_ + 2, step);              // NWS == 9
                }
            };
            if (prev && RES_DBG(a) != 3) {
                compute(s_ % NRING, ch, accP, [&](int step) {
                    // 4 RPW pieces per block, 2 RPW per stage, spread over the nine steps
                    if constexpr (RPW == 1) {
                        if (step == 1) epi_piece(accQ, blk - 1, 2 * ch);
                        if (step == 5) epi_piece(accQ, blk - 1, 2 * ch + 1);
                    } else {
                        if (step == 0 || step == 2 || step == 4 || step == 6) epi_piece(accQ, blk - 1, 4 * ch + step / 2);
                    }
                    dma_step(step);
                });
            } else {
                compute(s_ % NRING, ch, accP, [&](int step) { dma_step(step); });
            }
#ifdef FLAIR_TIMING_SWITCHES
            if (a.debug == 20) {
                __builtin_amdgcn_sched_barrier(0);
                stC += __builtin_amdgcn_s_memtime() - stT0;
            }
#endif
        }
    };
    for (int blk = 0; blk < nBlk; blk += 2) {
        block(blk, acc0, acc1);
        if (blk + 1 < nBlk) block(blk + 1, acc1, acc0);
    }
#ifdef FLAIR_TIMING_SWITCHES
    if (a.debug == 20 && a.stamps && lane == 0) {
        const unsigned long long tE = __builtin_amdgcn_s_memtime();
        unsigned long long* d = a.stamps + ((size_t)blockIdx.x * NW + wave) * 4;
        d[0] = stW; d[1] = stB; d[2] = stC; d[3] = tE - stStart;
    }
#endif
    // the last block's epilogue is exposed
    if (RES_DBG(a) == 3) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc0[j][0][r]), "v"(acc0[j][1][r]), "v"(acc1[j][0][r]), "v"(acc1[j][1][r]));
        return;
    }
    if (nBlk & 1) {
#pragma unroll
        for (int g = 0; g < 4 * RPW; ++g) epi_piece(acc0, nBlk - 1, g);
    } else {
#pragma unroll
        for (int g = 0; g < 4 * RPW; ++g) epi_piece(acc1, nBlk - 1, g);
    }
}

int launch_resident(const ChainArgs& c, hipStream_t s) {
    ResArgs a;
    a.x = c.x[0]; a.xLd = c.segLd[0]; a.xBytes = c.segBytes[0];
    a.w = c.wB; a.wBytes = c.wBBytes; a.bias = c.biasB; a.Cout = c.CoutB; a.act = c.actB;
    a.actParam = c.actParam; a.actPeriod = c.actPeriod; a.outScale = c.outScale;
    a.y = c.y; a.yLd = c.yLd; a.T = c.T; a.H = c.H; a.W = c.W;
    a.debug = 0;
    a.stamps = nullptr;
#ifdef FLAIR_TIMING_SWITCHES
    a.debug = getenv("FLAIR_RES_DEBUG") ? atoi(getenv("FLAIR_RES_DEBUG")) : 0;
    if (const char* e = getenv("FLAIR_RES_STAMPS")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16));
#endif
    constexpr size_t lds = 2 * 22 * 1024 + 3 * 36 * 1024 + 2048;
    const int grid = a.T * (a.H / 8) * (a.W / 32);
    // FLAIR_CONV_RESIDENT_WAVES = 4: four waves of two rows (default: eight waves of one row; same-box step 81.7 vs 82.1 ms)
    static const int nw = getenv("FLAIR_CONV_RESIDENT_WAVES") ? atoi(getenv("FLAIR_CONV_RESIDENT_WAVES")) : 8;
    auto go = [&](auto tag, auto nwTag) -> int {
        constexpr int ACT = decltype(tag)::value, NW = decltype(nwTag)::value;
        static LdsAttrOnce attr;
        const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&conv_resident_kernel<ACT, NW>));
        FLAIR_CHECK(e == hipSuccess, "flair_conv_chain: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL((conv_resident_kernel<ACT, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
        FLAIR_LAUNCH_CHECK();
        return FLAIR_OK;
    };
    auto pick = [&](auto tag) -> int {
        return nw == 8 ? go(tag, std::integral_constant<int, 8>{}) : go(tag, std::integral_constant<int, 4>{});
    };
    if (a.act == FLAIR_ACT_DCN_OFFSETS) return pick(std::integral_constant<int, 1>{});
    if (a.act == FLAIR_ACT_SILU) return pick(std::integral_constant<int, 2>{});
    return pick(std::integral_constant<int, 0>{});
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair of 3x3 convolutions c = 64 -> 64 -> 64 on LDS-DMA staging (round 4; FLAIR_CONV_PAIR=0 selects the older form): the HASA form of
// conv_chain_kernel above spends 7.8 k cycles of its 32.5 k on the first fetch (load -> register -> ds_write -> barrier), 11.0 k on
// a stage-A K loop whose MFMA floor is 6.9 k (two barriers per chunk), 4.1 k on moving the intermediate to LDS in 8-byte pieces and
// 7.2 k on stage B (floor 4.6 k) -- tools/probes/chain_probe.py.  Here, as in conv_resident_kernel:
//   * the (8+4) x 36 input halo of both 32-channel chunks arrives ONCE by LDS-DMA (2 x 27 KB, 64-byte rows, 16-byte pieces
//     XOR-swizzled by the pixel column on the source address);
//   * the four weight stages (A chunk 0 / 1, B chunk 0 / 1; 36 KB each) go through two ring slots, the next stage in flight while one is
//     multiplied, counted vmcnt, ONE s_barrier per stage;
//   * the intermediate (10 x 34 pixels x 64 channels, activation applied, zero outside the image, rounded to bf16 like the unfused
//     pair) is written over the input halo in 16-byte pieces (v_permlane32_swap gives a lane 8 consecutive channels) in the layout
//     stage B reads conflict-free;
//   * stage B = conv3x3_dma_kernel's one-row form, epilogue from the accumulators with both residual inputs requested before it.
// bf16, one 64-channel input segment, CoutB = 64, W % 32 == 0, H % 8 == 0, activations none / ReLU / LeakyReLU(0.1).
struct PairArgs {
    const void* x; int xLd; unsigned xBytes;      // one input frame
    const void* wA; const void* wB; unsigned wBytes;
    const float* biasA; const float* biasB;
    int actA, actB;
    float outScale;
    const void* res0; const void* res1; int res0Ld, res1Ld;
    void* y; int yLd;
    int T, H, W;
};

__global__ __launch_bounds__(512, 2) void conv_pair_kernel(PairArgs a) {
    prefetch_kernargs<sizeof(PairArgs)>();
    using E = bf16_t;
    constexpr int NW = 8;
    constexpr int WIN = 36, IN_INSTR = 27, IN_IMG = IN_INSTR * 1024;          // 12 x 36 = 432 halo pixels = 27 DMA instructions per chunk
    constexpr int WMID = 34, NPMID = 10 * WMID, MID_IMG = 22 * 1024;           // 340 intermediate pixels per chunk image (over the halo)
    constexpr int WINSTR = 36, SLOT = WINSTR * 1024;
    constexpr int RING = 2 * IN_IMG, BIAS = RING + 2 * SLOT;                   // + 2 KB: biases of stage A | stage B
    constexpr int NH = (IN_INSTR + NW - 1) / NW, NWS = (WINSTR + NW - 1) / NW; // 4, 5 DMA instructions per wave
    constexpr int NITEM = (NPMID + 31) / 32;                                   // 11 fragments of 32 intermediate pixels
    static_assert(2 * MID_IMG <= RING && NITEM <= 2 * NW, "layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int lrow = lane >> 2, lchunk = lane & 3;
    const int tilesW = a.W / 32, perFrame = tilesW * (a.H / 8);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int t = bid / perFrame, tile = bid - t * perFrame;
    const int h0 = (tile / tilesW) * 8, w0 = (tile % tilesW) * 32;

    const unsigned ldB = (unsigned)a.xLd * 2u;
    const u32x4_t xdesc = make_desc(reinterpret_cast<const char*>(a.x) + (size_t)t * a.H * a.W * ldB, a.xBytes);
    const u32x4_t wAdesc = make_desc(a.wA, a.wBytes), wBdesc = make_desc(a.wB, a.wBytes);
    const u32x4_t bAdesc = make_desc(a.biasA, a.biasA ? 256u : 0u), bBdesc = make_desc(a.biasB, a.biasB ? 256u : 0u);

    // ---- prologue DMA in the order of first use: biases, halo chunk 0, weights A0 | halo chunk 1, weights A1
    dma16(bAdesc, (unsigned)(lane * 16), (unsigned)BIAS);
    dma16(bBdesc, (unsigned)(lane * 16), (unsigned)(BIAS + 1024));
    auto issue_halo = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int k = (wave * NH + i) % IN_INSTR;
            const int R = k * 16 + lrow;
            const int hr = R / WIN, c = R - hr * WIN;
            const int hh = h0 - 2 + hr, ww = w0 - 2 + c;
            const bool ok = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
            const unsigned voff = ok ? (unsigned)(hh * a.W + ww) * ldB + (unsigned)(ch * 64) + (unsigned)((lchunk ^ ((c >> 2) & 3)) << 4) : FLAIR_OOB;
            dma16(xdesc, voff, (unsigned)(ch * IN_IMG + k * 1024));
        }
    };
    unsigned wlane[NWS];
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
        const int id = (wave * NWS + i) % WINSTR;
        const int row = id * 16 + lrow, tap9 = row >> 6, co = row & 63;
        wlane[i] = (unsigned)((co * 9 + tap9) * 64) * 2u + (unsigned)((lchunk ^ ((co >> 2) & 3)) << 4);
    }
    auto issue_w = [&](const u32x4_t& wd, int ch, int slot) {
#pragma unroll
        for (int i = 0; i < NWS; ++i) {
            const int id = (wave * NWS + i) % WINSTR;
            dma16(wd, (unsigned)(ch * 64) + wlane[i], (unsigned)(RING + slot * SLOT + id * 1024));
        }
    };
    issue_halo(0);
    issue_w(wAdesc, 0, 0);
    issue_halo(1);
    issue_w(wAdesc, 1, 1);

    // ---- stage A: work item it = 32 intermediate pixels q = 32 it + lr (row-major in the 10 x 34 region) x 64 channels;
    // wave w owns items w and w + 8 (items 8 .. 10: waves 0 .. 2)
    const bool two = wave + NW < NITEM;                                 // wave-uniform
    int q_[2], qr_[2], qc_[2];
    unsigned boffA[2][3][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        int q = (wave + ii * NW) * 32 + lr;
        if (q >= NPMID) q = NPMID - 1;                                  // tail lanes / idle second item recompute the last pixel (not stored)
        q_[ii] = q; qr_[ii] = q / WMID; qc_[ii] = q - qr_[ii] * WMID;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                boffA[ii][kw][ks] = (unsigned)((qr_[ii] * WIN + qc_[ii] + kw) * 64 + (((2 * ks + lh) ^ (((qc_[ii] + kw) >> 2) & 3)) << 4));
    }
    unsigned aoff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) aoff[ks] = (unsigned)(lr * 64 + (((2 * ks + lh) ^ ((lr >> 2) & 3)) << 4));

    const float* sbias = reinterpret_cast<const float*>(smem + BIAS);
    f32x16 accA[2][2];
    auto init_from_bias = [&](f32x16 (&acc)[2], const float* b) {
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(b + cf * 32 + 8 * g + 4 * lh);
                acc[cf][4 * g] = bq.x; acc[cf][4 * g + 1] = bq.y; acc[cf][4 * g + 2] = bq.z; acc[cf][4 * g + 3] = bq.w;
            }
    };
    // one chunk of stage A: per tap one set of weight fragments serves both items of the wave
    auto compute_A = [&](int slot, int ch) {
        const char* wb = smem + RING + slot * SLOT;
        const char* xb = smem + ch * IN_IMG;
        uint4 fa[2][2][2], fb[2][2][2];      // [set][cout fragment | item][k-step]
        auto load_tap = [&](int set, int tap9) {
            const int kh = tap9 / 3, kw = tap9 % 3;
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[set][cf][ks] = *reinterpret_cast<const uint4*>(wb + tap9 * 4096 + cf * 2048 + aoff[ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[set][0][ks] = *reinterpret_cast<const uint4*>(xb + kh * (WIN * 64) + boffA[0][kw][ks]);
            if (two) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[set][1][ks] = *reinterpret_cast<const uint4*>(xb + kh * (WIN * 64) + boffA[1][kw][ks]);
            }
        };
        load_tap(0, 0);
#pragma unroll
        for (int tap9 = 0; tap9 < 9; ++tap9) {
            const int set = tap9 & 1;
            if (tap9 < 8) load_tap(set ^ 1, tap9 + 1);
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    accA[0][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][cf][ks]), __builtin_bit_cast(bf16x8, fb[set][0][ks]), accA[0][cf], 0, 0, 0);
            if (two) {
#pragma unroll
                for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        accA[1][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][cf][ks]), __builtin_bit_cast(bf16x8, fb[set][1][ks]), accA[1][cf], 0, 0, 0);
            }
        }
    };

    // stage 0 (A, chunk 0): biases, halo chunk 0 and weights A0 have landed; halo chunk 1 and weights A1 (NH + NWS) stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH + NWS) : "memory");
    __builtin_amdgcn_s_barrier();
    init_from_bias(accA[0], sbias);
    init_from_bias(accA[1], sbias);
    compute_A(0, 0);
    // stage 1 (A, chunk 1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // everybody is done with ring slot 0
    issue_w(wBdesc, 0, 0);                               // weights B0 fly during this stage and the intermediate pass
    compute_A(1, 1);
    __builtin_amdgcn_s_barrier();                        // everybody is done with the input halo and ring slot 1
    issue_w(wBdesc, 1, 1);

    // ---- intermediate -> LDS over the halo: activation, zero outside the image, bf16; lane = pixel q, 8 consecutive channels per piece
    {
        const float slopeA = a.actA == FLAIR_ACT_NONE ? 1.f : a.actA == FLAIR_ACT_RELU ? 0.f : 0.1f;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            if (ii == 1 && !two) break;                  // wave-uniform
            const int qraw = (wave + ii * NW) * 32 + lr;
            const int hh = h0 - 1 + qr_[ii], ww = w0 - 1 + qc_[ii];
            const bool inimg = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
            const unsigned sw = (unsigned)((qc_[ii] >> 2) & 3);
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(accA[ii][cf][8 * jj + e]), __float_as_uint(accA[ii][cf][8 * jj + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw2[0]);
                        v[4 + e] = __uint_as_float(sw2[1]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = inimg ? fmaxf(v[e], v[e] * slopeA) : 0.f;
                    alignas(16) E out[8];
                    Vec16<E>::store(out, v);
                    // channels cf * 32 + 16 jj + 8 lh .. + 7: chunk image cf, 16-byte piece 2 jj + lh of the pixel's 64-byte row
                    if (qraw < NPMID)
                        *reinterpret_cast<uint4*>(smem + cf * MID_IMG + q_[ii] * 64 + (((2 * jj + lh) ^ sw) << 4)) = *reinterpret_cast<const uint4*>(out);
                }
        }
    }

    // ---- stage B: wave = output row `wave` of the tile x 32 pixels x 64 channels
    const long prow = ((long)t * a.H + h0 + wave) * a.W + w0;                 // first pixel of the wave's row
    const unsigned yLdB = (unsigned)a.yLd * 2u, r0LdB = (unsigned)a.res0Ld * 2u, r1LdB = (unsigned)a.res1Ld * 2u;
    const __amdgpu_buffer_rsrc_t yd = make_rsrc(reinterpret_cast<char*>(a.y) + prow * a.yLd * 2, (unsigned)a.W * yLdB);
    const __amdgpu_buffer_rsrc_t r0d = make_rsrc(a.res0 ? reinterpret_cast<const char*>(a.res0) + prow * a.res0Ld * 2 : nullptr, a.res0 ? (unsigned)a.W * r0LdB : 0u);
    const __amdgpu_buffer_rsrc_t r1d = make_rsrc(a.res1 ? reinterpret_cast<const char*>(a.res1) + prow * a.res1Ld * 2 : nullptr, a.res1 ? (unsigned)a.W * r1LdB : 0u);
    // both residual inputs of the wave's pixels (16 bytes per (cout group, lane)): requested now, used in the epilogue.  Without a
    // residual no load is issued (zero-sized resource + a uniform branch), so the counted waits below know how many are in flight.
    uint4 r0v[4], r1v[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) r0v[g] = r1v[g] = make_uint4(0u, 0u, 0u, 0u);
    const int nres = (a.res0 ? 4 : 0) + (a.res1 ? 4 : 0);                      // wave-uniform
    if (a.res0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) r0v[g] = buf_load16(r0d, (unsigned)lr * r0LdB + (unsigned)(32 * g + 16 * lh));
    }
    if (a.res1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) r1v[g] = buf_load16(r1d, (unsigned)lr * r1LdB + (unsigned)(32 * g + 16 * lh));
    }
    unsigned boffB[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            boffB[kw][ks] = (unsigned)((wave * WMID + kw + lr) * 64 + (((2 * ks + lh) ^ (((kw + lr) >> 2) & 3)) << 4));
    f32x16 accB[2];
    auto compute_B = [&](int slot, int ch) {
        const char* wb = smem + RING + slot * SLOT;
        const char* xb = smem + ch * MID_IMG;
        uint4 fb[2][3][2], fa[2][2][2];
        auto load_b = [&](int set, int kw) {
#pragma unroll
            for (int h = 0; h < 3; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[set][h][ks] = *reinterpret_cast<const uint4*>(xb + h * (WMID * 64) + boffB[kw][ks]);
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
                    accB[cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step & 1][cf][ks]), __builtin_bit_cast(bf16x8, fb[kq & 1][kh][ks]), accB[cf], 0, 0, 0);
        }
    };
    // stage 2 (B, chunk 0): weights B0 landed (younger: weights B1 and the residual loads), the intermediate is written
    if (nres == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWS) : "memory");
    else if (nres == 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWS + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NWS + 8) : "memory");
    __builtin_amdgcn_s_barrier();
    init_from_bias(accB, sbias + 256);
    compute_B(0, 0);
    // stage 3 (B, chunk 1)
    if (nres == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (nres == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute_B(1, 1);

    // ---- epilogue from the accumulators: activation, + res0 + res1, scale, 16-byte stores
    const float slopeB = a.actB == FLAIR_ACT_NONE ? 1.f : a.actB == FLAIR_ACT_RELU ? 0.f : 0.1f;
    const float scale = a.outScale;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int g = 2 * cf + jj;                   // couts 16 g + 8 lh .. + 7
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(accB[cf][8 * jj + e]), __float_as_uint(accB[cf][8 * jj + 4 + e]), false, false);
                v[e] = __uint_as_float(sw2[0]);
                v[4 + e] = __uint_as_float(sw2[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slopeB);
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
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd, (int)((unsigned)lr * yLdB + (unsigned)(32 * g + 16 * lh)), 0, 0);
        }
}

int launch_pair(const ChainArgs& c, hipStream_t s) {
    PairArgs a;
    a.x = c.x[0]; a.xLd = c.segLd[0]; a.xBytes = c.segBytes[0];
    a.wA = c.wA; a.wB = c.wB; a.wBytes = 64u * 9u * 64u * 2u;
    a.biasA = c.biasA; a.biasB = c.biasB; a.actA = c.actA; a.actB = c.actB; a.outScale = c.outScale;
    a.res0 = c.res0; a.res1 = c.res1; a.res0Ld = c.res0Ld; a.res1Ld = c.res1Ld;
    a.y = c.y; a.yLd = c.yLd; a.T = c.T; a.H = c.H; a.W = c.W;
    constexpr size_t lds = 2 * 27 * 1024 + 2 * 36 * 1024 + 2048;
    static LdsAttrOnce attr;
    const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&conv_pair_kernel));
    FLAIR_CHECK(e == hipSuccess, "flair_conv_chain: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(conv_pair_kernel, dim3(c.T * (c.H / 8) * (c.W / 32)), dim3(512), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

template <typename E, int C, int TH, int TW, bool HASA>
int launch_chain(const ChainArgs& a, hipStream_t s) {
    constexpr int BKE = MmaC<E>::BKE;
    constexpr size_t lds = (size_t)(C / BKE) * (TH + 2) * (TW + 2) * 80 + (size_t)C * 9 * 80 +
                           (HASA ? (size_t)(TH + 4) * (TW + 4) * 80 : 0) + (size_t)(C + CHAIN_MAX_COUTB) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static LdsAttrOnce attr;
    {
        const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&conv_chain_kernel<E, C, TH, TW, HASA>));
        FLAIR_CHECK(e == hipSuccess, "flair_conv_chain: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int grid = a.T * cdiv(a.H, TH) * (a.W / TW);
    hipLaunchKernelGGL((conv_chain_kernel<E, C, TH, TW, HASA>), dim3(grid), dim3(512), lds, s, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// Tile choice: 8 x 32 pixels (one image row per MFMA pixel group) when that still gives >= 256 workgroups
// per frame or the clip is long, else 8 x 8 (the 128x128 level with c = 128: 256 workgroups of 64 pixels);
// f32 halves the tile height so that the f32 intermediate fits the LDS.
template <typename E, bool HASA>
int dispatch_chain(const ChainArgs& a, int C, hipStream_t s) {
    constexpr int TALL = sizeof(E) == 4 ? 4 : 8;
    const bool wide = a.W % 32 == 0 && (long)a.T * cdiv(a.H, TALL) * (a.W / 32) >= 192;
    if (C == 64) return wide ? launch_chain<E, 64, TALL, 32, HASA>(a, s) : launch_chain<E, 64, TALL, 8, HASA>(a, s);
    return launch_chain<E, 128, TALL, 8, HASA>(a, s);
}

}  // namespace

extern "C" int flair_conv_chain(const flair_chain_params* p, const void* const* x, const void* wA, const float* biasA,
                                const void* wB, const float* biasB, const void* res0, const void* res1, void* y,
                                hipStream_t stream) {
    FLAIR_CHECK(p && x && wB && y, "flair_conv_chain: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_conv_chain: bad dtype %d", p->dtype);
    FLAIR_CHECK(p->nseg >= 1 && p->nseg <= 4, "flair_conv_chain: nseg %d not in 1..4", p->nseg);
    FLAIR_CHECK(p->T > 0 && p->H > 0 && p->W > 0, "flair_conv_chain: empty shape");
    FLAIR_CHECK(p->C == 64 || p->C == 128, "flair_conv_chain: intermediate width %d unsupported (64 or 128)", p->C);
    FLAIR_CHECK(p->W % 8 == 0, "flair_conv_chain: W %% 8 must be 0 (got %d x %d)", p->H, p->W);
    const int esz = p->dtype == FLAIR_BF16 ? 2 : 4, bke = p->dtype == FLAIR_BF16 ? 32 : 16;
    FLAIR_CHECK(p->CoutB > 0 && p->CoutB % (16 / esz) == 0 && p->CoutB <= CHAIN_MAX_COUTB,
                "flair_conv_chain: CoutB %d must be a multiple of %d and at most %d", p->CoutB, 16 / esz, CHAIN_MAX_COUTB);
    ChainArgs a{};
    int cin = 0;
    for (int i = 0; i < p->nseg; ++i) {
        FLAIR_CHECK(x[i], "flair_conv_chain: segment %d is null", i);
        FLAIR_CHECK(p->seg_c[i] > 0 && p->seg_c[i] % bke == 0, "flair_conv_chain: segment %d has %d channels; must be a "
                    "multiple of %d", i, p->seg_c[i], bke);
        FLAIR_CHECK(p->seg_ld[i] >= p->seg_c[i] && (p->seg_ld[i] * esz) % 16 == 0 && ((uintptr_t)x[i]) % 16 == 0,
                    "flair_conv_chain: segment %d stride/alignment", i);
        const unsigned long long bytes = (unsigned long long)p->H * p->W * p->seg_ld[i] * esz;
        FLAIR_CHECK(bytes < 0x40000000ull, "flair_conv_chain: one frame of segment %d spans %llu bytes (limit 1 GiB)", i, bytes);
        a.x[i] = x[i]; a.segC[i] = p->seg_c[i]; a.segLd[i] = p->seg_ld[i]; a.segBytes[i] = (unsigned)bytes;
        cin += p->seg_c[i];
    }
    a.nseg = p->nseg;
    const bool hasA = wA != nullptr;
    FLAIR_CHECK(hasA || (p->nseg == 1 && cin == p->C), "flair_conv_chain: without stage A the single input must have C channels");
    a.wA = wA; a.biasA = biasA; a.actA = p->actA; a.CinA = cin;
    a.wABytes = (unsigned)((unsigned long long)p->C * 9 * cin * esz);
    a.wB = wB; a.biasB = biasB; a.actB = p->actB; a.CoutB = p->CoutB;
    {
        const unsigned long long wb = (unsigned long long)p->CoutB * 9 * p->C * esz;
        FLAIR_CHECK(wb < 0x80000000ull, "flair_conv_chain: stage-B weights span %llu bytes", wb);
        a.wBBytes = (unsigned)wb;
    }
    FLAIR_CHECK(((uintptr_t)wB) % 16 == 0 && (!wA || ((uintptr_t)wA) % 16 == 0), "flair_conv_chain: weight alignment");
    FLAIR_CHECK(p->y_ld >= p->CoutB && (p->y_ld * esz) % 16 == 0 && ((uintptr_t)y) % 16 == 0,
                "flair_conv_chain: output stride/alignment");
    FLAIR_CHECK(!res0 || ((p->res_ld[0] * esz) % 16 == 0 && ((uintptr_t)res0) % 16 == 0), "flair_conv_chain: res0 alignment");
    FLAIR_CHECK(!res1 || ((p->res_ld[1] * esz) % 16 == 0 && ((uintptr_t)res1) % 16 == 0), "flair_conv_chain: res1 alignment");
    a.res0 = res0; a.res1 = res1; a.res0Ld = p->res_ld[0]; a.res1Ld = p->res_ld[1];
    a.outScale = p->out_scale;
    a.actParam = p->act_param; a.actPeriod = p->act_period;
    FLAIR_CHECK(p->actA != FLAIR_ACT_DCN_OFFSETS && (p->actB != FLAIR_ACT_DCN_OFFSETS ||
                                                       (p->act_period > 0 && p->act_period % 24 == 0)),
                "flair_conv_chain: FLAIR_ACT_DCN_OFFSETS is a stage-B activation with act_period = 3 * deform_groups");
    {   // act_vec implements NONE / RELU / LeakyReLU(0.1) / SiLU only: refuse the codes it would silently compute as LeakyReLU(0.1)
        auto plain = [](int c) { return c == FLAIR_ACT_NONE || c == FLAIR_ACT_RELU || c == FLAIR_ACT_LRELU01 || c == FLAIR_ACT_SILU; };
        FLAIR_CHECK(plain(p->actA) && (plain(p->actB) || p->actB == FLAIR_ACT_DCN_OFFSETS),
                    "flair_conv_chain: activation codes (%d, %d) unsupported: NONE / RELU / LRELU01 / SILU (+ DCN_OFFSETS for stage B)",
                    p->actA, p->actB);
    }
    a.y = y; a.yLd = p->y_ld;
    a.T = p->T; a.H = p->H; a.W = p->W;
    a.dbg = nullptr;
#ifdef FLAIR_CHAIN_STAMPS
    if (const char* e = getenv("FLAIR_CHAIN_DBG_PTR")) a.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 16));
#endif
    // round 4: the input-resident LDS-DMA form for the wide-output convolution without stage A (the c -> 27*G offset convolution)
    {
        static const bool useRes = !(getenv("FLAIR_CONV_RESIDENT") && atoi(getenv("FLAIR_CONV_RESIDENT")) == 0);
        if (useRes && !hasA && p->dtype == FLAIR_BF16 && p->C == 64 && p->W % 32 == 0 && p->H % 8 == 0 && !res0 && !res1 &&
            (p->y_ld * 2) % 16 == 0 && p->CoutB % 8 == 0 &&
            (unsigned long long)p->H * p->W * p->y_ld * 2 < 0x40000000ull)
            return launch_resident(a, stream);
    }
    // round 4: the c = 64 pair on LDS-DMA staging (conv_pair_kernel; FLAIR_CONV_PAIR=0: the register-staged form above).
    // Same box: fused-chain family 11.57 / 11.56 -> 10.59 / 10.55 ms per step (310 launches, 18.7 -> 15.5 us), step 72.84 / 72.76 -> 71.58 / 71.60 ms.
    {
        static const bool usePair = !(getenv("FLAIR_CONV_PAIR") && atoi(getenv("FLAIR_CONV_PAIR")) == 0);
        auto lin = [](int c) { return c == FLAIR_ACT_NONE || c == FLAIR_ACT_RELU || c == FLAIR_ACT_LRELU01; };
        if (usePair && hasA && p->dtype == FLAIR_BF16 && p->C == 64 && p->nseg == 1 && cin == 64 && p->CoutB == 64 && p->W % 32 == 0 &&
            p->H % 8 == 0 && lin(p->actA) && lin(p->actB) && (unsigned long long)p->H * p->W * p->y_ld * 2 < 0x40000000ull &&
            (!res0 || (unsigned long long)p->H * p->W * p->res_ld[0] * 2 < 0x40000000ull) &&
            (!res1 || (unsigned long long)p->H * p->W * p->res_ld[1] * 2 < 0x40000000ull))
            return launch_pair(a, stream);
    }
    // partial tiles in H are masked (hh < H); tile widths (32 or 8) divide W by the check above
    if (p->dtype == FLAIR_BF16)
        return hasA ? dispatch_chain<bf16_t, true>(a, p->C, stream) : dispatch_chain<bf16_t, false>(a, p->C, stream);
    return hasA ? dispatch_chain<float, true>(a, p->C, stream) : dispatch_chain<float, false>(a, p->C, stream);
}
