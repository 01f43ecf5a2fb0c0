// Attention kernels of the FLAIR UNet.
//
// (1) Spatial QKV attention per (frame, head) over the H*W tokens of a frame, head
//     width 64 -- replaces QKVAttentionLegacy / QKVAttention (guided_diffusion/
//     unet_new.py:540-605: einsum QK^T, f32 softmax, einsum AV).
//       * bf16: flash-style MFMA kernel.  S^T = K.Q^T is computed with keys on the
//         accumulator rows and queries on the lanes, so the softmax row statistics are
//         lane-local (+1 exchange with lane^32) and the probability tile is reused in
//         place as the B operand of O^T += V^T.P^T (cdna_hip_programming.md section 3,
//         "An accumulator tile as the next MFMA's operand").
//       * f32: one wavefront per query (head width 64 = one channel per lane), exact
//         f32 arithmetic; this is the tight-tolerance parity path, not a speed path.
// (2) Temporal window attention per pixel -- replaces TemporalAttention's
//     unfold + flash_attn_func (unet_new.py:473-515, nn.py:370-386): one query (own
//     frame) against the F-1 neighbouring frames (replicate padding at clip ends).
//     VALU / bandwidth bound: 8 lanes per (frame, pixel, head), 8 channels per lane.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct AttnArgs {
    const void* qkv;  // [frames][L][ld]
    void* out;        // [frames][L][outLd], channel = head*64 + d
    int ld, outLd;
    int L, heads;
    int qOff, kOff, vOff, headStride;  // channel offsets: x_off + head*headStride
    float scale;                        // applied to q.k
};

// ------------------------------------------------------------------ f32 / generic path
template <typename E>
__global__ void attn_rowwise_kernel(AttnArgs a) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int fh = blockIdx.y;
    const int f = fh / a.heads, hd = fh % a.heads;
    if (q >= a.L) return;
    const E* base = reinterpret_cast<const E*>(a.qkv) + (long)f * a.L * a.ld + hd * a.headStride;
    const float qv = ET<E>::ld(base + (long)q * a.ld + a.qOff + lane);
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int s = 0; s < a.L; ++s) {
        float d = qv * ET<E>::ld(base + (long)s * a.ld + a.kOff + lane);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
        d *= a.scale;
        const float mn = fmaxf(m, d);
        const float alpha = __expf(m - mn);
        const float p = __expf(d - mn);
        l = l * alpha + p;
        acc = acc * alpha + p * ET<E>::ld(base + (long)s * a.ld + a.vOff + lane);
        m = mn;
    }
    E* o = reinterpret_cast<E*>(a.out) + ((long)f * a.L + q) * a.outLd + hd * 64 + lane;
    ET<E>::st(o, acc / l);
}

// ------------------------------------------------------------------------- bf16 MFMA
// 4 waves x 32 queries per workgroup; KV tiles of 32 tokens shared through LDS.
__device__ __forceinline__ int k_off(int row, int chunk) {  // K tile: 128-byte rows, 8 chunks
    return row * 128 + ((chunk ^ (row & 7)) << 4);
}

__global__ __launch_bounds__(256) void attn_mfma_bf16_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sK[32 * 128];       // [kv][d]
    __shared__ __attribute__((aligned(16))) bf16_t sVt[64 * 40];     // [d][kv], row padded to 40
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int fh = blockIdx.y;
    const int f = fh / a.heads, hd = fh % a.heads;
    const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (long)f * a.L * a.ld + hd * a.headStride;
    const int q = blockIdx.x * 128 + wave * 32 + lr;
    const bool qok = q < a.L;

    // Q^T fragments (B operand): element j of k-step s = Q[q][16s + 8h + j]
    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = qok ? *reinterpret_cast<const uint4*>(base + (long)q * a.ld + a.qOff + 16 * s + 8 * lh)
                    : make_uint4(0, 0, 0, 0);

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -1e30f, l = 0.f;

    const int ntile = (a.L + 31) / 32;
    for (int t = 0; t < ntile; ++t) {
        // ---- stage K (row-major, swizzled) and V (transposed) for tokens 32t..32t+31
        {
            const int kv = tid >> 3, ch = tid & 7;
            const int tok = t * 32 + kv;
            uint4 kq = make_uint4(0, 0, 0, 0), vq = make_uint4(0, 0, 0, 0);
            if (tok < a.L) {
                kq = *reinterpret_cast<const uint4*>(base + (long)tok * a.ld + a.kOff + ch * 8);
                vq = *reinterpret_cast<const uint4*>(base + (long)tok * a.ld + a.vOff + ch * 8);
            }
            *reinterpret_cast<uint4*>(sK + k_off(kv, ch)) = kq;
            const bf16_t* ve = reinterpret_cast<const bf16_t*>(&vq);
#pragma unroll
            for (int e = 0; e < 8; ++e) sVt[(ch * 8 + e) * 40 + kv] = ve[e];
        }
        __syncthreads();

        // ---- S^T[kv][q] = K . Q^T
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 kf = *reinterpret_cast<const uint4*>(sK + k_off(lr, 2 * ks + lh));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                        __builtin_bit_cast(bf16x8, qf[ks]), s, 0, 0, 0);
        }
        // rows of this lane: kv = (r&3) + 8*(r>>2) + 4*lh
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            s[r] = kv < a.L ? s[r] * a.scale : -1e30f;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - mn);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l = l * alpha + psum;
        m = mn;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

        // ---- O^T[d][q] += V^T[d][kv] . P^T[kv][q]; P^T registers 8s..8s+7 are k-step s
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 pf;
            pf.x = pack2bf(s[8 * ks + 0], s[8 * ks + 1]);
            pf.y = pack2bf(s[8 * ks + 2], s[8 * ks + 3]);
            pf.z = pack2bf(s[8 * ks + 4], s[8 * ks + 5]);
            pf.w = pack2bf(s[8 * ks + 6], s[8 * ks + 7]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16_t* vr = sVt + (32 * i + lr) * 40 + 16 * ks + 4 * lh;
                const uint2 v0 = *reinterpret_cast<const uint2*>(vr);
                const uint2 v1 = *reinterpret_cast<const uint2*>(vr + 8);
                const uint4 vf = make_uint4(v0.x, v0.y, v1.x, v1.y);
                o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                               __builtin_bit_cast(bf16x8, pf), o[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!qok) return;
    const float inv = 1.f / l;
    bf16_t* op = reinterpret_cast<bf16_t*>(a.out) + ((long)f * a.L + q) * a.outLd + hd * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack2bf(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
            pk.y = pack2bf(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 32 * i + 8 * g + 4 * lh) = pk;
        }
}

// -------------------------------------------------------------- bf16 MFMA, pipelined
// Same mathematics and operand roles as attn_mfma_bf16_kernel above, restructured around the latency that
// kernel exposes once per 32-token tile (load K/V -> LDS -> barrier -> compute -> barrier):
//   * KV tiles of 64 tokens, two LDS stages, ONE barrier per tile; the next tile's K and V are requested into
//     registers before the current tile's MFMAs and written to the other stage afterwards;
//   * V stays row-major in LDS ([token][64 d], written with 16-byte stores) and its transposed MFMA fragments
//     come from ds_read_b64_tr_b16 (the hardware transpose read) instead of eight 2-byte scatter stores per
//     thread and tile; row pitch 192 B = 48 dwords, so the four rows of a transposed block hit four disjoint
//     16-bank quarters (conflict-free per 32-lane half);
//   * NW = 2 wavefronts (64 queries) per workgroup when 128-query workgroups would leave CUs idle (L = 256:
//     256 instead of 128 workgroups), else 4.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// (lo, hi) -> packed bf16 pair; inline asm so that the vectoriser cannot pair the conversions by register
// neighbourhood and re-interleave afterwards (measured in the ISA: 32 cvt + 32 fix-ups instead of 16 cvt)
__device__ __forceinline__ unsigned cvt_pk_bf16_asm(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 3) void attn_mfma_bf16_v2_kernel(AttnArgs a) {
    constexpr int NT = 64 * NW, KV = 64;
    constexpr int KPITCH = 128, VPITCH = 192;
    constexpr int STAGE = KV * KPITCH + KV * VPITCH;                  // 20 KB per stage
    constexpr int PIECES = KV * 8;                                   // 16-byte pieces of one K (or V) tile
    constexpr int LI = (PIECES + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int fh = blockIdx.y;
    const int f = fh / a.heads, hd = fh % a.heads;
    const bf16_t* base = reinterpret_cast<const bf16_t*>(a.qkv) + (long)f * a.L * a.ld + hd * a.headStride;
    const int q = blockIdx.x * (32 * NW) + wave * 32 + lr;
    const bool qok = q < a.L;

    uint4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = qok ? *reinterpret_cast<const uint4*>(base + (long)q * a.ld + a.qOff + 16 * s + 8 * lh)
                    : make_uint4(0, 0, 0, 0);

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    const float sc = a.scale * 1.4426950408889634f;                  // scores in log2 units: exp2 instead of exp

    uint4 kreg[LI], vreg[LI];
    auto issue = [&](int t) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int id = i * NT + tid;
            const int kv = id >> 3, ch = id & 7;
            const int tok = t * KV + kv;
            const bool ok = id < PIECES && tok < a.L;
            kreg[i] = ok ? *reinterpret_cast<const uint4*>(base + (long)tok * a.ld + a.kOff + ch * 8) : make_uint4(0, 0, 0, 0);
            vreg[i] = ok ? *reinterpret_cast<const uint4*>(base + (long)tok * a.ld + a.vOff + ch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage = [&](int buf) {
        char* sk = smem + buf * STAGE;
        char* sv = sk + KV * KPITCH;
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int id = i * NT + tid;
            if (id < PIECES) {
                const int kv = id >> 3, ch = id & 7;
                *reinterpret_cast<uint4*>(sk + k_off(kv, ch)) = kreg[i];
                *reinterpret_cast<uint4*>(sv + kv * VPITCH + ch * 16) = vreg[i];
            }
        }
    };

    const int ntile = (a.L + KV - 1) / KV;
    issue(0);
    stage(0);
    __syncthreads();
    // this lane's address inside a transposed 4-row x 16-column block of V (lane 4q+p of a 16-lane group supplies
    // row q, columns 4p..4p+3); group g = lane>>4: columns 16*(g&1).., k-half g>>1 (= lh)
    const int gl = lane & 15;
    const int trOff = ((gl >> 2) + 4 * lh) * VPITCH + (16 * ((lane >> 4) & 1) + 4 * (gl & 3)) * 2;
    // one KV tile out of LDS stage BUF (a compile-time constant: the two stages are two unrolled copies of the body,
    // so every fragment address is loop-invariant instead of 30 address adds per tile)
    auto tile = [&](int t, auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const char* sk = smem + BUF * STAGE;
        const char* sv = sk + KV * KPITCH;
        if (t + 1 < ntile) issue(t + 1);

        // ---- S^T[kv][q] = K . Q^T for the two 32-token halves of the tile
        f32x16 s[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[h2][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const uint4 kf = *reinterpret_cast<const uint4*>(sk + k_off(32 * h2 + lr, 2 * ks + lh));
                s[h2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                            __builtin_bit_cast(bf16x8, qf[ks]), s[h2], 0, 0, 0);
            }
        }
        // rows of this lane: kv = 32*h2 + (r&3) + 8*(r>>2) + 4*lh.  The softmax is the VALU bottleneck of this
        // kernel (32 scores per lane and tile against 16 MFMAs), so: tokens beyond L are masked only in the last tile
        // (wave-uniform branch), the scale rides in the exponent's FMA (p = 2^(s*sc - m), the running maximum is
        // kept in scaled units), v_exp_f32 directly, and the accumulators are rescaled only when some row's maximum grew
        if ((t + 1) * KV > a.L) {
            asm volatile("; partly masked last tile" ::: "memory");   // keeps this a branch (if-converted it is 64 VALU per tile)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = t * KV + 32 * h2 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (kv >= a.L) s[h2][r] = -1e30f;
                }
        }
        float tmax = -1e30f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[h2][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32)) * sc;
        const float mn = fmaxf(m, tmax);
        float ps[4] = {0.f, 0.f, 0.f, 0.f};                     // four short add chains instead of one of 32
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[h2][r] = __builtin_amdgcn_exp2f(fmaf(s[h2][r], sc, -mn));
                ps[r & 3] += s[h2][r];
            }
        float psum = (ps[0] + ps[1]) + (ps[2] + ps[3]);
        psum += __shfl_xor(psum, 32);
        if (__builtin_amdgcn_ballot_w64(mn > m) != 0) {          // some query's running maximum moved: rescale
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m = mn;
        }
        l += psum;

        // ---- O^T[d][q] += V^T[d][kv] . P^T[kv][q]: k-step (h2, ks2) covers tokens 32*h2 + 16*ks2 .. +15 in the
        // order (r&3) + 8*(r>>2) + 4*lh of the accumulator registers 8*ks2 .. 8*ks2+7 (any order, used on both sides)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 pf;
                pf.x = cvt_pk_bf16_asm(s[h2][8 * ks + 0], s[h2][8 * ks + 1]);
                pf.y = cvt_pk_bf16_asm(s[h2][8 * ks + 2], s[h2][8 * ks + 3]);
                pf.z = cvt_pk_bf16_asm(s[h2][8 * ks + 4], s[h2][8 * ks + 5]);
                pf.w = cvt_pk_bf16_asm(s[h2][8 * ks + 6], s[h2][8 * ks + 7]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    // elements 0..3: tokens row0 + 4*lh + {0..3}; elements 4..7: the same rows + 8
                    const char* vb = sv + (32 * h2 + 16 * ks) * VPITCH + 64 * i + trOff;
                    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vb));
                    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(vb + 8 * VPITCH));
                    uint4 vf;
                    vf.x = ((const unsigned*)&v0)[0]; vf.y = ((const unsigned*)&v0)[1];
                    vf.z = ((const unsigned*)&v1)[0]; vf.w = ((const unsigned*)&v1)[1];
                    o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                   __builtin_bit_cast(bf16x8, pf), o[i], 0, 0, 0);
                }
            }
        if (t + 1 < ntile) stage(BUF ^ 1);
        __syncthreads();
    };
    for (int t = 0; t < ntile; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < ntile) tile(t + 1, std::integral_constant<int, 1>{});
    }
    if (!qok) return;
    const float inv = 1.f / l;
    bf16_t* op = reinterpret_cast<bf16_t*>(a.out) + ((long)f * a.L + q) * a.outLd + hd * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 pk;
            pk.x = pack2bf(o[i][4 * g + 0] * inv, o[i][4 * g + 1] * inv);
            pk.y = pack2bf(o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv);
            *reinterpret_cast<uint2*>(op + 32 * i + 8 * g + 4 * lh) = pk;
        }
}

// ---------------------------------------------------------------- temporal window
template <typename E> __device__ __forceinline__ void load8(const E* p, float* v);
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    Vec16<float>::load(p, v);
    Vec16<float>::load(p + 4, v + 4);
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) { Vec16<bf16_t>::load(p, v); }
template <typename E> __device__ __forceinline__ void store8(E* p, const float* v);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
    Vec16<float>::store(p, v);
    Vec16<float>::store(p + 4, v + 4);
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) { Vec16<bf16_t>::store(p, v); }

__device__ __forceinline__ float rh(float v) { return __half2float(__float2half(v)); }

struct TAttnArgs {
    const void* qkv;  // [T][HW][ld]: q | k | v, each C wide, channel = head*64 + d
    const float* kpos;  // [window-1][C]: W_k . pe_j, added to k of window slot j
    void* out;          // [T][HW][outLd]
    int ld, outLd, T, C, window;
    long HW;
    int roundFp16;
    float scale;
};

template <typename E>
__global__ void temporal_attn_kernel(TAttnArgs a) {
    const int heads = a.C / 64;
    const long items = (long)a.T * a.HW * heads;  // one item = (t, pixel, head), 8 lanes each
    const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long item = gid >> 3;
    const int sub = (int)(gid & 7);
    if (item >= items) return;  // whole 8-lane groups leave together
    const int hd = (int)(item % heads);
    const long tp = item / heads;
    const long pix = tp % a.HW;
    const int t = (int)(tp / a.HW);
    const int c = hd * 64 + sub * 8;
    const E* base = reinterpret_cast<const E*>(a.qkv);
    float q[8];
    load8<E>(base + ((long)t * a.HW + pix) * a.ld + c, q);
    if (a.roundFp16)
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = rh(q[i]);
    const int half = a.window / 2;
    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    int slot = 0;
    for (int j = -half; j <= half; ++j) {
        if (j == 0) continue;
        int tt = t + j;
        tt = tt < 0 ? 0 : (tt >= a.T ? a.T - 1 : tt);
        const E* row = base + ((long)tt * a.HW + pix) * a.ld;
        float k[8], v[8];
        load8<E>(row + a.C + c, k);
        load8<E>(row + 2 * a.C + c, v);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float kk = k[i] + a.kpos[slot * a.C + c + i];
            if (a.roundFp16) {
                kk = rh(kk);
                v[i] = rh(v[i]);
            }
            d = fmaf(q[i], kk, d);
        }
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        d *= a.scale;
        const float mn = fmaxf(m, d);
        const float alpha = __expf(m - mn), p = __expf(d - mn);
        l = l * alpha + p;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] * alpha + p * v[i];
        m = mn;
        ++slot;
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] *= inv;
        if (a.roundFp16) acc[i] = rh(acc[i]);
    }
    store8<E>(reinterpret_cast<E*>(a.out) + ((long)t * a.HW + pix) * a.outLd + c, acc);
}

}  // namespace

extern "C" int flair_qkv_attention(const flair_attn_params* p, const void* qkv, void* out, hipStream_t stream) {
    FLAIR_CHECK(p && qkv && out, "flair_qkv_attention: null argument");
    FLAIR_CHECK(p->head_dim == 64, "flair_qkv_attention: head width %d unsupported (64 only)", p->head_dim);
    FLAIR_CHECK(p->frames > 0 && p->L > 0 && p->heads > 0, "flair_qkv_attention: empty shape");
    FLAIR_CHECK(p->ld % 8 == 0 && p->out_ld % 8 == 0 && p->q_off % 8 == 0 && p->k_off % 8 == 0 &&
                    p->v_off % 8 == 0 && p->head_stride % 8 == 0,
                "flair_qkv_attention: offsets/strides must be multiples of 8 elements");
    AttnArgs a;
    a.qkv = qkv; a.out = out; a.ld = p->ld; a.outLd = p->out_ld; a.L = p->L; a.heads = p->heads;
    a.qOff = p->q_off; a.kOff = p->k_off; a.vOff = p->v_off; a.headStride = p->head_stride;
    a.scale = p->scale;
    if (p->dtype == FLAIR_BF16) {
        static const int v2 = getenv("FLAIR_ATTN_V2") ? atoi(getenv("FLAIR_ATTN_V2")) : 1;
        const long wg128 = (long)((p->L + 127) / 128) * p->frames * p->heads;
        if (!v2)
            hipLaunchKernelGGL(attn_mfma_bf16_kernel, dim3((p->L + 127) / 128, p->frames * p->heads), dim3(256), 0,
                               stream, a);
        else if (wg128 >= 256)
            hipLaunchKernelGGL(attn_mfma_bf16_v2_kernel<4>, dim3((p->L + 127) / 128, p->frames * p->heads), dim3(256),
                               0, stream, a);
        else
            hipLaunchKernelGGL(attn_mfma_bf16_v2_kernel<2>, dim3((p->L + 63) / 64, p->frames * p->heads), dim3(128), 0,
                               stream, a);
    } else if (p->dtype == FLAIR_F32) {
        hipLaunchKernelGGL(attn_rowwise_kernel<float>, dim3((p->L + 3) / 4, p->frames * p->heads), dim3(256), 0,
                           stream, a);
    } else {
        FLAIR_CHECK(false, "flair_qkv_attention: bad dtype");
    }
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_temporal_attention(const flair_tattn_params* p, const void* qkv, const float* kpos, void* out,
                                        hipStream_t stream) {
    FLAIR_CHECK(p && qkv && kpos && out, "flair_temporal_attention: null argument");
    FLAIR_CHECK(p->C % 64 == 0 && p->window % 2 == 1 && p->window >= 3, "flair_temporal_attention: C=%d window=%d",
                p->C, p->window);
    FLAIR_CHECK(p->ld >= 3 * p->C && p->ld % 8 == 0 && p->out_ld % 8 == 0, "flair_temporal_attention: strides");
    TAttnArgs a;
    a.qkv = qkv; a.kpos = kpos; a.out = out; a.ld = p->ld; a.outLd = p->out_ld; a.T = p->T; a.C = p->C;
    a.window = p->window; a.HW = (long)p->H * p->W; a.roundFp16 = p->round_fp16; a.scale = p->scale;
    const long threads = (long)p->T * a.HW * (p->C / 64) * 8;
    const int grid = (int)((threads + 255) / 256);
    if (p->dtype == FLAIR_BF16)
        hipLaunchKernelGGL(temporal_attn_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, a);
    else if (p->dtype == FLAIR_F32)
        hipLaunchKernelGGL(temporal_attn_kernel<float>, dim3(grid), dim3(256), 0, stream, a);
    else
        FLAIR_CHECK(false, "flair_temporal_attention: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
