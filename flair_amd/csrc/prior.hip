// Kernels of the CodeFormer auxiliary prior (reference guided_diffusion/codeformer.py)
// that the UNet path does not already have: LayerNorm over channels, attention with one wide head (AttnBlock,
// C = 512 at 16x16), arg-max + codebook lookup, adaptive instance normalisation, and the SFT fusion.  All of them
// are small next to the prior's convolutions (>= 97 % of its ~620 GFLOP per 512x512 face, run by conv.hip); they
// are written for correctness and coalesced access, f32 arithmetic on f32 / bf16 storage.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// ---- LayerNorm over the C channels of each pixel (nn.LayerNorm(512), codeformer.py:541-542,639): one wave per
// pixel, the row held in registers; optional second output y2 = y + pos[pixel % posRows] (the q = k input of
// TransformerSALayer, codeformer.py:561-562)
template <typename E, int NV>   // NV 16-byte pieces per lane: C <= 64 * NV * VEC
__global__ __launch_bounds__(256) void layernorm_kernel(const E* x, int xLd, long rows, int C, const float* gamma,
                                                        const float* beta, float eps, E* y, int yLd, const float* pos,
                                                        int posRows, E* y2, int y2Ld) {
    constexpr int VEC = ET<E>::VEC;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[NV][VEC];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * VEC;
        if (c < C) {
            Vec16<E>::load(x + row * xLd + c, v[j]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum += v[j][e];
        }
    }
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * VEC;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[j][e] -= mean;
                sq = fmaf(v[j][e], v[j][e], sq);
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    const float* pr = pos ? pos + (row % posRows) * (long)C : nullptr;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (j * 64 + lane) * VEC;
        if (c < C) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = fmaf(v[j][e] * rstd, gamma[c + e], beta[c + e]);
            Vec16<E>::store(y + row * yLd + c, o);
            if (y2) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) o[e] += pr[c + e];
                Vec16<E>::store(y2 + row * y2Ld + c, o);
            }
        }
    }
}

// ---- attention with wide heads (AttnBlock.forward, codeformer.py:217-241: one head of width C = 512 over the 256
// pixels of the 16x16 level).  A workgroup owns QT = 16 queries of one (frame, head): scores for every key in LDS
// (thread per key, the query tile broadcast from LDS), row softmax by one wave per 4 queries, then P.V with a
// thread per output channel (coalesced V rows).  ~0.27 GFLOP per 16x16 frame: not worth MFMA tiles.
struct WideAttn {
    const void* qkv; void* out;
    int ld, outLd, L, heads, d;
    int qOff, kOff, vOff, headStride;
    float scale;
};

template <typename E>
__global__ __launch_bounds__(256) void attn_wide_kernel(WideAttn a) {
    constexpr int QT = 16, VEC = ET<E>::VEC;
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    float* qs = wsm;                       // [QT][d]
    float* sc = wsm + QT * a.d;            // [QT][L]
    const int fh = blockIdx.y, f = fh / a.heads, h = fh % a.heads;
    const int q0 = blockIdx.x * QT;
    const E* base = reinterpret_cast<const E*>(a.qkv) + (long)f * a.L * a.ld + h * a.headStride;
    const int dv = a.d / VEC;
    for (int i = threadIdx.x; i < QT * dv; i += 256) {
        const int q = i / dv, c = (i % dv) * VEC;
        float v[VEC];
        if (q0 + q < a.L)
            Vec16<E>::load(base + (long)(q0 + q) * a.ld + a.qOff + c, v);
        else
            for (int e = 0; e < VEC; ++e) v[e] = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) qs[q * a.d + c + e] = v[e] * a.scale;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.L; k += 256) {
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0.f;
        const E* kr = base + (long)k * a.ld + a.kOff;
        for (int c = 0; c < a.d; c += VEC) {
            float kv[VEC];
            Vec16<E>::load(kr + c, kv);
#pragma unroll
            for (int q = 0; q < QT; ++q) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[q] = fmaf(qs[q * a.d + c + e], kv[e], acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < QT; ++q) sc[q * a.L + k] = acc[q];
    }
    __syncthreads();
    {   // softmax of each query row: wave w owns rows 4w .. 4w+3
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int q = 4 * w; q < 4 * w + 4; ++q) {
            float m = -INFINITY;
            for (int k = lane; k < a.L; k += 64) m = fmaxf(m, sc[q * a.L + k]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            float s = 0.f;
            for (int k = lane; k < a.L; k += 64) {
                const float e = __expf(sc[q * a.L + k] - m);
                sc[q * a.L + k] = e;
                s += e;
            }
            const float inv = 1.f / wave_sum(s);
            for (int k = lane; k < a.L; k += 64) sc[q * a.L + k] *= inv;
        }
    }
    __syncthreads();
    E* ob = reinterpret_cast<E*>(a.out) + (long)f * a.L * a.outLd + h * a.d;
    for (int c = threadIdx.x; c < a.d; c += 256) {
        float acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0.f;
        const E* vc = base + a.vOff + c;
        for (int k = 0; k < a.L; ++k) {
            const float v = ET<E>::ld(vc + (long)k * a.ld);
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] = fmaf(sc[q * a.L + k], v, acc[q]);
        }
#pragma unroll
        for (int q = 0; q < QT; ++q)
            if (q0 + q < a.L) ET<E>::st(ob + (long)(q0 + q) * a.outLd + c, acc[q]);
    }
}

// ---- code selection: arg-max of each token's logits (softmax + topk(1) of codeformer.py:727-728 picks the same
// entry; first index on ties) and lookup of that codebook row (get_codebook_feat, :82-94): one wave per token
template <typename E>
__global__ __launch_bounds__(256) void argmax_codebook_kernel(const E* logits, int ld, long rows, int N, const float* codebook,
                                                              int D, const int* forced, int* idx, E* y, int yLd) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    int best = 0;
    if (forced) {
        best = forced[row];
        best = best < 0 ? 0 : (best >= N ? N - 1 : best);   // an out-of-range code index must not read outside the codebook
    } else {
        float bv = -INFINITY;
        best = N;
        for (int n = lane; n < N; n += 64) {
            const float v = ET<E>::ld(logits + row * ld + n);
            if (v > bv) {
                bv = v;
                best = n;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(best, off);
            if (ov > bv || (ov == bv && oi < best)) {
                bv = ov;
                best = oi;
            }
        }
        if (best >= N) best = 0;       // a row of NaNs: defined behaviour instead of an out-of-range read
    }
    if (lane == 0 && idx) idx[row] = best;
    for (int c = lane; c < D; c += 64) ET<E>::st(y + row * yLd + c, codebook[(long)best * D + c]);
}

// ---- adaptive_instance_normalization (codeformer.py:437-470): per (frame, channel) statistics of content and
// style over the HW pixels (unbiased variance + eps), y = (content - mc) / sc * ss + ms.  Block = 64 channels x 4
// pixel lanes; two passes (mean, then centred squares): the 16x16x256 tensors are L2 resident.
template <typename E>
__global__ __launch_bounds__(256) void adain_kernel(const E* content, int cLd, const E* style, int sLd, int HW, int C,
                                                    float eps, E* y, int yLd) {
    __shared__ float red[4][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6, f = blockIdx.y;
    const bool ok = c < C;
    const E* cp = content + (long)f * HW * cLd + c;
    const E* sp = style + (long)f * HW * sLd + c;
    float s0 = 0.f, s1 = 0.f;
    if (ok)
        for (int p = pl; p < HW; p += 4) {
            s0 += ET<E>::ld(cp + (long)p * cLd);
            s1 += ET<E>::ld(sp + (long)p * sLd);
        }
    red[0][pl][threadIdx.x & 63] = s0;
    red[1][pl][threadIdx.x & 63] = s1;
    __syncthreads();
    const int cl = threadIdx.x & 63;
    const float mc = (red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl]) / HW;
    const float ms = (red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl]) / HW;
    float q0 = 0.f, q1 = 0.f;
    if (ok)
        for (int p = pl; p < HW; p += 4) {
            const float dc = ET<E>::ld(cp + (long)p * cLd) - mc, ds = ET<E>::ld(sp + (long)p * sLd) - ms;
            q0 = fmaf(dc, dc, q0);
            q1 = fmaf(ds, ds, q1);
        }
    red[2][pl][cl] = q0;
    red[3][pl][cl] = q1;
    __syncthreads();
    const float vc = (red[2][0][cl] + red[2][1][cl] + red[2][2][cl] + red[2][3][cl]) / (HW - 1) + eps;
    const float vs = (red[3][0][cl] + red[3][1][cl] + red[3][2][cl] + red[3][3][cl]) / (HW - 1) + eps;
    const float sdc = sqrtf(vc), sds = sqrtf(vs);
    if (ok)
        for (int p = pl; p < HW; p += 4) {
            const float v = (ET<E>::ld(cp + (long)p * cLd) - mc) / sdc * sds + ms;
            ET<E>::st(y + ((long)f * HW + p) * yLd + c, v);
        }
}

// ---- Fuse_sft_block tail (codeformer.py:595-596): y = dec + w * (dec * scale + shift), dense NHWC tensors
template <typename E>
__global__ __launch_bounds__(256) void sft_fuse_kernel(const E* dec, const E* scale, const E* shift, float w, E* y, long nvec) {
    constexpr int VEC = ET<E>::VEC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float d[VEC], sc[VEC], sh[VEC];
        Vec16<E>::load(dec + i * VEC, d);
        Vec16<E>::load(scale + i * VEC, sc);
        Vec16<E>::load(shift + i * VEC, sh);
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] = d[e] + w * fmaf(d[e], sc[e], sh[e]);
        Vec16<E>::store(y + i * VEC, d);
    }
}

}  // namespace

extern "C" int flair_layernorm_nhwc(const void* x, int dtype, int x_ld, long rows, int C, const float* gamma,
                                    const float* beta, float eps, void* y, int y_ld, const float* pos, int pos_rows,
                                    void* y2, int y2_ld, hipStream_t stream) {
    FLAIR_CHECK(x && gamma && beta && y, "flair_layernorm_nhwc: null argument");
    FLAIR_CHECK(dtype == FLAIR_F32 || dtype == FLAIR_BF16, "flair_layernorm_nhwc: bad dtype %d", dtype);
    const int vec = dtype == FLAIR_BF16 ? 8 : 4;
    FLAIR_CHECK(rows > 0 && C > 0 && C % vec == 0 && C <= 64 * 4 * vec, "flair_layernorm_nhwc: C=%d unsupported", C);
    FLAIR_CHECK(x_ld % vec == 0 && y_ld % vec == 0 && (!y2 || (y2_ld % vec == 0 && pos && pos_rows > 0)),
                "flair_layernorm_nhwc: strides / positional table");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL((layernorm_kernel<bf16_t, 4>), grid, dim3(256), 0, stream, (const bf16_t*)x, x_ld, rows, C, gamma,
                           beta, eps, (bf16_t*)y, y_ld, pos, pos_rows, (bf16_t*)y2, y2_ld);
    else
        hipLaunchKernelGGL((layernorm_kernel<float, 4>), grid, dim3(256), 0, stream, (const float*)x, x_ld, rows, C, gamma,
                           beta, eps, (float*)y, y_ld, pos, pos_rows, (float*)y2, y2_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_attention_wide(const flair_attn_params* p, const void* qkv, void* out, hipStream_t stream) {
    FLAIR_CHECK(p && qkv && out, "flair_attention_wide: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_attention_wide: bad dtype %d", p->dtype);
    const int vec = p->dtype == FLAIR_BF16 ? 8 : 4;
    FLAIR_CHECK(p->frames > 0 && p->L > 0 && p->heads > 0 && p->head_dim > 0 && p->head_dim % vec == 0,
                "flair_attention_wide: shape");
    FLAIR_CHECK(p->ld % vec == 0 && p->q_off % vec == 0 && p->k_off % vec == 0 && p->v_off % vec == 0 &&
                    p->head_stride % vec == 0,
                "flair_attention_wide: offsets/strides must be multiples of %d elements", vec);
    const size_t lds = (size_t)16 * (p->head_dim + p->L) * sizeof(float);
    FLAIR_CHECK(lds <= 128 * 1024, "flair_attention_wide: head_dim %d + L %d tokens exceed the LDS tile", p->head_dim, p->L);
    WideAttn a;
    a.qkv = qkv; a.out = out; a.ld = p->ld; a.outLd = p->out_ld; a.L = p->L; a.heads = p->heads; a.d = p->head_dim;
    a.qOff = p->q_off; a.kOff = p->k_off; a.vOff = p->v_off; a.headStride = p->head_stride; a.scale = p->scale;
    static LdsAttrOnce attr0, attr1;
    {
        const hipError_t e0 = flair_max_lds_once(attr0, reinterpret_cast<const void*>(&attn_wide_kernel<bf16_t>), 128 * 1024);
        const hipError_t e1 = flair_max_lds_once(attr1, reinterpret_cast<const void*>(&attn_wide_kernel<float>), 128 * 1024);
        FLAIR_CHECK(e0 == hipSuccess && e1 == hipSuccess, "flair_attention_wide: hipFuncSetAttribute failed");
    }
    const dim3 grid((p->L + 15) / 16, p->frames * p->heads);
    if (p->dtype == FLAIR_BF16)
        hipLaunchKernelGGL(attn_wide_kernel<bf16_t>, grid, dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL(attn_wide_kernel<float>, grid, dim3(256), lds, stream, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_argmax_codebook(const void* logits, int dtype, int ld, long rows, int N, const float* codebook, int D,
                                     const int* forced_idx, int* idx, void* y, int y_ld, hipStream_t stream) {
    FLAIR_CHECK(logits && codebook && y, "flair_argmax_codebook: null argument");
    FLAIR_CHECK(dtype == FLAIR_F32 || dtype == FLAIR_BF16, "flair_argmax_codebook: bad dtype %d", dtype);
    FLAIR_CHECK(rows > 0 && N > 0 && D > 0 && ld >= N && y_ld >= D, "flair_argmax_codebook: shape");
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(argmax_codebook_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)logits, ld, rows, N,
                           codebook, D, forced_idx, idx, (bf16_t*)y, y_ld);
    else
        hipLaunchKernelGGL(argmax_codebook_kernel<float>, grid, dim3(256), 0, stream, (const float*)logits, ld, rows, N,
                           codebook, D, forced_idx, idx, (float*)y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_adain_nhwc(const void* content, int c_ld, const void* style, int s_ld, int dtype, int frames, int HW,
                                int C, float eps, void* y, int y_ld, hipStream_t stream) {
    FLAIR_CHECK(content && style && y, "flair_adain_nhwc: null argument");
    FLAIR_CHECK(dtype == FLAIR_F32 || dtype == FLAIR_BF16, "flair_adain_nhwc: bad dtype %d", dtype);
    FLAIR_CHECK(frames > 0 && HW > 1 && C > 0 && c_ld >= C && s_ld >= C && y_ld >= C, "flair_adain_nhwc: shape");
    const dim3 grid((C + 63) / 64, frames);
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(adain_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)content, c_ld, (const bf16_t*)style,
                           s_ld, HW, C, eps, (bf16_t*)y, y_ld);
    else
        hipLaunchKernelGGL(adain_kernel<float>, grid, dim3(256), 0, stream, (const float*)content, c_ld, (const float*)style,
                           s_ld, HW, C, eps, (float*)y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_sft_fuse(const void* dec, const void* scale, const void* shift, float w, int dtype, long n, void* y,
                              hipStream_t stream) {
    FLAIR_CHECK(dec && scale && shift && y, "flair_sft_fuse: null argument");
    FLAIR_CHECK(dtype == FLAIR_F32 || dtype == FLAIR_BF16, "flair_sft_fuse: bad dtype %d", dtype);
    const int vec = dtype == FLAIR_BF16 ? 8 : 4;
    FLAIR_CHECK(n > 0 && n % vec == 0, "flair_sft_fuse: n=%ld must be a multiple of %d", n, vec);
    const long nvec = n / vec;
    long blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(sft_fuse_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)dec,
                           (const bf16_t*)scale, (const bf16_t*)shift, w, (bf16_t*)y, nvec);
    else
        hipLaunchKernelGGL(sft_fuse_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)dec,
                           (const float*)scale, (const float*)shift, w, (float*)y, nvec);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
