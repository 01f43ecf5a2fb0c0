// Streaming / small kernels around the UNet: layout conversion at the API edge,
// timestep embedding, the f32 embedding linears and the fused sampler update.
// All HBM- or launch-bound; 16-byte accesses on the NHWC side.
#include "common.h"

namespace {

// ---- NCHW f32 (API edge) -> NHWC clip tensor, written at a channel offset ----------
// One thread per (pixel, channel-of-source); reads are strided over channel planes but
// each plane is contiguous over pixels (coalesced per channel), writes are 2/4-byte
// scattered into a <= 64-byte pixel row: only used on 3..6-channel images.
template <typename E>
__global__ void nchw_to_nhwc_kernel(const float* src, int N, int C, long HW, E* dst, int ld, int coff) {
    const long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i % HW;
        for (int c = 0; c < C; ++c) ET<E>::st(dst + i * ld + coff + c, src[(n * C + c) * HW + p]);
    }
}

template <typename E>
__global__ void nhwc_to_nchw_kernel(const E* src, int ld, int coff, int N, int C, long HW, float* dst) {
    const long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW, p = i % HW;
        for (int c = 0; c < C; ++c) dst[(n * C + c) * HW + p] = ET<E>::ld(src + i * ld + coff + c);
    }
}

// ---- sinusoidal embedding: out[n] = [cos(t*f_i) | sin(t*f_i)], f_i = P^(-i/half) ---
__global__ void timestep_embedding_kernel(const float* t, int N, int dim, float maxPeriod, int sinFirst,
                                          float* out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half) return;
    const int n = i / half, k = i % half;
    // same operation order as the reference: exp(-log(P) * k / half) in f32
    const float freq = expf(-logf(maxPeriod) * (float)k / (float)half);
    const float arg = t[n] * freq;
    out[(long)n * dim + (sinFirst ? half : 0) + k] = cosf(arg);
    out[(long)n * dim + (sinFirst ? 0 : half) + k] = sinf(arg);
    if ((dim & 1) && k == 0) out[(long)n * dim + dim - 1] = 0.f;
}

// ---- y[m][n] = act_out( sum_k act_in(x[m][k]) * w[n][k] + b[n] ), f32, M <= 32 -----
// One wavefront per output feature at a time, `fpw` consecutive features per wavefront: weights are read once,
// coalesced; the M (activated) input rows are staged in LDS once per workgroup.  A GEMV-shaped,
// weight-bandwidth-bound op (M = frames of a clip); eight features per wave and eight weight loads in flight per lane:
// 281 -> 215 us for the 16 x 512 -> 47k embedding matrix of unet_new (still M LDS reads per weight: an MFMA 16x16x4
// formulation would be the next step; it is 0.2 % of the step).
template <int MMAX>
__global__ void linear_f32_kernel(const float* x, int M, int K, const float* w, const float* b, int N, int fpw,
                                  int actIn, int actOut, float* y, int yLd) {
    extern __shared__ float xs[];  // [M][K]
    for (int i = threadIdx.x; i < M * K; i += blockDim.x) xs[i] = apply_act(x[i], actIn);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * fpw;
    for (int n = n0; n < n0 + fpw && n < N; ++n) {
        float acc[MMAX];
#pragma unroll
        for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
        const float* wr = w + (long)n * K;
        for (int k0 = lane; k0 < K; k0 += 64 * 8) {          // 8 weight loads in flight per lane
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = k0 + 64 * j < K ? wr[k0 + 64 * j] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 64 * j < K ? k0 + 64 * j : lane;
#pragma unroll
                for (int m = 0; m < MMAX; ++m)
                    if (m < M) acc[m] = fmaf(xs[m * K + k], wv[j], acc[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            float v = acc[m];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && m < M) y[(long)m * yLd + n] = apply_act(v + (b ? b[n] : 0.f), actOut);
        }
    }
}

// ---- fused sampler update (gaussian_diffusion.py:325-327,465-470,507-515) ------------
// phase 1: x0 = clamp(c_recip*x - c_recipm1*eps)                       (predict_xstart)
// phase 2: x0 = clamp(x0 - gamma*restored); optional aux blend; then
//          eps' = (c_recip*x - x0)/c_recipm1;  x_prev = c_prev*x0 + nz*co*(sqrt(1-rho)*eps' + sqrt(rho)*z)
__global__ void predict_xstart_kernel(const float* x, const float* modelOut, int N, int C, int Cm, long HW,
                                      float cRecip, float cRecipm1, int clip, float* x0) {
    const long total = (long)N * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / (C * HW), r = i % (C * HW);
        const float eps = modelOut[n * Cm * HW + r];
        float v = cRecip * x[i] - cRecipm1 * eps;
        if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        x0[i] = v;
    }
}

__global__ void sampler_update_kernel(const float* x, float* x0, const float* restored, const float* aux,
                                      const float* z, const float* prev, long frameElems, int T, int Tp,
                                      long total, float gamma, float wAux, float cRecip,
                                      float cRecipm1, float cPrev, float coNoise, float sq1mRho, float sqRho,
                                      int clip, int nonzero, float* xPrev) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = x0[i];
        if (restored) {
            v = v - gamma * restored[i];
            if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        }
        if (aux) {
            float f = aux[i];
            if (clip) f = fminf(fmaxf(f, -1.f), 1.f);
            v = wAux * v + (1.f - wAux) * f;
        }
        if (prev) {  // first Tp frames of every clip are pinned to the previous window's result
            const long n = i / frameElems, r = i % frameElems;
            const long b = n / T;
            const int t = (int)(n % T);
            if (t < Tp) v = prev[(b * Tp + t) * frameElems + r];
        }
        x0[i] = v;
        const float xi = x[i];
        const float eps = (cRecip * xi - v) / cRecipm1;
        float out = cPrev * v;
        if (nonzero) out += sq1mRho * coNoise * eps + sqRho * coNoise * z[i];
        xPrev[i] = out;
    }
}

// generic: out = clamp(a*x + b*y, lo, hi)  (f32, flat)
__global__ void axpby_kernel(const float* x, const float* y, float a, float b, float lo, float hi, long n,
                             float* out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = fminf(fmaxf(a * x[i] + (y ? b * y[i] : 0.f), lo), hi);
}

// per-pixel multiply of an NHWC tensor by a [pixels] f32 map (BasicVSR++ vsrpp_weights)
template <typename E>
__global__ void scale_pixels_kernel(E* x, int ld, int C, long P, const float* wmap) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = P * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / cv;
        const int c0 = (int)(i % cv) * VEC;
        float v[VEC];
        Vec16<E>::load(x + p * ld + c0, v);
        const float s = wmap[p];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] *= s;
        Vec16<E>::store(x + p * ld + c0, v);
    }
}


// out[p][c] = (clamp(x[p][c]*a + b, lo, hi) - sub[c]) * mul[c]   (f32 clip tensors; SPyNet
// input normalisation, unet_new.py:1300 + mmedit SPyNet mean/std)
__global__ void affine_channels_kernel(const float* x, int xLd, int C, long P, float a, float b, float lo,
                                       float hi, const float* sub, const float* mul, float* y, int yLd) {
    const long total = P * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / C;
        const int c = (int)(i % C);
        float v = fminf(fmaxf(x[p * xLd + c] * a + b, lo), hi);
        y[p * yLd + c] = (v - (sub ? sub[c] : 0.f)) * (mul ? mul[c] : 1.f);
    }
}

// x[f][p][c] += bias[f][c]  (AttentionbottleBlock: h + emb_out, unet_new.py:426-428)
template <typename E>
__global__ void add_frame_bias_kernel(E* x, int ld, int C, int F, long HW, const float* bias, int bLd) {
    const long total = (long)F * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const long f = p / HW;
        E* e = x + p * ld + c;
        ET<E>::st(e, ET<E>::ld(e) + bias[f * bLd + c]);
    }
}

// y = x + sigmoid(g[f][c]) * (m - x)
template <typename E>
__global__ void gated_blend_kernel(const E* x, int xLd, const E* m, int mLd, const float* gate, int gLd, int C, int F,
                                   long HW, E* y, int yLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = (long)F * HW * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        const long f = p / HW;
        float a[VEC], b[VEC];
        Vec16<E>::load(x + p * xLd + c0, a);
        Vec16<E>::load(m + p * mLd + c0, b);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float s = 1.f / (1.f + __expf(-gate[f * gLd + c0 + k]));
            a[k] = (1.f - s) * a[k] + s * b[k];
        }
        Vec16<E>::store(y + p * yLd + c0, a);
    }
}

// dst[p][coff + c] = (E) src[p][c]  for c < C   (f32 flow fields into a conv input segment)
template <typename E>
__global__ void cast_channels_kernel(const float* src, int sLd, int C, long P, E* dst, int dLd, int coff) {
    const long total = P * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / C;
        const int c = (int)(i % C);
        ET<E>::st(dst + p * dLd + coff + c, src[p * sLd + c]);
    }
}

// y = act(x0 + x1) on [P] pixels of C channels (x1 may be null: y = act(x0)).  The residual sum of a torchvision Bottleneck
// (`out += identity; out = relu(out)`: the conv epilogues add residuals AFTER the activation) and FPN's lateral sums.
template <typename E>
__global__ void add_act_kernel(const E* x0, int ld0, const E* x1, int ld1, int C, long P, int act, E* y, int yLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = P * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        float a[VEC], b[VEC];
        Vec16<E>::load(x0 + p * ld0 + c0, a);
        if (x1) {
            Vec16<E>::load(x1 + p * ld1 + c0, b);
#pragma unroll
            for (int k = 0; k < VEC; ++k) a[k] += b[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) a[k] = apply_act(a[k], act);
        Vec16<E>::store(y + p * yLd + c0, a);
    }
}

// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on [F][H][W][C] (torchvision ResNet stem): out (H + 1) / 2 x (W + 1) / 2,
// padding positions do not take part (-inf)
template <typename E>
__global__ void maxpool3s2_kernel(const E* x, int xLd, int F, int H, int W, int C, E* y, int yLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long total = (long)F * Ho * Wo * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * VEC;
        long q = i / cv;
        const int wo = (int)(q % Wo);
        q /= Wo;
        const int ho = (int)(q % Ho), f = (int)(q / Ho);
        float m[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) m[k] = -INFINITY;
        for (int dh = -1; dh <= 1; ++dh) {
            const int h = 2 * ho + dh;
            if ((unsigned)h >= (unsigned)H) continue;
            for (int dw = -1; dw <= 1; ++dw) {
                const int w = 2 * wo + dw;
                if ((unsigned)w >= (unsigned)W) continue;
                float v[VEC];
                Vec16<E>::load(x + (((long)f * H + h) * W + w) * xLd + c0, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        }
        Vec16<E>::store(y + (((long)f * Ho + ho) * Wo + wo) * yLd + c0, m);
    }
}

inline int grid_for(long n, int block = 256, int cap = 2048) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int flair_nchw_f32_to_nhwc(const float* src, int N, int C, int H, int W, void* dst, int dtype,
                                      int dst_ld, int dst_coff, hipStream_t stream) {
    FLAIR_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && dst_coff >= 0 && dst_coff + C <= dst_ld,
                "flair_nchw_f32_to_nhwc: bad argument");
    const long HW = (long)H * W;
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(N * HW)), dim3(256), 0, stream, src, N, C, HW,
                           (bf16_t*)dst, dst_ld, dst_coff);
    else if (dtype == FLAIR_F32)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(N * HW)), dim3(256), 0, stream, src, N, C, HW,
                           (float*)dst, dst_ld, dst_coff);
    else
        FLAIR_CHECK(false, "flair_nchw_f32_to_nhwc: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_nhwc_to_nchw_f32(const void* src, int dtype, int src_ld, int src_coff, int N, int C, int H,
                                      int W, float* dst, hipStream_t stream) {
    FLAIR_CHECK(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && src_coff >= 0 && src_coff + C <= src_ld,
                "flair_nhwc_to_nchw_f32: bad argument");
    const long HW = (long)H * W;
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(N * HW)), dim3(256), 0, stream,
                           (const bf16_t*)src, src_ld, src_coff, N, C, HW, dst);
    else if (dtype == FLAIR_F32)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(N * HW)), dim3(256), 0, stream,
                           (const float*)src, src_ld, src_coff, N, C, HW, dst);
    else
        FLAIR_CHECK(false, "flair_nhwc_to_nchw_f32: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_timestep_embedding(const float* t, int N, int dim, float max_period, int sin_first,
                                        float* out, hipStream_t stream) {
    FLAIR_CHECK(t && out && N > 0 && dim >= 2, "flair_timestep_embedding: bad argument");
    const int n = N * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, t, N, dim, max_period,
                       sin_first, out);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_linear_f32(const float* x, int M, int K, const float* w, const float* bias, int N, int act_in,
                                int act_out, float* y, int y_ld, hipStream_t stream) {
    FLAIR_CHECK(x && w && y && M > 0 && M <= 32 && K > 0 && N > 0 && y_ld >= N, "flair_linear_f32: bad argument (M<=32)");
    const size_t lds = (size_t)M * K * sizeof(float);
    FLAIR_CHECK(lds <= 64 * 1024, "flair_linear_f32: M*K too large for LDS staging");
    const int wavesPerBlock = 4;
    const int fpw = N >= 8192 ? 8 : 1;                       // features per wavefront (keep >= 256 workgroups)
    const int grid = (N + wavesPerBlock * fpw - 1) / (wavesPerBlock * fpw);
    if (M <= 8)
        hipLaunchKernelGGL(linear_f32_kernel<8>, dim3(grid), dim3(256), lds, stream, x, M, K, w, bias, N, fpw,
                           act_in, act_out, y, y_ld);
    else if (M <= 16)
        hipLaunchKernelGGL(linear_f32_kernel<16>, dim3(grid), dim3(256), lds, stream, x, M, K, w, bias, N, fpw,
                           act_in, act_out, y, y_ld);
    else
        hipLaunchKernelGGL(linear_f32_kernel<32>, dim3(grid), dim3(256), lds, stream, x, M, K, w, bias, N, fpw,
                           act_in, act_out, y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_predict_xstart(const float* x, const float* model_out, int N, int C, int Cm, int H, int W,
                                    float c_recip, float c_recipm1, int clip, float* x0, hipStream_t stream) {
    FLAIR_CHECK(x && model_out && x0 && N > 0 && C > 0 && Cm >= C, "flair_predict_xstart: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(predict_xstart_kernel, dim3(grid_for((long)N * C * HW)), dim3(256), 0, stream, x, model_out, N,
                       C, Cm, HW, c_recip, c_recipm1, clip, x0);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_sampler_update(const flair_sampler_coefs* c, const float* x, float* x0, const float* restored,
                                    const float* aux, const float* z, const float* prev_recon, long n,
                                    float* x_prev, hipStream_t stream) {
    FLAIR_CHECK(c && x && x0 && x_prev && n > 0, "flair_sampler_update: null argument");
    FLAIR_CHECK(!c->nonzero || z, "flair_sampler_update: noise required when t != 0");
    FLAIR_CHECK(!prev_recon || (c->frame_elems > 0 && c->frames > 0 && c->prev_frames > 0 &&
                                c->prev_frames <= c->frames && n % (c->frame_elems * c->frames) == 0),
                "flair_sampler_update: prev_recon geometry");
    hipLaunchKernelGGL(sampler_update_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, x0, restored, aux, z,
                       prev_recon, (long)c->frame_elems, c->frames, c->prev_frames, n,
                       c->gamma, c->w_aux, c->sqrt_recip_alphas_cumprod, c->sqrt_recipm1_alphas_cumprod,
                       c->sqrt_alphas_cumprod_prev, c->sqrt_one_minus_alphas_cumprod_prev, c->sqrt_one_minus_rho,
                       c->sqrt_rho, c->clip_denoised, c->nonzero, x_prev);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_axpby_f32(const float* x, const float* y, float a, float b, float lo, float hi, long n,
                               float* out, hipStream_t stream) {
    FLAIR_CHECK(x && out && n > 0, "flair_axpby_f32: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, a, b, lo, hi, n, out);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_scale_pixels(void* x, int dtype, int ld, int C, long P, const float* wmap, hipStream_t stream) {
    FLAIR_CHECK(x && wmap && P > 0 && C > 0, "flair_scale_pixels: bad argument");
    if (dtype == FLAIR_BF16) {
        FLAIR_CHECK(C % 8 == 0, "flair_scale_pixels: C %% 8");
        hipLaunchKernelGGL(scale_pixels_kernel<bf16_t>, dim3(grid_for(P * (C / 8))), dim3(256), 0, stream,
                           (bf16_t*)x, ld, C, P, wmap);
    } else {
        FLAIR_CHECK(C % 4 == 0, "flair_scale_pixels: C %% 4");
        hipLaunchKernelGGL(scale_pixels_kernel<float>, dim3(grid_for(P * (C / 4))), dim3(256), 0, stream, (float*)x,
                           ld, C, P, wmap);
    }
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_affine_channels_f32(const float* x, int x_ld, int C, long P, float a, float b, float lo,
                                         float hi, const float* sub, const float* mul, float* y, int y_ld,
                                         hipStream_t stream) {
    FLAIR_CHECK(x && y && C > 0 && P > 0 && x_ld >= C && y_ld >= C, "flair_affine_channels_f32: bad argument");
    hipLaunchKernelGGL(affine_channels_kernel, dim3(grid_for(P * C)), dim3(256), 0, stream, x, x_ld, C, P, a, b, lo,
                       hi, sub, mul, y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_add_frame_bias(void* x, int dtype, int ld, int C, int F, long HW, const float* bias,
                                    int bias_ld, hipStream_t stream) {
    FLAIR_CHECK(x && bias && C > 0 && F > 0 && HW > 0 && ld >= C && bias_ld >= C, "flair_add_frame_bias: bad argument");
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(add_frame_bias_kernel<bf16_t>, dim3(grid_for(F * HW * C)), dim3(256), 0, stream,
                           (bf16_t*)x, ld, C, F, HW, bias, bias_ld);
    else if (dtype == FLAIR_F32)
        hipLaunchKernelGGL(add_frame_bias_kernel<float>, dim3(grid_for(F * HW * C)), dim3(256), 0, stream, (float*)x,
                           ld, C, F, HW, bias, bias_ld);
    else
        FLAIR_CHECK(false, "flair_add_frame_bias: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_cast_channels(const float* src, int src_ld, int C, long P, void* dst, int dtype, int dst_ld,
                                   int dst_coff, hipStream_t stream) {
    FLAIR_CHECK(src && dst && C > 0 && P > 0 && src_ld >= C && dst_coff >= 0 && dst_coff + C <= dst_ld,
                "flair_cast_channels: bad argument");
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(cast_channels_kernel<bf16_t>, dim3(grid_for(P * C)), dim3(256), 0, stream, src, src_ld, C,
                           P, (bf16_t*)dst, dst_ld, dst_coff);
    else if (dtype == FLAIR_F32)
        hipLaunchKernelGGL(cast_channels_kernel<float>, dim3(grid_for(P * C)), dim3(256), 0, stream, src, src_ld, C, P,
                           (float*)dst, dst_ld, dst_coff);
    else
        FLAIR_CHECK(false, "flair_cast_channels: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

namespace {
// LEARNED_RANGE variance (gaussian_diffusion.py:285-292): frac=(v+1)/2,
// logvar = frac*max_log + (1-frac)*min_log, var = exp(logvar); v = channels [C,2C) of the model output.
__global__ void learned_range_kernel(const float* modelOut, int N, int C, long HW, float minLog, float maxLog,
                                     float* var, float* logvar) {
    const long total = (long)N * C * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / (C * HW), r = i % (C * HW);
        const float v = modelOut[(n * 2 * C + C) * HW + r];
        const float frac = (v + 1.f) / 2.f;
        const float lv = frac * maxLog + (1.f - frac) * minLog;
        logvar[i] = lv;
        var[i] = expf(lv);
    }
}
}  // namespace

extern "C" int flair_learned_range_variance(const float* model_out, int N, int C, int H, int W, float min_log,
                                            float max_log, float* variance, float* log_variance,
                                            hipStream_t stream) {
    FLAIR_CHECK(model_out && variance && log_variance && N > 0 && C > 0, "flair_learned_range_variance: bad argument");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(learned_range_kernel, dim3(grid_for((long)N * C * HW)), dim3(256), 0, stream, model_out, N, C,
                       HW, min_log, max_log, variance, log_variance);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_gated_blend(const void* x, int x_ld, const void* m, int m_ld, const float* gate, int gate_ld,
                                 int dtype, int C, int F, long HW, void* y, int y_ld, hipStream_t stream) {
    FLAIR_CHECK(x && m && gate && y && C > 0 && F > 0 && HW > 0 && gate_ld >= C, "flair_gated_blend: bad argument");
    if (dtype == FLAIR_BF16) {
        FLAIR_CHECK(C % 8 == 0, "flair_gated_blend: C %% 8");
        hipLaunchKernelGGL(gated_blend_kernel<bf16_t>, dim3(grid_for(F * HW * (C / 8))), dim3(256), 0, stream,
                           (const bf16_t*)x, x_ld, (const bf16_t*)m, m_ld, gate, gate_ld, C, F, HW, (bf16_t*)y, y_ld);
    } else if (dtype == FLAIR_F32) {
        FLAIR_CHECK(C % 4 == 0, "flair_gated_blend: C %% 4");
        hipLaunchKernelGGL(gated_blend_kernel<float>, dim3(grid_for(F * HW * (C / 4))), dim3(256), 0, stream,
                           (const float*)x, x_ld, (const float*)m, m_ld, gate, gate_ld, C, F, HW, (float*)y, y_ld);
    } else {
        FLAIR_CHECK(false, "flair_gated_blend: bad dtype");
    }
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_add_act_nhwc(const void* x0, int x0_ld, const void* x1, int x1_ld, int dtype, int C, long P, int act, void* y,
                                  int y_ld, hipStream_t stream) {
    FLAIR_CHECK(x0 && y && C > 0 && P > 0 && x0_ld >= C && y_ld >= C && (!x1 || x1_ld >= C), "flair_add_act_nhwc: bad argument");
    FLAIR_CHECK(act >= FLAIR_ACT_NONE && act <= FLAIR_ACT_GELU && act != FLAIR_ACT_DCN_OFFSETS, "flair_add_act_nhwc: activation %d", act);
    const int vec = dtype == FLAIR_BF16 ? 8 : 4, esz = dtype == FLAIR_BF16 ? 2 : 4;
    FLAIR_CHECK(dtype == FLAIR_BF16 || dtype == FLAIR_F32, "flair_add_act_nhwc: bad dtype");
    FLAIR_CHECK(C % vec == 0 && (x0_ld * esz) % 16 == 0 && (y_ld * esz) % 16 == 0 && (!x1 || (x1_ld * esz) % 16 == 0) &&
                    ((uintptr_t)x0) % 16 == 0 && ((uintptr_t)y) % 16 == 0 && ((uintptr_t)x1) % 16 == 0,
                "flair_add_act_nhwc: C %% %d, 16-byte aligned rows", vec);
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(add_act_kernel<bf16_t>, dim3(grid_for(P * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x0, x0_ld,
                           (const bf16_t*)x1, x1_ld, C, P, act, (bf16_t*)y, y_ld);
    else
        hipLaunchKernelGGL(add_act_kernel<float>, dim3(grid_for(P * (C / 4))), dim3(256), 0, stream, (const float*)x0, x0_ld,
                           (const float*)x1, x1_ld, C, P, act, (float*)y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_maxpool3x3s2_nhwc(const void* x, int x_ld, int dtype, int F, int H, int W, int C, void* y, int y_ld,
                                       hipStream_t stream) {
    FLAIR_CHECK(x && y && F > 0 && H > 0 && W > 0 && C > 0 && x_ld >= C && y_ld >= C, "flair_maxpool3x3s2_nhwc: bad argument");
    FLAIR_CHECK(dtype == FLAIR_BF16 || dtype == FLAIR_F32, "flair_maxpool3x3s2_nhwc: bad dtype");
    const int vec = dtype == FLAIR_BF16 ? 8 : 4, esz = dtype == FLAIR_BF16 ? 2 : 4;
    FLAIR_CHECK(C % vec == 0 && (x_ld * esz) % 16 == 0 && (y_ld * esz) % 16 == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0,
                "flair_maxpool3x3s2_nhwc: C %% %d, 16-byte aligned rows", vec);
    const long n = (long)F * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(maxpool3s2_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, stream, (const bf16_t*)x, x_ld, F, H, W, C, (bf16_t*)y, y_ld);
    else
        hipLaunchKernelGGL(maxpool3s2_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, (const float*)x, x_ld, F, H, W, C, (float*)y, y_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
