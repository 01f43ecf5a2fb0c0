// Degradation operators of FLAIR's data-consistency step (restore_fn) on (N,C,H,W) f32
// images -- the sampler-side tensors, not the UNet's NHWC activations.
//
//  * flair_depthwise_filter : pseudoSR's Filter_Layer family (guided_diffusion/pseudoSR.py:15-44,
//    174-246): a per-channel 2-D cross-correlation with one shared filter after replication
//    padding, with the three access patterns A_pinv needs --
//       Down    : replicate-pad, filter, keep samples [pre::f, pre::f]        (:226-244)
//       InvHtH  : replicate-pad, filter                                       (:183-195)
//       Up      : zero-stuff by f at offset pre, replicate-pad, filter        (:199-225)
//    expressed as  out[i][j] = sum_{u,v} K[u][v] * IN(i*so + off + u - pad, j*so + off + v - pad)
//    where IN clamps its coordinates to the (virtual, zero-stuffed) image and returns 0
//    at stuffed positions.
//  * flair_jpeg_roundtrip   : jpeg_decode(jpeg_encode(x, qf), qf) of guided_diffusion/jpeg.py:72-167
//    fused into one kernel per 16x16 macro-block: RGB->YCbCr, 4:2:0, 8x8 ortho DCT-II,
//    quantise + round-half-even, dequantise, IDCT, chroma replication, YCbCr->RGB.
//  * flair_matmul_f32       : small dense C = A.B (SRConv's separable operators,
//    restore_util.py:102-227) -- tiled through LDS, f32.
// All are tiny next to the UNet (<0.1 % of a step); they are written for coalesced plane
// access and zero host round trips, not for peak throughput.
#include "common.h"

namespace {

__device__ __forceinline__ int pad_index(int a, int n, int reflect) {
    if (reflect) {   // F.pad(mode="reflect"): -1 -> 1, n -> n-2
        a = a < 0 ? -a : a;
        a = a > n - 1 ? 2 * (n - 1) - a : a;
        return a < 0 ? 0 : a;
    }
    return a < 0 ? 0 : (a > n - 1 ? n - 1 : a);
}

__global__ void depthwise_filter_kernel(const float* x, int planes, int Hin, int Win, const float* K, int kh, int kw,
                                        int pad, int so, int off, int stuff, int stuffOff, int Hout, int Wout,
                                        int reflect, float* y) {
    extern __shared__ float ks[];
    for (int i = threadIdx.x; i < kh * kw; i += blockDim.x) ks[i] = K[i];
    __syncthreads();
    const int Hv = Hin * stuff, Wv = Win * stuff;
    const long total = (long)planes * Hout * Wout;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Wout);
        const int i = (int)((idx / Wout) % Hout);
        const long pl = idx / ((long)Wout * Hout);
        const float* xp = x + pl * Hin * Win;
        float acc = 0.f;
        for (int u = 0; u < kh; ++u) {
            const int a = pad_index(i * so + off + u - pad, Hv, reflect);
            if (stuff > 1 && (a % stuff) != stuffOff) continue;
            const float* row = xp + (long)(a / stuff) * Win;
            for (int v = 0; v < kw; ++v) {
                const int b = pad_index(j * so + off + v - pad, Wv, reflect);
                if (stuff > 1 && (b % stuff) != stuffOff) continue;
                acc = fmaf(ks[u * kw + v], row[b / stuff], acc);
            }
        }
        y[idx] = acc;
    }
}

// Stride-1, un-stuffed form for one filter size (KS x KS = 39 x 39: pseudoSR's inverse-(H^T H) filter on the low-resolution frames,
// pseudoSR.py:183-195, once per denoising step): a workgroup owns a 32 x 32 output tile of one plane with its (32 + KS - 1)^2 input
// patch (padding already applied) and the filter in LDS; a thread finishes four horizontally adjacent outputs, so every input value
// it reads serves up to four taps.  Same tap order and the same fused multiply-adds per output as the generic kernel above
// (bit-identical results); that one spends ~20 instructions per tap on index arithmetic and a cached global load: 300 us per call.
template <int KS>
__global__ __launch_bounds__(256) void depthwise_filter_tiled_kernel(const float* x, int Hin, int Win, const float* K, int pad, int off,
                                                                     int Hout, int Wout, int reflect, float* y) {
    constexpr int PW = 32 + KS - 1, PITCH = (PW + 3 + 3) / 4 * 4, NSEG = (KS + 3 + 3) / 4;       // patch width, LDS pitch, float4 reads per row
    __shared__ __attribute__((aligned(16))) float patch[PW * PITCH];
    __shared__ float ks[KS * KS];
    const int tilesW = Wout / 32, tilesH = Hout / 32;
    const int pl = blockIdx.x / (tilesW * tilesH), t = blockIdx.x % (tilesW * tilesH);
    const int i0 = (t / tilesW) * 32, j0 = (t % tilesW) * 32;
    const float* xp = x + (long)pl * Hin * Win;
    for (int i = threadIdx.x; i < KS * KS; i += 256) ks[i] = K[i];
    for (int i = threadIdx.x; i < PW * PITCH; i += 256) {
        const int r = i / PITCH, c = i % PITCH;
        float v = 0.f;
        if (c < PW) v = xp[(long)pad_index(i0 + off + r - pad, Hin, reflect) * Win + pad_index(j0 + off + c - pad, Win, reflect)];
        patch[i] = v;
    }
    __syncthreads();
    const int row = threadIdx.x >> 3, cg = (threadIdx.x & 7) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < KS; ++u) {
        float seg[NSEG * 4];
        const float4* src = reinterpret_cast<const float4*>(patch + (row + u) * PITCH + cg);
#pragma unroll
        for (int m = 0; m < NSEG; ++m) {
            const float4 q = src[m];
            seg[4 * m] = q.x; seg[4 * m + 1] = q.y; seg[4 * m + 2] = q.z; seg[4 * m + 3] = q.w;
        }
#pragma unroll
        for (int v = 0; v < KS; ++v) {
            const float kv = ks[u * KS + v];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(kv, seg[v + r], acc[r]);
        }
    }
    *reinterpret_cast<float4*>(y + ((long)pl * Hout + i0 + row) * Wout + j0 + cg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ------------------------------------------------------------------------------ JPEG
struct JpegTables {
    float q1[64], q2[64];  // luma / chroma quantisation (already scaled for the quality factor)
    float D[64];           // 8x8 orthonormal DCT-II matrix D[k][n]
};

// forward: out = D * blk * D^T ; inverse: out = D^T * blk * D   (LinearDCT + apply_linear_2d)
__device__ __forceinline__ void dct8x8(const float* D, const float (&in)[8][8], float (&out)[8][8], bool inverse) {
    float tmp[8][8];
    // apply along the last dim: tmp[r][k] = sum_n in[r][n] * M[k][n]
    for (int r = 0; r < 8; ++r)
        for (int k = 0; k < 8; ++k) {
            float s = 0.f;
            for (int n = 0; n < 8; ++n) s = fmaf(in[r][n], inverse ? D[n * 8 + k] : D[k * 8 + n], s);
            tmp[r][k] = s;
        }
    // then along the first dim
    for (int k = 0; k < 8; ++k)
        for (int c = 0; c < 8; ++c) {
            float s = 0.f;
            for (int n = 0; n < 8; ++n) s = fmaf(tmp[n][c], inverse ? D[n * 8 + k] : D[k * 8 + n], s);
            out[k][c] = s;
        }
}

// lv (or null): where the quantised integer levels of this block go (row stride lvLd) -- what jpeg_encode returns
__device__ __forceinline__ void codec8x8(const JpegTables& t, const float* q, float (&blk)[8][8], float* lv, int lvLd) {
    float c[8][8];
    dct8x8(t.D, blk, c, false);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            const float level = rintf(c[i][j] / q[i * 8 + j]);  // round half to even
            if (lv) lv[i * lvLd + j] = level;
            c[i][j] = level * q[i * 8 + j];
        }
    dct8x8(t.D, c, blk, true);
}

// one thread per 8x8 luma block / per 8x8 chroma block (two launches' worth of work in one
// kernel: blockIdx.y selects luma (0) or chroma plane (1,2)); planes staged in a YCbCr buffer.
__global__ void jpeg_to_ycbcr_kernel(const float* x, int N, int S, float* ycc) {
    const long total = (long)N * S * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ((long)S * S), p = i % ((long)S * S);
        const float* b = x + n * 3 * S * S + p;
        const float r = (b[0] + 1.f) / 2.f * 255.f, g = (b[(long)S * S] + 1.f) / 2.f * 255.f,
                    bl = (b[2L * S * S] + 1.f) / 2.f * 255.f;
        float* o = ycc + n * 3 * S * S + p;
        o[0] = 0.299f * r + 0.587f * g + 0.114f * bl;
        o[(long)S * S] = -0.1687f * r + -0.3313f * g + 0.5f * bl + 128.f;
        o[2L * S * S] = 0.5f * r + -0.4187f * g + -0.0813f * bl + 128.f;
    }
}

__global__ void jpeg_blocks_kernel(float* ycc, int N, int S, JpegTables t, float* lumaQ, float* chromaQ) {
    // work items: luma blocks N*(S/8)^2, then chroma blocks 2*N*(S/16)^2
    const int lb = S / 8, cb = S / 16;
    const long nl = (long)N * lb * lb, nc = (long)N * 2 * cb * cb;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nl + nc; i += (long)gridDim.x * blockDim.x) {
        float blk[8][8];
        if (i < nl) {
            const long n = i / (lb * lb);
            const int by = (int)((i / lb) % lb), bx = (int)(i % lb);
            float* p = ycc + n * 3 * S * S + (long)by * 8 * S + bx * 8;
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) blk[r][c] = p[r * S + c] - 128.f;
            codec8x8(t, t.q1, blk, lumaQ ? lumaQ + n * S * S + (long)by * 8 * S + bx * 8 : nullptr, S);
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) p[r * S + c] = blk[r][c] + 128.f;
        } else {
            const long k = i - nl;
            const long n = k / (2 * cb * cb);
            const int ch = (int)((k / (cb * cb)) % 2);
            const int by = (int)((k / cb) % cb), bx = (int)(k % cb);
            // chroma is subsampled [::2, ::2]; the decoded value is replicated over each 2x2
            float* p = ycc + (n * 3 + 1 + ch) * S * S + (long)by * 16 * S + bx * 16;
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) blk[r][c] = p[2 * r * S + 2 * c] - 128.f;
            const int hs = S / 2;
            codec8x8(t, t.q2, blk, chromaQ ? chromaQ + (n * 2 + ch) * hs * hs + (long)by * 8 * hs + bx * 8 : nullptr, hs);
            for (int r = 0; r < 8; ++r)
                for (int c = 0; c < 8; ++c) {
                    const float v = blk[r][c] + 128.f;
                    float* q = p + 2 * r * S + 2 * c;
                    q[0] = v; q[1] = v; q[S] = v; q[S + 1] = v;
                }
        }
    }
}

__global__ void jpeg_to_rgb_kernel(const float* ycc, int N, int S, float* y) {
    const long total = (long)N * S * S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ((long)S * S), p = i % ((long)S * S);
        const float* b = ycc + n * 3 * S * S + p;
        const float Y = b[0], cb = b[(long)S * S] - 128.f, cr = b[2L * S * S] - 128.f;
        float* o = y + n * 3 * S * S + p;
        o[0] = (1.f * Y + -3.68199903e-05f * cb + 1.40198758f * cr) / 255.f * 2.f - 1.f;
        o[(long)S * S] = (1.f * Y + -3.44113281e-01f * cb + -7.14103821e-01f * cr) / 255.f * 2.f - 1.f;
        o[2L * S * S] = (1.f * Y + 1.77197812f * cb + -1.34583413e-04f * cr) / 255.f * 2.f - 1.f;
    }
}

// -------------------------------------------------------------------- small matmul
// C[b] (M x N) = A[b or shared] (M x K) * B[b or shared] (K x N), row-major, 16x16 LDS tiles
__global__ void matmul_kernel(const float* A, long aStride, const float* B, long bStride, float* C, int M, int N,
                              int K) {
    __shared__ float as[16][17], bs[16][17];
    const int b = blockIdx.z;
    const float* Ab = A + b * aStride;
    const float* Bb = B + b * bStride;
    const int row = blockIdx.y * 16 + threadIdx.y, col = blockIdx.x * 16 + threadIdx.x;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        as[threadIdx.y][threadIdx.x] = (row < M && k0 + threadIdx.x < K) ? Ab[(long)row * K + k0 + threadIdx.x] : 0.f;
        bs[threadIdx.y][threadIdx.x] = (col < N && k0 + threadIdx.y < K) ? Bb[(long)(k0 + threadIdx.y) * N + col] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(as[threadIdx.y][k], bs[k][threadIdx.x], acc);
        __syncthreads();
    }
    if (row < M && col < N) C[(long)b * M * N + (long)row * N + col] = acc;
}

// Resizer gather-MAC along one axis (guided_diffusion/resizer.py:54-73): for a tensor viewed as
// [outer][L][inner], out[o][i][n] = sum_k w[k][i] * x[o][fov[k][i]][n]
__global__ void gather_mac_kernel(const float* x, long outer, int Lin, long inner, const int* fov, const float* w,
                                  int taps, int Lout, float* y) {
    const long total = outer * Lout * inner;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long n = idx % inner;
        const int i = (int)((idx / inner) % Lout);
        const long o = idx / (inner * Lout);
        float acc = 0.f;
        for (int k = 0; k < taps; ++k) acc += x[(o * Lin + fov[k * Lout + i]) * inner + n] * w[k * Lout + i];
        y[idx] = acc;
    }
}

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int flair_depthwise_filter(const float* x, int planes, int Hin, int Win, const float* filt, int kh, int kw,
                                      int pad, int out_stride, int out_offset, int stuff, int stuff_offset, int Hout,
                                      int Wout, int reflect, float* y, hipStream_t stream) {
    FLAIR_CHECK(x && filt && y && planes > 0 && Hin > 0 && Win > 0 && kh > 0 && kw > 0 && kh * kw <= 4096,
                "flair_depthwise_filter: bad argument");
    FLAIR_CHECK(out_stride >= 1 && stuff >= 1 && stuff_offset >= 0 && stuff_offset < stuff && Hout > 0 && Wout > 0,
                "flair_depthwise_filter: bad sampling parameters");
    if (kh == 39 && kw == 39 && stuff == 1 && out_stride == 1 && Hout % 32 == 0 && Wout % 32 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
        hipLaunchKernelGGL(depthwise_filter_tiled_kernel<39>, dim3(planes * (Hout / 32) * (Wout / 32)), dim3(256), 0, stream, x, Hin, Win,
                           filt, pad, out_offset, Hout, Wout, reflect, y);
        FLAIR_LAUNCH_CHECK();
        return FLAIR_OK;
    }
    hipLaunchKernelGGL(depthwise_filter_kernel, dim3(grid_for((long)planes * Hout * Wout)), dim3(256),
                       (size_t)kh * kw * sizeof(float), stream, x, planes, Hin, Win, filt, kh, kw, pad, out_stride,
                       out_offset, stuff, stuff_offset, Hout, Wout, reflect, y);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_jpeg_roundtrip(const float* x, int N, int S, const float* q_luma, const float* q_chroma,
                                    const float* dct8, float* workspace, float* y, float* luma_q, float* chroma_q,
                                    hipStream_t stream) {
    FLAIR_CHECK(x && q_luma && q_chroma && dct8 && workspace && y && N > 0 && S > 0 && S % 16 == 0,
                "flair_jpeg_roundtrip: bad argument (S must be a multiple of 16)");
    JpegTables t;
    for (int i = 0; i < 64; ++i) {
        t.q1[i] = q_luma[i];
        t.q2[i] = q_chroma[i];
        t.D[i] = dct8[i];
    }
    hipLaunchKernelGGL(jpeg_to_ycbcr_kernel, dim3(grid_for((long)N * S * S)), dim3(256), 0, stream, x, N, S, workspace);
    FLAIR_LAUNCH_CHECK();
    const long blocks = (long)N * (S / 8) * (S / 8) + (long)N * 2 * (S / 16) * (S / 16);
    FLAIR_CHECK(!luma_q == !chroma_q, "flair_jpeg_roundtrip: give both level planes or neither");
    hipLaunchKernelGGL(jpeg_blocks_kernel, dim3(grid_for(blocks)), dim3(64), 0, stream, workspace, N, S, t, luma_q, chroma_q);
    FLAIR_LAUNCH_CHECK();
    hipLaunchKernelGGL(jpeg_to_rgb_kernel, dim3(grid_for((long)N * S * S)), dim3(256), 0, stream, workspace, N, S, y);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_matmul_f32(const float* A, long a_batch_stride, const float* B, long b_batch_stride, float* C,
                                int batch, int M, int N, int K, hipStream_t stream) {
    FLAIR_CHECK(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0, "flair_matmul_f32: bad argument");
    hipLaunchKernelGGL(matmul_kernel, dim3((N + 15) / 16, (M + 15) / 16, batch), dim3(16, 16), 0, stream, A,
                       a_batch_stride, B, b_batch_stride, C, M, N, K);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_gather_mac_f32(const float* x, long outer, int Lin, long inner, const int* fov, const float* w,
                                    int taps, int Lout, float* y, hipStream_t stream) {
    FLAIR_CHECK(x && fov && w && y && outer > 0 && Lin > 0 && inner > 0 && taps > 0 && Lout > 0,
                "flair_gather_mac_f32: bad argument");
    hipLaunchKernelGGL(gather_mac_kernel, dim3(grid_for(outer * Lout * inner)), dim3(256), 0, stream, x, outer, Lin,
                       inner, fov, w, taps, Lout, y);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
