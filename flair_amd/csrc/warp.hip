// Bilinear resampling kernels on NHWC clips.
//
//  * flair_flow_warp      -- mmedit flow_warp = F.grid_sample(bilinear, align_corners=True,
//                            zeros|border) on pixel coordinates + flow; call sites
//                            guided_diffusion/unet_new.py:706,718,719 (BasicVSR++ propagation)
//                            and SPyNet's per-level warp.
//  * flair_flow_compose   -- flow_n2 = flow_n1 + warp(flow_n2, flow_n1) (unet_new.py:716-718)
//                            on 2-channel f32 flow fields.
//  * flair_resize_nhwc    -- F.interpolate bilinear (align_corners 0/1) / bicubic
//                            (align_corners=False, A=-0.75) / avg_pool2d 2x2 used by SPyNet's
//                            pyramid and by the flow-input resize (unet_new.py:1336-1345).
// All are gather-bound: one thread per (output pixel, 16-byte channel chunk).
#include "common.h"

namespace {

// PyTorch's grid_sample round trip: pixel -> [-1,1] -> pixel (align_corners=True).
// The result passes through an empty asm: `ix - floor(ix)` must see the ROUNDED coordinate like the reference does, and under
// -ffp-contract=fast whether hipcc fuses the last multiply into that subtraction depends on the code around the inlined call --
// two kernels that must agree bit for bit, vsrpp_prep and vsrpp_warp2, came out one ulp apart in their weights.
__device__ __forceinline__ float gs_coord(float pix, int size) {
    const float denom = (float)(size > 1 ? size - 1 : 1);
    const float g = 2.0f * pix / denom - 1.0f;
    float r = ((g + 1.f) / 2.f) * (float)(size - 1);
    asm("" : "+v"(r));
    return r;
}

template <typename E>
__global__ void flow_warp_kernel(const E* x, int xLd, const float* flow, int fLd, int F, int H, int W, int C, int border,
                                 E* y, int yLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = (long)F * H * W * cv;
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, (unsigned)((size_t)F * H * W * xLd * sizeof(E)));     // < 2 GiB: host check
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long f = p / ((long)W * H);
        const float2 fl = *reinterpret_cast<const float2*>(flow + p * fLd);
        float ix = gs_coord((float)w + fl.x, W);
        float iy = gs_coord((float)h + fl.y, H);
        if (border) {
            ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
            iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
        }
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = ix - fx, ay = iy - fy;
        const float wgt[4] = {(1.f - ax) * (1.f - ay), ax * (1.f - ay), (1.f - ax) * ay, ax * ay};
        // all four corner loads in flight at once: a whole-clip descriptor, out-of-range offset for corners outside the frame
        // (behind `if (inside)` branches every load was followed by its own `s_waitcnt vmcnt(0)`)
        uint4 cq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
            const bool in = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
            cq[q] = buf_load16(xr, in ? (unsigned)(((f * H + yy) * W + xx) * xLd + c0) * (unsigned)sizeof(E) : FLAIR_OOB);
        }
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[VEC];
            Vec16<E>::load(reinterpret_cast<const E*>(&cq[q]), v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wgt[q], v[k], acc[k]);
        }
        Vec16<E>::store(y + p * yLd + c0, acc);
    }
}

// out = f1 + warp(f2, f1), zeros padding; flows are [F][H][W][2] f32
__global__ void flow_compose_kernel(const float* f1, const float* f2, int F, int H, int W, float* out) {
    const long total = (long)F * H * W;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        const long f = p / ((long)W * H);
        const float2 fl = *reinterpret_cast<const float2*>(f1 + p * 2);
        const float ix = gs_coord((float)w + fl.x, W), iy = gs_coord((float)h + fl.y, H);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = ix - fx, ay = iy - fy;
        const float wgt[4] = {(1.f - ax) * (1.f - ay), ax * (1.f - ay), (1.f - ax) * ay, ax * ay};
        float ox = 0.f, oy = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
            if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) {
                const float2 v = *reinterpret_cast<const float2*>(f2 + ((f * H + yy) * W + xx) * 2);
                ox = fmaf(wgt[q], v.x, ox);
                oy = fmaf(wgt[q], v.y, oy);
            }
        }
        *reinterpret_cast<float2*>(out + p * 2) = make_float2(fl.x + ox, fl.y + oy);
    }
}


// ---- BasicVSR++ alignment inputs of one propagation step in ONE launch (unet_new.py:704-722):
//   cond_n1 = warp(prop, flow_n1);  flow_n2 = flow_n1 + warp(flow_prev, flow_n1);
//   cond_n2 = warp(feat_n2, flow_n2);  flowpad[p][0..3] = (flow_n1, flow_n2) cast to E
// One thread per (pixel, 16-byte channel chunk); the flow composition is recomputed per chunk
// (2 channels, cheap) instead of being a launch of its own.
__device__ __forceinline__ void bil_setup(float px, float py, int W, int H, int (&xy)[2], float (&wgt)[4]) {
    const float ix = gs_coord(px, W), iy = gs_coord(py, H);
    const float fx = floorf(ix), fy = floorf(iy);
    xy[0] = (int)fminf(fmaxf(fx, -2.f), (float)W);
    xy[1] = (int)fminf(fmaxf(fy, -2.f), (float)H);
    const float ax = ix - fx, ay = iy - fy;
    wgt[0] = (1.f - ax) * (1.f - ay); wgt[1] = ax * (1.f - ay); wgt[2] = (1.f - ax) * ay; wgt[3] = ax * ay;
}

template <typename E>
__global__ void vsrpp_prep_kernel(const E* prop, int propLd, const E* feat2, int feat2Ld, const float* flow1,
                                  const float* flowPrev, int H, int W, int C, E* cond1, int cond1Ld, E* cond2,
                                  int cond2Ld, float* flow2Out, E* flowpad, int padLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = (long)H * W * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        const int w = (int)(p % W), h = (int)(p / W);
        const float2 f1 = *reinterpret_cast<const float2*>(flow1 + p * 2);
        int xy[2];
        float wg[4];
        bil_setup((float)w + f1.x, (float)h + f1.y, W, H, xy, wg);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        float2 f2 = make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = xy[0] + (q & 1), yy = xy[1] + (q >> 1);
            if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) {
                float v[VEC];
                Vec16<E>::load(prop + ((long)yy * W + xx) * propLd + c0, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wg[q], v[k], acc[k]);
                if (flowPrev) {
                    const float2 fp = *reinterpret_cast<const float2*>(flowPrev + ((long)yy * W + xx) * 2);
                    f2.x = fmaf(wg[q], fp.x, f2.x);
                    f2.y = fmaf(wg[q], fp.y, f2.y);
                }
            }
        }
        Vec16<E>::store(cond1 + p * cond1Ld + c0, acc);
        if (flowPrev) {
            f2.x += f1.x;
            f2.y += f1.y;
            bil_setup((float)w + f2.x, (float)h + f2.y, W, H, xy, wg);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int xx = xy[0] + (q & 1), yy = xy[1] + (q >> 1);
                if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) {
                    float v[VEC];
                    Vec16<E>::load(feat2 + ((long)yy * W + xx) * feat2Ld + c0, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wg[q], v[k], acc[k]);
                }
            }
            Vec16<E>::store(cond2 + p * cond2Ld + c0, acc);
        }
        if (c0 == 0) {
            if (flowPrev) *reinterpret_cast<float2*>(flow2Out + p * 2) = f2;
            ET<E>::st(flowpad + p * padLd + 0, f1.x);
            ET<E>::st(flowpad + p * padLd + 1, f1.y);
            ET<E>::st(flowpad + p * padLd + 2, f2.x);
            ET<E>::st(flowpad + p * padLd + 3, f2.y);
        }
    }
}

// Two independent warps of one propagation step with BOTH flows given (the second-order flow
// flow1 + warp(flow_prev, flow1) depends on the flows only, so the caller composes it once per clip
// instead of once per denoising step): cond1 = warp(prop, flow1), cond2 = warp(feat2, flow2).
// blockIdx.y selects the warp, so the two gather chains run side by side instead of one after the other.
template <typename E>
__global__ void vsrpp_warp2_kernel(const E* prop, int propLd, const E* feat2, int feat2Ld, const float* flow1,
                                   const float* flow2, int H, int W, int C, E* cond1, int cond1Ld, E* cond2,
                                   int cond2Ld) {
    prefetch_kernargs<96>();
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const long total = (long)H * W * cv;
    const bool second = blockIdx.y == 1;
    const E* src = second ? feat2 : prop;
    const int srcLd = second ? feat2Ld : propLd;
    const float* flow = second ? flow2 : flow1;
    E* dst = second ? cond2 : cond1;
    const int dstLd = second ? cond2Ld : cond1Ld;
    // The four corners are unconditional buffer loads on a descriptor of the source frame (corners outside the image get an
    // out-of-range offset and read 0, which adds nothing): as plain loads under `if (inside)` branches hipcc followed each
    // one with `s_waitcnt vmcnt(0)` -- four serial round trips to memory per pixel in a kernel that is nothing but latency.
    const __amdgpu_buffer_rsrc_t sr = make_rsrc(src, (unsigned)((size_t)H * W * srcLd * sizeof(E)));
    // (32-bit index arithmetic: one frame has < 2^31 pieces, and 64-bit division is a loop on this hardware)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total; i += gridDim.x * blockDim.x) {
        const unsigned p = i / (unsigned)cv;
        const int c0 = (int)(i - p * (unsigned)cv) * VEC;
        const int h = (int)(p / (unsigned)W), w = (int)(p - (unsigned)h * (unsigned)W);
        const float2 f = *reinterpret_cast<const float2*>(flow + (size_t)p * 2);
        int xy[2];
        float wg[4];
        bil_setup((float)w + f.x, (float)h + f.y, W, H, xy, wg);
        uint4 cq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = xy[0] + (q & 1), yy = xy[1] + (q >> 1);
            const bool in = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
            cq[q] = buf_load16(sr, in ? (unsigned)((yy * W + xx) * srcLd + c0) * (unsigned)sizeof(E) : FLAIR_OOB);
        }
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[VEC];
            Vec16<E>::load(reinterpret_cast<const E*>(&cq[q]), v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = fmaf(wg[q], v[k], acc[k]);
        }
        Vec16<E>::store(dst + (size_t)p * dstLd + c0, acc);
    }
}

// ---- generic scalar-channel resize (few channels: images and flows) ------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// nearest x2 (F.interpolate(scale_factor=2) of Upsample layers): a pure copy, 16 bytes per thread
template <typename E>
__global__ void nearest2x_kernel(const E* x, int xLd, long F, int Hi, int Wi, int C, E* y, int yLd) {
    constexpr int VEC = ET<E>::VEC;
    const int cv = C / VEC;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const long total = F * Ho * Wo * cv;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * VEC;
        const long p = i / cv;
        const int wo = (int)(p % Wo);
        const int ho = (int)((p / Wo) % Ho);
        const long f = p / ((long)Wo * Ho);
        *reinterpret_cast<uint4*>(y + p * yLd + c) =
            *reinterpret_cast<const uint4*>(x + ((f * Hi + (ho >> 1)) * Wi + (wo >> 1)) * xLd + c);
    }
}

// mode: 0 bilinear align_corners=False, 1 bilinear align_corners=True, 2 bicubic
// (align_corners=False, A=-0.75, border-clamped taps), 3 2x2 average pool, 4 nearest.
template <typename E>
__global__ void resize_kernel(const E* x, int xLd, int F, int Hi, int Wi, int C, int mode, int Ho, int Wo, E* y,
                              int yLd, float outScaleX, float outScaleY) {
    const long total = (long)F * Ho * Wo * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int wo = (int)(p % Wo);
        const int ho = (int)((p / Wo) % Ho);
        const long f = p / ((long)Wo * Ho);
        const E* fb = x + f * Hi * Wi * xLd + c;
        float r = 0.f;
        if (mode == 4) {  // nearest (F.interpolate default): src = floor(dst * in/out)
            int yy = (int)floorf((float)ho * ((float)Hi / (float)Ho));
            int xx = (int)floorf((float)wo * ((float)Wi / (float)Wo));
            yy = yy > Hi - 1 ? Hi - 1 : yy;
            xx = xx > Wi - 1 ? Wi - 1 : xx;
            r = ET<E>::ld(fb + ((long)yy * Wi + xx) * xLd);
        } else if (mode == 3) {
            for (int q = 0; q < 4; ++q)
                r += ET<E>::ld(fb + ((long)(2 * ho + (q >> 1)) * Wi + 2 * wo + (q & 1)) * xLd);
            r *= 0.25f;
        } else if (mode == 2) {
            const float sx = (float)Wi / (float)Wo, sy = (float)Hi / (float)Ho;
            const float rx = sx * ((float)wo + 0.5f) - 0.5f, ry = sy * ((float)ho + 0.5f) - 0.5f;
            const float fx = floorf(rx), fy = floorf(ry);
            const float tx = rx - fx, ty = ry - fy;
            const float A = -0.75f;
            const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
            const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
            for (int j = 0; j < 4; ++j) {
                int yy = (int)fy - 1 + j;
                yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
                float rowv = 0.f;
                for (int k = 0; k < 4; ++k) {
                    int xx = (int)fx - 1 + k;
                    xx = xx < 0 ? 0 : (xx > Wi - 1 ? Wi - 1 : xx);
                    rowv += cx[k] * ET<E>::ld(fb + ((long)yy * Wi + xx) * xLd);
                }
                r += cy[j] * rowv;
            }
        } else {
            float rx, ry;
            if (mode == 1) {
                const float sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
                const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
                rx = sx * (float)wo;
                ry = sy * (float)ho;
            } else {
                const float sx = (float)Wi / (float)Wo, sy = (float)Hi / (float)Ho;
                rx = fmaxf(sx * ((float)wo + 0.5f) - 0.5f, 0.f);
                ry = fmaxf(sy * ((float)ho + 0.5f) - 0.5f, 0.f);
            }
            const int x0 = (int)rx, y0 = (int)ry;
            const int x1 = x0 + (x0 < Wi - 1 ? 1 : 0), y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
            const float ax = rx - (float)x0, ay = ry - (float)y0;
            const float v00 = ET<E>::ld(fb + ((long)y0 * Wi + x0) * xLd), v01 = ET<E>::ld(fb + ((long)y0 * Wi + x1) * xLd);
            const float v10 = ET<E>::ld(fb + ((long)y1 * Wi + x0) * xLd), v11 = ET<E>::ld(fb + ((long)y1 * Wi + x1) * xLd);
            r = (1.f - ay) * ((1.f - ax) * v00 + ax * v01) + ay * ((1.f - ax) * v10 + ax * v11);
        }
        r *= (c == 0 ? outScaleX : (c == 1 ? outScaleY : 1.f));
        ET<E>::st(y + p * yLd + c, r);
    }
}

inline int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int flair_flow_warp(const void* x, int dtype, int x_ld, const float* flow, int flow_ld, int F, int H,
                               int W, int C, int border, void* y, int y_ld, hipStream_t stream) {
    FLAIR_CHECK(x && flow && y && F > 0 && H > 0 && W > 0 && C > 0 && flow_ld >= 2 && flow_ld % 2 == 0,
                "flair_flow_warp: bad argument");
    FLAIR_CHECK((unsigned long long)F * H * W * x_ld * (dtype == FLAIR_BF16 ? 2 : 4) < 0x80000000ull,
                "flair_flow_warp: the source clip spans >= 2 GiB (32-bit gather offsets): call per frame");
    if (dtype == FLAIR_BF16) {
        FLAIR_CHECK(C % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0, "flair_flow_warp: bf16 needs C %% 8 == 0");
        hipLaunchKernelGGL(flow_warp_kernel<bf16_t>, dim3(grid_for((long)F * H * W * (C / 8))), dim3(256), 0, stream,
                           (const bf16_t*)x, x_ld, flow, flow_ld, F, H, W, C, border, (bf16_t*)y, y_ld);
    } else if (dtype == FLAIR_F32) {
        FLAIR_CHECK(C % 4 == 0 && x_ld % 4 == 0 && y_ld % 4 == 0, "flair_flow_warp: f32 needs C %% 4 == 0");
        hipLaunchKernelGGL(flow_warp_kernel<float>, dim3(grid_for((long)F * H * W * (C / 4))), dim3(256), 0, stream,
                           (const float*)x, x_ld, flow, flow_ld, F, H, W, C, border, (float*)y, y_ld);
    } else {
        FLAIR_CHECK(false, "flair_flow_warp: bad dtype");
    }
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_flow_compose(const float* f1, const float* f2, int F, int H, int W, float* out,
                                  hipStream_t stream) {
    FLAIR_CHECK(f1 && f2 && out && F > 0 && H > 0 && W > 0, "flair_flow_compose: bad argument");
    hipLaunchKernelGGL(flow_compose_kernel, dim3(grid_for((long)F * H * W)), dim3(256), 0, stream, f1, f2, F, H, W,
                       out);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_resize_nhwc(const void* x, int dtype, int x_ld, int F, int Hi, int Wi, int C, int mode, int Ho,
                                 int Wo, void* y, int y_ld, float scale_c0, float scale_c1, hipStream_t stream) {
    FLAIR_CHECK(x && y && F > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && mode >= 0 && mode <= 4,
                "flair_resize_nhwc: bad argument");
    FLAIR_CHECK(mode != 3 || (Hi == 2 * Ho && Wi == 2 * Wo), "flair_resize_nhwc: avg-pool needs exact 2x");
    const long n = (long)F * Ho * Wo * C;
    {
        const int vec = dtype == FLAIR_BF16 ? 8 : 4;
        if (mode == 4 && Ho == 2 * Hi && Wo == 2 * Wi && C % vec == 0 && x_ld % vec == 0 && y_ld % vec == 0 &&
            scale_c0 == 1.f && scale_c1 == 1.f && (dtype == FLAIR_BF16 || dtype == FLAIR_F32)) {
            if (dtype == FLAIR_BF16)
                hipLaunchKernelGGL(nearest2x_kernel<bf16_t>, dim3(grid_for(n / vec)), dim3(256), 0, stream, (const bf16_t*)x,
                                   x_ld, (long)F, Hi, Wi, C, (bf16_t*)y, y_ld);
            else
                hipLaunchKernelGGL(nearest2x_kernel<float>, dim3(grid_for(n / vec)), dim3(256), 0, stream, (const float*)x,
                                   x_ld, (long)F, Hi, Wi, C, (float*)y, y_ld);
            FLAIR_LAUNCH_CHECK();
            return FLAIR_OK;
        }
    }
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(resize_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, stream, (const bf16_t*)x, x_ld, F,
                           Hi, Wi, C, mode, Ho, Wo, (bf16_t*)y, y_ld, scale_c0, scale_c1);
    else if (dtype == FLAIR_F32)
        hipLaunchKernelGGL(resize_kernel<float>, dim3(grid_for(n)), dim3(256), 0, stream, (const float*)x, x_ld, F,
                           Hi, Wi, C, mode, Ho, Wo, (float*)y, y_ld, scale_c0, scale_c1);
    else
        FLAIR_CHECK(false, "flair_resize_nhwc: bad dtype");
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_vsrpp_warp2(const void* prop, int prop_ld, const void* feat2, int feat2_ld, const float* flow1,
                                 const float* flow2, int dtype, int H, int W, int C, void* cond1, int cond1_ld,
                                 void* cond2, int cond2_ld, hipStream_t stream) {
    FLAIR_CHECK(prop && flow1 && cond1 && H > 0 && W > 0 && C > 0, "flair_vsrpp_warp2: bad argument");
    FLAIR_CHECK(!flow2 || (feat2 && cond2), "flair_vsrpp_warp2: second-order inputs incomplete");
    const int vec = dtype == FLAIR_BF16 ? 8 : 4;
    FLAIR_CHECK(dtype == FLAIR_BF16 || dtype == FLAIR_F32, "flair_vsrpp_warp2: bad dtype");
    FLAIR_CHECK(C % vec == 0, "flair_vsrpp_warp2: C %% %d", vec);
    FLAIR_CHECK((unsigned long long)H * W * prop_ld * (16 / vec) < 0x80000000ull && (unsigned long long)H * W * feat2_ld * (16 / vec) < 0x80000000ull,
                "flair_vsrpp_warp2: a source frame spans >= 2 GiB");
    const dim3 grid(grid_for((long)H * W * (C / vec)), flow2 ? 2 : 1);
    if (dtype == FLAIR_BF16)
        hipLaunchKernelGGL(vsrpp_warp2_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)prop, prop_ld,
                           (const bf16_t*)feat2, feat2_ld, flow1, flow2, H, W, C, (bf16_t*)cond1, cond1_ld,
                           (bf16_t*)cond2, cond2_ld);
    else
        hipLaunchKernelGGL(vsrpp_warp2_kernel<float>, grid, dim3(256), 0, stream, (const float*)prop, prop_ld,
                           (const float*)feat2, feat2_ld, flow1, flow2, H, W, C, (float*)cond1, cond1_ld,
                           (float*)cond2, cond2_ld);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

extern "C" int flair_vsrpp_prep(const void* prop, int prop_ld, const void* feat2, int feat2_ld, const float* flow1,
                                const float* flow_prev, int dtype, int H, int W, int C, void* cond1, int cond1_ld,
                                void* cond2, int cond2_ld, float* flow2_out, void* flowpad, int pad_ld,
                                hipStream_t stream) {
    FLAIR_CHECK(prop && flow1 && cond1 && flowpad && H > 0 && W > 0 && C > 0 && pad_ld >= 4,
                "flair_vsrpp_prep: bad argument");
    FLAIR_CHECK(!flow_prev || (feat2 && cond2 && flow2_out), "flair_vsrpp_prep: second-order inputs incomplete");
    if (dtype == FLAIR_BF16) {
        FLAIR_CHECK(C % 8 == 0, "flair_vsrpp_prep: C %% 8");
        hipLaunchKernelGGL(vsrpp_prep_kernel<bf16_t>, dim3(grid_for((long)H * W * (C / 8))), dim3(256), 0, stream,
                           (const bf16_t*)prop, prop_ld, (const bf16_t*)feat2, feat2_ld, flow1, flow_prev, H, W, C,
                           (bf16_t*)cond1, cond1_ld, (bf16_t*)cond2, cond2_ld, flow2_out, (bf16_t*)flowpad, pad_ld);
    } else if (dtype == FLAIR_F32) {
        FLAIR_CHECK(C % 4 == 0, "flair_vsrpp_prep: C %% 4");
        hipLaunchKernelGGL(vsrpp_prep_kernel<float>, dim3(grid_for((long)H * W * (C / 4))), dim3(256), 0, stream,
                           (const float*)prop, prop_ld, (const float*)feat2, feat2_ld, flow1, flow_prev, H, W, C,
                           (float*)cond1, cond1_ld, (float*)cond2, cond2_ld, flow2_out, (float*)flowpad, pad_ld);
    } else {
        FLAIR_CHECK(false, "flair_vsrpp_prep: bad dtype");
    }
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
