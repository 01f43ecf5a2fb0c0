// GroupNorm(32 groups) over a whole clip + fused affine/FiLM + SiLU (+ 2x resample).
//
// Replaces GroupNorm32 behind LazyReshaper3D (guided_diffusion/nn_new.py:17-19,
// nn.py:359-367) as used by ResBlock.in_layers/out_layers (unet_new.py:237-278,
// 309-329: SiLU, (1+scale)*norm+shift, Upsample/Downsample h_upd/x_upd), the
// attention norms (:358,:408,:461) and the output head (:1216-1222).
//
// Statistics are joint over (C/groups) x frames_per_stat x H x W.  HBM-bound: three
// launches per norm, two passes over the tensor:
//   gn_partial  : every workgroup streams a pixel range with 16-byte loads, keeps
//                 per-channel f32 sum / sum-of-squares in registers, folds the rows and
//                 the channels of a group through LDS and writes one partial per group;
//   gn_finalize : one wavefront per (stat, group) adds the partials in f64 (4 loads in
//                 flight per lane, 64-lane shuffle tree) -> mean, rstd.  A separate launch
//                 on purpose: folding it into gn_partial with a "last workgroup" ticket needs
//                 a device-scope fence in every workgroup, which on the 8-XCD part writes
//                 back / invalidates the L2 and cost 5x the whole statistics pass;
//   gn_apply    : y = act(x*A + B) with A,B folded per (frame, channel) from
//                 mean/rstd/gamma/beta/(scale,shift); optionally writes the 2x2
//                 average-pooled or nearest-upsampled result and the resampled
//                 raw input in the same pass.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

struct GnSrc {
    const void* x[2];
    int c[2];   // channels per segment (sum = C)
    int ld[2];  // pixel stride per segment
};

template <typename E>
__global__ void gn_partial_kernel(GnSrc s, int C, int groups, long pixPerStat, int blocksPerStat, float* part) {
    prefetch_kernargs<sizeof(GnSrc) + 32>();
    constexpr int VEC = ET<E>::VEC;
    extern __shared__ float red[];  // [2][rows][C]; row 0 of each plane ends up holding the channel totals
    const int cv = C / VEC;
    const int rows = blockDim.x / cv;
    const int slot = threadIdx.x % cv, r = threadIdx.x / cv;
    const int stat = blockIdx.x / blocksPerStat, blk = blockIdx.x % blocksPerStat;
    const long per = (pixPerStat + blocksPerStat - 1) / blocksPerStat;
    const long beg = stat * pixPerStat + (long)blk * per;
    long end = beg + per;
    const long lim = (stat + 1) * pixPerStat;
    if (end > lim) end = lim;

    int c0 = slot * VEC;
    const E* base;
    int ld;
    if (c0 < s.c[0]) {
        base = reinterpret_cast<const E*>(s.x[0]) + c0;
        ld = s.ld[0];
    } else {
        base = reinterpret_cast<const E*>(s.x[1]) + (c0 - s.c[0]);
        ld = s.ld[1];
    }
    float sum[VEC], sq[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) sum[i] = sq[i] = 0.f;
#pragma unroll 4
    for (long p = beg + r; p < end; p += rows) {
        float v[VEC];
        Vec16<E>::load(base + p * ld, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            sum[i] += v[i];
            sq[i] = fmaf(v[i], v[i], sq[i]);
        }
    }
    float* rs = red;
    float* rq = red + rows * C;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        rs[r * C + c0 + i] = sum[i];
        rq[r * C + c0 + i] = sq[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < rows; ++k) {
            a += rs[k * C + c];
            b += rq[k * C + c];
        }
        rs[c] = a;      // column c is read and overwritten by this thread only
        rq[c] = b;
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < cpg; ++k) {
            a += rs[g * cpg + k];
            b += rq[g * cpg + k];
        }
        part[((long)blockIdx.x * 2 + 0) * groups + g] = a;
        part[((long)blockIdx.x * 2 + 1) * groups + g] = b;
    }
}

// one wave per (stat, group)
__global__ void gn_finalize_kernel(const float* part, int groups, int cpg, int blocksPerStat, long pixPerStat,
                                   float eps, float* stats) {
    const int lane = threadIdx.x & 63;
    const int sg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int stat = sg / groups, g = sg % groups;
    const float* pp = part + ((long)stat * blocksPerStat * 2) * groups + g;
    const long st = 2L * groups;
    double a = 0.0, b = 0.0;
    int i = lane;
    for (; i + 192 < blocksPerStat; i += 256) {
        const float a0 = pp[i * st], b0 = pp[i * st + groups];
        const float a1 = pp[(i + 64) * st], b1 = pp[(i + 64) * st + groups];
        const float a2 = pp[(i + 128) * st], b2 = pp[(i + 128) * st + groups];
        const float a3 = pp[(i + 192) * st], b3 = pp[(i + 192) * st + groups];
        a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; i < blocksPerStat; i += 64) {
        a += (double)pp[i * st];
        b += (double)pp[i * st + groups];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if (lane == 0) {
        const double cnt = (double)pixPerStat * cpg;
        const double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0) var = 0;
        stats[sg * 2 + 0] = (float)mean;
        stats[sg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

struct GnApply {
    GnSrc s;
    int C, groups;
    int F, H, W;           // frames, input frame size
    int framesPerStat;
    const float* stats;    // [nstat][groups][2]
    const float* gamma;    // [C]
    const float* beta;     // [C]
    const float* film;     // [F][filmLd] rows of (scale[C] | shift[C]) or null
    int filmLd;
    int act;
    int resample;          // 0 none, 1 avg-pool 2x2, 2 nearest x2
    void* y;  int yLd;     // activated (and resampled) output
    void* raw; int rawLd;  // optional: resampled un-normalised input
    int blocksPerFrame;
    int filmVec;           // gn_group_kernel: film rows are 16-byte aligned (float4 loads)
};

template <typename E>
__global__ __launch_bounds__(512) void gn_apply_kernel(GnApply a) {
    prefetch_kernargs<sizeof(GnApply)>();
    constexpr int VEC = ET<E>::VEC;
    const int cv = a.C / VEC;
    const int rows = blockDim.x / cv;
    const int slot = threadIdx.x % cv, r = threadIdx.x / cv;
    const int f = blockIdx.x / a.blocksPerFrame, blk = blockIdx.x % a.blocksPerFrame;
    const int c0 = slot * VEC;
    const E* base;
    int ld;
    if (c0 < a.s.c[0]) {
        base = reinterpret_cast<const E*>(a.s.x[0]) + c0;
        ld = a.s.ld[0];
    } else {
        base = reinterpret_cast<const E*>(a.s.x[1]) + (c0 - a.s.c[0]);
        ld = a.s.ld[1];
    }
    // fold normalisation + affine + FiLM into y = x*A + B.  All parameter loads of the thread's VEC channels are issued
    // before the first use (statistics, gamma, beta; then, under ONE uniform branch, the FiLM row): written channel by channel
    // with the `if (film)` inside, every channel was two serial round trips to the L2 -- 16 of them before a workgroup
    // touched its first pixel, on a pass that lasts ~40 us.
    float A[VEC], B[VEC];
    const int cpg = a.C / a.groups;
    const int stat = f / a.framesPerStat;
    {
        float2 st[VEC];
        float gm[VEC], bt[VEC], sc[VEC], sh[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = c0 + i;
            st[i] = *reinterpret_cast<const float2*>(a.stats + (stat * a.groups + c / cpg) * 2);
            gm[i] = a.gamma[c];
            bt[i] = a.beta[c];
            sc[i] = 0.f;
            sh[i] = 0.f;
        }
        if (a.film) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sc[i] = a.film[(long)f * a.filmLd + c0 + i];
                sh[i] = a.film[(long)f * a.filmLd + a.C + c0 + i];
            }
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float ga = gm[i] * st[i].y;
            float be = bt[i] - st[i].x * ga;
            if (a.film) {
                const float m = 1.f + sc[i];
                ga *= m;
                be = be * m + sh[i];
            }
            A[i] = ga;
            B[i] = be;
        }
    }
    // The streaming loops are compiled once per activation CLASS (none / SiLU / anything else), chosen by one wave-uniform
    // branch: with apply_act's switch inside them the kernel was 12 400 instructions with a branch per element, and SiLU's IEEE
    // divide (ten instructions) made the pass VALU-bound next to its HBM time; bf16 outputs take v_rcp_f32 (1 ulp) instead.
    auto run = [&](auto actTag) {
        constexpr int ACTC = decltype(actTag)::value;
        auto act1 = [&](float x) -> float {
            if constexpr (ACTC == 0) return x;
            else if constexpr (ACTC == 1) {
                if constexpr (sizeof(E) == 2) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
                else return silu_f(x);
            } else return apply_act(x, a.act);
        };
        E* y = reinterpret_cast<E*>(a.y);
        E* raw = reinterpret_cast<E*>(a.raw);
        const long fin = (long)f * a.H * a.W;
        if (a.resample == 0) {
            const long n = (long)a.H * a.W;
            const long per = (n + a.blocksPerFrame - 1) / a.blocksPerFrame;
            long end = (blk + 1) * per;
            if (end > n) end = n;
            // y never aliases x (header contract): keep 4 loads in flight before the first store
            long p = blk * per + r;
            for (; p + 3 * rows < end; p += 4 * rows) {
                uint4 raw4[4];
    #pragma unroll
                for (int u = 0; u < 4; ++u)
                    raw4[u] = *reinterpret_cast<const uint4*>(base + (fin + p + (long)u * rows) * ld);
    #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v[VEC];
                    Vec16<E>::load(reinterpret_cast<const E*>(&raw4[u]), v);
    #pragma unroll
                    for (int i = 0; i < VEC; ++i) v[i] = act1(fmaf(v[i], A[i], B[i]));
                    Vec16<E>::store(y + (fin + p + (long)u * rows) * a.yLd + c0, v);
                }
            }
            for (; p < end; p += rows) {
                float v[VEC];
                Vec16<E>::load(base + (fin + p) * ld, v);
    #pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = act1(fmaf(v[i], A[i], B[i]));
                Vec16<E>::store(y + (fin + p) * a.yLd + c0, v);
            }
        } else if (a.resample == 1) {
            const int Ho = a.H / 2, Wo = a.W / 2;
            const long n = (long)Ho * Wo;
            const long per = (n + a.blocksPerFrame - 1) / a.blocksPerFrame;
            long end = (blk + 1) * per;
            if (end > n) end = n;
            const long fout = (long)f * n;
            for (long p = blk * per + r; p < end; p += rows) {
                const int ho = (int)(p / Wo), wo = (int)(p % Wo);
                float accv[VEC], accr[VEC];
    #pragma unroll
                for (int i = 0; i < VEC; ++i) accv[i] = accr[i] = 0.f;
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long pi = fin + (long)(2 * ho + (q >> 1)) * a.W + 2 * wo + (q & 1);
                    float v[VEC];
                    Vec16<E>::load(base + pi * ld, v);
    #pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        accr[i] += v[i];
                        accv[i] += act1(fmaf(v[i], A[i], B[i]));
                    }
                }
    #pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    accv[i] *= 0.25f;
                    accr[i] *= 0.25f;
                }
                Vec16<E>::store(y + (fout + p) * a.yLd + c0, accv);
                if (raw) Vec16<E>::store(raw + (fout + p) * a.rawLd + c0, accr);
            }
        } else {
            const long n = (long)a.H * a.W;
            const long per = (n + a.blocksPerFrame - 1) / a.blocksPerFrame;
            long end = (blk + 1) * per;
            if (end > n) end = n;
            const int Wo = a.W * 2;
            const long fout = (long)f * n * 4;
            for (long p = blk * per + r; p < end; p += rows) {
                const int h = (int)(p / a.W), w = (int)(p % a.W);
                float v[VEC], u[VEC];
                Vec16<E>::load(base + (fin + p) * ld, u);
    #pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = act1(fmaf(u[i], A[i], B[i]));
    #pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long po = fout + (long)(2 * h + (q >> 1)) * Wo + 2 * w + (q & 1);
                    Vec16<E>::store(y + po * a.yLd + c0, v);
                    if (raw) Vec16<E>::store(raw + po * a.rawLd + c0, u);
                }
            }
        }
    };
    if (a.act == FLAIR_ACT_NONE) run(std::integral_constant<int, 0>{});
    else if (a.act == FLAIR_ACT_SILU) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
}

// ---- small tensors: the whole norm in ONE launch, one workgroup per (statistic, group) ------------------
// 93 of the 205 norms of a forward are on the 16x16 .. 4x4 levels, where the three dependent launches above cost
// ~20 us for microseconds of traffic.  Here a workgroup owns one group (cpg = C/groups channels =
// one or more 16-byte pieces per pixel): it streams the group's pieces once into LDS (<= 128 KB per group), reduces
// sum / sum of squares in f32 per thread and in f64 across the workgroup, then applies y = act(x*A + B) from LDS.
// No cross-workgroup reduction, no finalize launch.  Kernel trace inside the bench (r02n): 10.5 us against 19.7 us
// for the three launches on the same tensors; a two-pass form for 256 KB groups (the 32x32 level) measured 100 us
// (every 16-byte piece sits in its own 128-byte line, 32 workgroups only) and was dropped.
template <typename E>
__global__ __launch_bounds__(1024) void gn_group_kernel(GnApply a, float eps) {
    prefetch_kernargs<sizeof(GnApply) + 8>();
    constexpr int VEC = ET<E>::VEC;
    constexpr int NT = 1024, U = 4;
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    __shared__ double redS[16], redQ[16];
    __shared__ float statS[2];
    const int cpg = a.C / a.groups;
    const int pp = cpg / VEC;                                   // 16-byte pieces per pixel of this group (1, 2, 4: divides NT)
    const int stat = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    const long pixPerStat = (long)a.framesPerStat * a.H * a.W;
    const long pix0 = stat * pixPerStat;
    const int k = threadIdx.x % pp;                             // this thread's piece of the group: fixed channels
    const int rows = NT / pp;
    const int c0 = g * cpg + k * VEC;
    // a group lies inside one input segment (segment widths are multiples of cpg: checked by the host)
    const bool seg1 = c0 >= a.s.c[0];
    const E* base = reinterpret_cast<const E*>(seg1 ? a.s.x[1] : a.s.x[0]) + (seg1 ? c0 - a.s.c[0] : c0);
    const int ld = seg1 ? a.s.ld[1] : a.s.ld[0];
    uint4* cache = reinterpret_cast<uint4*>(gsm);
    float ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
        *reinterpret_cast<float4*>(&ga[e]) = *reinterpret_cast<const float4*>(a.gamma + c0 + e);
        *reinterpret_cast<float4*>(&be[e]) = *reinterpret_cast<const float4*>(a.beta + c0 + e);
    }

    float sum = 0.f, sq = 0.f;
    auto accumulate = [&](const uint4& raw) {
        float v[VEC];
        Vec16<E>::load(reinterpret_cast<const E*>(&raw), v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sum += v[e];
            sq = fmaf(v[e], v[e], sq);
        }
    };
    long q = threadIdx.x / pp;
    for (; q + (U - 1) * rows < pixPerStat; q += U * rows) {
        uint4 raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (pix0 + q + (long)u * rows) * ld);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cache[(q + (long)u * rows) * pp + k] = raw[u];
            accumulate(raw[u]);
        }
    }
    for (; q < pixPerStat; q += rows) {
        const uint4 raw = *reinterpret_cast<const uint4*>(base + (pix0 + q) * ld);
        cache[q * pp + k] = raw;
        accumulate(raw);
    }
    double ds = (double)sum, dq = (double)sq;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ds += __shfl_xor(ds, off);
        dq += __shfl_xor(dq, off);
    }
    if ((threadIdx.x & 63) == 0) {
        redS[threadIdx.x >> 6] = ds;
        redQ[threadIdx.x >> 6] = dq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        for (int w = 0; w < NT / 64; ++w) {
            ts += redS[w];
            tq += redQ[w];
        }
        const double cnt = (double)pixPerStat * cpg;
        const double mean = ts / cnt;
        double var = tq / cnt - mean * mean;
        if (var < 0) var = 0;
        statS[0] = (float)mean;
        statS[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const float mean = statS[0], rstd = statS[1];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        ga[e] *= rstd;
        be[e] -= mean * ga[e];
    }
    E* y = reinterpret_cast<E*>(a.y);
    const int hw = a.H * a.W;
    // FiLM rows of a pixel's frame: requested for TWO of the thread's pixels before either is finished (one load round trip
    // per pair instead of one per pixel; these launches are a few microseconds long), activation per class as in gn_apply
    auto film_load = [&](long p, float (&sc)[VEC], float (&sh)[VEC]) {
        const float* fr = a.film + (p / hw) * a.filmLd + c0;
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
            if (a.filmVec) {
                const float4 s4 = *reinterpret_cast<const float4*>(fr + e);
                const float4 h4 = *reinterpret_cast<const float4*>(fr + a.C + e);
                sc[e] = s4.x; sc[e + 1] = s4.y; sc[e + 2] = s4.z; sc[e + 3] = s4.w;
                sh[e] = h4.x; sh[e + 1] = h4.y; sh[e + 2] = h4.z; sh[e + 3] = h4.w;
            } else {
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    sc[e + jx] = fr[e + jx];
                    sh[e + jx] = fr[a.C + e + jx];
                }
            }
        }
    };
    auto run = [&](auto actTag) {
        constexpr int ACTC = decltype(actTag)::value;
        auto act1 = [&](float x) -> float {
            if constexpr (ACTC == 0) return x;
            else if constexpr (ACTC == 1) {
                if constexpr (sizeof(E) == 2) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
                else return silu_f(x);
            } else return apply_act(x, a.act);
        };
        auto finish = [&](const uint4& raw, long p, const float (&sc)[VEC], const float (&sh)[VEC]) {
            float v[VEC];
            Vec16<E>::load(reinterpret_cast<const E*>(&raw), v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float A = ga[e], B = be[e];
                if (a.film) {
                    const float m = 1.f + sc[e];
                    A *= m;
                    B = B * m + sh[e];
                }
                v[e] = act1(fmaf(v[e], A, B));
            }
            Vec16<E>::store(y + p * a.yLd + c0, v);
        };
        long qq = threadIdx.x / pp;
        for (; qq + rows < pixPerStat; qq += 2 * rows) {
            float sc0[VEC] = {}, sh0[VEC] = {}, sc1[VEC] = {}, sh1[VEC] = {};
            if (a.film) {
                film_load(pix0 + qq, sc0, sh0);
                film_load(pix0 + qq + rows, sc1, sh1);
            }
            finish(cache[qq * pp + k], pix0 + qq, sc0, sh0);
            finish(cache[(qq + rows) * pp + k], pix0 + qq + rows, sc1, sh1);
        }
        for (; qq < pixPerStat; qq += rows) {
            float sc0[VEC] = {}, sh0[VEC] = {};
            if (a.film) film_load(pix0 + qq, sc0, sh0);
            finish(cache[qq * pp + k], pix0 + qq, sc0, sh0);
        }
    };
    if (a.act == FLAIR_ACT_NONE) run(std::integral_constant<int, 0>{});
    else if (a.act == FLAIR_ACT_SILU) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 2>{});
}

}  // namespace

// Workgroups per statistic: ~16 pixels per thread row, so small tensors still spread over many CUs.
static int gn_blocks_per_stat(const flair_gn_params* p) {
    const int vec = p->dtype == FLAIR_BF16 ? 8 : 4;
    const int cv = p->C / vec > 0 ? p->C / vec : 1;
    const int rows = 256 / cv > 0 ? 256 / cv : 1;
    const long pix = (long)p->frames_per_stat * p->H * p->W;
    long bps = (pix + (long)rows * 16 - 1) / ((long)rows * 16);
    // cap of statistics workgroups per statistic (FLAIR_GN_PARTIAL_BLOCKS; 1024 until round 3: 512 streams as fast with half the
    // partials for gn_finalize: whole norm 75.6 -> 73.3 us on 16x256^2x64, 44.8 -> 41.8 us on 16x128^2x128; 256: equal, 128: slower)
    static const int cap = [] {
        const char* e = getenv("FLAIR_GN_PARTIAL_BLOCKS");
        const int v = e ? atoi(e) : 512;
        return v >= 1 && v <= 4096 ? v : 512;          // an unparsable or out-of-range value keeps the default
    }();
    if (bps > cap) bps = cap;
    if (bps < 1) bps = 1;
    return (int)bps;
}

extern "C" size_t flair_groupnorm_workspace_bytes(const flair_gn_params* p) {
    if (!p) return 0;
    const int nstat = p->F / p->frames_per_stat;
    const int bps = gn_blocks_per_stat(p);
    // partials [nstat*bps][2][groups] + stats [nstat][groups][2]
    return ((size_t)nstat * bps * 2 * p->groups + (size_t)nstat * p->groups * 2) * sizeof(float);
}

extern "C" int flair_groupnorm_nhwc(const flair_gn_params* p, const void* x0, const void* x1,
                                    const float* gamma, const float* beta, const float* film,
                                    void* y, void* raw, void* workspace, hipStream_t stream) {
    FLAIR_CHECK(p && x0 && gamma && beta && y && workspace, "flair_groupnorm_nhwc: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_groupnorm_nhwc: bad dtype");
    const int vec = p->dtype == FLAIR_BF16 ? 8 : 4;
    const int C = p->C;
    FLAIR_CHECK(C > 0 && C % vec == 0 && C % p->groups == 0, "flair_groupnorm_nhwc: C=%d groups=%d", C, p->groups);
    FLAIR_CHECK(p->c0 > 0 && p->c0 <= C && p->c0 % vec == 0 && (p->c0 == C || x1),
                "flair_groupnorm_nhwc: bad segment split c0=%d", p->c0);
    FLAIR_CHECK(p->F > 0 && p->frames_per_stat > 0 && p->F % p->frames_per_stat == 0,
                "flair_groupnorm_nhwc: frames %d / frames_per_stat %d", p->F, p->frames_per_stat);
    FLAIR_CHECK(p->resample >= 0 && p->resample <= 2, "flair_groupnorm_nhwc: resample mode");
    FLAIR_CHECK(p->resample != 1 || (p->H % 2 == 0 && p->W % 2 == 0), "flair_groupnorm_nhwc: odd size for pooling");
    const int cv = C / vec;
    FLAIR_CHECK(cv <= 512, "flair_groupnorm_nhwc: C=%d too wide", C);
    const int rows = cv <= 256 ? 256 / cv : 1;      // wide tensors: one pixel row per workgroup of cv threads
    const int threads = rows * cv;
    const int nstat = p->F / p->frames_per_stat;
    const long pix = (long)p->frames_per_stat * p->H * p->W;
    const int bps = gn_blocks_per_stat(p);
    FLAIR_CHECK(!film || p->film_ld >= 2 * C, "flair_groupnorm_nhwc: film_ld");
    float* part = reinterpret_cast<float*>(workspace);
    float* stats = part + (size_t)nstat * bps * 2 * p->groups;

    GnSrc s;
    s.x[0] = x0; s.x[1] = x1;
    s.c[0] = p->c0; s.c[1] = C - p->c0;
    s.ld[0] = p->ld0; s.ld[1] = p->ld1;
    {   // small tensors: one launch, one workgroup per (statistic, group)
        static const long fusedMax = getenv("FLAIR_GN_FUSED_MAX") ? atol(getenv("FLAIR_GN_FUSED_MAX")) : (128l << 10);
        const int cpg = C / p->groups;
        const int esz = p->dtype == FLAIR_BF16 ? 2 : 4;
        const long groupBytes = pix * cpg * esz;                  // one group's data
        if (p->resample == 0 && !raw && cpg % vec == 0 && 1024 % (cpg / vec) == 0 && p->c0 % cpg == 0 && groupBytes <= fusedMax && groupBytes <= (128l << 10) &&
            (long)nstat * p->groups >= 16) {
            GnApply a;
            a.s = s;
            a.C = C; a.groups = p->groups;
            a.F = p->F; a.H = p->H; a.W = p->W;
            a.framesPerStat = p->frames_per_stat;
            a.stats = nullptr; a.gamma = gamma; a.beta = beta; a.film = film; a.filmLd = p->film_ld;
            a.act = p->act; a.resample = 0;
            a.y = y; a.yLd = p->y_ld; a.raw = nullptr; a.rawLd = 0; a.blocksPerFrame = 0;
            a.filmVec = film && p->film_ld % 4 == 0 && (reinterpret_cast<uintptr_t>(film) & 15) == 0;
            static LdsAttrOnce attr0, attr1;
            {
                const hipError_t e0 = flair_max_lds_once(attr0, reinterpret_cast<const void*>(&gn_group_kernel<bf16_t>), 128 * 1024);
                const hipError_t e1 = flair_max_lds_once(attr1, reinterpret_cast<const void*>(&gn_group_kernel<float>), 128 * 1024);
                FLAIR_CHECK(e0 == hipSuccess && e1 == hipSuccess, "flair_groupnorm_nhwc: hipFuncSetAttribute failed");
            }
            const int grid = nstat * p->groups;
            if (p->dtype == FLAIR_BF16)
                hipLaunchKernelGGL(gn_group_kernel<bf16_t>, dim3(grid), dim3(1024), (size_t)groupBytes, stream, a, p->eps);
            else
                hipLaunchKernelGGL(gn_group_kernel<float>, dim3(grid), dim3(1024), (size_t)groupBytes, stream, a, p->eps);
            FLAIR_LAUNCH_CHECK();
            return FLAIR_OK;
        }
    }
    const size_t lds = (size_t)2 * rows * C * sizeof(float);
    if (p->dtype == FLAIR_BF16)
        hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, dim3(nstat * bps), dim3(threads), lds, stream, s, C, p->groups, pix, bps,
                           part);
    else
        hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(nstat * bps), dim3(threads), lds, stream, s, C, p->groups, pix, bps,
                           part);
    FLAIR_LAUNCH_CHECK();
    const int sg = nstat * p->groups;
    FLAIR_CHECK(sg % 4 == 0 || sg < 4, "flair_groupnorm_nhwc: groups");
    const int wavesPerBlock = sg >= 4 ? 4 : sg;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(sg / wavesPerBlock), dim3(64 * wavesPerBlock), 0, stream,
                       part, p->groups, C / p->groups, bps, pix, p->eps, stats);
    FLAIR_LAUNCH_CHECK();

    GnApply a;
    a.s = s;
    a.C = C; a.groups = p->groups;
    a.F = p->F; a.H = p->H; a.W = p->W;
    a.framesPerStat = p->frames_per_stat;
    a.stats = stats; a.gamma = gamma; a.beta = beta; a.film = film; a.filmLd = p->film_ld;
    a.act = p->act; a.resample = p->resample;
    a.y = y; a.yLd = p->y_ld; a.raw = raw; a.rawLd = p->raw_ld;
    const long outPix = p->resample == 1 ? (long)p->H * p->W / 4 : (long)p->H * p->W;
    int bpf = (int)((outPix + (long)rows * 4 - 1) / ((long)rows * 4));
    // ~4 workgroups per CU, each streaming a long pixel range: measured on the 16x256^2x64 / 16x128^2x128 clip
    // tensors 4096 blocks 105 / 101 us, 1024 blocks 91 / 63 us for the whole norm (tools/bench_gn.py, r02)
    static const int totalBlocks = getenv("FLAIR_GN_BLOCKS") ? atoi(getenv("FLAIR_GN_BLOCKS")) : 1024;
    const int maxBpf = totalBlocks / p->F > 0 ? totalBlocks / p->F : 1;
    if (bpf > maxBpf) bpf = maxBpf;
    if (bpf < 1) bpf = 1;
    a.blocksPerFrame = bpf;
    if (p->dtype == FLAIR_BF16)
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(p->F * bpf), dim3(threads), 0, stream, a);
    else
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(p->F * bpf), dim3(threads), 0, stream, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
