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
//   raw  : [F][H][W][rawLd]  conv_offset output, 27*G channels in TAP-MAJOR order (the host
//          permutes the output channels of the last conv_offset convolution when it packs
//          its weights, which is free): tap k owns the 3*G contiguous channels [3*G*k, 3*G*(k+1)),
//            raw[3*G*k + 2*g + {0,1}] = (dy,dx) pre-activation of group g, tap k
//            raw[3*G*k + 2*G + g]     = mask pre-activation
//          (reference order: 2*(g*9+k)+{0,1} and 18*G + g*9 + k), so one tap of one pixel is a
//          single 96-byte run (bf16) and consecutive taps share cache lines
//   dy,dx(g,k) = M*tanh(raw..) + flow_g.{y,x};  flow_g = flow1 (g < G/2) else flow2
//   m(g,k)     = sigmoid(raw..)
//   Y[p][co]   = bias[co] + sum_{k,ci} Wt[co][k][ci] * m(g(ci),k) * bilinear(X[ci], p + pk + d(g(ci),k))
//
// Same MFMA operand roles as conv.hip (weights = A operand, gathered pixels = B operand,
// XOR-swizzled LDS rows); the activation staging is replaced by a 4-corner NHWC gather of
// 16-byte channel chunks (one deformable group = 8 or 16 contiguous channels).

#include <stdlib.h>

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
    int activated;       // raw holds finished residues / masks (FLAIR_ACT_DCN_OFFSETS upstream)
    long P;
    unsigned xBytes[2], rawBytes, wBytes;
};

template <typename E> struct MmaD;
template <> struct MmaD<bf16_t> {
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * i + half; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[0]),
                                                      __builtin_bit_cast(bf16x8, b[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[1]),
                                                      __builtin_bit_cast(bf16x8, b[1]), acc, 0, 0, 0);
    }
};
template <> struct MmaD<float> {
    static __device__ __forceinline__ int chunk(int i, int half) { return 2 * half + i; }
    static __device__ __forceinline__ void run(const uint4 (&a)[2], const uint4 (&b)[2], f32x16& acc) {
        const float* af = reinterpret_cast<const float*>(a);
        const float* bf = reinterpret_cast<const float*>(b);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
    }
};

// LDS tile rows hold one K step (CPR 16-byte chunks) of one cout / pixel; the chunk index is
// XORed with a row-derived value so that 16 consecutive rows read the same logical chunk from
// 16 different bank groups (conflict-free ds_read_b128) for 64/128/256-byte rows.
template <int CPR>
__device__ __forceinline__ int tile_off(int row, int chunk) {
    const int s = CPR == 4 ? (row >> 2) & 3 : CPR == 8 ? (row >> 1) & 7 : row & 15;
    return row * (CPR * 16) + ((chunk ^ s) << 4);
}

// full-rate signed 24-bit multiply (the compiler turns __mul24 into the quarter-rate v_mul_lo_u32 here)
__device__ __forceinline__ int mul_i24(int a, int b) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// (lo, hi) -> packed bf16 pair, round-to-nearest-even; as inline asm so that the vectoriser cannot regroup the
// conversions by accumulator pair and re-interleave the halves afterwards (4 extra VALU per 8 channels)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(e^{2x}+1); |err| ~ 1e-7 relative, saturates cleanly for large |x|
    const float e = __expf(2.f * x);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);   // v_rcp_f32 (1 ulp) instead of an IEEE divide
}

// Workgroup = TP = 32*NPF consecutive pixels of one frame x ALL output channels (NCF
// fragments of 32), so every bilinear gather is done exactly once; TPP threads share one
// pixel, each owning one 16-byte channel chunk of the K step (one tap, TPP*16 bytes of
// channels).  A per-frame call only has H*W*TPP threads of gather parallelism, so the tile
// shapes are picked to keep >= 2048 waves in flight at 256^2 / 128^2 while the weight tile
// (re-read by every workgroup) is amortised over as many pixels as possible.
//
// Per tap, the 3*G raw offset/mask values of the TP pixels are staged in LDS (double
// buffered, fetched one tap ahead).  Per K step every thread turns its group's staged values
// into 4 corner addresses and issues 4 unconditional buffer loads (out-of-range corners read
// 0) plus its share of the weight tile.  Two register sets alternate, so the loads of step
// k+1 are in flight while step k is blended, written to LDS and multiplied on the matrix cores.
// ONEFRAME (F == 1 and H*W a multiple of the pixel tile: every per-frame call of the recurrence): the feature
// buffer resources cover exactly one frame, so rows above / below the image fall outside the resource and read 0
// by the hardware range check; columns left / right of it get weight 0.  That removes the per-corner address
// selects of the general path (the kernel is VALU-issue bound: ~75 % VALU-active, profiles/r02b_dcn_pmc_sq.txt).
// DOT2 (bf16, round 4): the four-corner blend on v_dot2c_f32_bf16 -- corner pairs of one channel packed by v_perm_b32 against the
// pair of bilinear weights rounded to bf16 (f32 accumulation): 38 vector instructions per 8 channels instead of 52 (32 bf16 -> f32
// unpacks + 16 v_pk_fma_f32, which issue at 1.4x the cost of a plain instruction: tools/probes/valu_rate_probe.hip).  The weights lose
// 16 mantissa bits: |error| <= 2^-9 * sum_k |w_k v_k| <= 2^-9 max|v| per blended value, the size of the bf16 rounding the value gets
// anyway when it is staged for the MFMA; f32 tensors keep the exact path.
// PFD = gather register sets in flight (2: the loads of step k+1 fly while step k is blended; 3: also step k+2 --
// for the c = 128 tiles, where only 8 wavefronts per CU exist to hide the gather round trip; 1 (round 4, the c = 64 per-frame form): the
// loads of a step are waited for in the same step, at 64 VGPRs, and a SECOND workgroup on the CU hides them).
template <typename E, int NCF, int NPF, int TPP, bool ACTIVATED, bool ONEFRAME, int PFD, bool DOT2>
__global__ __launch_bounds__(32 * NPF * TPP, (PFD == 1 ? 8 : NCF >= 4 ? 32 * NPF * TPP / 256 : 4)) void dcn_kernel(DcnArgs a) {
    prefetch_kernargs<sizeof(DcnArgs)>();
    constexpr int NT = 32 * NPF * TPP, NW = NT / 64;
    constexpr int TP = 32 * NPF, TC = 32 * NCF, CPR = TPP;
    constexpr int VEC = ET<E>::VEC;
    constexpr unsigned ESZ = sizeof(E);
    constexpr int BKE = CPR * VEC;                            // K elements per step
    // (cout-frag, pixel-frag) pairs per wave; with more waves than pairs, KSPLIT waves share a pair and
    // each multiplies its slice of the step's 64-byte K sub-steps (summed through LDS at the end)
    constexpr int NPAIR = NCF * NPF;
    constexpr int KSPLIT = NW > NPAIR ? NW / NPAIR : 1;
    constexpr int PAIRS = NW > NPAIR ? 1 : NPAIR / NW;
    constexpr int KSUB = CPR / 4 / KSPLIT;                    // K sub-steps per wave and step
    static_assert((NW > NPAIR ? NPAIR * KSPLIT == NW : PAIRS * NW == NPAIR) && KSUB >= 1 && KSUB * KSPLIT * 4 == CPR,
                  "tile shape");
    constexpr int WR = (TC * CPR + NT - 1) / NT;              // weight pieces per thread
    constexpr int RAWR = (TP * 3 * (int)ESZ + NT - 1) / NT;   // raw pieces per thread and tap (G <= 16)
    constexpr int BUF = (TC + TP) * CPR * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = a.G;
    const int slabPieces = 3 * G * (int)ESZ / 16;
    const int rawPitch = 3 * G * (int)ESZ + 16;               // +16 B: spreads pixels over banks
    char* sraw = smem;
    char* stile = smem + 2 * TP * rawPitch;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // each XCD (own L2) works on one contiguous eighth of the frame: the gathered features of
    // its tiles stay in that L2 instead of every L2 fetching the whole frame
    const long p0 = (long)xcd_remap(blockIdx.x, gridDim.x) * TP;
    const int q = tid % TPP, srow = tid / TPP;
    long p = p0 + srow;
    const bool pvalid = p < a.P;
    if (!pvalid) p = a.P - 1;
    const int pw = (int)(p % a.W);
    const int ph = (int)((p / a.W) % a.H);
    // ONEFRAME: the tile lies in one frame, taken from the block-uniform tile origin so that the per-frame buffer
    // resources below are provably uniform (a per-thread frame index costs a waterfall loop around every load)
    const int pf = (int)((ONEFRAME ? p0 : p) / ((long)a.W * a.H));
    float2 fl1 = make_float2(0.f, 0.f), fl2 = make_float2(0.f, 0.f);
    if (a.flow1) fl1 = *reinterpret_cast<const float2*>(a.flow1 + p * 2);
    if (a.flow2) fl2 = *reinterpret_cast<const float2*>(a.flow2 + p * 2);

    // ---- per-tap staging of the raw conv_offset values.  Runs in the post-barrier (MFMA)
    // phase of a K step, where the consumed gather registers are dead, so it adds nothing to
    // the register peak; the mapping is recomputed per tap (once every cbPerTap steps).
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.raw, a.rawBytes);
    auto raw_issue = [&](int tap, uint4 (&reg)[RAWR]) {
#pragma unroll
        for (int j = 0; j < RAWR; ++j) {
            const int id = tid + j * NT;
            const int px = id / slabPieces, pc = id - px * slabPieces;
            const long pp = p0 + px;
            const unsigned within = (unsigned)(3 * tap * G) * ESZ + pc * 16;
            reg[j] = buf_load16(rr, px < TP && pp < a.P ? (unsigned)(pp * a.rawLd * ESZ) + within : FLAIR_OOB);
        }
    };
    auto raw_write = [&](int tap, const uint4 (&reg)[RAWR]) {
#pragma unroll
        for (int j = 0; j < RAWR; ++j) {
            const int id = tid + j * NT;
            const int px = id / slabPieces, pc = id - px * slabPieces;
            if (px < TP) *reinterpret_cast<uint4*>(sraw + ((tap & 1) * TP + px) * rawPitch + pc * 16) = reg[j];
        }
    };
    auto stage_raw = [&](int tap) {
        uint4 reg[RAWR];
        raw_issue(tap, reg);
        raw_write(tap, reg);
    };

    const int cpg = a.Cin / G;
    int cpgShift = 0;                       // cpg is a power of two (8 or 16 channels per group)
    while ((1 << cpgShift) < cpg) ++cpgShift;
    const int halfC = a.Cin / 2;
    const int cbPerTap = a.Cin / BKE;       // even: both input halves are multiples of BKE
    const int nk = 9 * cbPerTap;
    // ONEFRAME: xBytes is the size of ONE frame, and the tile's frame is block-uniform (H*W % TP == 0)
    const size_t fo0 = ONEFRAME ? (size_t)pf * a.H * a.W * a.xLd[0] * ESZ : 0, fo1 = ONEFRAME ? (size_t)pf * a.H * a.W * a.xLd[1] * ESZ : 0;
    const __amdgpu_buffer_rsrc_t xr0 = make_rsrc(reinterpret_cast<const char*>(a.x[0]) + fo0, a.xBytes[0]);
    const __amdgpu_buffer_rsrc_t xr1 = make_rsrc(reinterpret_cast<const char*>(a.x[1]) + fo1, a.xBytes[1]);
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.w, a.wBytes);
    const unsigned frameBase = (unsigned)pf * a.H * a.W;

    struct Regs {
        uint4 c[4];
        uint4 w[WR];
        float wt[4];
    } rs[PFD];

    // per-thread part of the weight offsets (fixed over the K loop)
    unsigned wrow[WR];
#pragma unroll
    for (int j = 0; j < WR; ++j) {
        const int id = tid + j * NT;
        const int n = id / CPR, ch = id % CPR;
        wrow[j] = n < a.Cout && n < TC ? (unsigned)(n * 9 * a.Cin + ch * VEC) * ESZ : FLAIR_OOB;
    }
    int itap = 0, icb = 0, ikh = 0, ikw = 0;   // K-loop position of the NEXT issue (block uniform)
    const int rawRow = srow * rawPitch;        // this thread's pixel inside a staged raw slab
    const float fH = (float)a.H, fW = (float)a.W;
    const float phm1 = (float)(ph - 1), pwm1 = (float)(pw - 1);
    auto issue = [&](Regs& r) {
        const int tap = itap, cb = icb;
        const int c = cb * BKE + q * VEC;
        const int g = c >> cpgShift;
        const E* myraw = reinterpret_cast<const E*>(sraw + (tap & 1) * TP * rawPitch + rawRow);
        const float ry = ET<E>::ld(myraw + 2 * g), rx = ET<E>::ld(myraw + 2 * g + 1);
        const float rm = ET<E>::ld(myraw + 2 * G + g);
        // groups below G/2 take flow1, the others flow2; a K step lies in one input half (halfC % BKE == 0), so this is the
        // block-uniform `second` below
        const bool second = cb * BKE >= halfC;                // block-uniform (halfC % BKE == 0)
        const float2 fl = second ? fl2 : fl1;
        // ACTIVATED (compile time): the residues and the mask are used as stored, no transcendental per group
        float sy = phm1 + (float)ikh + fl.y, sx = pwm1 + (float)ikw + fl.x, mk;
        if constexpr (ACTIVATED) {
            sy += ry;
            sx += rx;
            mk = rm;
        } else {
            sy += a.maxMag * fast_tanh(ry);
            sx += a.maxMag * fast_tanh(rx);
            mk = __builtin_amdgcn_rcpf(1.f + __expf(-rm));
        }
        const float fy = floorf(sy), fx = floorf(sx);
        const float ay = sy - fy, ax = sx - fx;
        // clamp before the int conversion so wild offsets cannot overflow (one v_med3 each)
        const int y0 = (int)__builtin_amdgcn_fmed3f(fy, -2.f, fH), x0 = (int)__builtin_amdgcn_fmed3f(fx, -2.f, fW);
        const unsigned ld = (unsigned)(second ? a.xLd[1] : a.xLd[0]) * ESZ;
        const unsigned coff = (unsigned)(c - (second ? halfC : 0)) * ESZ;
        const __amdgpu_buffer_rsrc_t xr = second ? xr1 : xr0;
        const unsigned rowStep = (unsigned)a.W * ld;
        // one multiply for the top-left corner, the other three are adds; SIGNED 24-bit multiplies are full rate
        // (v_mul_lo_u32 is quarter rate) and keep the arithmetic consistent mod 2^32 when the top-left corner is
        // outside the image while a neighbour is inside: |pixel index| < 2^23 and byte strides < 2^23 (host check)
        if constexpr (ONEFRAME) {
            const float wx0 = (unsigned)x0 < (unsigned)a.W ? (1.f - ax) * mk : 0.f;
            const float wx1 = (unsigned)(x0 + 1) < (unsigned)a.W ? ax * mk : 0.f;
            r.wt[0] = (1.f - ay) * wx0;
            r.wt[1] = (1.f - ay) * wx1;
            r.wt[2] = ay * wx0;
            r.wt[3] = ay * wx1;
            const unsigned o00 =
                (unsigned)mul_i24(mul_i24(y0, a.W) + x0, (int)ld) + coff;
            r.c[0] = buf_load16(xr, o00);
            r.c[1] = buf_load16(xr, o00 + ld);
            r.c[2] = buf_load16(xr, o00 + rowStep);
            r.c[3] = buf_load16(xr, o00 + rowStep + ld);
        } else {
            r.wt[0] = (1.f - ay) * (1.f - ax) * mk;
            r.wt[1] = (1.f - ay) * ax * mk;
            r.wt[2] = ay * (1.f - ax) * mk;
            r.wt[3] = ay * ax * mk;
            const unsigned o00 = (unsigned)mul_i24(
                                     (int)frameBase + mul_i24(y0, a.W) + x0, (int)ld) + coff;
            const bool yok0 = pvalid && (unsigned)y0 < (unsigned)a.H, yok1 = pvalid && (unsigned)(y0 + 1) < (unsigned)a.H;
            const bool xok0 = (unsigned)x0 < (unsigned)a.W, xok1 = (unsigned)(x0 + 1) < (unsigned)a.W;
            r.c[0] = buf_load16(xr, yok0 && xok0 ? o00 : FLAIR_OOB);
            r.c[1] = buf_load16(xr, yok0 && xok1 ? o00 + ld : FLAIR_OOB);
            r.c[2] = buf_load16(xr, yok1 && xok0 ? o00 + rowStep : FLAIR_OOB);
            r.c[3] = buf_load16(xr, yok1 && xok1 ? o00 + rowStep + ld : FLAIR_OOB);
        }
        const unsigned wk = (unsigned)(tap * a.Cin + cb * BKE) * ESZ;      // block-uniform part of the weight offset
#pragma unroll
        for (int j = 0; j < WR; ++j) r.w[j] = buf_load16(wrs, wrow[j] == FLAIR_OOB ? FLAIR_OOB : wrow[j] + wk);
        if (++icb == cbPerTap) {
            icb = 0;
            ++itap;
            if (++ikw == 3) {
                ikw = 0;
                ++ikh;
            }
        }
    };
    auto blend_and_stage = [&](const Regs& r, int buf) {
        char* base = stile + buf * BUF;
        uint4 outv;
        if constexpr (sizeof(E) == 2 && DOT2) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const bf16x2_t w01 = __builtin_bit_cast(bf16x2_t, cvt_pk_bf16(r.wt[0], r.wt[1]));
            const bf16x2_t w23 = __builtin_bit_cast(bf16x2_t, cvt_pk_bf16(r.wt[2], r.wt[3]));
            unsigned o[4];
            const unsigned c0[4] = {r.c[0].x, r.c[0].y, r.c[0].z, r.c[0].w}, c1[4] = {r.c[1].x, r.c[1].y, r.c[1].z, r.c[1].w};
            const unsigned c2[4] = {r.c[2].x, r.c[2].y, r.c[2].z, r.c[2].w}, c3[4] = {r.c[3].x, r.c[3].y, r.c[3].z, r.c[3].w};
            float lo[4], hi[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // (even channel of corner a | even channel of corner b), (odd | odd): one v_perm_b32 each
                const bf16x2_t tlo = __builtin_bit_cast(bf16x2_t, __builtin_amdgcn_perm(c1[d], c0[d], 0x05040100u));
                const bf16x2_t thi = __builtin_bit_cast(bf16x2_t, __builtin_amdgcn_perm(c1[d], c0[d], 0x07060302u));
                const bf16x2_t blo = __builtin_bit_cast(bf16x2_t, __builtin_amdgcn_perm(c3[d], c2[d], 0x05040100u));
                const bf16x2_t bhi = __builtin_bit_cast(bf16x2_t, __builtin_amdgcn_perm(c3[d], c2[d], 0x07060302u));
                lo[d] = __builtin_amdgcn_fdot2_f32_bf16(tlo, w01, 0.f, false);
                hi[d] = __builtin_amdgcn_fdot2_f32_bf16(thi, w01, 0.f, false);
                lo[d] = __builtin_amdgcn_fdot2_f32_bf16(blo, w23, lo[d], false);
                hi[d] = __builtin_amdgcn_fdot2_f32_bf16(bhi, w23, hi[d], false);
            }
            // HAZARD (gfx950, ROCm 7.2): a vector instruction other than another v_dot2c on the same accumulator that reads a
            // v_dot2c_f32_bf16 result within 3 wait states gets the OLD register value -- the hardware does not interlock and
            // hipcc inserts nothing (tools/probes/dot2_hazard.hip).  All eight sums pass through one statement that ends the
            // last dot and carries the wait states; the conversions below cannot be scheduled above it.
            asm volatile("s_nop 2" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = cvt_pk_bf16(lo[d], hi[d]);
            outv = make_uint4(o[0], o[1], o[2], o[3]);
        } else if constexpr (sizeof(E) == 2) {
            // even / odd channels of each bf16 pair accumulate separately, so the result packs with one
            // v_cvt_pk_bf16_f32 per dword (no re-interleaving)
            // (two-wide vectors: v_pk_fma_f32, one instruction per two channels)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 lo[2] = {{0.f, 0.f}, {0.f, 0.f}}, hi[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned d[4] = {r.c[i].x, r.c[i].y, r.c[i].z, r.c[i].w};
                const f32x2 w = {r.wt[i], r.wt[i]};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x2 vl = {__uint_as_float(d[2 * j] << 16), __uint_as_float(d[2 * j + 1] << 16)};
                    const f32x2 vh = {__uint_as_float(d[2 * j] & 0xffff0000u), __uint_as_float(d[2 * j + 1] & 0xffff0000u)};
                    lo[j] = __builtin_elementwise_fma(w, vl, lo[j]);
                    hi[j] = __builtin_elementwise_fma(w, vh, hi[j]);
                }
            }
            outv = make_uint4(cvt_pk_bf16(lo[0].x, hi[0].x), cvt_pk_bf16(lo[0].y, hi[0].y), cvt_pk_bf16(lo[1].x, hi[1].x),
                              cvt_pk_bf16(lo[1].y, hi[1].y));
        } else {
            float acc[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[VEC];
                Vec16<E>::load(reinterpret_cast<const E*>(&r.c[i]), v);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fmaf(r.wt[i], v[e], acc[e]);
            }
            alignas(16) E out[VEC];
            Vec16<E>::store(out, acc);
            outv = *reinterpret_cast<const uint4*>(out);
        }
        *reinterpret_cast<uint4*>(base + TC * CPR * 16 + tile_off<CPR>(srow, q)) = outv;
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            const int id = tid + j * NT;
            if (id < TC * CPR) *reinterpret_cast<uint4*>(base + tile_off<CPR>(id / CPR, id % CPR)) = r.w[j];
        }
    };

    f32x16 acc[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // Tap T's raw values are staged in the post-barrier phase of step T*cbPerTap - (PFD+1): after the last reader of
    // that LDS buffer (tap T-2, issued PFD-1 steps ahead) and with one barrier to go before their first reader
    // (the issue for the first step of tap T, at the top of step T*cbPerTap - (PFD-1)).
    constexpr int LEAD = PFD + 1;
    int stageTap = 0;
    {   // taps 0 (and 1) before the loop: both requests in flight before either is written (one round trip, not two)
        uint4 r0[RAWR], r1[RAWR];
        const bool two = cbPerTap - LEAD < 0;
        raw_issue(0, r0);
        if (two) raw_issue(1, r1);
        raw_write(0, r0);
        if (two) raw_write(1, r1);
        stageTap = two ? 2 : 1;
    }
    int stageAt = stageTap * cbPerTap - LEAD;
    __syncthreads();
    auto mfma_step = [&](int buf) {
        const char* wb = stile + buf * BUF;
        const char* xb = wb + TC * CPR * 16;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int pair = KSPLIT > 1 ? wave % NPAIR : wave * PAIRS + i;
            const int kpart = KSPLIT > 1 ? wave / NPAIR : 0;
            const int pfrag = pair % NPF, cf = pair / NPF;
#pragma unroll
            for (int kk = 0; kk < KSUB; ++kk) {
                const int ks = kpart * KSUB + kk;
                uint4 af[2], bf[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ch = ks * 4 + MmaD<E>::chunk(h, lh);
                    bf[h] = *reinterpret_cast<const uint4*>(xb + tile_off<CPR>(pfrag * 32 + lr, ch));
                    af[h] = *reinterpret_cast<const uint4*>(wb + tile_off<CPR>(cf * 32 + lr, ch));
                }
                MmaD<E>::run(af, bf, acc[i]);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < PFD - 1; ++u)
        if (u < nk) issue(rs[u]);
    constexpr int UNR = 2 * PFD;                       // register sets cycle mod PFD, LDS tile buffers mod 2
    for (int k = 0; k < nk; k += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (k + u < nk) {                          // block-uniform
                if (k + u + PFD - 1 < nk) issue(rs[(u + PFD - 1) % PFD]);
                blend_and_stage(rs[u % PFD], u & 1);
                __syncthreads();
                mfma_step(u & 1);
                if (k + u == stageAt && stageTap < 9) {
                    stage_raw(stageTap);
                    ++stageTap;
                    stageAt += cbPerTap;
                }
            }
        }
    }
    if constexpr (KSPLIT > 1) {
        // sum the K slices of each pair: slices 1.. park their accumulators in LDS, slice 0 adds them
        __syncthreads();                                   // the last step's tile reads are done
        float* red = reinterpret_cast<float*>(stile);
        const int pair0 = wave % NPAIR, kpart0 = wave / NPAIR;
        if (kpart0 > 0) {
            float* dst = red + ((kpart0 - 1) * NPAIR + pair0) * 64 * 16 + lane * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<float4*>(dst + r) = make_float4(acc[0][r], acc[0][r + 1], acc[0][r + 2], acc[0][r + 3]);
        }
        __syncthreads();
        if (kpart0 > 0) return;
#pragma unroll
        for (int kp = 1; kp < KSPLIT; ++kp) {
            const float* src = red + ((kp - 1) * NPAIR + pair0) * 64 * 16 + lane * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + r);
                acc[0][r] += v.x; acc[0][r + 1] += v.y; acc[0][r + 2] += v.z; acc[0][r + 3] += v.w;
            }
        }
    }
    E* y = reinterpret_cast<E*>(a.y);
    if constexpr (sizeof(E) == 2) {
        // bf16, Cout % 8 == 0 and 16-byte aligned rows: the biases of all the wave's store groups are requested in one batch
        // (buffer loads: zero-sized descriptor without a bias, out-of-range offset past Cout -- loaded per element under
        // branches, every group of four was a serial round trip that also waited for the store before it), and
        // v_permlane32_swap pairs give each lane 8 consecutive couts of its pixel: 16-byte stores, half as many.
        if ((a.Cout & 7) == 0 && (a.yLd & 7) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0) {
            const __amdgpu_buffer_rsrc_t bd = make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
            const __amdgpu_buffer_rsrc_t yd = make_rsrc(a.y, (unsigned)a.P * a.yLd * 2u);      // < 2 GiB: host check
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                const int pair = KSPLIT > 1 ? wave % NPAIR : wave * PAIRS + i;
                const int pfrag = pair % NPF, cf = pair / NPF;
                const long po = p0 + pfrag * 32 + lr;              // lanes l and l + 32 share the pixel: the swaps pair lanes with equal predicates
                uint4 bq[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = cf * 32 + 16 * j + 8 * lh;
                    bq[j][0] = buf_load16(bd, co < a.Cout ? 4u * co : FLAIR_OOB);
                    bq[j][1] = buf_load16(bd, co < a.Cout ? 4u * co + 16u : FLAIR_OOB);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int co = cf * 32 + 16 * j + 8 * lh;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][8 * j + e]),
                                                                          __float_as_uint(acc[i][8 * j + 4 + e]), false, false);
                        v[e] = __uint_as_float(sw2[0]);
                        v[4 + e] = __uint_as_float(sw2[1]);
                    }
                    v[0] += __uint_as_float(bq[j][0].x); v[1] += __uint_as_float(bq[j][0].y);
                    v[2] += __uint_as_float(bq[j][0].z); v[3] += __uint_as_float(bq[j][0].w);
                    v[4] += __uint_as_float(bq[j][1].x); v[5] += __uint_as_float(bq[j][1].y);
                    v[6] += __uint_as_float(bq[j][1].z); v[7] += __uint_as_float(bq[j][1].w);
                    alignas(16) E out[8];
                    Vec16<E>::store(out, v);
                    const uint4 ov = *reinterpret_cast<const uint4*>(out);
                    // (a buffer store with an out-of-range offset for tail pixels / padding couts: under an EXEC-masked branch
                    // hipcc waits for the previous store's acknowledgement at the join)
                    const unsigned yo = po < a.P && co < a.Cout ? ((unsigned)po * a.yLd + co) * 2u : FLAIR_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{ov.x, ov.y, ov.z, ov.w}, yd, (int)yo, 0, 0);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int pair = KSPLIT > 1 ? wave % NPAIR : wave * PAIRS + i;
        const int pfrag = pair % NPF, cf = pair / NPF;
        const long po = p0 + pfrag * 32 + lr;
        if (po >= a.P) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = cf * 32 + 8 * g + 4 * lh;
            if (co >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g + e] + (a.bias ? a.bias[co + e] : 0.f);
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
}

}  // namespace

template <typename E, int NCF, int NPF, int TPP, bool ACTIVATED, bool ONEFRAME, bool DOT2, int PFDSEL = 0>
static int launch_dcn_v2(const DcnArgs& a0, hipStream_t stream) {
    constexpr int PFD = PFDSEL > 0 ? PFDSEL : NCF >= 4 && TPP <= 8 ? 3 : 2;
    DcnArgs a = a0;
    if (ONEFRAME) {      // resources cover one frame: rows outside the image fall outside the resource
        const unsigned long long half = a.Cin / 2, hw = (unsigned long long)a.H * a.W;
        a.xBytes[0] = (unsigned)(((hw - 1) * a.xLd[0] + half) * sizeof(E));
        a.xBytes[1] = (unsigned)(((hw - 1) * a.xLd[1] + half) * sizeof(E));
    }
    constexpr int NT = 32 * NPF * TPP, TP = 32 * NPF, TC = 32 * NCF;
    const int rawPitch = 3 * a.G * (int)sizeof(E) + 16;
    const size_t lds = (size_t)2 * TP * rawPitch + 2 * (TC + TP) * TPP * 16;
    static LdsAttrOnce attr;
    {
        const hipError_t e = flair_max_lds_once(attr, reinterpret_cast<const void*>(&dcn_kernel<E, NCF, NPF, TPP, ACTIVATED, ONEFRAME, PFD, DOT2>));
        FLAIR_CHECK(e == hipSuccess, "flair_dcn_align: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL((dcn_kernel<E, NCF, NPF, TPP, ACTIVATED, ONEFRAME, PFD, DOT2>), dim3(cdiv(a.P, TP)), dim3(NT), lds, stream, a);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}

// FLAIR_DCN_DOT2 = 0: the f32 blend for bf16 tensors too (A/B switch; the per-frame activated form of the recurrence only)
template <typename E, int NCF, int NPF, int TPP, bool ACTIVATED, bool ONEFRAME>
static int launch_dcn_v(const DcnArgs& a, hipStream_t stream) {
    if constexpr (sizeof(E) == 2 && ACTIVATED && ONEFRAME) {
        static const bool dot2 = !(getenv("FLAIR_DCN_DOT2") && atoi(getenv("FLAIR_DCN_DOT2")) == 0);
        if constexpr (NCF == 2 && NPF == 4) {
            // ONE gather register set (64 VGPRs instead of 106) so that TWO 16-wave workgroups share a CU (2 x 78 KB of LDS, 8 waves per SIMD): at
            // 106 VGPRs a 256^2 frame's 512 workgroups ran as two rounds of one workgroup per CU, every barrier idling the CU for the skew of
            // its 16 waves; now the other workgroup's waves fill the barrier skew and the gather round trips that the second register set hid.
            // Alignment family 7.86 / 7.98 -> 7.03 / 7.10 ms per step, step 78.07 / 77.61 -> 77.13 / 77.00 ms (same box).  FLAIR_DCN_PFD1=0: two sets.
            static const bool pfd1 = !(getenv("FLAIR_DCN_PFD1") && atoi(getenv("FLAIR_DCN_PFD1")) == 0);
            if (dot2 && pfd1) return launch_dcn_v2<E, NCF, NPF, TPP, ACTIVATED, ONEFRAME, true, 1>(a, stream);
        }
        if (dot2) return launch_dcn_v2<E, NCF, NPF, TPP, ACTIVATED, ONEFRAME, true>(a, stream);
    }
    return launch_dcn_v2<E, NCF, NPF, TPP, ACTIVATED, ONEFRAME, false>(a, stream);
}

template <typename E, int NCF, int NPF, int TPP>
static int launch_dcn(const DcnArgs& a, hipStream_t stream) {
    // the recurrence calls per frame with activated offsets; everything else takes the general kernel
    const bool oneframe = ((long)a.H * a.W) % (32 * NPF) == 0;
    if (a.activated && oneframe) return launch_dcn_v<E, NCF, NPF, TPP, true, true>(a, stream);
    if (a.activated) return launch_dcn_v<E, NCF, NPF, TPP, true, false>(a, stream);
    return launch_dcn_v<E, NCF, NPF, TPP, false, false>(a, stream);
}

extern "C" int flair_dcn_align(const flair_dcn_params* p, const void* x0, const void* x1, const void* raw,
                               const float* flow1, const float* flow2, const void* w, const float* bias, void* y,
                               hipStream_t stream) {
    FLAIR_CHECK(p && x0 && x1 && raw && w && y, "flair_dcn_align: null argument");
    FLAIR_CHECK(p->dtype == FLAIR_F32 || p->dtype == FLAIR_BF16, "flair_dcn_align: bad dtype");
    const int vec = p->dtype == FLAIR_BF16 ? 8 : 4;
    const int bke = vec * (p->Cout <= 64 && p->dtype != FLAIR_BF16 ? 4 : 8);      // K elements per step of the chosen tile
    const int esz = p->dtype == FLAIR_BF16 ? 2 : 4;
    const int cpg = p->G > 0 ? p->Cin / p->G : 0;
    FLAIR_CHECK(p->G > 0 && p->G % 8 == 0 && p->G <= 16 && p->Cin % p->G == 0 && cpg % vec == 0 &&
                    (cpg & (cpg - 1)) == 0 && (p->Cin / 2) % bke == 0,
                "flair_dcn_align: Cin=%d G=%d not supported", p->Cin, p->G);
    FLAIR_CHECK(p->Cout % 4 == 0 && p->Cout <= 128 && p->raw_ld >= 27 * p->G && (p->raw_ld * esz) % 16 == 0,
                "flair_dcn_align: Cout (<=128) / raw_ld");
    DcnArgs a;
    a.x[0] = x0; a.x[1] = x1; a.xLd[0] = p->x_ld[0]; a.xLd[1] = p->x_ld[1];
    a.Cin = p->Cin; a.raw = raw; a.rawLd = p->raw_ld; a.flow1 = flow1; a.flow2 = flow2;
    a.w = w; a.bias = bias; a.y = y; a.yLd = p->y_ld;
    a.F = p->F; a.H = p->H; a.W = p->W; a.Cout = p->Cout; a.G = p->G; a.maxMag = p->max_residue_magnitude;
    a.activated = p->raw_activated;
    a.P = (long)p->F * p->H * p->W;
    const unsigned long long half = p->Cin / 2;
    const unsigned long long b0 = ((unsigned long long)(a.P - 1) * p->x_ld[0] + half) * esz;
    const unsigned long long b1 = ((unsigned long long)(a.P - 1) * p->x_ld[1] + half) * esz;
    const unsigned long long br = ((unsigned long long)(a.P - 1) * p->raw_ld + 27 * p->G) * esz;
    const unsigned long long by = (unsigned long long)a.P * p->y_ld * esz;
    FLAIR_CHECK(b0 < 0x80000000ull && b1 < 0x80000000ull && br < 0x80000000ull && by < 0x80000000ull, "flair_dcn_align: tensor spans >= 2 GiB");
    FLAIR_CHECK(a.P + 2l * p->W + 2 < (1l << 23) && (long)p->x_ld[0] * esz < (1l << 23) && (long)p->x_ld[1] * esz < (1l << 23),
                "flair_dcn_align: F*H*W = %ld pixels (limit 2^23 per call: call per frame)", a.P);
    a.xBytes[0] = (unsigned)b0; a.xBytes[1] = (unsigned)b1; a.rawBytes = (unsigned)((br + 15) / 16 * 16);
    a.wBytes = (unsigned)((unsigned long long)p->Cout * 9 * p->Cin * esz);
    // Tile choice (measured on per-frame 256^2 / 128^2 calls, profiles/README.md).  c=64: 64 pixels x 64
    // couts with 8 threads per pixel, so one gather instruction fetches the whole 128-byte channel row of a
    // corner (two waves share each MFMA pair, K split); c=128: 32 pixels x 128 couts, 8 threads per pixel.
    if (p->dtype == FLAIR_BF16) {
        if (p->Cout > 64) {
            // c = 128: every workgroup streams the whole 590 KB weight matrix, so the pixel tile sets the weight traffic
            // (32-pixel tiles: 512 workgroups x 590 KB = 302 MB per 128x128 frame against 151 MB of gathers).
            // FLAIR_DCN_TILE_C128 = 64: 64-pixel tiles (8 waves, one workgroup per CU on a 128x128 frame) halve it.
            static const int tile128 = getenv("FLAIR_DCN_TILE_C128") ? atoi(getenv("FLAIR_DCN_TILE_C128")) : 1664;
            // 1664 (default since round 4): 64-pixel tiles with SIXTEEN threads per pixel -- one K step = 128 channels = one input half
            // of one tap (18 steps instead of 36), 16 waves per workgroup, one workgroup per CU on a 128x128 frame: twice the waves
            // in flight per CU, half the barriers, half the weight traffic of the 32-pixel tiles: 43.0 -> 32.7 us per launch
            // (tools/bench_dcn.py, same box).  32 / 64: the round-2/3 forms.
            if (tile128 == 1664 && a.P >= 64 * 256 && (p->Cin / 2) % 128 == 0) return launch_dcn<bf16_t, 4, 2, 16>(a, stream);
            if (tile128 >= 64 && a.P >= 64 * 256) return launch_dcn<bf16_t, 4, 2, 8>(a, stream);
            return launch_dcn<bf16_t, 4, 1, 8>(a, stream);
        }
        // 64-pixel tiles need 47 KB of LDS: 3 workgroups per CU, so a 256x256 frame (1024 tiles) runs as 768 + 256
        // (a second, one-third-full round).  128-pixel tiles (78 KB, 2 per CU, 16 wavefronts each) make it one full round.
        static const int wide = getenv("FLAIR_DCN_TILE") ? atoi(getenv("FLAIR_DCN_TILE")) : 128;
        return (wide >= 128 && a.P >= 128 * 512) ? launch_dcn<bf16_t, 2, 4, 8>(a, stream) : launch_dcn<bf16_t, 2, 2, 8>(a, stream);
    }
    return p->Cout <= 64 ? launch_dcn<float, 2, 2, 4>(a, stream) : launch_dcn<float, 4, 1, 8>(a, stream);
}
