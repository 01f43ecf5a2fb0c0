// Shared device/host helpers for the FLAIR MI355X (gfx950) kernels.
// Everything here is written for CDNA4 only: 64-wide wavefronts, MFMA, 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/flair_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

typedef uint16_t bf16_t;  // storage type of a bfloat16 element in HBM / LDS

#define FLAIR_WAVE 64

// ---- error plumbing (thread-local message, negative return codes) -------------
void flair_set_error(const char* fmt, ...);
#define FLAIR_CHECK(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            flair_set_error(__VA_ARGS__); \
            return FLAIR_ERR_ARG;         \
        }                                 \
    } while (0)
#define FLAIR_LAUNCH_CHECK()                                                  \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            flair_set_error("HIP launch failed: %s", hipGetErrorString(e_));  \
            return FLAIR_ERR_HIP;                                             \
        }                                                                     \
    } while (0)

// ---- bf16 <-> f32 --------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// Plain cast: hipcc emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving).
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// Element traits: E = float or bf16_t.  VEC = elements per 16-byte access.
template <typename E> struct ET;
template <> struct ET<float> {
    static constexpr int VEC = 4;
    static constexpr int DT = FLAIR_F32;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ET<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = FLAIR_BF16;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of elements <-> VEC floats
template <typename E> struct Vec16;
template <> struct Vec16<float> {
    static __device__ __forceinline__ void load(const float* p, float* o) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* o) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct Vec16<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
        o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
        o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
        uint4 v;
        v.x = pack2bf(o[0], o[1]); v.y = pack2bf(o[2], o[3]);
        v.z = pack2bf(o[4], o[5]); v.w = pack2bf(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case FLAIR_ACT_RELU: return v > 0.f ? v : 0.f;
        case FLAIR_ACT_LRELU01: return v > 0.f ? v : 0.1f * v;
        case FLAIR_ACT_SILU: return silu_f(v);
        case FLAIR_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
        case FLAIR_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        default: return v;
    }
}

// FLAIR_ACT_DCN_OFFSETS on VEC consecutive channels starting at co (VEC divides 8, period is a multiple of 24, so
// all of them fall in the same class): residues mag * tanh(v) = mag * (1 - 2 / (e^{2v} + 1)), masks 1 / (1 + e^{-v}).
template <int N>
__device__ __forceinline__ void dcn_offset_act(float (&v)[N], int co, float mag, int period) {
    const bool residue = 3 * (co % period) < 2 * period;
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const float r = __builtin_amdgcn_rcpf(1.f + __expf(residue ? 2.f * v[e] : -v[e]));
        v[e] = residue ? mag * (1.f - 2.f * r) : r;
    }
}

// The same activation with per-lane constants (the N consecutive channels of a call share their class): out = A * rcp(1 + 2^(k v)) + B,
// residues: mag tanh(v) = mag - 2 mag / (1 + e^{2v}); masks: 1 / (1 + e^{-v}) -- three vector and two transcendental instructions
// per value instead of eight and two; `scale` (the convolution's out_scale) is folded into A and B.
template <int N>
__device__ __forceinline__ void dcn_offset_act_fast(float (&v)[N], int co, float mag, int period, float scale) {
    const bool residue = 3 * (co % period) < 2 * period;
    const float kk = residue ? 2.885390081777927f : -1.4426950408889634f;       // 2 log2(e) | -log2(e)
    const float A = residue ? -2.f * mag * scale : scale, B = residue ? mag * scale : 0.f;
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = fmaf(__builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(kk * v[e])), A, B);
}

// ---- buffer loads: out-of-range offsets return 0, which gives zero padding for free and keeps
// the loads unconditional (a select or branch on a load result makes hipcc wait vmcnt(0) at once).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define FLAIR_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byteOff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byteOff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (observed
// round-robin dispatch), so give each XCD a contiguous chunk of the logical grid.
// Bijective for any grid size (cdna_hip_programming.md section 5, "XCD swizzle").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Kernel-argument prefetch.  hipcc loads the by-value argument struct of a kernel (ConvArgs: 304 bytes = five 64-byte lines
// of the kernarg segment) lazily, one s_load + s_waitcnt at a time as the prologue needs the fields: three to five DEPENDENT
// scalar-cache misses, each a round trip to memory, before the first vector load can be issued.  A replayed forward is
// ~2 600 such launches per step.  This touches every line of the argument block in ONE batch (one round trip); the
// compiler's own loads then hit the scalar cache.  NBYTES = size of the explicit arguments (+ the implicit block behind them).
template <int NBYTES>
__device__ __forceinline__ void prefetch_kernargs() {
#ifdef FLAIR_NO_KARG_PREFETCH
    return;
#endif
    const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LINES = (NBYTES + 63) / 64 + 1;
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    if constexpr (LINES <= 2)
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1) : "s"(ka) : "memory");
    else if constexpr (LINES <= 4)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(ka) : "memory");
    else if constexpr (LINES <= 6)
        asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\ts_load_dword %3, %6, 0xc0\n\t"
                     "s_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5) : "s"(ka) : "memory");
    else
        asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
                     "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7) : "s"(ka) : "memory");
}

// One LDS-DMA wave instruction: 64 lanes x 16 bytes from `desc` (buffer descriptor words in SGPRs) at per-lane byte
// offset `voff` (out-of-range offsets deliver zeros) to LDS bytes [ldsAddr, ldsAddr + 1024).  Inline asm so that hipcc
// neither counts it in its own vmcnt bookkeeping nor waits for it ahead of unrelated ds_reads (it drains every LDS-DMA
// before the next LDS read it cannot tell apart from the DMA's target; cdna_hip_programming.md section 5, trap (a)): the
// kernel waits with its own `s_waitcnt vmcnt(0)` before the barrier that publishes a stage.  M0 is saved and restored.
__device__ __forceinline__ void dma16(u32x4_t desc, unsigned voff, unsigned ldsAddr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(ldsAddr), "s"(desc)
                 : "memory");
}

__device__ __forceinline__ u32x4_t make_desc(const void* base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    u32x4_t d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    d.z = __builtin_amdgcn_readfirstlane(bytes);
    d.w = 0x00020000u;
    return d;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are PER DEVICE in HIP: cache them per (kernel instantiation,
// device), not per process, so that a process which later runs on a second GPU configures that device too.  (Races between
// host threads are benign: the calls are idempotent.)
struct LdsAttrOnce {
    unsigned long long done = 0;      // bit d: device d configured (devices >= 64: configured on every call)
};
static inline hipError_t flair_max_lds_once(LdsAttrOnce& st, const void* fn, int bytes = 160 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && ((st.done >> dev) & 1ull)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) st.done |= 1ull << dev;
    return e;
}
static inline int flair_cu_count() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
