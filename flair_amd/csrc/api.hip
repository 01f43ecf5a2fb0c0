// Error plumbing + version of the C ABI (include/flair_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void flair_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* flair_last_error(void) { return g_err; }
extern "C" int flair_abi_version(void) { return 6; }

// ---- multi-GPU start-up: the one collective of the path (replaces dist_util.py:40-79's pickled-chunk
// load_state_dict + per-parameter sync_params).  RCCL is bound at run time (dlopen of the librccl the process already has:
// a host that never distributes weights needs no RCCL at all).
#include <dlfcn.h>

typedef int (*flair_nccl_bcast_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);

extern "C" int flair_bcast_weights(void* blob, size_t bytes, int root, void* rccl_comm, hipStream_t stream) {
    FLAIR_CHECK(blob && bytes > 0 && rccl_comm && root >= 0, "flair_bcast_weights: bad argument (blob, bytes, root, communicator)");
    static flair_nccl_bcast_fn fn = nullptr;
    if (!fn) {
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);          // the copy the host process (e.g. torch) already loaded
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        FLAIR_CHECK(h, "flair_bcast_weights: librccl.so not found (%s)", dlerror());
        fn = reinterpret_cast<flair_nccl_bcast_fn>(dlsym(h, "ncclBroadcast"));
        FLAIR_CHECK(fn, "flair_bcast_weights: ncclBroadcast not found in librccl");
    }
    // one message per <= 512 MiB (xGMI links are point-to-point: RCCL pipelines a broadcast over its ring / tree per message)
    const size_t chunk = (size_t)512 << 20;
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = bytes - off < chunk ? bytes - off : chunk;
        char* p = reinterpret_cast<char*>(blob) + off;
        const int rc = fn(p, p, n, /*ncclUint8*/ 1, root, rccl_comm, stream);
        if (rc != 0) {
            flair_set_error("flair_bcast_weights: ncclBroadcast returned %d", rc);
            return FLAIR_ERR_HIP;
        }
    }
    return FLAIR_OK;
}

// ---- calibration launches (round 4): what THIS device sustains with nothing in the way, so that a roofline fraction against the
// data-sheet peaks (2.5 PFLOP/s at 2.4 GHz, 8 TB/s) can be read beside the rate the silicon holds under load.
// Measured on the pool (profiles/r04_attainable_peaks.txt): independent 32x32x16 bf16 MFMA chains out of registers reach 1.72 PFLOP/s in a
// 78 us launch (1.66 GHz) and 2.13 PFLOP/s sustained (2.04 GHz); a plain copy of 1 GiB moves 4.8-5.1 TB/s.
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
typedef __attribute__((ext_vector_type(16))) float probe_f32x16;

__global__ __launch_bounds__(256) void probe_matrix_kernel(float* sink, int iters) {
    probe_f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    probe_bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (float)((threadIdx.x + e) & 255));
        b[e] = (__bf16)(0.002f * (float)((threadIdx.x + 2 * e) & 255));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;          // never true: keeps the chains alive without a store per thread
}

__global__ __launch_bounds__(256) void probe_stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n, int mode) {
    unsigned sum = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (mode == 1) {
            dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
        } else {
            const uint4 v = src[i];
            if (mode == 2) dst[i] = v;
            else sum ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (mode == 0 && sum == 0x12345679u) dst[0] = make_uint4(sum, 0u, 0u, 0u);
}
}  // namespace

extern "C" int flair_probe_matrix_rate(int iters, int workgroups_per_cu, float* sink, double* flop_out, hipStream_t stream) {
    FLAIR_CHECK(iters > 0 && workgroups_per_cu >= 1 && workgroups_per_cu <= 8 && sink, "flair_probe_matrix_rate: bad argument");
    const int grid = flair_cu_count() * workgroups_per_cu;
    hipLaunchKernelGGL(probe_matrix_kernel, dim3(grid), dim3(256), 0, stream, sink, iters);
    FLAIR_LAUNCH_CHECK();
    if (flop_out) *flop_out = (double)grid * 4 /*waves*/ * iters * 16 /*MFMAs*/ * (2.0 * 32 * 32 * 16);
    return FLAIR_OK;
}

extern "C" int flair_probe_stream_rate(const void* src, void* dst, size_t bytes, int mode, hipStream_t stream) {
    FLAIR_CHECK(dst && (mode == 1 || src) && bytes >= 16 && mode >= 0 && mode <= 2, "flair_probe_stream_rate: bad argument (mode 0 read, 1 write, 2 copy)");
    FLAIR_CHECK((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0, "flair_probe_stream_rate: 16-byte aligned buffers");
    hipLaunchKernelGGL(probe_stream_kernel, dim3(flair_cu_count() * 8), dim3(256), 0, stream, reinterpret_cast<const uint4*>(src),
                       reinterpret_cast<uint4*>(dst), bytes / 16, mode);
    FLAIR_LAUNCH_CHECK();
    return FLAIR_OK;
}
