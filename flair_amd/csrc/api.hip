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
extern "C" int flair_abi_version(void) { return 5; }

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
