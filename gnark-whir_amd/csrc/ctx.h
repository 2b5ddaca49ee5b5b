// Library-internal context shared by the HIP translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstdio>
#include "../../include/mi355x_groth16.h"

struct DevBuf {             // growable device scratch owned by the ctx (no hipMalloc in the hot path
    void *p = nullptr;      // after warm-up: buffers only ever grow)
    size_t cap = 0;
};

struct mi_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    mi_stats stats{};
    hipEvent_t ev[24]{};
    // scratch
    alignas(16) unsigned char ntt_state[256];  // NttState (ntt.hip): root tables + plan knobs
    alignas(16) unsigned char msm_knobs[64];   // MsmKnobs (msm.hip)
    DevBuf ws[24];          // MSM / prove workspaces, see msm.hip / prove.hip
    int cu_count = 256;
};

#define MI_CHECK_HIP(ctx, call)                                                                       \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                          \
            return e__ == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP;                                  \
        }                                                                                             \
    } while (0)
#define MI_FAIL(ctx, code, msg)                                                                       \
    do {                                                                                              \
        (ctx)->err = (msg);                                                                           \
        return (code);                                                                                \
    } while (0)
#define MI_TRY(expr)                                                                                  \
    do {                                                                                              \
        int32_t rc__ = (expr);                                                                        \
        if (rc__ != MI_OK) return rc__;                                                               \
    } while (0)

// grow-only scratch
static inline int32_t mi_reserve(mi_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return MI_OK;
    if (b.p) { MI_CHECK_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    MI_CHECK_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return MI_OK;
}

// internal entry points implemented across translation units
int32_t mi_ntt_dev_impl(mi_ctx *ctx, mi_fr *inout_dev, uint32_t log_n, uint32_t flags);
int32_t mi_compute_h_dev_impl(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                              size_t n_constraints, mi_fr *h_out);
void mi_ntt_state_init(mi_ctx *ctx);
void mi_ntt_state_free(mi_ctx *ctx);
void mi_msm_state_init(mi_ctx *ctx);
// MSM returning the XYZZ result on the host (used by prove.hip); pts/scalars are device pointers
int32_t mi_msm_g1_xyzz(mi_ctx *ctx, const void *pts_dev, const void *scalars_dev, size_t n, uint32_t flags, void *out_xyzz_host);
int32_t mi_msm_g2_xyzz(mi_ctx *ctx, const void *pts_dev, const void *scalars_dev, size_t n, uint32_t flags, void *out_xyzz_host);
