// Library-internal context shared by the HIP translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <functional>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include "../../include/mi355x_groth16.h"
#include "../../include/mi355x_groth16_group.h"
#include "../../include/mi355x_whir_ingest.h"
#include "../../include/mi355x_groth16_debug.h"

struct DevBuf {             // growable device scratch owned by the ctx (no hipMalloc in the hot path
    void *p = nullptr;      // after warm-up: buffers only ever grow)
    size_t cap = 0;
};

#define MI_MSM_SLOTS 6
#define MI_ZHOOK_SLOT 4    // the slot whose sort may take its count from computeH's last launch (mi_ctx::zhook)
struct MsmSlot {            // one in-flight MSM (msm.hip): own stream, events, workspaces, pinned result
    hipStream_t stream = nullptr;
    hipEvent_t ev[7]{};     // 0: sort done, 1/2: around the level-1 accumulate launch, 3/4: whole job, 5: bucket sums ready (deferred reduce),
                            // 6: the largest bucket's size has landed in host memory (exact level count, msm.hip)
    bool max_pending = false;       // this slot's sort has a "largest bucket" word on its way to the host (ev[6])
    uint32_t max_key_count = 0;     // entries of the fullest bucket of this slot's sort, once fetched
    DevBuf buf[18];
    void *host_wsum = nullptr;
    uint32_t n = 0, c = 0, G = 0;
    uint64_t entries_cap = 0;   // entries the sort of this slot is sized for: windows * n, or the counted number (MI_MSM_EXACT_SIZE)
    uint64_t stat_pairs = 0;    // (point, scalar) pairs this MSM really has (A and K run over per-wire expanded arrays with holes)
    uint32_t nwin_keys = 0;     // windows in the key space: ceil(256/c) for the generic MSM, 1 for the fixed-base one
    uint32_t nwin_digits = 0;   // digits per scalar = ceil(256/c)
    bool active = false, timed = false;
    bool deferred = false;      // accumulate stage done up to the bucket sums, reduce not yet enqueued (MI_MSM_DEFER_REDUCE)
    uint32_t tail_seg = 0;      // buckets per bucket-reduce thread
    const uint32_t *entries_src = nullptr;   // HOST word holding the number of sorted entries (stats): in the pinned memory of the slot that owns the sort
    // Set by the caller before mi_msm_enqueue, consumed by it: called on the enqueueing thread right before the bucket accumulation is
    // enqueued (after the sort); blocks until the event it returns HAS BEEN RECORDED and the slot's stream then waits for that event
    // (null result = no wait).  prove.hip holds the wire MSMs' accumulations back until computeH is done this way.
    const std::function<hipEvent_t()> *accum_gate = nullptr;
};

struct mi_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;   // host-pointer inputs of mi_groth16_prove: pageable copies run on a stream that carries nothing else
                                         // (no event between two of them: a marker takes the copies behind it off the runtime's fast path)
    std::string err;
    std::mutex err_m;          // a prove enqueues its MSM groups from helper threads (prove.hip): failures there report through mi_set_err
    mi_stats stats{};
    hipEvent_t ev[24]{};
    // scratch
    alignas(16) unsigned char ntt_state[384];  // NttState (ntt.hip): root tables + plan knobs
    alignas(16) unsigned char msm_knobs[128];   // MsmKnobs (msm.hip)
    DevBuf ws[24];
    MsmSlot msm[MI_MSM_SLOTS];          // MSM / prove workspaces, see msm.hip / prove.hip
    int cu_count = 256;
    int prio_scheme = 0;      // MI_PRIO_*: how the context's streams rank (api.hip, msm.hip)
    uint32_t fixed_knob[3] = {0, 0, 0};  // prove's fixed-base tables for A+K / B / Z: 0 = automatic, 1 = never, 17..22 = forced (prove.hip)
    // The Z MSM's digit count riding in computeH's LAST launch ("the count pass of Z could ride in computeH's last store"):
    // prove.hip arms it right before that launch (mi_msm_z_count_arm: the sort's shape and its count matrix C1), ntt.hip's fused last
    // kernel counts every h coefficient it stores -- its contiguous tile IS a slice (or two) of the sort -- and sets `done`; the Z sort
    // (msm2_sort_enqueue) then skips its own count pass: h is read once less.  Disarmed again as soon as the launch is enqueued.
    // Behind the knob "z_count_fused" (on by default: throughput equal, a single proof slightly shorter).
    // (only slot MI_ZHOOK_SLOT -- prove's Z MSM -- ever looks at it, and the thread that arms it is the one that enqueues that slot)
    std::atomic<uint64_t> dense_item_sorts{0};   // accumulations whose item size the dense-sort rule chose (msm.hip; helper threads enqueue too)
    uint64_t z_count_fused_launches = 0;   // computeH last launches that carried the count (mi_debug_get_counter: the tests' proof that the path ran)
    struct ZCountHook { bool armed = false, done = false; int slot = -1; uint32_t n = 0, c = 0; alignas(8) unsigned char shape[64]; uint32_t *C1 = nullptr; const void *h = nullptr; /* the vector whose digits were counted */ } zhook;
    uint32_t hold_accum = 0;             // prove: 1 = the wire MSMs' bucket accumulations wait for computeH (mi_debug_set_prove_schedule; measured: no gain)
};

// Fault injection for the error-path tests (mi_debug_inject_hip_failure, api.hip): the n-th MI_CHECK_HIP from now reports
// hipErrorUnknown INSTEAD of running its call.  Disabled (<= 0) it costs one relaxed atomic load per checked call.
#include <atomic>
extern std::atomic<int> mi_fault_countdown;
static inline bool mi_fault_hit() {
    if (mi_fault_countdown.load(std::memory_order_relaxed) <= 0) return false;
    return mi_fault_countdown.fetch_sub(1, std::memory_order_relaxed) == 1;
}
static inline void mi_set_err(struct mi_ctx *ctx, const std::string &msg);
#define MI_CHECK_HIP(ctx, call)                                                                       \
    do {                                                                                              \
        hipError_t e__ = mi_fault_hit() ? hipErrorUnknown : (call);                                   \
        if (e__ != hipSuccess) {                                                                      \
            mi_set_err((ctx), std::string(#call) + ": " + hipGetErrorString(e__));                    \
            return e__ == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP;                                  \
        }                                                                                             \
    } while (0)
#define MI_FAIL(ctx, code, msg)                                                                       \
    do {                                                                                              \
        mi_set_err((ctx), (msg));                                                                     \
        return (code);                                                                                \
    } while (0)
#define MI_TRY(expr)                                                                                  \
    do {                                                                                              \
        int32_t rc__ = (expr);                                                                        \
        if (rc__ != MI_OK) return rc__;                                                               \
    } while (0)

// A helper thread that works on a context beside the thread that owns the call (the prover pool's ProveKnowledge enqueue, pool.hip)
// records its failures in a string of its own: set for the duration of its work, it replaces ctx->err for THIS thread, so the two
// threads never write one std::string and each failure reaches the job with the message of the thread that met it.
// roctx ranges around the host-side phases of a call (mi_debug_set_trace_ranges; off by default: one relaxed load per site).  rocprofv3
// --marker-trace shows them beside the kernel trace.  api.hip loads the roctx library on first use; nothing links against it.
extern std::atomic<int> mi_ranges_on;
void mi_range_push(const char *name);
void mi_range_pop();
struct MiRange {
    bool on;
    explicit MiRange(const char *name) : on(mi_ranges_on.load(std::memory_order_relaxed) != 0) { if (on) mi_range_push(name); }
    ~MiRange() { if (on) mi_range_pop(); }
    MiRange(const MiRange &) = delete;
    MiRange &operator=(const MiRange &) = delete;
};
extern thread_local std::string *mi_err_sink;
static inline void mi_set_err(struct mi_ctx *ctx, const std::string &msg) {
    if (mi_err_sink) { *mi_err_sink = msg; return; }
    std::lock_guard<std::mutex> lk(ctx->err_m);
    ctx->err = msg;
}

// grow-only scratch
static inline int32_t mi_reserve(mi_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return MI_OK;
    if (b.p) { MI_CHECK_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    MI_CHECK_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return MI_OK;
}

// the same for scratch a faster path would like but the slower one does without: false (and no error) when the memory is not there
static inline bool mi_try_reserve(DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return true;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return false; }
    b.cap = want;
    return true;
}

// ctx->copy_stream, created on first use (api.hip)
int32_t mi_copy_stream(mi_ctx *ctx, hipStream_t *out);
// internal entry points implemented across translation units
int32_t mi_ntt_dev_impl(mi_ctx *ctx, mi_fr *inout_dev, uint32_t log_n, uint32_t flags);
int32_t mi_compute_h_dev_impl(mi_ctx *ctx, uint32_t log_n, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                              size_t n_constraints, mi_fr *h_out);
// computeH one input vector at a time (ntt.hip): part 0 = a, 1 = b, 2 = c (src = that vector), 3 = the rest (src unused)
// part 2 with src2 != null: c is not given; it is formed as src o src2 (the original a and b) on the way into its transform
int32_t mi_compute_h_part(mi_ctx *ctx, uint32_t log_n, int part, const mi_fr *src, size_t n_constraints, mi_fr *h_out, const mi_fr *src2 = nullptr);
void mi_ntt_state_init(mi_ctx *ctx);
size_t mi_ntt_table_bytes(mi_ctx *ctx);
void mi_ntt_state_free(mi_ctx *ctx);
bool mi_ntt_set_knob(mi_ctx *ctx, const char *name, int64_t value);   // the NTT's share of mi_debug_set_knob: true = name known and value accepted
void mi_ntt_state_trim(mi_ctx *ctx);   // frees every table and marks them unbuilt (mi_ctx_trim)
int32_t mi_msm_state_init(mi_ctx *ctx);   // MI_OK or the first failing HIP call; partial state is freed by mi_msm_state_free
// Stream priority schemes (3 hardware levels; comment in msm.hip).  A context on its own ranks computeH high, the wire MSMs
// normal and Z low.  The contexts of a prover pool are staggered on top of that: the first runs nearly as if alone, the
// others fill what it leaves (28.2-28.4 vs 27.4 proofs/s with three in flight; two staggered contexts reach what three
// equal ones did).
enum { MI_PRIO_SOLO = 0,          // computeH high, A/B1/B2/K normal, Z low
       MI_PRIO_POOL_FIRST = 1,    // computeH high, A/B1/B2/K high,   Z normal
       MI_PRIO_POOL_SECOND = 2,   // computeH normal, A/B1/B2/K normal, Z low
       MI_PRIO_POOL_REST = 3 };   // everything low
void mi_msm_state_free(mi_ctx *ctx);
// Asynchronous MSM on slot `slot` (curve 1 = G1, 2 = G2; device pointers).  sort_slot < 0: sort the scalars
// here; otherwise reuse the sort already enqueued on that slot (same scalars, other bases).  wait_ev (may be
// null) orders the slot's stream after the producer of its inputs.  mi_msm_finish blocks on the slot and
// returns the XYZZ result on the host.
// precomp_c != 0: pts_dev holds the fixed-base window copies [ceil(256/c)][n] built by mi_msm_precompute (msm2_core.cuh).
int32_t mi_msm_enqueue(mi_ctx *ctx, int slot, int sort_slot, int curve, const void *pts_dev, const void *scalars_dev, size_t n,
                       uint32_t flags, hipEvent_t wait_ev, bool timed, uint32_t precomp_c = 0, size_t stat_pairs = 0,
                       uint32_t generic_c = 0 /* window bits of the generic path when the parts of a sharded MSM must agree (0 = from n) */);
// pre[w][i] = 2^(c*w) * base[i] for w < ceil(256/c) (affine), on ctx->stream.  pre must hold ceil(256/c) * n points.
int32_t mi_msm_precompute(mi_ctx *ctx, int curve, const void *base_dev, void *pre_dev, size_t n, uint32_t c);
int32_t mi_msm_finish(mi_ctx *ctx, int slot, int curve, void *out_xyzz_host);
// arms ctx->zhook for the fixed-base MSM that `slot` is about to run over n scalars with c-bit windows (no-op when the knob is off or the
// shape does not fit: the sort then counts by itself, as always)
int32_t mi_msm_z_count_arm(mi_ctx *ctx, int slot, size_t n, uint32_t c);
// Internal flag of mi_msm_enqueue (next to MI_MSM_SCALARS_CANONICAL): stop at the bucket sums.  mi_msm_bucket_view then
// exposes them, mi_msm_reduce_enqueue runs the rest (bucket reduce, window sums, copy to the host) and mi_msm_finish collects.
#define MI_MSM_DEFER_REDUCE 0x100u
// Internal flag: size the entry-indexed workspaces by the COUNTED number of non-zero digits (one synchronisation of the slot's
// stream after the count pass) instead of windows * n.  For scalars that are available when the MSM is enqueued (wire values).
#define MI_MSM_EXACT_SIZE 0x200u
// Internal flag: the points are in the R' = 2^261 packed form (msm_curve_ops.h to_rprime): G1 level 1 runs the 9 x 29-bit kernel.
#define MI_MSM_PTS_RPRIME 0x400u
struct MsmBucketView {
    void *bucket = nullptr;     // XYZZ[nkeys] on the slot's device; null for an empty MSM
    size_t nkeys = 0, xyzz_bytes = 0;
    uint32_t seg = 0, c = 0, nwin = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr; // recorded on `stream` after the last accumulate level
};
int32_t mi_msm_bucket_view(mi_ctx *ctx, int slot, int curve, MsmBucketView *v);
int32_t mi_msm_reduce_enqueue(mi_ctx *ctx, int slot, int curve);
bool mi_msm_limb29_enabled(mi_ctx *ctx);   // the G1 level-1 kernel in 9 x 29-bit limbs is in use (mi_debug_set_msm_limb29)
uint32_t mi_msm_auto_c(size_t n);   // the generic path's window bits for n pairs
const struct MsmCurveOps &mi_msm_ops(int curve);

