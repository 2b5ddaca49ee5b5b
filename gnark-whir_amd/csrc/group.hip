// Device groups: the prove path and a single MSM point-sharded over the GPUs of one node (BASELINE.json
// configs[4]; north_star: "a single large MSM shards its base points across the 8 GPUs of one node with an RCCL all-reduce of
// partial bucket sums over xGMI").  The reference has one call, groth16.Prove at /root/reference/mt.go:496, and no notion of
// devices; this is what its Go caller binds to spread that one call over several MI355X.
//
//   partitioning   pk points are static: mi_pk_load_sharded cuts the WIRES into `world` contiguous ranges (and the N - 1 pairs
//                  of the Z MSM likewise); rank r keeps the A / B1 / B2 / K points of its wires and its slice of pk.G1.Z
//                  resident (tables included), so a proof moves no point over any link.  Per proof every rank receives its slice
//                  of W from the host, the lead rank runs computeH (NTT = replicas only) and hands each rank its
//                  slice of h device-to-device.
//   exchange       EC addition is not an ncclRedOp, so the "all-reduce of partial bucket sums" is byte-typed:
//                  mode 0 (SURVEY 8e option i)  every rank finishes its Pippenger locally and contributes ONE partial sum per
//                         MSM (128 / 256 B XYZZ); combine = world point additions (single process: on the host, the window sums
//                         land in pinned host memory anyway; one rank per process: ncclAllGather of the partials).
//                  mode 1 (option ii, the north_star's wording)  every rank stops at its BUCKET sums; rank r owns the keys
//                         [r K / world, (r+1) K / world) and receives that slice from every other rank -- a reduce-scatter written
//                         as grouped ncclSend / ncclRecv, one hop on the full xGMI mesh, all 7 links at once -- adds the slices
//                         (k_msm_sum_slices), runs the bucket reduce on its slice only, and the per-rank results are combined as
//                         in mode 0 (Horner over the windows is linear, so combining after it is the same sum).
//   transport      RCCL (ncclCommInitAll in one process, ncclCommInitRank for one rank per process).  A group that names the
//                  same device twice (the 1-GPU rehearsal the tests run) cannot have an RCCL communicator and moves the same
//                  slices with hipMemcpyPeerAsync instead.
#include "prove_internal.h"
#include "msm_curve_ops.h"
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <future>
#include <thread>
#include <vector>

struct mi_group {
    int world = 0;                 // ranks in the group
    int rank0 = 0;                 // global rank of the first local context
    std::vector<int> dev;          // device of every LOCAL rank
    std::vector<mi_ctx *> ctx;     // one context per local rank
    std::vector<ncclComm_t> comm;  // RCCL communicator per local rank; empty = peer copies inside this process
    std::vector<hipStream_t> xs;   // per local rank: the exchange stream (sends, receives, slice sums)
    std::vector<hipEvent_t> ev_x, ev_in, ev_done, ev_h;
    std::vector<DevBuf> recv;      // per local rank: bucket slices received from the other ranks
    std::vector<DevBuf> stage;     // per local rank: small staging area for the partial-sum all-gather
    std::string err;
    std::atomic<bool> busy{false};   // calls on one group must not overlap: an entry point that finds it set returns MI_EINVAL (GroupCall)
    int n_local() const { return (int)ctx.size(); }
    bool local(int r) const { return r >= rank0 && r < rank0 + n_local(); }
};
struct mi_pk_sharded {
    std::vector<mi_pk *> part;     // one per local rank
    u32 log_n = 0;
    u64 nb_wires = 0;
    bool uniform = false;          // every local part chose the same MSM plan per group (needed by mode 1)
};

#define G_FAIL(g, code, msg) do { (g)->err = (msg); return (code); } while (0)
#define G_HIP(g, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e__); \
                            return e__ == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP; } } while (0)
#define G_NCCL(g, call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { (g)->err = std::string(#call) + ": " + ncclGetErrorString(r__); \
                             return MI_EHIP; } } while (0)
#define G_CTX(g, i, expr) do { int32_t rc__ = (expr); if (rc__ != MI_OK) { (g)->err = mi_last_error((g)->ctx[i]); return rc__; } } while (0)

// Entry-point guard: the group's exchange streams, receive buffers and per-rank contexts serve ONE call at a time.  A second call that
// arrives while one is running is refused (MI_EINVAL, the running call's error text is left alone) instead of corrupting g->recv.
struct GroupCall {
    mi_group *g; bool ok;
    explicit GroupCall(mi_group *g_) : g(g_), ok(g_ && !g_->busy.exchange(true, std::memory_order_acquire)) {}
    ~GroupCall() { if (ok) g->busy.store(false, std::memory_order_release); }
};
#define G_ENTER(g) GroupCall call__(g); if (!call__.ok) return MI_EINVAL

static void range_of(u64 total, int world, int r, u64 &lo, u64 &hi) { lo = total * (u64)r / (u64)world; hi = total * (u64)(r + 1) / (u64)world; }

// ---------------------------------------------------------------- point-to-point batches
struct Xfer { int src, dst; const void *sp; void *dp; size_t bytes; };   // global ranks; a pointer is meaningful in its owner's process only
// Runs the batch on the exchange streams xs[local rank].  Afterwards xs[i] is ordered after every transfer rank i sends or receives.
static int32_t run_xfers(mi_group *g, const std::vector<Xfer> &xs_list, const std::vector<hipStream_t> &xs) {
    if (!g->comm.empty()) {
        G_NCCL(g, ncclGroupStart());
        for (const Xfer &x : xs_list) {
            if (!x.bytes) continue;
            if (g->local(x.src)) { (void)hipSetDevice(g->dev[x.src - g->rank0]); G_NCCL(g, ncclSend(x.sp, x.bytes, ncclUint8, x.dst, g->comm[x.src - g->rank0], xs[x.src - g->rank0])); }
            if (g->local(x.dst)) { (void)hipSetDevice(g->dev[x.dst - g->rank0]); G_NCCL(g, ncclRecv(x.dp, x.bytes, ncclUint8, x.src, g->comm[x.dst - g->rank0], xs[x.dst - g->rank0])); }
        }
        G_NCCL(g, ncclGroupEnd());
        return MI_OK;
    }
    // same process, no communicator (a device named twice): peer copies on the source's stream, then every stream waits for all.
    // A copy is issued by its SOURCE into the destination's buffer, so the sources must first wait for whatever the destinations
    // still do with those buffers (the slice sums of the previous exchange): a stream-level barrier on entry as well.
    for (int i = 0; i < g->n_local(); i++) { (void)hipSetDevice(g->dev[i]); G_HIP(g, hipEventRecord(g->ev_in[i], xs[i])); }
    for (int i = 0; i < g->n_local(); i++) {
        (void)hipSetDevice(g->dev[i]);
        for (int j = 0; j < g->n_local(); j++) if (j != i) G_HIP(g, hipStreamWaitEvent(xs[i], g->ev_in[j], 0));
    }
    for (const Xfer &x : xs_list) {
        if (!x.bytes) continue;
        if (!g->local(x.src) || !g->local(x.dst)) G_FAIL(g, MI_EINVAL, "group: peer-copy transport reached a remote rank");
        const int s = x.src - g->rank0, d = x.dst - g->rank0;
        (void)hipSetDevice(g->dev[s]);
        if (g->dev[s] == g->dev[d]) G_HIP(g, hipMemcpyAsync(x.dp, x.sp, x.bytes, hipMemcpyDeviceToDevice, xs[s]));
        else G_HIP(g, hipMemcpyPeerAsync(x.dp, g->dev[d], x.sp, g->dev[s], x.bytes, xs[s]));
    }
    for (int i = 0; i < g->n_local(); i++) { (void)hipSetDevice(g->dev[i]); G_HIP(g, hipEventRecord(g->ev_x[i], xs[i])); }
    for (int i = 0; i < g->n_local(); i++) {
        (void)hipSetDevice(g->dev[i]);
        for (int j = 0; j < g->n_local(); j++) if (j != i) G_HIP(g, hipStreamWaitEvent(xs[i], g->ev_x[j], 0));
    }
    return MI_OK;
}

// ---------------------------------------------------------------- lifecycle
static int32_t group_finish_init(mi_group *g) {
    const int n = g->n_local();
    g->xs.assign(n, nullptr); g->ev_x.assign(n, nullptr); g->ev_in.assign(n, nullptr); g->ev_done.assign(n, nullptr); g->ev_h.assign(n, nullptr);
    g->recv.assign(n, DevBuf{}); g->stage.assign(n, DevBuf{});
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(g->dev[i]);
        G_HIP(g, hipStreamCreateWithFlags(&g->xs[i], hipStreamNonBlocking));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_x[i], hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_in[i], hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_done[i], hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_h[i], hipEventDisableTiming));
    }
    return MI_OK;
}

extern "C" {

int32_t mi_group_destroy(mi_group *g) {
    if (!g) return MI_EINVAL;
    for (int i = 0; i < g->n_local(); i++) {
        (void)hipSetDevice(g->dev[i]);
        if (g->ctx[i]) (void)hipStreamSynchronize(g->ctx[i]->stream);
        if (i < (int)g->xs.size() && g->xs[i]) { (void)hipStreamSynchronize(g->xs[i]); (void)hipStreamDestroy(g->xs[i]); }
        if (i < (int)g->comm.size() && g->comm[i]) (void)ncclCommDestroy(g->comm[i]);
        for (auto *v : {&g->ev_x, &g->ev_in, &g->ev_done, &g->ev_h}) if (i < (int)v->size() && (*v)[i]) (void)hipEventDestroy((*v)[i]);
        if (i < (int)g->recv.size() && g->recv[i].p) (void)hipFree(g->recv[i].p);
        if (i < (int)g->stage.size() && g->stage[i].p) (void)hipFree(g->stage[i].p);
        if (g->ctx[i]) mi_shutdown(g->ctx[i]);
    }
    delete g;
    return MI_OK;
}

// One process, n_dev contexts (SURVEY 8b's mi_init(dev_ids, n_dev, ...)): what a Go caller uses.
int32_t mi_group_create(const int *dev_ids, int n_dev, mi_group **out) {
    if (!dev_ids || !out || n_dev < 1 || n_dev > 64) return MI_EINVAL;
    *out = nullptr;
    mi_group *g = new (std::nothrow) mi_group();
    if (!g) return MI_ENOMEM;
    g->world = n_dev; g->rank0 = 0;
    bool distinct = true;
    for (int i = 0; i < n_dev; i++) for (int j = 0; j < i; j++) if (dev_ids[i] == dev_ids[j]) distinct = false;
    for (int i = 0; i < n_dev; i++) {
        mi_ctx *c = nullptr;
        int32_t rc = mi_init(dev_ids[i], &c);
        if (rc != MI_OK) { mi_group_destroy(g); return rc; }
        g->dev.push_back(dev_ids[i]); g->ctx.push_back(c);
    }
    int32_t rc = group_finish_init(g);
    if (rc == MI_OK && distinct) {
        // full-mesh xGMI: let every device map every other one (peer copies of the h slices; RCCL does its own set-up)
        for (int i = 0; i < n_dev; i++) for (int j = 0; j < n_dev; j++) if (i != j) {
            (void)hipSetDevice(dev_ids[i]);
            hipError_t e = hipDeviceEnablePeerAccess(dev_ids[j], 0);
            if (e != hipSuccess) (void)hipGetLastError();   // already enabled, or no direct link: hipMemcpyPeerAsync still works (staged)
        }
        g->comm.assign(n_dev, nullptr);
        ncclResult_t r = ncclCommInitAll(g->comm.data(), n_dev, dev_ids);
        if (r != ncclSuccess) { g->comm.clear(); mi_group_destroy(g); return MI_EHIP; }
    }
    if (rc != MI_OK) { mi_group_destroy(g); return rc; }
    *out = g;
    return MI_OK;
}

// One rank per process (torch.distributed.run, MPI): rank 0 makes the id, the caller's own channel distributes its 128 bytes.
int32_t mi_group_unique_id(uint8_t id[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!id) return MI_EINVAL;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return MI_EHIP;
    std::memcpy(id, &u, 128);
    return MI_OK;
}
int32_t mi_group_create_rank(int device_id, int rank, int world, const uint8_t id[128], mi_group **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return MI_EINVAL;
    *out = nullptr;
    mi_group *g = new (std::nothrow) mi_group();
    if (!g) return MI_ENOMEM;
    g->world = world; g->rank0 = rank;
    mi_ctx *c = nullptr;
    int32_t rc = mi_init(device_id, &c);
    if (rc != MI_OK) { delete g; return rc; }
    g->dev.push_back(device_id); g->ctx.push_back(c);
    rc = group_finish_init(g);
    if (rc == MI_OK) {
        ncclUniqueId u;
        std::memcpy(&u, id, 128);
        (void)hipSetDevice(device_id);
        g->comm.assign(1, nullptr);
        if (ncclCommInitRank(&g->comm[0], world, u, rank) != ncclSuccess) { g->comm.clear(); rc = MI_EHIP; }
    }
    if (rc != MI_OK) { mi_group_destroy(g); return rc; }
    *out = g;
    return MI_OK;
}
int32_t mi_group_world(const mi_group *g) { return g ? g->world : 0; }
int32_t mi_group_local(const mi_group *g) { return g ? g->n_local() : 0; }
mi_ctx *mi_group_ctx(mi_group *g, int local_rank) { return g && local_rank >= 0 && local_rank < g->n_local() ? g->ctx[local_rank] : nullptr; }
const char *mi_group_last_error(mi_group *g) { return g ? g->err.c_str() : "null group"; }
// 1 = RCCL communicator, 2 = peer copies inside this process (a device named twice)
int32_t mi_group_transport(const mi_group *g) { return !g ? 0 : (g->comm.empty() ? 2 : 1); }

// Every local rank sends a distinct pattern of `bytes` bytes to every rank of the group (itself included) and checks what it
// received: the transport (RCCL grouped send / recv, or peer copies) in isolation.  All ranks of the group call it together.
int32_t mi_group_exchange_selftest(mi_group *g, size_t bytes) {
    if (!g || !bytes || bytes > ((size_t)1 << 28)) return MI_EINVAL;
    G_ENTER(g);
    const int nl = g->n_local(), W = g->world;
    std::vector<void *> sbuf(nl, nullptr), rbuf(nl, nullptr);
    std::vector<hipStream_t> xs(nl);
    std::vector<Xfer> list;
    auto pat = [](int src, int dst, size_t k) { return (unsigned char)(17 * src + 101 * dst + 3 * k + (k >> 8)); };
    int32_t rc = MI_OK;
    auto body = [&]() -> int32_t {
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            G_HIP(g, hipMalloc(&sbuf[i], bytes * W)); G_HIP(g, hipMalloc(&rbuf[i], bytes * W));
            std::vector<unsigned char> h(bytes * W);
            for (int d = 0; d < W; d++) for (size_t k = 0; k < bytes; k++) h[d * bytes + k] = pat(g->rank0 + i, d, k);
            xs[i] = g->xs[i];
            G_HIP(g, hipMemcpyAsync(sbuf[i], h.data(), h.size(), hipMemcpyHostToDevice, xs[i]));
            G_HIP(g, hipMemsetAsync(rbuf[i], 0, bytes * W, xs[i]));
            G_HIP(g, hipStreamSynchronize(xs[i]));
        }
        for (int s = 0; s < W; s++) for (int d = 0; d < W; d++) {
            Xfer x{s, d, nullptr, nullptr, bytes};
            if (g->local(s)) x.sp = (char *)sbuf[s - g->rank0] + (size_t)d * bytes;
            if (g->local(d)) x.dp = (char *)rbuf[d - g->rank0] + (size_t)s * bytes;
            list.push_back(x);
        }
        MI_TRY(run_xfers(g, list, xs));
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            std::vector<unsigned char> h(bytes * W);
            G_HIP(g, hipMemcpyAsync(h.data(), rbuf[i], h.size(), hipMemcpyDeviceToHost, xs[i]));
            G_HIP(g, hipStreamSynchronize(xs[i]));
            for (int s = 0; s < W; s++) for (size_t k = 0; k < bytes; k++)
                if (h[s * bytes + k] != pat(s, g->rank0 + i, k)) G_FAIL(g, MI_EHIP, "group: exchange self-test received wrong bytes");
        }
        return MI_OK;
    };
    rc = body();
    for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); if (sbuf[i]) (void)hipFree(sbuf[i]); if (rbuf[i]) (void)hipFree(rbuf[i]); }
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------- mode 1: reduce-scatter of the bucket sums of one MSM slot
// Every local rank has enqueued the MSM of `slot` with MI_MSM_DEFER_REDUCE.  Afterwards every rank's reduce is enqueued.
static int32_t exchange_buckets(mi_group *g, int slot, int curve) {
    const int nl = g->n_local(), W = g->world;
    const MsmCurveOps &ops = mi_msm_ops(curve);
    std::vector<MsmBucketView> v(nl);
    for (int i = 0; i < nl; i++) {
        G_CTX(g, i, mi_msm_bucket_view(g->ctx[i], slot, curve, &v[i]));
        if (!v[i].bucket) G_FAIL(g, MI_EINVAL, "group: mode 1 (bucket exchange) needs every rank to hold at least one pair of every MSM");
        if (v[i].nkeys != v[0].nkeys || v[i].c != v[0].c) G_FAIL(g, MI_EINVAL, "group: the ranks disagree on the bucket layout of an MSM");
    }
    const size_t K = v[0].nkeys, B = ops.xyzz_bytes;
    const size_t own_max = (K + W - 1) / W + 1;
    std::vector<hipStream_t> xs(nl);
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        xs[i] = g->xs[i];
        G_CTX(g, i, mi_reserve(g->ctx[i], g->recv[i], (size_t)(W > 1 ? W - 1 : 1) * own_max * B));
        G_HIP(g, hipStreamWaitEvent(xs[i], v[i].ready, 0));
    }
    std::vector<Xfer> list;
    for (int s = 0; s < W; s++) for (int d = 0; d < W; d++) {
        if (s == d) continue;
        u64 lo, hi;
        range_of(K, W, d, lo, hi);
        Xfer x{s, d, nullptr, nullptr, (size_t)(hi - lo) * B};
        if (g->local(s)) x.sp = (const char *)v[s - g->rank0].bucket + lo * B;
        if (g->local(d)) x.dp = (char *)g->recv[d - g->rank0].p + (size_t)(s < d ? s : s - 1) * (hi - lo) * B;
        list.push_back(x);
    }
    MI_TRY(run_xfers(g, list, xs));
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        u64 lo, hi;
        range_of(K, W, g->rank0 + i, lo, hi);
        char *bk = (char *)v[i].bucket;
        ops.sum_slices(xs[i], bk + lo * B, g->recv[i].p, (u32)(W - 1), (u32)(hi - lo));
        G_HIP(g, hipGetLastError());
        // keys of other owners: their sums live there now; here they read as infinity for the reduce
        if (lo) G_HIP(g, hipMemsetAsync(bk, 0, lo * B, xs[i]));
        if (hi < K) G_HIP(g, hipMemsetAsync(bk + hi * B, 0, (K - hi) * B, xs[i]));
        G_HIP(g, hipEventRecord(g->ev_done[i], xs[i]));
        G_HIP(g, hipStreamWaitEvent(v[i].stream, g->ev_done[i], 0));
        G_CTX(g, i, mi_msm_reduce_enqueue(g->ctx[i], slot, curve));
    }
    return MI_OK;
}

// Sum of the per-rank partial results of one MSM (XYZZ on the host).  Single process: plain additions.  One rank per process:
// byte-typed ncclAllGather of the partials, then the same additions (in rank order) on every rank.
template <class F>
static int32_t combine_partials(mi_group *g, const std::vector<XYZZ<F>> &local, XYZZ<F> *out) {
    XYZZ<F> acc = XYZZ<F>::inf();
    if (g->n_local() == g->world) {
        for (const auto &p : local) xyzz_add(acc, p);
        *out = acc;
        return MI_OK;
    }
    if (g->n_local() != 1 || g->comm.size() != 1) G_FAIL(g, MI_EINVAL, "group: a process holds either all ranks or exactly one");
    const size_t B = sizeof(XYZZ<F>);
    std::vector<XYZZ<F>> all((size_t)g->world);
    (void)hipSetDevice(g->dev[0]);
    G_CTX(g, 0, mi_reserve(g->ctx[0], g->stage[0], B * (size_t)(g->world + 1)));
    char *st = (char *)g->stage[0].p;
    hipStream_t s = g->xs[0];
    G_HIP(g, hipMemcpyAsync(st, &local[0], B, hipMemcpyHostToDevice, s));
    G_NCCL(g, ncclAllGather(st, st + B, B, ncclUint8, g->comm[0], s));
    G_HIP(g, hipMemcpyAsync(all.data(), st + B, B * (size_t)g->world, hipMemcpyDeviceToHost, s));
    G_HIP(g, hipStreamSynchronize(s));
    for (const auto &p : all) xyzz_add(acc, p);
    *out = acc;
    return MI_OK;
}

// Group-wide minimum and maximum of one 64-bit value per local rank (plans every rank must agree on: table budgets, window widths).
// Single process: over the local values.  One rank per process: ncclAllGather of the 8 bytes.
static int32_t group_min_max(mi_group *g, const std::vector<u64> &local, u64 *mn, u64 *mx) {
    std::vector<u64> all = local;
    if (g->n_local() != g->world) {
        if (g->n_local() != 1 || g->comm.size() != 1) G_FAIL(g, MI_EINVAL, "group: a process holds either all ranks or exactly one");
        all.assign((size_t)g->world, 0);
        (void)hipSetDevice(g->dev[0]);
        G_CTX(g, 0, mi_reserve(g->ctx[0], g->stage[0], 8 * (size_t)(g->world + 1)));
        char *st = (char *)g->stage[0].p;
        hipStream_t s = g->xs[0];
        G_HIP(g, hipMemcpyAsync(st, &local[0], 8, hipMemcpyHostToDevice, s));
        G_NCCL(g, ncclAllGather(st, st + 8, 8, ncclUint8, g->comm[0], s));
        G_HIP(g, hipMemcpyAsync(all.data(), st + 8, 8 * (size_t)g->world, hipMemcpyDeviceToHost, s));
        G_HIP(g, hipStreamSynchronize(s));
    }
    *mn = ~(u64)0; *mx = 0;
    for (u64 v : all) { if (v < *mn) *mn = v; if (v > *mx) *mx = v; }
    return MI_OK;
}

template <class F, class JacT>
static void write_jac(const XYZZ<F> &r, JacT *out) {
    Jac<F> j;
    if (r.is_inf()) j = Jac<F>{F::one(), F::one(), F::zero()};
    else { Affine<F> a = xyzz_to_affine(r); j = Jac<F>{a.x, a.y, F::one()}; }
    std::memcpy(out, &j, sizeof(j));
}

// One MSM whose (point, scalar) pairs are already spread over the local ranks' devices.
template <class F, class JacT>
static int32_t msm_sharded_dev(mi_group *g, int curve, const void *const *pts_dev, const void *const *sc_dev, const size_t *n_local, size_t n_total,
                               uint32_t flags, uint32_t mode, JacT *out) {
    if (!g || !pts_dev || !sc_dev || !n_local || !out || (flags & ~1u) || mode > 1) return MI_EINVAL;
    const int nl = g->n_local();
    // every rank must cut its scalars into the same windows: width from the largest share, not from the local count
    const u32 c = mi_msm_auto_c((n_total + g->world - 1) / g->world);
    const uint32_t df = mode == 1 ? MI_MSM_DEFER_REDUCE : 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        std::memset(&ctx->stats, 0, sizeof(ctx->stats));
        G_HIP(g, hipEventRecord(ctx->ev[0], ctx->stream));
        G_CTX(g, i, mi_msm_enqueue(ctx, 0, -1, curve, pts_dev[i], sc_dev[i], n_local[i], flags | df, ctx->ev[0], curve == 1, 0, 0, c));
    }
    if (mode == 1) MI_TRY(exchange_buckets(g, 0, curve));
    std::vector<XYZZ<F>> part((size_t)nl);
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        G_CTX(g, i, mi_msm_finish(g->ctx[i], 0, curve, &part[i]));
    }
    XYZZ<F> total;
    MI_TRY(combine_partials<F>(g, part, &total));
    write_jac<F>(total, out);
    g->ctx[0]->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MI_OK;
}

// Host arrays, one process: cut into contiguous slices, upload slice r to rank r, run the sharded MSM.
template <class F, class AffT, class JacT>
static int32_t msm_sharded_host(mi_group *g, int curve, const AffT *pts, const mi_fr *scalars, size_t n, uint32_t flags, uint32_t mode, JacT *out) {
    if (!g || !out || ((!pts || !scalars) && n)) return MI_EINVAL;
    if (g->n_local() != g->world) G_FAIL(g, MI_EINVAL, "group: host-array entry points need all ranks in this process");
    const int nl = g->n_local();
    std::vector<const void *> pp(nl), ss(nl);
    std::vector<size_t> nn(nl);
    for (int i = 0; i < nl; i++) {
        u64 lo, hi;
        range_of(n, g->world, i, lo, hi);
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        G_CTX(g, i, mi_reserve(ctx, ctx->ws[2], (hi - lo) * sizeof(AffT) + 64));
        G_CTX(g, i, mi_reserve(ctx, ctx->ws[3], (hi - lo) * sizeof(mi_fr) + 64));
        if (hi > lo) {
            G_HIP(g, hipMemcpyAsync(ctx->ws[2].p, pts + lo, (hi - lo) * sizeof(AffT), hipMemcpyHostToDevice, ctx->stream));
            G_HIP(g, hipMemcpyAsync(ctx->ws[3].p, scalars + lo, (hi - lo) * sizeof(mi_fr), hipMemcpyHostToDevice, ctx->stream));
        }
        pp[i] = ctx->ws[2].p; ss[i] = ctx->ws[3].p; nn[i] = hi - lo;
    }
    return msm_sharded_dev<F>(g, curve, pp.data(), ss.data(), nn.data(), n, flags, mode, out);
}

extern "C" {

int32_t mi_msm_g1_sharded_dev(mi_group *g, const mi_g1_affine *const *pts_dev, const mi_fr *const *scalars_dev, const size_t *n_local, size_t n_total,
                              uint32_t flags, uint32_t mode, mi_g1_jac *out) {
    G_ENTER(g);
    return msm_sharded_dev<Fp>(g, 1, (const void *const *)pts_dev, (const void *const *)scalars_dev, n_local, n_total, flags, mode, out);
}
int32_t mi_msm_g2_sharded_dev(mi_group *g, const mi_g2_affine *const *pts_dev, const mi_fr *const *scalars_dev, const size_t *n_local, size_t n_total,
                              uint32_t flags, uint32_t mode, mi_g2_jac *out) {
    G_ENTER(g);
    return msm_sharded_dev<Fp2>(g, 2, (const void *const *)pts_dev, (const void *const *)scalars_dev, n_local, n_total, flags, mode, out);
}
int32_t mi_msm_g1_sharded(mi_group *g, const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, uint32_t mode, mi_g1_jac *out) {
    G_ENTER(g);
    return msm_sharded_host<Fp>(g, 1, pts, scalars, n, flags, mode, out);
}
int32_t mi_msm_g2_sharded(mi_group *g, const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags, uint32_t mode, mi_g2_jac *out) {
    G_ENTER(g);
    return msm_sharded_host<Fp2>(g, 2, pts, scalars, n, flags, mode, out);
}

// ---------------------------------------------------------------- sharded proving key
int32_t mi_pk_sharded_free(mi_group *g, mi_pk_sharded *spk) {
    if (!g || !spk) return MI_EINVAL;
    G_ENTER(g);
    for (size_t i = 0; i < spk->part.size(); i++) if (spk->part[i]) { (void)hipSetDevice(g->dev[i]); mi_pk_free(g->ctx[i], spk->part[i]); }
    delete spk;
    return MI_OK;
}

}  // extern "C"

// Splits pk.G1.{A,B,K,Z} and pk.G2.B into `world` contiguous slices (by wire; Z by index) and makes slice r resident on rank r.
//   host arrays    descs = ONE whole-key descriptor (what mi_pk_load takes); every process of a multi-process group passes it.
//   device arrays  descs = one descriptor per LOCAL rank: header and masks of the WHOLE key (host), point arrays = that rank's
//                  slices already on that rank's device (counts = points of the slice), adopted by reference as mi_pk_load_dev does.
static int32_t pk_load_sharded_impl(mi_group *g, const mi_pk_desc *descs, bool device_points, mi_pk_sharded **out) {
    if (!g || !descs || !out) return MI_EINVAL;
    *out = nullptr;
    const mi_pk_desc *d = descs;
    if (d->log_n > 28 || !d->infinity_a || !d->infinity_b || d->nb_public > d->nb_wires) G_FAIL(g, MI_EINVAL, "pk: bad header");
    const int nl = g->n_local(), W = g->world;
    if (device_points) for (int i = 1; i < nl; i++)
        if (descs[i].log_n != d->log_n || descs[i].nb_wires != d->nb_wires || descs[i].nb_public != d->nb_public || !descs[i].infinity_a || !descs[i].infinity_b)
            G_FAIL(g, MI_EINVAL, "pk: the per-rank descriptors disagree on the key's header");
    const u64 N = (u64)1 << d->log_n;
    mi_pk_sharded *spk = new (std::nothrow) mi_pk_sharded();
    if (!spk) return MI_ENOMEM;
    spk->part.assign(nl, nullptr); spk->log_n = d->log_n; spk->nb_wires = d->nb_wires;
    // window widths of the generic path that all parts share (mode 1 needs equal bucket layouts): from the LARGEST part of each MSM
    // ... and whether EVERY rank holds at least one pair of every MSM: mode 1 exchanges bucket slices rank to rank and a rank without
    // buckets would leave its peers waiting in the collective -- decided here from the masks every process holds, so that every
    // process refuses mode 1 alike (spk->uniform) instead of one rank failing while the others hang
    u64 max_w = 0, max_b = 0, max_z = 0;
    bool all_nonempty = true;
    for (int r = 0; r < W; r++) {
        u64 lo, hi, zlo, zhi, nb = 0;
        range_of(d->nb_wires, W, r, lo, hi); range_of(N - 1, W, r, zlo, zhi);
        for (u64 j = lo; j < hi; j++) nb += d->infinity_b[j] ? 0 : 1;
        if (hi - lo > max_w) max_w = hi - lo;
        if (nb > max_b) max_b = nb;
        if (zhi - zlo > max_z) max_z = zhi - zlo;
        if (hi == lo || nb == 0 || zhi == zlo) all_nonempty = false;
    }
    // ONE fixed-base plan for all parts (mode 1 exchanges buckets, so the parts must cut their scalars alike; and a part just
    // under the 2^20-point threshold next to one just over it would otherwise pick different paths): the rule of mi_pk_load
    // (prove.hip: tables for an MSM of >= 2^20 points while they fit in a third of the free memory, smallest group first),
    // applied to the LARGEST part and the tightest device OF THE WHOLE GROUP (one rank per process: the budgets are all-gathered),
    // then forced on every context through its knobs.  Knobs the caller set (mi_debug_set_prove_fixed_base) are kept.
    u32 plan[3] = {1, 1, 1};   // A+K, B, Z: 1 = no tables
    auto fail = [&](int32_t rc) { for (size_t i = 0; i < spk->part.size(); i++) if (spk->part[i]) { (void)hipSetDevice(g->dev[i]); mi_pk_free(g->ctx[i], spk->part[i]); } delete spk; return rc; };
    {
        std::vector<u64> budgets(nl, 0);
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            size_t fr = 0, tot = 0, sharers = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
            for (int j = 0; j < nl; j++) sharers += g->dev[j] == g->dev[i] ? 1 : 0;
            budgets[i] = fr / 3 / sharers;
        }
        u64 bmin = 0, bmax = 0;
        int32_t rc = group_min_max(g, budgets, &bmin, &bmax);
        if (rc != MI_OK) return fail(rc);
        size_t budget = (size_t)bmin;
        auto nwin_of = [](u32 c) { return (size_t)((256 + c - 1) / c); };
        auto choose = [&](u32 c_auto, u64 n_max, size_t bytes_per_point) -> u32 {
            const size_t need = nwin_of(c_auto) * n_max * bytes_per_point;
            if (n_max < ((u64)1 << 20) || need > budget) return 1;
            budget -= need;
            return c_auto;
        };
        plan[2] = choose(20, max_z, sizeof(G1Aff));
        plan[1] = choose(17, max_b, sizeof(G1Aff) + sizeof(G2Aff));
        plan[0] = choose(19, max_w, 2 * sizeof(G1Aff));
    }
    std::vector<int32_t> rcs(nl, MI_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < nl; i++) th.emplace_back([&, i] {   // uploads (and table builds) of the parts run side by side, one host thread per device
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        ShardRange sr;
        range_of(d->nb_wires, W, g->rank0 + i, sr.w_lo, sr.w_hi); range_of(N - 1, W, g->rank0 + i, sr.z_lo, sr.z_hi);
        u32 saved[3];
        for (int k = 0; k < 3; k++) { saved[k] = ctx->fixed_knob[k]; if (!saved[k]) ctx->fixed_knob[k] = plan[k]; }
        rcs[i] = mi_pk_load_range(ctx, device_points ? &descs[i] : d, &spk->part[i], device_points, &sr);
        for (int k = 0; k < 3; k++) ctx->fixed_knob[k] = saved[k];
    });
    for (auto &t : th) t.join();
    // every process reaches the agreement below even when a local part failed (a collective that only some ranks enter would hang)
    int32_t first_bad = MI_OK;
    for (int i = 0; i < nl; i++) if (rcs[i] != MI_OK && first_bad == MI_OK) { g->err = mi_last_error(g->ctx[i]); first_bad = rcs[i]; }
    // all parts on the same plan?  (a part whose tables could not be allocated after all fell back to the generic path by itself)
    std::vector<u64> sig(nl, 0);
    for (int i = 0; i < nl; i++) {
        mi_pk *p = spk->part[i];
        if (p) { p->gen_c_ak = mi_msm_auto_c(max_w); p->gen_c_b = mi_msm_auto_c(max_b); p->gen_c_z = mi_msm_auto_c(max_z); }
        sig[i] = p ? ((u64)1 << 32 | (u64)p->c_ak << 16 | (u64)p->c_b << 8 | (u64)p->c_z) : 0;   // 0 = this part failed to load
    }
    u64 smin = 0, smax = 0;
    int32_t rc = group_min_max(g, sig, &smin, &smax);
    if (first_bad != MI_OK) return fail(first_bad);
    if (rc != MI_OK) return fail(rc);
    if (smin == 0) { g->err = "pk: another rank of the group failed to load its part"; return fail(MI_EHIP); }
    spk->uniform = smin == smax && all_nonempty;
    *out = spk;
    return MI_OK;
}

// One proof over the ranks of the group (groth16.Prove, mt.go:496).  Inputs either in host memory (host = true: W is the WHOLE wire
// vector, a process reads only the ranges of its local ranks; a, b, c are read by the process that holds rank 0) or already on the
// devices (W_dev[i] = the wire range of local rank i on its device; a, b, c on rank 0's device).
// mode 0: per-rank partial sums (option i); mode 1: bucket reduce-scatter before the reduce (option ii).
static int32_t prove_sharded_impl(mi_group *g, mi_pk_sharded *spk, bool host, const mi_fr *W_host, const mi_fr *const *W_dev, size_t n_wires,
                                  const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m,
                                  uint32_t mode, mi_proof_out *out, mi_stats *stats) {
    if (!g || !spk || !r_m || !s_m || !out || mode > 1 || (host ? !W_host : !W_dev)) return MI_EINVAL;
    const int nl = g->n_local(), W = g->world;
    if ((nl != W && nl != 1) || (int)spk->part.size() != nl) G_FAIL(g, MI_EINVAL, "group: a process holds either all ranks or exactly one");
    const bool lead_here = g->rank0 == 0;   // global rank 0 runs computeH and owns a, b, c
    if (lead_here && (!a || !b || !c)) return MI_EINVAL;
    const size_t N = (size_t)1 << spk->log_n;
    if (n_wires != spk->nb_wires || n_constraints > N) G_FAIL(g, MI_EINVAL, "prove: witness size does not match the proving key");
    if (mode == 1 && !spk->uniform) G_FAIL(g, MI_EINVAL, "group: mode 1 needs every part to use the same MSM plan and every rank to hold pairs of every MSM");
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t cb = n_constraints * sizeof(mi_fr);
    // workspaces first, each on its own device: W slice (+ a, b, c on the lead) for host inputs; h (whole on the lead, a slice elsewhere)
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        mi_pk *pk = spk->part[i];
        const bool lead = g->rank0 + i == 0;
        std::memset(&ctx->stats, 0, sizeof(ctx->stats));
        if (host) G_CTX(g, i, mi_reserve(ctx, ctx->ws[16], pk->nb_wires * sizeof(mi_fr) + (lead ? 3 * cb : 0) + 128));
        G_CTX(g, i, mi_reserve(ctx, ctx->ws[14], (lead ? N : pk->n_z_msm + 1) * sizeof(Fr)));
    }
    const bool defer = mode == 1;
    std::vector<int32_t> rcs(nl, MI_OK);
    // every local rank: its slice of W, its wire MSMs; the lead also a, b, c, computeH and its own Z MSM.  One host thread per rank:
    // enqueueing the wire MSMs waits once for the count pass of their sorts (msm.hip, MI_MSM_EXACT_SIZE)
    auto rank_main = [&](int i) -> int32_t {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        mi_pk *pk = spk->part[i];
        hipEvent_t *ev = ctx->ev;
        const bool lead = g->rank0 + i == 0;
        const size_t wb = pk->nb_wires * sizeof(mi_fr);
        const mi_fr *Wd = host ? (const mi_fr *)ctx->ws[16].p : W_dev[i];
        if (!Wd && wb) MI_FAIL(ctx, MI_EINVAL, "prove: null wire slice");
        // host inputs: pageable copies on the context's copy stream (which carries nothing else), ordered by synchronising it on this
        // thread -- no event between two of them (prove.hip, pool.hip: a marker slows every copy behind it)
        const auto t_up = std::chrono::steady_clock::now();
        if (host && wb) {
            MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)Wd, W_host + pk->wire_lo, wb, hipMemcpyHostToDevice, ctx->copy_stream));
            MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        }
        MI_CHECK_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
        MI_TRY(mi_prove_enqueue_wire_msms(ctx, pk, Wd, ev[2], defer));
        if (!lead) return MI_OK;
        // lead: a, b, c arrive while the wire MSMs run; computeH; its own slice of h feeds its Z MSM straight away
        const mi_fr *da = a, *db = b, *dc = c;
        if (host) {
            char *base = (char *)ctx->ws[16].p;
            da = (mi_fr *)(base + wb); db = (mi_fr *)(base + wb + cb); dc = (mi_fr *)(base + wb + 2 * cb);
            if (cb) {
                MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)da, a, cb, hipMemcpyHostToDevice, ctx->copy_stream));
                MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)db, b, cb, hipMemcpyHostToDevice, ctx->copy_stream));
                MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)dc, c, cb, hipMemcpyHostToDevice, ctx->copy_stream));
                MI_CHECK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
            }
            ctx->stats.h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_up).count();
        }
        MI_CHECK_HIP(ctx, hipEventRecord(ev[11], ctx->stream));
        Fr *h = (Fr *)ctx->ws[14].p;
        MI_TRY(mi_compute_h_dev_impl(ctx, pk->log_n, da, db, dc, n_constraints, (mi_fr *)h));
        MI_CHECK_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
        return mi_prove_enqueue_z_msm(ctx, pk, (const mi_fr *)(h + pk->z_lo), ev[3], defer);
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < nl; i++) th.emplace_back([&, i] { rcs[i] = rank_main(i); });
        rcs[0] = rank_main(0);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(g->dev[0]);
    for (int i = 0; i < nl; i++) if (rcs[i] != MI_OK) { g->err = mi_last_error(g->ctx[i]); return rcs[i]; }
    // h: rank 0 hands every other rank its slice device to device, as one batch of the group's transport (grouped ncclSend / ncclRecv,
    // or same-process copies) on the exchange streams; the events that order the Z MSMs behind it are recorded by each RECEIVER on its
    // own stream (an event is recorded only on a stream of the device it was created on)
    if (W > 1) {
        std::vector<Xfer> list;
        for (int j = 1; j < W; j++) {
            u64 zlo, zhi;
            range_of(N - 1, W, j, zlo, zhi);
            Xfer x{0, j, nullptr, nullptr, (size_t)(zhi - zlo) * sizeof(Fr)};
            if (g->local(0)) x.sp = (const char *)g->ctx[0 - g->rank0]->ws[14].p + zlo * sizeof(Fr);
            if (g->local(j)) x.dp = g->ctx[j - g->rank0]->ws[14].p;
            list.push_back(x);
        }
        if (lead_here) { (void)hipSetDevice(g->dev[0]); G_HIP(g, hipStreamWaitEvent(g->xs[0], g->ctx[0]->ev[3], 0)); }
        MI_TRY(run_xfers(g, list, g->xs));
        for (int i = 0; i < nl; i++) {
            if (g->rank0 + i == 0) continue;
            (void)hipSetDevice(g->dev[i]);
            G_HIP(g, hipEventRecord(g->ev_h[i], g->xs[i]));
            G_CTX(g, i, mi_prove_enqueue_z_msm(g->ctx[i], spk->part[i], (const mi_fr *)g->ctx[i]->ws[14].p, g->ev_h[i], defer));
        }
    }
    if (defer) {
        // same order on every rank: A, B1, B2, K, Z
        static const int slots[5] = {0, 1, 2, 3, 4}, curves[5] = {1, 1, 2, 1, 1};
        for (int k = 0; k < 5; k++) MI_TRY(exchange_buckets(g, slots[k], curves[k]));
    }
    ProofAssembler as;
    as.start(spk->part[0], r_m, s_m);
    // collect: per MSM the sum of the ranks' partial results (every process ends with the same five sums)
    G1X sum_a, sum_b1, sum_k, sum_z;
    G2X sum_b2;
    auto collect1 = [&](int slot, G1X *acc) -> int32_t {
        std::vector<G1X> part((size_t)nl);
        for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); G_CTX(g, i, mi_msm_finish(g->ctx[i], slot, 1, &part[i])); }
        return combine_partials<Fp>(g, part, acc);
    };
    auto collect2 = [&](int slot, G2X *acc) -> int32_t {
        std::vector<G2X> part((size_t)nl);
        for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); G_CTX(g, i, mi_msm_finish(g->ctx[i], slot, 2, &part[i])); }
        return combine_partials<Fp2>(g, part, acc);
    };
    MI_TRY(collect1(0, &sum_a));
    MI_TRY(collect1(1, &sum_b1));
    as.have_a_b1(sum_a, sum_b1);
    MI_TRY(collect1(3, &sum_k));
    MI_TRY(collect2(2, &sum_b2));
    MI_TRY(collect1(4, &sum_z));
    for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); G_HIP(g, hipStreamSynchronize(g->ctx[i]->stream)); G_HIP(g, hipStreamSynchronize(g->xs[i])); }
    const auto t_gpu_done = std::chrono::steady_clock::now();
    as.finish(sum_k, sum_b2, sum_z, out);
    const auto t_end = std::chrono::steady_clock::now();
    (void)hipSetDevice(g->dev[0]);
    mi_stats &st = g->ctx[0]->stats;
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<float, std::milli>(y - x).count(); };
    if (lead_here) {
        G_HIP(g, hipEventElapsedTime(&st.compute_h_ms, g->ctx[0]->ev[11], g->ctx[0]->ev[3]));
    }
    st.assemble_ms = ms(t_gpu_done, t_end);
    st.total_ms = ms(t_begin, t_end);
    if (stats) *stats = st;
    return MI_OK;
}

extern "C" {

int32_t mi_pk_load_sharded(mi_group *g, const mi_pk_desc *d, mi_pk_sharded **out) {
    G_ENTER(g);
    return pk_load_sharded_impl(g, d, false, out);
}
int32_t mi_pk_load_sharded_dev(mi_group *g, const mi_pk_desc *slice_descs, mi_pk_sharded **out) {
    G_ENTER(g);
    return pk_load_sharded_impl(g, slice_descs, true, out);
}
int32_t mi_groth16_prove_sharded(mi_group *g, mi_pk_sharded *spk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                                 size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m, uint32_t mode, mi_proof_out *out, mi_stats *stats) {
    G_ENTER(g);
    return prove_sharded_impl(g, spk, true, W, nullptr, n_wires, a, b, c, n_constraints, r_m, s_m, mode, out, stats);
}
int32_t mi_groth16_prove_sharded_dev(mi_group *g, mi_pk_sharded *spk, const mi_fr *const *W_dev, size_t n_wires, const mi_fr *a_dev, const mi_fr *b_dev,
                                     const mi_fr *c_dev, size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m, uint32_t mode, mi_proof_out *out,
                                     mi_stats *stats) {
    G_ENTER(g);
    return prove_sharded_impl(g, spk, false, nullptr, W_dev, n_wires, a_dev, b_dev, c_dev, n_constraints, r_m, s_m, mode, out, stats);
}

}  // extern "C"
