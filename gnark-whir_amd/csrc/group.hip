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
//                         MSM (128 / 256 B XYZZ); combine = world point additions (the partials of all five MSMs travel in ONE
//                         all-gather together with every rank's status).
//                  mode 1 (option ii, the north_star's wording)  every rank stops at its BUCKET sums; rank r owns the keys
//                         [r K / world, (r+1) K / world) and receives that slice from every other rank -- a reduce-scatter written
//                         as grouped ncclSend / ncclRecv, one hop on the full xGMI mesh, all 7 links at once -- adds the slices
//                         (k_msm_sum_slices), runs the bucket reduce on its slice only, and the per-rank results are combined as
//                         in mode 0 (Horner over the windows is linear, so combining after it is the same sum).
//   transport      1  RCCL (ncclCommInitAll in one process, ncclCommInitRank for one rank per process).
//                  2  same-process copies (hipMemcpyPeerAsync / device-to-device): a single-process group that names a device twice
//                     (the 1-GPU rehearsal of the tests) cannot have an RCCL communicator.
//                  3  host-staged: one rank per process, the processes meet in a POSIX shared-memory segment named after the group
//                     id; slices travel device -> segment -> device through per-(source, destination) chunk rings, small values
//                     (status words, partial sums) through an all-gather area.  For ranks that cannot have an RCCL communicator
//                     between them -- two processes on ONE device (how a 1-GPU box rehearses the one-rank-per-process flow: RCCL
//                     refuses two ranks per device, the bookkeeping does not care) or a box without a working RCCL fabric.  Every
//                     wait has a deadline (MI_GROUP_TIMEOUT_MS, default 60 s): a peer that died is an error, not a hang.
//   failures       with one rank per process every entry point is a collective, and a rank that returned early from a local failure
//                  would leave its peers waiting in the next exchange.  So every phase that can fail locally (argument checks,
//                  workspaces, uploads, enqueues) is followed by an AGREEMENT -- an all-gather of the ranks' status words -- before
//                  the next exchange starts, and the last all-gather (the partial sums) carries the status too: either every rank
//                  enters an exchange or every rank returns an error (its own, or "another rank failed") and its MSM slots are drained.
#include "prove_internal.h"
#include "msm_curve_ops.h"
#include "ntt_cross.h"
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

// ---------------------------------------------------------------- transport 3: the shared-memory segment
static constexpr uint32_t SHM_MAGIC = 0x6d693335u, SHM_MAX_WORLD = 64, SHM_AG_MAX = 1024;
struct alignas(64) ShmRing { std::atomic<uint64_t> head; char pad0[56]; std::atomic<uint64_t> tail; char pad1[56]; };   // chunks produced (by the source) / consumed (by the destination)
struct alignas(64) ShmAg { std::atomic<uint64_t> seq; char pad[56]; unsigned char data[2][SHM_AG_MAX]; };              // collective k writes data[k & 1], then seq = k
struct ShmHeader {
    std::atomic<uint32_t> magic;      // set by rank 0 when the header is initialised
    uint32_t world, nslot;
    uint64_t chunk;                   // bytes per ring slot
    std::atomic<uint32_t> attached;   // ranks that have mapped the segment
    std::atomic<uint32_t> poisoned;   // a rank hit a transport error: everyone waiting gives up at once
    std::atomic<uint64_t> beat;       // rank 0 counts here while it waits for the others to attach: a segment whose count stands still has no creator any more
    ShmAg ag[SHM_MAX_WORLD];
    ShmRing ring[SHM_MAX_WORLD * SHM_MAX_WORLD];   // ring[src * world + dst]
};
struct ShmLink {
    ShmHeader *h = nullptr;
    unsigned char *data = nullptr;    // ring r, slot k: data + ((size_t)r * nslot + k) * chunk
    size_t map_bytes = 0;
    uint64_t ag_seq = 0;              // collectives of the all-gather area this rank has entered (all ranks in lockstep)
    int timeout_ms = 60000;
    std::vector<void *> registered;   // the ring regions pinned for the device (hipHostRegister): this rank's row and column
};

struct mi_group {
    int world = 0;                 // ranks in the group
    int rank0 = 0;                 // global rank of the first local context
    int transport = 0;             // 1 = RCCL, 2 = same-process copies, 3 = host-staged (shared memory)
    std::vector<int> dev;          // device of every LOCAL rank
    std::vector<mi_ctx *> ctx;     // one context per local rank
    std::vector<ncclComm_t> comm;  // transport 1: RCCL communicator per local rank
    ShmLink shm;                   // transport 3
    std::vector<hipStream_t> xs;   // per local rank: the exchange stream (sends, receives, slice sums)
    std::vector<hipEvent_t> ev_x, ev_in, ev_done, ev_h;
    std::vector<DevBuf> recv;      // per local rank: bucket slices received from the other ranks
    std::vector<DevBuf> stage;     // per local rank: small staging area for the all-gathers (transport 1)
    // computeH over the ranks (compute_h_sharded): per local rank the cross-rank tables, three slice vectors + the exchange area + the h
    // slice with one slot in front (the previous rank's last coefficient: the Z cut of a sharded key starts one element early), two events
    bool sharded_h = false;        // mi_group_set_sharded_compute_h: mi_groth16_prove_sharded runs computeH over all ranks
    std::vector<CrossNttTables> xt;
    std::vector<DevBuf> hx[3], hy, hy2, hh;
    std::vector<hipEvent_t> ev_c0, ev_c1, ev_r;
    uint32_t lead_share = 0xffffffffu;   // permille of an even wire share that rank 0 -- which also runs computeH -- takes (mi_group_set_lead_share; all ones = automatic)
    int timeout_ms = 60000;        // how long a rank waits for its peers without anything completing (MI_GROUP_TIMEOUT_MS; both transports)
    bool nonblocking = false;      // transport 1, one rank per process: the communicator is non-blocking and every wait on it is a deadline poll
    std::string err;
    bool broken = false;           // a transport call failed half-way: rings / communicator are in an unknown state, every later call is refused
    std::atomic<bool> busy{false};   // calls on one group must not overlap: an entry point that finds it set returns MI_EINVAL (GroupCall)
    int n_local() const { return (int)ctx.size(); }
    bool local(int r) const { return r >= rank0 && r < rank0 + n_local(); }
};
struct mi_pk_sharded {
    std::vector<mi_pk *> part;     // one per local rank
    u32 log_n = 0;
    u64 nb_wires = 0;
    bool uniform = false;          // every local part chose the same MSM plan per group (needed by mode 1)
    uint32_t lead_share = 1000;    // the wire cut this key was loaded with (permille, resolved): a prove under another cut is refused
};

#define G_FAIL(g, code, msg) do { (g)->err = (msg); return (code); } while (0)
#define G_HIP(g, call) do { hipError_t e__ = mi_fault_hit() ? hipErrorUnknown : (call); if (e__ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e__); \
                            return e__ == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP; } } while (0)
#define G_CTX(g, i, expr) do { int32_t rc__ = (expr); if (rc__ != MI_OK) { (g)->err = mi_last_error((g)->ctx[i]); return rc__; } } while (0)

// Entry-point guard: the group's exchange streams, receive buffers and per-rank contexts serve ONE call at a time.  A second call that
// arrives while one is running is refused (MI_EINVAL, the running call's error text is left alone) instead of corrupting g->recv.
struct GroupCall {
    mi_group *g; bool ok;
    explicit GroupCall(mi_group *g_) : g(g_), ok(g_ && !g_->busy.exchange(true, std::memory_order_acquire)) {}
    ~GroupCall() { if (ok) g->busy.store(false, std::memory_order_release); }
};
#define G_ENTER(g) GroupCall call__(g); if (!call__.ok) return MI_EINVAL; \
                   if ((g)->broken) G_FAIL(g, MI_EHIP, "group: an earlier exchange failed half-way; destroy the group and create a new one")

static void range_of(u64 total, int world, int r, u64 &lo, u64 &hi) { lo = total * (u64)r / (u64)world; hi = total * (u64)(r + 1) / (u64)world; }
// The WIRE cut of a sharded key.  Rank 0 (the lead) also runs computeH, which no other rank can help with (NTT = replicas only, SURVEY
// 8e), and the other ranks cannot start their Z MSMs before h exists: giving the lead a smaller share of the wire MSMs shortens the
// proof's critical path (the projection).  share = permille of the even share: the lead takes share / 1000 of
// nb_wires / world wires, the other ranks split the rest evenly; 1000 = the even cut of range_of, bit for bit.  The N - 1 pairs of the
// Z MSM stay cut evenly (Z starts on every rank at the same moment, when h arrives).
// Automatic: 1000 * max(0, 1 - rho (world - 1)) with rho = computeH time / wire-MSM time = 1/2 (measured at N = 2^26 on one MI355X
// with the WHIR witness mix: 60 of 115 ms) -- 1000 for one rank, 500 for two, 0 from three ranks on.
static uint32_t lead_share_of(const mi_group *g) {
    if (g->lead_share <= 1000) return g->lead_share;
    const int w = g->world;
    return w <= 1 ? 1000u : w == 2 ? 500u : 0u;
}
static void wire_range_of(u64 nb_wires, int world, int r, uint32_t share, u64 &lo, u64 &hi) {
    if (share >= 1000 || world <= 1) { range_of(nb_wires, world, r, lo, hi); return; }
    const u64 lead_n = nb_wires * share / (1000ull * (u64)world), rest = nb_wires - lead_n;
    if (r == 0) { lo = 0; hi = lead_n; return; }
    u64 a, b;
    range_of(rest, world - 1, r - 1, a, b);
    lo = lead_n + a; hi = lead_n + b;
}

// ---------------------------------------------------------------- transport 3: shared-memory link
static std::string shm_name_of(const uint8_t id[128]) {
    uint64_t h = 1469598103934665603ull;   // FNV-1a over the 128 id bytes
    for (int i = 0; i < 128; i++) { h ^= id[i]; h *= 1099511628211ull; }
    char buf[64];
    snprintf(buf, sizeof buf, "/mi355x_grp_%016llx", (unsigned long long)h);
    return buf;
}
struct Deadline {
    std::chrono::steady_clock::time_point t;
    explicit Deadline(int ms) : t(std::chrono::steady_clock::now() + std::chrono::milliseconds(ms)) {}
    bool passed() const { return std::chrono::steady_clock::now() > t; }
};
static void shm_pause(unsigned &spins) {
    if (++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
}
// Rank 0 creates and initialises the segment, the others map it as soon as it shows its magic; once every rank is attached rank 0
// unlinks the name, so that nothing outlives the processes whatever way they end.
static int32_t shm_attach(mi_group *g, const uint8_t id[128]) {
    ShmLink &L = g->shm;
    L.timeout_ms = g->timeout_ms;
    uint64_t chunk = (uint64_t)1 << 20;
    uint32_t nslot = 4;
    if (const char *e = getenv("MI_GROUP_SHM_CHUNK_KB")) { const long v = atol(e); if (v >= 4 && v <= (1 << 16)) chunk = (uint64_t)v << 10; }
    const int W = g->world, rank = g->rank0;
    if (W > (int)SHM_MAX_WORLD) G_FAIL(g, MI_EINVAL, "group: the host-staged transport serves at most 64 ranks");
    const size_t hdr = (sizeof(ShmHeader) + 4095) & ~(size_t)4095;
    const std::string name = shm_name_of(id);
    const Deadline dl(L.timeout_ms);
    if (rank == 0) {
        (void)shm_unlink(name.c_str());   // a leftover of a run that died before everyone had attached
        const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) G_FAIL(g, MI_EHIP, "group: shm_open (create) failed");
        L.map_bytes = hdr + (size_t)W * W * nslot * chunk;   // tmpfs allocates a page when it is first touched: rings nobody uses cost nothing
        if (ftruncate(fd, (off_t)L.map_bytes) != 0) { close(fd); (void)shm_unlink(name.c_str()); G_FAIL(g, MI_ENOMEM, "group: ftruncate of the shared segment failed"); }
        void *m = mmap(nullptr, L.map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { (void)shm_unlink(name.c_str()); G_FAIL(g, MI_ENOMEM, "group: mmap of the shared segment failed"); }
        L.h = (ShmHeader *)m;
        L.data = (unsigned char *)m + hdr;
        // (a fresh tmpfs file reads as zeros: rings, sequence numbers and flags start at 0)
        L.h->world = (uint32_t)W; L.h->nslot = nslot; L.h->chunk = chunk;
        L.h->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        // The name may still point at the segment of an EARLIER group with the same id (a run that died before rank 0 unlinked it, or rank
        // 0 of this run has not replaced it yet): its magic and shape look right.  A segment is this group's only if, once its magic shows,
        // the NAME still leads to the same file -- rank 0 unlinks a leftover before it creates, and unlinks its own only after every rank
        // (this one included) has attached.  Anything else is unmapped and the name is tried again until the deadline.
        unsigned spins = 0;
        for (;;) {
            if (dl.passed()) G_FAIL(g, MI_EHIP, "group: rank 0 never created (or initialised) the shared segment (timeout)");
            const int fd = shm_open(name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < hdr) { if (fd >= 0) close(fd); shm_pause(spins); continue; }   // not there / not sized yet
            const size_t bytes = (size_t)sb.st_size;
            void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (m == MAP_FAILED) G_FAIL(g, MI_ENOMEM, "group: mmap of the shared segment failed");
            ShmHeader *h = (ShmHeader *)m;
            bool mine = false;
            while (!dl.passed()) {
                if (h->magic.load(std::memory_order_acquire) == SHM_MAGIC) {
                    const int fd2 = shm_open(name.c_str(), O_RDWR, 0600);
                    struct stat sb2;
                    mine = fd2 >= 0 && fstat(fd2, &sb2) == 0 && sb2.st_ino == sb.st_ino && sb2.st_dev == sb.st_dev;
                    if (fd2 >= 0) close(fd2);
                    if (mine) {   // ... and its creator is alive: rank 0 counts h->beat while it waits for us (a leftover's stands still)
                        const uint64_t b0 = h->beat.load(std::memory_order_acquire);
                        const Deadline alive(L.timeout_ms < 1000 ? L.timeout_ms : 1000);
                        while (h->beat.load(std::memory_order_acquire) == b0 && !alive.passed()) shm_pause(spins);
                        mine = h->beat.load(std::memory_order_acquire) != b0;
                    }
                    break;
                }
                shm_pause(spins);
            }
            if (mine && (int)h->world == W && hdr + (size_t)W * W * h->nslot * h->chunk <= bytes) { L.h = h; L.data = (unsigned char *)m + hdr; L.map_bytes = bytes; break; }
            (void)munmap(m, bytes);
            if (mine) G_FAIL(g, MI_EINVAL, "group: the shared segment belongs to a group of another shape");
            shm_pause(spins);   // a leftover: try the name again
        }
    }
    L.h->attached.fetch_add(1, std::memory_order_acq_rel);
    unsigned spins = 0;
    while ((int)L.h->attached.load(std::memory_order_acquire) < W) {
        if (dl.passed()) { if (rank == 0) (void)shm_unlink(name.c_str()); G_FAIL(g, MI_EHIP, "group: not every rank attached to the shared segment (timeout)"); }
        if (rank == 0) L.h->beat.fetch_add(1, std::memory_order_release);
        shm_pause(spins);
    }
    if (rank == 0) (void)shm_unlink(name.c_str());
    // The rings THIS rank sources (row `rank`, contiguous) or sinks (column `rank`) as pinned memory: copies to and from them run at the
    // bus rate and do not stage a second time (best effort).  Registering pins -- and therefore allocates -- every page it covers: the
    // whole area would be W * W rings for every rank (256 MB at W = 8, 16 GB at 64), these are 2 W - 1.
    {
        const size_t ring_bytes = (size_t)L.h->nslot * L.h->chunk;
        auto reg = [&](size_t first_ring, size_t n_rings) {
            void *p = L.data + first_ring * ring_bytes;
            if (hipHostRegister(p, n_rings * ring_bytes, hipHostRegisterDefault) == hipSuccess) L.registered.push_back(p);
            else (void)hipGetLastError();
        };
        reg((size_t)rank * W, (size_t)W);
        for (int s = 0; s < W; s++) if (s != rank) reg((size_t)s * W + rank, 1);
    }
    return MI_OK;
}
static void shm_detach(mi_group *g) {
    ShmLink &L = g->shm;
    if (!L.h) return;
    for (void *p : L.registered) (void)hipHostUnregister(p);
    L.registered.clear();
    (void)munmap((void *)L.h, L.map_bytes);
    L.h = nullptr; L.data = nullptr;
}
static int32_t shm_fail(mi_group *g, const char *msg) {
    g->shm.h->poisoned.store(1, std::memory_order_release);
    g->broken = true;
    G_FAIL(g, MI_EHIP, msg);
}
// all-gather of `bytes` (<= SHM_AG_MAX) per rank through the segment's all-gather area
static int32_t shm_allgather(mi_group *g, const void *local, size_t bytes, void *all) {
    ShmLink &L = g->shm;
    if (bytes > SHM_AG_MAX) G_FAIL(g, MI_EINVAL, "group: all-gather payload too large");
    const uint64_t k = ++L.ag_seq;
    ShmAg &mine = L.h->ag[g->rank0];
    std::memcpy(mine.data[k & 1], local, bytes);
    mine.seq.store(k, std::memory_order_release);
    const Deadline dl(L.timeout_ms);
    for (int r = 0; r < g->world; r++) {
        unsigned spins = 0;
        // (a rank may be one collective ahead -- it then wrote the OTHER buffer -- never two: collective k + 1 completes only once
        //  every rank has entered it, i.e. has finished reading k)
        while (L.h->ag[r].seq.load(std::memory_order_acquire) < k) {
            if (L.h->poisoned.load(std::memory_order_acquire)) { g->broken = true; G_FAIL(g, MI_EHIP, "group: another rank reported a transport failure"); }
            if (dl.passed()) return shm_fail(g, "group: a rank did not reach the all-gather (timeout; did its process end?)");
            shm_pause(spins);
        }
        std::memcpy((char *)all + (size_t)r * bytes, L.h->ag[r].data[k & 1], bytes);
    }
    return MI_OK;
}

// ---------------------------------------------------------------- transport 1 with one rank per process: deadlines
// The communicator of a per-rank group is NON-BLOCKING (ncclCommInitRankConfig, blocking = 0): no RCCL call may hold this thread for
// longer than it takes to queue work, and every wait -- for the communicator to come up, for a group of sends / receives to be
// issued, for the exchange stream to drain -- is a poll of ncclCommGetAsyncError / hipStreamQuery against MI_GROUP_TIMEOUT_MS.  When
// the deadline passes (a peer's process ended, a link went down) or RCCL reports an asynchronous error, the communicator is ABORTED
// (ncclCommAbort: its kernels leave the stream), the group is marked broken and the call returns an error: a timeout, not a hang.
// Single-process groups (ncclCommInitAll) keep blocking communicators: all their ranks live and die with this process.
static inline bool nccl_ok(ncclResult_t r) { return r == ncclSuccess || r == ncclInProgress; }
static int32_t nccl_abort_all(mi_group *g, const std::string &msg) {
    for (auto &c : g->comm) if (c) { (void)ncclCommAbort(c); c = nullptr; }
    g->broken = true;
    G_FAIL(g, MI_EHIP, msg);
}
// waits until communicator i has left the "in progress" state (the last non-blocking call on it has been carried out)
static int32_t nccl_settle(mi_group *g, int i, const char *what) {
    if (!g->nonblocking) return MI_OK;
    const Deadline dl(g->timeout_ms);
    unsigned spins = 0;
    for (;;) {
        ncclResult_t st = ncclSuccess;
        const ncclResult_t r = ncclCommGetAsyncError(g->comm[i], &st);
        if (r != ncclSuccess) return nccl_abort_all(g, std::string("group: ncclCommGetAsyncError failed ") + what + ": " + ncclGetErrorString(r));
        if (st == ncclSuccess) return MI_OK;
        if (st != ncclInProgress) return nccl_abort_all(g, std::string("group: RCCL reported an asynchronous error ") + what + ": " + ncclGetErrorString(st));
        if (dl.passed()) return nccl_abort_all(g, std::string("group: RCCL did not finish ") + what + " (timeout; did a peer's process end?)");
        shm_pause(spins);
    }
}
// waits for stream s of local rank i to drain, watching the communicator meanwhile
static int32_t nccl_wait_stream(mi_group *g, int i, hipStream_t s, const char *what) {
    if (!g->nonblocking) { G_HIP(g, hipStreamSynchronize(s)); return MI_OK; }
    const Deadline dl(g->timeout_ms);
    unsigned spins = 0;
    bool injected = mi_fault_hit();   // ONE count per wait: the call number an injected failure fires at does not depend on how often this loop polls
    for (;;) {
        const hipError_t q = injected ? hipErrorUnknown : hipStreamQuery(s);
        if (q == hipSuccess) return MI_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return nccl_abort_all(g, std::string("group: the exchange stream failed ") + what + ": " + hipGetErrorString(q)); }
        ncclResult_t st = ncclSuccess;
        if (ncclCommGetAsyncError(g->comm[i], &st) != ncclSuccess || (st != ncclSuccess && st != ncclInProgress))
            return nccl_abort_all(g, std::string("group: RCCL reported an asynchronous error ") + what + ": " + ncclGetErrorString(st));
        if (dl.passed()) return nccl_abort_all(g, std::string("group: the exchange did not complete ") + what + " (timeout; did a peer's process end?)");
        shm_pause(spins);
    }
}

// ---------------------------------------------------------------- point-to-point batches
struct Xfer { int src, dst; const void *sp; void *dp; size_t bytes; };   // global ranks; a pointer is meaningful in its owner's process only
// transport 3: every transfer this process takes part in advances chunk by chunk through ring[src][dst] -- the source copies a chunk
// device -> slot and publishes it (head), the destination copies slot -> device and releases it (tail).  A process both sends and
// receives in one batch and the rings are finite, so the two directions are interleaved in ONE progress loop (two ranks that each
// first sent everything would wait for each other's free slots forever).  Transfers of one (src, dst) pair complete in list order
// (the ring is a FIFO and both sides build the same list).
static int32_t run_xfers_shm(mi_group *g, const std::vector<Xfer> &list, const std::vector<hipStream_t> &xs) {
    ShmLink &L = g->shm;
    const int W = g->world, me = g->rank0;
    const size_t chunk = (size_t)L.h->chunk;
    const uint32_t nslot = L.h->nslot;
    (void)hipSetDevice(g->dev[0]);
    G_HIP(g, hipStreamSynchronize(xs[0]));   // what this rank sends is complete; what it receives into is no longer read
    struct St { size_t sent = 0, rcvd = 0; };
    std::vector<St> st(list.size());
    Deadline dl(L.timeout_ms);   // the time this rank may WAIT for its peers without any chunk moving (renewed by every chunk)
    unsigned spins = 0;
    for (;;) {
        bool all_done = true, progress = false;
        std::vector<char> pair_send_busy((size_t)W * W, 0), pair_recv_busy((size_t)W * W, 0);
        for (size_t k = 0; k < list.size(); k++) {
            const Xfer &x = list[k];
            if (!x.bytes) continue;
            const size_t pr = (size_t)x.src * W + x.dst;
            ShmRing &ring = L.h->ring[pr];
            unsigned char *slots = L.data + pr * nslot * chunk;
            if (x.src == me && st[k].sent < x.bytes) {
                all_done = false;
                if (!pair_send_busy[pr]) {
                    pair_send_busy[pr] = 1;   // later transfers of this pair wait for this one
                    uint64_t head = ring.head.load(std::memory_order_relaxed);
                    while (st[k].sent < x.bytes && head - ring.tail.load(std::memory_order_acquire) < nslot) {
                        const size_t nb = x.bytes - st[k].sent < chunk ? x.bytes - st[k].sent : chunk;
                        G_HIP(g, hipMemcpyAsync(slots + (head % nslot) * chunk, (const char *)x.sp + st[k].sent, nb, hipMemcpyDeviceToHost, xs[0]));
                        G_HIP(g, hipStreamSynchronize(xs[0]));
                        st[k].sent += nb;
                        ring.head.store(++head, std::memory_order_release);
                        progress = true;
                    }
                }
            }
            if (x.dst == me && st[k].rcvd < x.bytes) {
                all_done = false;
                if (!pair_recv_busy[pr]) {
                    pair_recv_busy[pr] = 1;
                    uint64_t tail = ring.tail.load(std::memory_order_relaxed);
                    while (st[k].rcvd < x.bytes && tail < ring.head.load(std::memory_order_acquire)) {
                        const size_t nb = x.bytes - st[k].rcvd < chunk ? x.bytes - st[k].rcvd : chunk;
                        G_HIP(g, hipMemcpyAsync((char *)x.dp + st[k].rcvd, slots + (tail % nslot) * chunk, nb, hipMemcpyHostToDevice, xs[0]));
                        G_HIP(g, hipStreamSynchronize(xs[0]));
                        st[k].rcvd += nb;
                        ring.tail.store(++tail, std::memory_order_release);
                        progress = true;
                    }
                }
            }
        }
        if (all_done) return MI_OK;
        if (progress) { spins = 0; dl = Deadline(L.timeout_ms); continue; }
        if (L.h->poisoned.load(std::memory_order_acquire)) { g->broken = true; G_FAIL(g, MI_EHIP, "group: another rank reported a transport failure"); }
        if (dl.passed()) return shm_fail(g, "group: a peer did not take part in the exchange (timeout; did its process end?)");
        shm_pause(spins);
    }
}
// Runs the batch on the exchange streams xs[local rank].  Afterwards xs[i] is ordered after every transfer rank i sends or receives.
// A failure inside a batch leaves the transport in an unknown state (some ranks may be waiting in it): the group is marked broken.
static int32_t run_xfers_impl(mi_group *g, const std::vector<Xfer> &xs_list, const std::vector<hipStream_t> &xs) {
    const MiRange range_fn("mi.group.exchange");
    if (g->transport == 3) return run_xfers_shm(g, xs_list, xs);
    if (g->transport == 1) {
        // every ncclGroupStart is closed by its ncclGroupEnd whatever happens in between (an open group would swallow this thread's next
        // RCCL calls); the first failure is reported after the group has been closed
        ncclResult_t bad = ncclSuccess;
        const char *where = "";
        ncclResult_t r = ncclGroupStart();
        if (!nccl_ok(r)) { g->err = std::string("ncclGroupStart: ") + ncclGetErrorString(r); return MI_EHIP; }
        for (const Xfer &x : xs_list) {
            if (!x.bytes || bad != ncclSuccess) continue;
            if (g->local(x.src)) {
                (void)hipSetDevice(g->dev[x.src - g->rank0]);
                r = mi_fault_hit() ? ncclInternalError : ncclSend(x.sp, x.bytes, ncclUint8, x.dst, g->comm[x.src - g->rank0], xs[x.src - g->rank0]);
                if (!nccl_ok(r)) { bad = r; where = "ncclSend"; continue; }
            }
            if (g->local(x.dst)) {
                (void)hipSetDevice(g->dev[x.dst - g->rank0]);
                r = ncclRecv(x.dp, x.bytes, ncclUint8, x.src, g->comm[x.dst - g->rank0], xs[x.dst - g->rank0]);
                if (!nccl_ok(r)) { bad = r; where = "ncclRecv"; }
            }
        }
        r = ncclGroupEnd();
        if (bad == ncclSuccess && !nccl_ok(r)) { bad = r; where = "ncclGroupEnd"; }
        if (bad != ncclSuccess) {
            const std::string msg = std::string("group: ") + where + ": " + ncclGetErrorString(bad);
            if (g->nonblocking) return nccl_abort_all(g, msg);   // part of the batch may be queued: the communicator cannot be used again
            G_FAIL(g, MI_EHIP, msg);
        }
        if (g->nonblocking) {
            // one rank per process: the batch is issued AND complete (or the group is broken) before anything else is built on it -- a peer
            // that ended mid-exchange surfaces here, within the deadline, not in some later synchronisation without one
            MI_TRY(nccl_settle(g, 0, "while issuing an exchange"));
            MI_TRY(nccl_wait_stream(g, 0, xs[0], "in an exchange"));
        }
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
static int32_t run_xfers(mi_group *g, const std::vector<Xfer> &xs_list, const std::vector<hipStream_t> &xs) {
    const int32_t rc = run_xfers_impl(g, xs_list, xs);
    if (rc != MI_OK && g->n_local() != g->world) {   // the peers may be inside the exchange this rank just left
        g->broken = true;
        if (g->transport == 3) g->shm.h->poisoned.store(1, std::memory_order_release);
    }
    return rc;
}

// ---------------------------------------------------------------- all-gather of small host values, agreement
// local: n_local x bytes (this process's ranks, in order); all: world x bytes in rank order.  Single process: a copy.  One rank
// per process: ncclAllGather(ncclUint8) through a staging area (transport 1) or the shared segment (transport 3).
static int32_t group_allgather(mi_group *g, const void *local, size_t bytes, void *all) {
    const MiRange range_fn("mi.group.allgather");
    // (a per-rank RCCL group of ONE rank goes through its communicator all the same: that is how a 1-GPU box runs the polled path)
    if (g->n_local() == g->world && !g->nonblocking) { std::memcpy(all, local, bytes * (size_t)g->world); return MI_OK; }
    if (g->n_local() != 1) G_FAIL(g, MI_EINVAL, "group: a process holds either all ranks or exactly one");
    int32_t rc;
    if (g->transport == 3) rc = shm_allgather(g, local, bytes, all);
    else if (g->transport != 1 || g->comm.size() != 1) G_FAIL(g, MI_EINVAL, "group: no transport between the processes of this group");
    else rc = [&]() -> int32_t {
        (void)hipSetDevice(g->dev[0]);
        G_CTX(g, 0, mi_reserve(g->ctx[0], g->stage[0], bytes * (size_t)(g->world + 1)));
        char *st = (char *)g->stage[0].p;
        hipStream_t s = g->xs[0];
        G_HIP(g, hipMemcpyAsync(st, local, bytes, hipMemcpyHostToDevice, s));
        const ncclResult_t r = mi_fault_hit() ? ncclInternalError : ncclAllGather(st, st + bytes, bytes, ncclUint8, g->comm[0], s);
        if (!nccl_ok(r)) {
            const std::string msg = std::string("group: ncclAllGather: ") + ncclGetErrorString(r);
            if (g->nonblocking) return nccl_abort_all(g, msg);
            G_FAIL(g, MI_EHIP, msg);
        }
        MI_TRY(nccl_settle(g, 0, "while issuing an all-gather"));
        G_HIP(g, hipMemcpyAsync(all, st + bytes, bytes * (size_t)g->world, hipMemcpyDeviceToHost, s));
        MI_TRY(nccl_wait_stream(g, 0, s, "in an all-gather"));
        return MI_OK;
    }();
    if (rc != MI_OK) g->broken = true;   // some ranks may have got through, others not: nothing collective can follow
    return rc;
}
// What every rank says before the group enters an exchange: its status and up to ten values all ranks must agree on.
struct Agree { int32_t rc; uint32_t check_n; uint64_t check[10]; };
// local_rc / local_err: the first failure among this process's ranks (MI_OK: none).  Every rank of the group returns MI_OK or every
// rank returns an error: the failing rank its own, the others "another rank failed".  check[0..check_n): values that must be EQUAL on
// all ranks (bucket layouts of an exchange).
static int32_t group_agree(mi_group *g, int32_t local_rc, const std::string &local_err, const char *phase, const uint64_t *check = nullptr, uint32_t check_n = 0) {
    std::vector<Agree> mine((size_t)g->n_local()), all((size_t)g->world);
    for (auto &a : mine) {
        std::memset(&a, 0, sizeof a);
        a.rc = local_rc; a.check_n = check_n;
        for (uint32_t k = 0; k < check_n && k < 10; k++) a.check[k] = check[k];
    }
    MI_TRY(group_allgather(g, mine.data(), sizeof(Agree), all.data()));
    for (int r = 0; r < g->world; r++)
        if (all[r].rc != MI_OK) {
            if (local_rc != MI_OK) { g->err = local_err; return local_rc; }
            g->err = std::string("group: rank ") + std::to_string(r) + " failed " + phase + " (status " + std::to_string(all[r].rc) + "); nothing was exchanged";
            return all[r].rc;
        }
    for (int r = 1; r < g->world; r++)
        if (all[r].check_n != all[0].check_n || std::memcmp(all[r].check, all[0].check, sizeof all[0].check) != 0)
            G_FAIL(g, MI_EINVAL, std::string("group: the ranks disagree on the shape of the exchange ") + phase);
    return MI_OK;
}
// Group-wide minimum and maximum of one 64-bit value per local rank (plans every rank must agree on: table budgets, window widths).
static int32_t group_min_max(mi_group *g, const std::vector<u64> &local, u64 *mn, u64 *mx) {
    std::vector<u64> all((size_t)g->world, 0);
    MI_TRY(group_allgather(g, local.data(), 8, all.data()));
    *mn = ~(u64)0; *mx = 0;
    for (u64 v : all) { if (v < *mn) *mn = v; if (v > *mx) *mx = v; }
    return MI_OK;
}

// ---------------------------------------------------------------- lifecycle
static int32_t group_finish_init(mi_group *g) {
    const int n = g->n_local();
    g->xs.assign(n, nullptr); g->ev_x.assign(n, nullptr); g->ev_in.assign(n, nullptr); g->ev_done.assign(n, nullptr); g->ev_h.assign(n, nullptr);
    g->recv.assign(n, DevBuf{}); g->stage.assign(n, DevBuf{});
    g->xt.assign(n, CrossNttTables{}); g->hy.assign(n, DevBuf{}); g->hy2.assign(n, DevBuf{}); g->hh.assign(n, DevBuf{}); g->ev_c0.assign(n, nullptr); g->ev_c1.assign(n, nullptr);
    g->ev_r.assign((size_t)n * 7, nullptr);   // computeH over the ranks: seven "ready" events a local rank, created by compute_h_sharded_prepare
    for (auto &v : g->hx) v.assign(n, DevBuf{});
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(g->dev[i]);
        G_HIP(g, hipEventCreateWithFlags(&g->ev_c0[i], hipEventDisableTiming));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_c1[i], hipEventDisableTiming));
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
        // (a broken group's communicator was aborted where it broke; one that is merely unused is destroyed -- on a non-blocking
        //  communicator neither call holds this thread)
        if (i < (int)g->comm.size() && g->comm[i]) (void)(g->broken ? ncclCommAbort(g->comm[i]) : ncclCommDestroy(g->comm[i]));
        for (auto *v : {&g->ev_x, &g->ev_in, &g->ev_done, &g->ev_h}) if (i < (int)v->size() && (*v)[i]) (void)hipEventDestroy((*v)[i]);
        for (auto *v : {&g->ev_c0, &g->ev_c1}) if (i < (int)v->size() && (*v)[i]) (void)hipEventDestroy((*v)[i]);
        for (size_t k = (size_t)i * 7; k < (size_t)(i + 1) * 7 && k < g->ev_r.size(); k++) if (g->ev_r[k]) (void)hipEventDestroy(g->ev_r[k]);
        for (auto *v : {&g->hx[0], &g->hx[1], &g->hx[2], &g->hy, &g->hy2, &g->hh}) if (i < (int)v->size() && (*v)[i].p) (void)hipFree((*v)[i].p);
        if (i < (int)g->xt.size()) mi_cross_tables_free(&g->xt[i]);
        if (i < (int)g->recv.size() && g->recv[i].p) (void)hipFree(g->recv[i].p);
        if (i < (int)g->stage.size() && g->stage[i].p) (void)hipFree(g->stage[i].p);
    }
    shm_detach(g);
    for (int i = 0; i < g->n_local(); i++) if (g->ctx[i]) { (void)hipSetDevice(g->dev[i]); mi_shutdown(g->ctx[i]); }
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
    g->transport = distinct ? 1 : 2;
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
int32_t mi_group_create_rank_ex(int device_id, int rank, int world, const uint8_t id[128], int transport, mi_group **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world || (transport != MI_GROUP_TRANSPORT_RCCL && transport != MI_GROUP_TRANSPORT_HOST)) return MI_EINVAL;
    *out = nullptr;
    mi_group *g = new (std::nothrow) mi_group();
    if (!g) return MI_ENOMEM;
    g->world = world; g->rank0 = rank; g->transport = transport;
    if (const char *e = getenv("MI_GROUP_TIMEOUT_MS")) { const int v = atoi(e); if (v > 0) g->timeout_ms = v; }
    mi_ctx *c = nullptr;
    int32_t rc = mi_init(device_id, &c);
    if (rc != MI_OK) { delete g; return rc; }
    g->dev.push_back(device_id); g->ctx.push_back(c);
    rc = group_finish_init(g);
    if (rc == MI_OK && transport == MI_GROUP_TRANSPORT_RCCL) {
        ncclUniqueId u;
        std::memcpy(&u, id, 128);
        (void)hipSetDevice(device_id);
        g->comm.assign(1, nullptr);
        ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
        cfg.blocking = 0;   // (see "transport 1 with one rank per process: deadlines")
        g->nonblocking = true;
        const ncclResult_t r = ncclCommInitRankConfig(&g->comm[0], world, u, rank, &cfg);
        if (!nccl_ok(r) || !g->comm[0]) { g->comm.clear(); g->err = std::string("ncclCommInitRankConfig: ") + ncclGetErrorString(r); rc = MI_EHIP; }
        else rc = nccl_settle(g, 0, "while the communicator was coming up (every rank of the group must call mi_group_create_rank)");
    }
    if (rc == MI_OK && transport == MI_GROUP_TRANSPORT_HOST) { (void)hipSetDevice(device_id); rc = shm_attach(g, id); }
    if (rc != MI_OK) { mi_group_destroy(g); return rc; }
    *out = g;
    return MI_OK;
}
int32_t mi_group_create_rank(int device_id, int rank, int world, const uint8_t id[128], mi_group **out) {
    return mi_group_create_rank_ex(device_id, rank, world, id, MI_GROUP_TRANSPORT_RCCL, out);
}
int32_t mi_group_set_lead_share(mi_group *g, uint32_t permille) {
    if (!g || (permille > 1000 && permille != MI_LEAD_SHARE_AUTO)) return MI_EINVAL;
    G_ENTER(g);
    g->lead_share = permille;
    return MI_OK;
}
int32_t mi_group_wire_range(const mi_group *g, uint64_t nb_wires, int rank, uint64_t *lo, uint64_t *hi) {
    if (!g || !lo || !hi || rank < 0 || rank >= g->world) return MI_EINVAL;
    u64 a, b;
    wire_range_of(nb_wires, g->world, rank, lead_share_of(g), a, b);
    *lo = a; *hi = b;
    return MI_OK;
}
int32_t mi_group_world(const mi_group *g) { return g ? g->world : 0; }
int32_t mi_group_rank(const mi_group *g) { return g ? g->rank0 : -1; }
int32_t mi_group_local(const mi_group *g) { return g ? g->n_local() : 0; }
mi_ctx *mi_group_ctx(mi_group *g, int local_rank) { return g && local_rank >= 0 && local_rank < g->n_local() ? g->ctx[local_rank] : nullptr; }
const char *mi_group_last_error(mi_group *g) { return g ? g->err.c_str() : "null group"; }
// 1 = RCCL communicator, 2 = copies inside this process (a device named twice), 3 = host-staged through shared memory
int32_t mi_group_transport(const mi_group *g) { return !g ? 0 : g->transport; }
// What the communicator ITSELF says about the group (observed, not derived from the arguments of mi_group_create_rank): the number of
// ranks RCCL joined (ncclCommCount of this process's first communicator), 0 when the transport is not RCCL, negative on an RCCL error.
int32_t mi_group_comm_ranks(const mi_group *g) {
    if (!g) return MI_EINVAL;
    if (g->transport != MI_GROUP_TRANSPORT_RCCL || g->comm.empty() || !g->comm[0]) return 0;
    int n = 0;
    return ncclCommCount(g->comm[0], &n) == ncclSuccess ? n : MI_EHIP;
}
// PCI bus id ("0000:c1:00.0") of the device local rank `local_rank` runs on, as the HIP runtime reports it: what tells two ranks on two
// GPUs from two ranks on one.
int32_t mi_group_device_pci(const mi_group *g, int local_rank, char out[32]) {
    if (!g || !out || local_rank < 0 || local_rank >= g->n_local()) return MI_EINVAL;
    out[0] = 0;
    return hipDeviceGetPCIBusId(out, 32, g->dev[local_rank]) == hipSuccess ? MI_OK : MI_EHIP;
}

// Every local rank sends a distinct pattern of `bytes` bytes to every rank of the group (itself included) and checks what it
// received: the transport (RCCL grouped send / recv, peer copies, or the shared-memory rings) in isolation, then the all-gather of
// host values the agreements and the partial-sum combine use.  All ranks of the group call it together.
int32_t mi_group_exchange_selftest(mi_group *g, size_t bytes) {
    if (!g || !bytes || bytes > ((size_t)1 << 28)) return MI_EINVAL;
    G_ENTER(g);
    const int nl = g->n_local(), W = g->world;
    std::vector<void *> sbuf(nl, nullptr), rbuf(nl, nullptr);
    std::vector<hipStream_t> xs(nl);
    std::vector<Xfer> list;
    auto pat = [](int src, int dst, size_t k) { return (unsigned char)(17 * src + 101 * dst + 3 * k + (k >> 8)); };
    int32_t rc = MI_OK;
    auto prepare = [&]() -> int32_t {
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
        return MI_OK;
    };
    auto verify = [&]() -> int32_t {
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
    auto body = [&]() -> int32_t {
        int32_t lrc = prepare();
        MI_TRY(group_agree(g, lrc, g->err, "while preparing the self-test"));   // a rank without buffers must not leave the others in the exchange
        for (int s = 0; s < W; s++) for (int d = 0; d < W; d++) {
            Xfer x{s, d, nullptr, nullptr, bytes};
            if (g->local(s)) x.sp = (char *)sbuf[s - g->rank0] + (size_t)d * bytes;
            if (g->local(d)) x.dp = (char *)rbuf[d - g->rank0] + (size_t)s * bytes;
            list.push_back(x);
        }
        MI_TRY(run_xfers(g, list, xs));
        lrc = verify();
        // the all-gather of host values: every rank contributes (rank, 3 rank + 1) and must read that back from everyone
        std::vector<u64> mine(2 * (size_t)nl), all(2 * (size_t)W);
        for (int i = 0; i < nl; i++) { mine[2 * i] = (u64)(g->rank0 + i); mine[2 * i + 1] = 3 * (u64)(g->rank0 + i) + 1; }
        MI_TRY(group_allgather(g, mine.data(), 16, all.data()));
        for (int r = 0; r < W && lrc == MI_OK; r++)
            if (all[2 * r] != (u64)r || all[2 * r + 1] != 3 * (u64)r + 1) { g->err = "group: all-gather self-test received wrong values"; lrc = MI_EHIP; }
        return group_agree(g, lrc, g->err, "in the transport self-test");
    };
    rc = body();
    for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); if (sbuf[i]) (void)hipFree(sbuf[i]); if (rbuf[i]) (void)hipFree(rbuf[i]); }
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------- mode 1: reduce-scatter of the bucket sums of MSM slots
// Every local rank has enqueued the MSMs of `slots` with MI_MSM_DEFER_REDUCE.  prepare (local, may fail): bucket views, receive
// buffer, the layout values the ranks must agree on.  run (after the agreement): ONE batch of the transport for all the slots,
// then per slot the sum of the received slices and the reduce.  Local failures inside run are returned AFTER the rank has taken part
// in the batch (the peers are in it).
struct BucketExchange {
    std::vector<int> slots, curves;
    std::vector<std::vector<MsmBucketView>> v;   // [slot index][local rank]
    std::vector<size_t> off;                     // receive-buffer offset of every slot's region
};
static int32_t exchange_prepare(mi_group *g, BucketExchange &bx, const int *slots, const int *curves, int n_slots, std::vector<uint64_t> &check) {
    const int nl = g->n_local(), W = g->world;
    bx.slots.assign(slots, slots + n_slots); bx.curves.assign(curves, curves + n_slots);
    bx.v.assign(n_slots, std::vector<MsmBucketView>(nl));
    bx.off.assign(n_slots + 1, 0);
    for (int k = 0; k < n_slots; k++) {
        const MsmCurveOps &ops = mi_msm_ops(curves[k]);
        for (int i = 0; i < nl; i++) {
            G_CTX(g, i, mi_msm_bucket_view(g->ctx[i], slots[k], curves[k], &bx.v[k][i]));
            if (!bx.v[k][i].bucket) G_FAIL(g, MI_EINVAL, "group: mode 1 (bucket exchange): a rank has no bucket array for one of the MSMs");
            if (bx.v[k][i].nkeys != bx.v[k][0].nkeys || bx.v[k][i].c != bx.v[k][0].c) G_FAIL(g, MI_EINVAL, "group: the ranks disagree on the bucket layout of an MSM");
        }
        const size_t K = bx.v[k][0].nkeys, own_max = (K + W - 1) / W + 1;
        bx.off[k + 1] = bx.off[k] + (size_t)(W > 1 ? W - 1 : 1) * own_max * ops.xyzz_bytes;
        check.push_back(((uint64_t)K << 8) | bx.v[k][0].c);
    }
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        // the previous batch's slice sums still read the buffer on xs[i]: growing it (hipFree) synchronises the device, reusing it is ordered by xs[i]
        G_CTX(g, i, mi_reserve(g->ctx[i], g->recv[i], bx.off[n_slots] + 256));
    }
    return MI_OK;
}
static int32_t exchange_run(mi_group *g, const BucketExchange &bx, int first, int count) {
    const int nl = g->n_local(), W = g->world;
    int32_t local_rc = MI_OK;
    std::string local_err;
    auto note = [&](int32_t rc, const std::string &msg) { if (rc != MI_OK && local_rc == MI_OK) { local_rc = rc; local_err = msg; } };
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        for (int k = first; k < first + count; k++) {
            const hipError_t e = hipStreamWaitEvent(g->xs[i], bx.v[k][i].ready, 0);
            if (e != hipSuccess) note(MI_EHIP, std::string("hipStreamWaitEvent (bucket sums ready): ") + hipGetErrorString(e));
        }
    }
    std::vector<Xfer> list;
    for (int k = first; k < first + count; k++) {
        const size_t K = bx.v[k][0].nkeys, B = mi_msm_ops(bx.curves[k]).xyzz_bytes;
        for (int s = 0; s < W; s++) for (int d = 0; d < W; d++) {
            if (s == d) continue;
            u64 lo, hi;
            range_of(K, W, d, lo, hi);
            Xfer x{s, d, nullptr, nullptr, (size_t)(hi - lo) * B};
            if (g->local(s)) x.sp = (const char *)bx.v[k][s - g->rank0].bucket + lo * B;
            if (g->local(d)) x.dp = (char *)g->recv[d - g->rank0].p + bx.off[k] + (size_t)(s < d ? s : s - 1) * (hi - lo) * B;
            list.push_back(x);
        }
    }
    MI_TRY(run_xfers(g, list, g->xs));   // a transport failure: the group is broken, nothing below matters
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        for (int k = first; k < first + count; k++) {
            const MsmCurveOps &ops = mi_msm_ops(bx.curves[k]);
            const size_t K = bx.v[k][0].nkeys, B = ops.xyzz_bytes;
            const auto post = [&]() -> int32_t {
                u64 lo, hi;
                range_of(K, W, g->rank0 + i, lo, hi);
                char *bk = (char *)bx.v[k][i].bucket;
                ops.sum_slices(g->xs[i], bk + lo * B, (const char *)g->recv[i].p + bx.off[k], (u32)(W - 1), (u32)(hi - lo));
                G_HIP(g, hipGetLastError());
                // keys of other owners: their sums live there now; here they read as infinity for the reduce
                if (lo) G_HIP(g, hipMemsetAsync(bk, 0, lo * B, g->xs[i]));
                if (hi < K) G_HIP(g, hipMemsetAsync(bk + hi * B, 0, (K - hi) * B, g->xs[i]));
                G_HIP(g, hipEventRecord(g->ev_done[i], g->xs[i]));
                G_HIP(g, hipStreamWaitEvent(bx.v[k][i].stream, g->ev_done[i], 0));
                G_CTX(g, i, mi_msm_reduce_enqueue(g->ctx[i], bx.slots[k], bx.curves[k]));
                return MI_OK;
            };
            const int32_t rc = post();
            note(rc, g->err);
        }
    }
    if (local_rc != MI_OK) g->err = local_err;
    return local_rc;
}

// Whatever a failed call left on a context's MSM slots is collected (a deferred slot runs its reduce over its local buckets first), so
// that the next call finds every slot idle.  Errors here change nothing any more.
static void drain_slots(mi_group *g, const int *slots, const int *curves, int n) {
    for (int i = 0; i < g->n_local(); i++) {
        (void)hipSetDevice(g->dev[i]);
        G1X t1; G2X t2;
        for (int k = 0; k < n; k++) {
            (void)mi_msm_reduce_enqueue(g->ctx[i], slots[k], curves[k]);
            (void)mi_msm_finish(g->ctx[i], slots[k], curves[k], curves[k] == 1 ? (void *)&t1 : (void *)&t2);
        }
        (void)hipStreamSynchronize(g->ctx[i]->stream);
        (void)hipStreamSynchronize(g->xs[i]);
    }
}

template <class F, class JacT>
static void write_jac(const XYZZ<F> &r, JacT *out) {
    Jac<F> j;
    if (r.is_inf()) j = Jac<F>{F::one(), F::one(), F::zero()};
    else { Affine<F> a = xyzz_to_affine(r); j = Jac<F>{a.x, a.y, F::one()}; }
    std::memcpy(out, &j, sizeof(j));
}
// runs fn(i) for every local rank, rank 0 on this thread and the others on threads of their own (an MSM enqueue waits once on the
// host for its sort's largest bucket, msm.hip: one after the other, rank i + 1 would not even start its sort before rank i's has
// counted); returns the first failure in rank order with its text in *err
template <class Fn>
static int32_t for_each_local_rank(mi_group *g, std::string *err, Fn fn) {
    const int nl = g->n_local();
    std::vector<int32_t> rcs((size_t)nl, MI_OK);
    auto guarded = [&](int i) { try { (void)hipSetDevice(g->dev[i]); rcs[i] = fn(i); } catch (...) { rcs[i] = MI_ENOMEM; } };
    std::vector<std::thread> th;
    for (int i = 1; i < nl; i++) {
        try { th.emplace_back(guarded, i); } catch (...) { guarded(i); }   // no thread to be had: in line
    }
    guarded(0);
    for (auto &t : th) t.join();
    (void)hipSetDevice(g->dev[0]);
    for (int i = 0; i < nl; i++) if (rcs[i] != MI_OK) { *err = mi_last_error(g->ctx[i]); return rcs[i]; }
    return MI_OK;
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
    const int slots[1] = {0}, curves[1] = {curve};
    std::string lerr;
    int32_t lrc = for_each_local_rank(g, &lerr, [&](int i) -> int32_t {
        mi_ctx *ctx = g->ctx[i];
        std::memset(&ctx->stats, 0, sizeof(ctx->stats));
        MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        return mi_msm_enqueue(ctx, 0, -1, curve, pts_dev[i], sc_dev[i], n_local[i], flags | df, ctx->ev[0], curve == 1, 0, 0, c);
    });
    BucketExchange bx;
    std::vector<uint64_t> check{(uint64_t)n_total, (uint64_t)mode, (uint64_t)flags};
    if (mode == 1 && lrc == MI_OK) { lrc = exchange_prepare(g, bx, slots, curves, 1, check); lerr = g->err; }
    int32_t rc = group_agree(g, lrc, lerr, "while enqueueing its share of the MSM", check.data(), (uint32_t)check.size());
    if (rc == MI_OK && mode == 1) { lrc = exchange_run(g, bx, 0, 1); lerr = g->err; if (g->broken) rc = lrc; }
    // the partial sums and the status of every rank in ONE all-gather: all ranks return the sum, or all return an error
    struct Part { int32_t rc; uint32_t pad[3]; XYZZ<F> p; };
    std::vector<Part> mine((size_t)nl), all((size_t)g->world);
    if (rc == MI_OK) {
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            std::memset(&mine[i], 0, sizeof(Part));
            const int32_t r = lrc != MI_OK ? lrc : mi_msm_finish(g->ctx[i], 0, curve, &mine[i].p);
            if (r != MI_OK && lrc == MI_OK) { lrc = r; lerr = mi_last_error(g->ctx[i]); }
            mine[i].rc = lrc;
        }
        rc = group_allgather(g, mine.data(), sizeof(Part), all.data());
        if (rc == MI_OK)
            for (int r = 0; r < g->world && rc == MI_OK; r++)
                if (all[r].rc != MI_OK) {
                    rc = all[r].rc;
                    g->err = lrc != MI_OK ? lerr : std::string("group: rank ") + std::to_string(r) + " failed in its share of the MSM (status " + std::to_string(rc) + ")";
                    if (lrc != MI_OK) rc = lrc;
                }
    }
    if (rc != MI_OK) { const std::string keep = g->err; drain_slots(g, slots, curves, 1); g->err = keep; return rc; }
    XYZZ<F> total = XYZZ<F>::inf();
    for (const Part &p : all) xyzz_add(total, p.p);
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

// ---------------------------------------------------------------- computeH over the ranks of the group (csrc/ntt_cross.hip has the maths)
// Data: rank r holds the natural-order slice [r M, (r + 1) M) of a, b (and c) in g->hx[0..2][local], M = N / world.  Result: the
// coefficients of h in gnark's bit-reversed order, slice r on rank r, in g->hh[local] + 1 (slot 0 = the previous rank's last coefficient,
// fetched by one more batch: the Z pairs of a sharded key are cut over N - 1, so every slice but the first starts one element early).
// Six transforms as on one device (h = den FFT^-1_coset(ca cb) - den FFT^-1(c)); each is a local size-M transform and one cross-rank
// step between two all-to-alls over the group's transport -- 9 batches of (world - 1) / world of a slice per rank: the pair between
// the coset transforms of a, b and the last transform cancels.
// Local failures are CARRIED THROUGH every batch (the peers are in them) and returned at the end; a transport failure breaks the group.
static bool sharded_h_possible(const mi_group *g, u32 log_n) {
    const int W = g->world;
    if (W < 2 || W > 16 || (W & (W - 1))) return false;
    u32 lw = 0;
    while ((1 << lw) < W) lw++;
    return log_n >= 2 * lw && log_n <= 28;
}
// Seven "ready" events per local rank: a transfer waits for the ONE kernel that produced what it moves, not for whatever else has
// been enqueued on the context's stream since -- that is what lets the next vector's arithmetic run under this vector's transfer.
enum { E_IN, E_DA, E_DB, E_DC, E_FA, E_FB, E_MID, E_COUNT };
// Everything compute_h_sharded allocates -- the cross-rank tables, the two exchange vectors, the h slice, the events -- for ONE local rank.
// Called in the LOCAL phase of the callers, before the group agrees to proceed: an allocation that fails here is an MI_ENOMEM
// every rank returns with the group intact, not a null pointer under kernels and transfers the peers are already waiting for.
static int32_t compute_h_sharded_prepare(mi_group *g, int i, u32 log_n) {
    const int nl = g->n_local();
    u32 log_w = 0;
    while ((1 << log_w) < g->world) log_w++;
    const size_t M = (size_t)1 << (log_n - log_w);
    mi_ctx *c = g->ctx[i];
    MI_TRY(mi_cross_tables_build(c, log_n, log_w, (u32)(g->rank0 + i), &g->xt[i]));
    MI_TRY(mi_reserve(c, g->hy[i], M * sizeof(Fr)));
    MI_TRY(mi_reserve(c, g->hy2[i], M * sizeof(Fr)));
    MI_TRY(mi_reserve(c, g->hh[i], (M + 1) * sizeof(Fr)));
    if (g->ev_r.size() != (size_t)nl * E_COUNT) MI_FAIL(c, MI_EINVAL, "sharded computeH: the group's event table was not sized at creation");
    for (int k = 0; k < E_COUNT; k++) {   // (slots of local rank i only: local ranks prepare from their own threads)
        hipEvent_t &e = g->ev_r[(size_t)i * E_COUNT + k];
        if (!e) MI_CHECK_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    return MI_OK;
}
static bool compute_h_sharded_prepared(const mi_group *g, int i) {
    if (g->ev_r.size() != (size_t)g->n_local() * E_COUNT || !g->hy[i].p || !g->hy2[i].p || !g->hh[i].p || !g->xt[i].s_fwd || !g->xt[i].s_inv) return false;
    for (int k = 0; k < E_COUNT; k++) if (!g->ev_r[(size_t)i * E_COUNT + k]) return false;
    return true;
}
static int32_t compute_h_sharded(mi_group *g, u32 log_n, bool derive_c) {
    const MiRange range_fn("mi.group.computeH.enqueue");
    const int nl = g->n_local(), W = g->world;
    u32 log_w = 0;
    while ((1 << log_w) < W) log_w++;
    const u32 log_m = log_n - log_w;
    const size_t M = (size_t)1 << log_m, cnt = M >> log_w, row = cnt * sizeof(Fr);
    int32_t lrc = MI_OK;
    std::string lerr;
    auto note = [&](int32_t rc, const std::string &msg) { if (rc != MI_OK && lrc == MI_OK) { lrc = rc; lerr = msg; } };
    auto each = [&](const std::function<int32_t(int, mi_ctx *)> &fn) {
        for (int i = 0; i < nl; i++) { (void)hipSetDevice(g->dev[i]); const int32_t rc = fn(i, g->ctx[i]); note(rc, mi_last_error(g->ctx[i])); }
    };
    // the callers prepared every local rank and the GROUP agreed on the outcome: nothing is allocated from here on.  (A caller that skipped
    // that would have kernels and transfers run on null buffers: refused before anything is enqueued.)
    for (int i = 0; i < nl; i++) if (!compute_h_sharded_prepared(g, i)) { g->err = "sharded computeH: buffers were not prepared before the agreement"; return MI_EINVAL; }
    auto mark = [&](int k) {
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            hipEvent_t e = g->ev_r[(size_t)i * E_COUNT + k];
            if (!e || hipEventRecord(e, g->ctx[i]->stream) != hipSuccess) note(MI_EHIP, "sharded computeH: hipEventRecord failed");
        }
    };
    // all-to-all: block d of the source rank's buffer -> block s of rank d's buffer (the same rule in both directions); the exchange
    // streams pick up behind event `after` of their context's stream and hand back to it when the batch is done
    auto a2a = [&](std::vector<DevBuf> &from, std::vector<DevBuf> &to, int after) -> int32_t {
        std::vector<Xfer> list;
        for (int s = 0; s < W; s++) for (int d = 0; d < W; d++) {
            Xfer x{s, d, nullptr, nullptr, row};
            if (g->local(s)) x.sp = (const char *)from[s - g->rank0].p + (size_t)d * row;
            if (g->local(d)) x.dp = (char *)to[d - g->rank0].p + (size_t)s * row;
            list.push_back(x);
        }
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            hipEvent_t e = g->ev_r[(size_t)i * E_COUNT + after];
            if (!e || hipStreamWaitEvent(g->xs[i], e, 0) != hipSuccess) note(MI_EHIP, "sharded computeH: stream hand-over failed");
        }
        MI_TRY(run_xfers(g, list, g->xs));
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            if (hipEventRecord(g->ev_c1[i], g->xs[i]) != hipSuccess || hipStreamWaitEvent(g->ctx[i]->stream, g->ev_c1[i], 0) != hipSuccess) note(MI_EHIP, "sharded computeH: stream hand-over failed");
        }
        return MI_OK;
    };
    // the pieces of a transform (csrc/ntt_cross.hip header): cross-rank step of FFTInverse on the columns in Y; local size-M FFTInverse;
    // coset factor + local size-M FFT (DIT) of a coefficient slice
    auto cross_inv = [&](std::vector<DevBuf> &Y, bool den_scale) { each([&](int i, mi_ctx *c) { return mi_cross_dft(c, c->stream, Y[i].p, g->xt[i], 0, den_scale); }); };
    auto local_inv = [&](std::vector<DevBuf> &X) { each([&](int i, mi_ctx *c) { return mi_ntt_dev_impl(c, (mi_fr *)X[i].p, log_m, MI_NTT_INVERSE); }); };
    auto local_fwd = [&](std::vector<DevBuf> &X) {
        each([&](int i, mi_ctx *c) -> int32_t {
            MI_TRY(mi_cross_mul(c, c->stream, X[i].p, X[i].p, g->xt[i].s_fwd, nullptr, M));
            return mi_ntt_dev_impl(c, (mi_fr *)X[i].p, log_m, MI_NTT_DIT);
        });
    };
    std::vector<DevBuf> &Xa = g->hx[0], &Xb = g->hx[1], &Xc = g->hx[2], &Y0 = g->hy, &Y1 = g->hy2;
    // The six transforms, interleaved so that a batch of the transport moves one vector while the kernels of another run (a batch
    // blocks this thread until it is complete -- the transports' deadline rule -- so what should overlap it is enqueued BEFORE it):
    //   FFTInverse(a), (b), (c) = [rows -> columns] cross-rank step [columns -> rows] local transform;  FFT_coset(a), (b) = factor, local
    //   transform [rows -> columns];  then the middle kernel (cross-rank steps of a and b, product, cross-rank step of the last
    //   FFTInverse: the all-to-all pair in between cancels)  [columns -> rows]  local transform.   9 batches.
    if (derive_c) each([&](int i, mi_ctx *c) { return mi_cross_mul(c, c->stream, Xc[i].p, Xa[i].p, Xb[i].p, nullptr, M); });
    mark(E_IN);
    MI_TRY(a2a(Xa, Y0, E_IN));
    cross_inv(Y0, false); mark(E_DA);
    MI_TRY(a2a(Xb, Y1, E_IN));                 // under a's cross-rank step
    cross_inv(Y1, false); mark(E_DB);
    MI_TRY(a2a(Y0, Xa, E_DA));                 // under b's cross-rank step
    local_inv(Xa); local_fwd(Xa); mark(E_FA);
    MI_TRY(a2a(Xc, Y0, E_IN));                 // under a's two local transforms
    cross_inv(Y0, true); mark(E_DC);           // den * coefficients of c
    MI_TRY(a2a(Y1, Xb, E_DB));
    local_inv(Xb); local_fwd(Xb); mark(E_FB);
    MI_TRY(a2a(Y0, Xc, E_DC));                 // under b's two local transforms
    local_inv(Xc);
    MI_TRY(a2a(Xa, Y0, E_FA));                 // under c's local transform
    MI_TRY(a2a(Xb, Y1, E_FB));
    each([&](int i, mi_ctx *c) { return mi_cross_mid(c, c->stream, Y0[i].p, Y1[i].p, g->xt[i]); });
    mark(E_MID);
    MI_TRY(a2a(Y0, Xa, E_MID));
    local_inv(Xa);
    // h = den g^-k (.) - den c_k, into the h slice behind its front slot
    each([&](int i, mi_ctx *c) { return mi_cross_mul(c, c->stream, (char *)g->hh[i].p + sizeof(Fr), g->hx[0][i].p, g->xt[i].s_inv, g->hx[2][i].p, M); });
    {   // every rank's last coefficient -> the front slot of the next rank
        std::vector<Xfer> list;
        for (int r = 0; r + 1 < W; r++) {
            Xfer x{r, r + 1, nullptr, nullptr, sizeof(Fr)};
            if (g->local(r)) x.sp = (const char *)g->hh[r - g->rank0].p + M * sizeof(Fr);
            if (g->local(r + 1)) x.dp = g->hh[r + 1 - g->rank0].p;
            list.push_back(x);
        }
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            if (hipEventRecord(g->ev_c0[i], g->ctx[i]->stream) != hipSuccess || hipStreamWaitEvent(g->xs[i], g->ev_c0[i], 0) != hipSuccess) note(MI_EHIP, "sharded computeH: stream hand-over failed");
        }
        MI_TRY(run_xfers(g, list, g->xs));
        for (int i = 0; i < nl; i++) {
            (void)hipSetDevice(g->dev[i]);
            if (hipEventRecord(g->ev_c1[i], g->xs[i]) != hipSuccess || hipStreamWaitEvent(g->ctx[i]->stream, g->ev_c1[i], 0) != hipSuccess) note(MI_EHIP, "sharded computeH: stream hand-over failed");
        }
    }
    if (lrc != MI_OK) g->err = lerr;
    return lrc;
}
// fills g->hx[which][i] with local rank i's slice of a vector of n_valid elements: rows [r M, (r + 1) M), zero beyond n_valid.
// src: host (whole vector) or device (THIS rank's slice, i.e. the elements [r M, min((r + 1) M, n_valid)))
static int32_t load_h_slice(mi_group *g, int i, int which, u32 log_m, const mi_fr *src, bool host_whole, size_t n_valid, hipStream_t st) {
    mi_ctx *c = g->ctx[i];
    const size_t M = (size_t)1 << log_m, lo = (size_t)(g->rank0 + i) * M;
    const size_t have = n_valid > lo ? (n_valid - lo < M ? n_valid - lo : M) : 0;
    MI_TRY(mi_reserve(c, g->hx[which][i], M * sizeof(Fr)));
    char *dst = (char *)g->hx[which][i].p;
    if (have) {
        if (!src) MI_FAIL(c, MI_EINVAL, "sharded computeH: null input vector");
        if (host_whole) MI_CHECK_HIP(c, hipMemcpyAsync(dst, src + lo, have * sizeof(Fr), hipMemcpyHostToDevice, st));
        else MI_CHECK_HIP(c, hipMemcpyAsync(dst, src, have * sizeof(Fr), hipMemcpyDeviceToDevice, st));
    }
    if (have < M) MI_CHECK_HIP(c, hipMemsetAsync(dst + have * sizeof(Fr), 0, (M - have) * sizeof(Fr), st));
    return MI_OK;
}

extern "C" {
int32_t mi_group_set_sharded_compute_h(mi_group *g, uint32_t on) {
    if (!g || on > 1) return MI_EINVAL;
    G_ENTER(g);
    g->sharded_h = on != 0;
    return MI_OK;
}
// computeH alone over the ranks: every local rank passes ITS slices (device pointers; c_sl == NULL: c = a o b) and receives its slice
// of h (M elements of gnark's bit-reversed order).  A collective: all ranks call it together.
int32_t mi_compute_h_sharded_dev(mi_group *g, uint32_t log_n, const mi_fr *const *a_sl, const mi_fr *const *b_sl, const mi_fr *const *c_sl,
                                 size_t n_constraints, mi_fr *const *h_sl) {
    if (!g || !a_sl || !b_sl || !h_sl) return MI_EINVAL;
    G_ENTER(g);
    int32_t lrc = MI_OK;
    std::string lerr;
    const bool ok_shape = sharded_h_possible(g, log_n) && n_constraints <= ((size_t)1 << log_n);
    if (!ok_shape) { lrc = MI_EINVAL; lerr = "sharded computeH: needs 2, 4, 8 or 16 ranks, N >= ranks^2 and n_constraints <= N"; }
    u32 log_w = 0;
    while ((1 << log_w) < g->world) log_w++;
    const u32 log_m = log_n - log_w;
    for (int i = 0; i < g->n_local() && lrc == MI_OK; i++) {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *c = g->ctx[i];
        int32_t rc = load_h_slice(g, i, 0, log_m, a_sl[i], false, n_constraints, c->stream);
        if (rc == MI_OK) rc = load_h_slice(g, i, 1, log_m, b_sl[i], false, n_constraints, c->stream);
        if (rc == MI_OK) rc = c_sl ? load_h_slice(g, i, 2, log_m, c_sl[i], false, n_constraints, c->stream) : mi_reserve(c, g->hx[2][i], sizeof(Fr) << log_m);
        if (rc == MI_OK) rc = compute_h_sharded_prepare(g, i, log_n);   // every buffer of the collective exists BEFORE the ranks agree to enter it
        if (rc != MI_OK) { lrc = rc; lerr = mi_last_error(c); }
    }
    const uint64_t check[3] = {log_n, (uint64_t)n_constraints, c_sl ? 1u : 0u};
    MI_TRY(group_agree(g, lrc, lerr, "while loading its slices for computeH", check, 3));
    lrc = compute_h_sharded(g, log_n, c_sl == nullptr);
    if (g->broken) return lrc;
    lerr = g->err;
    for (int i = 0; i < g->n_local() && lrc == MI_OK; i++) {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *c = g->ctx[i];
        hipError_t e = hipMemcpyAsync(h_sl[i], (const char *)g->hh[i].p + sizeof(Fr), sizeof(Fr) << log_m, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { lrc = MI_EHIP; lerr = std::string("sharded computeH: copying h out failed: ") + hipGetErrorString(e); }
    }
    return group_agree(g, lrc, lerr, "in the sharded computeH");
}
}  // extern "C"

// ---------------------------------------------------------------- sharded proving key
int32_t mi_pk_sharded_free(mi_group *g, mi_pk_sharded *spk) {
    if (!g || !spk) return MI_EINVAL;
    GroupCall call__(g);
    if (!call__.ok) return MI_EINVAL;   // (a broken group still frees its keys)
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
    const int nl = g->n_local(), W = g->world;
    {   // the header every process passes must be acceptable AND the same everywhere before anything collective starts
        int32_t lrc = MI_OK;
        std::string lerr;
        if (d->log_n > 28 || !d->infinity_a || !d->infinity_b || d->nb_public > d->nb_wires) { lrc = MI_EINVAL; lerr = "pk: bad header"; }
        if (lrc == MI_OK && device_points) for (int i = 1; i < nl; i++)
            if (descs[i].log_n != d->log_n || descs[i].nb_wires != d->nb_wires || descs[i].nb_public != d->nb_public || !descs[i].infinity_a || !descs[i].infinity_b) {
                lrc = MI_EINVAL; lerr = "pk: the per-rank descriptors disagree on the key's header";
            }
        const uint64_t check[4] = {d->log_n, d->nb_wires, d->nb_public, lead_share_of(g)};   // (the lead's wire share must be the same on every rank)
        MI_TRY(group_agree(g, lrc, lerr, "while checking the key's header", check, 4));
    }
    const u64 N = (u64)1 << d->log_n;
    mi_pk_sharded *spk = new (std::nothrow) mi_pk_sharded();
    if (!spk) return MI_ENOMEM;
    spk->part.assign(nl, nullptr); spk->log_n = d->log_n; spk->nb_wires = d->nb_wires;
    // window widths of the generic path that all parts share (mode 1 needs equal bucket layouts): from the LARGEST part of each MSM
    // (a rank WITHOUT pairs of some MSM -- the lead with a wire share of 0, a tiny key -- still takes part in mode 1: an empty deferred MSM
    //  leaves a zeroed bucket array of the agreed shape, msm.hip mi_msm_enqueue)
    const uint32_t share = lead_share_of(g);
    spk->lead_share = share;
    u64 max_w = 0, max_b = 0, max_z = 0;
    for (int r = 0; r < W; r++) {
        u64 lo, hi, zlo, zhi, nb = 0;
        wire_range_of(d->nb_wires, W, r, share, lo, hi); range_of(N - 1, W, r, zlo, zhi);
        for (u64 j = lo; j < hi; j++) nb += d->infinity_b[j] ? 0 : 1;
        if (hi - lo > max_w) max_w = hi - lo;
        if (nb > max_b) max_b = nb;
        if (zhi - zlo > max_z) max_z = zhi - zlo;
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
        wire_range_of(d->nb_wires, W, g->rank0 + i, share, sr.w_lo, sr.w_hi); range_of(N - 1, W, g->rank0 + i, sr.z_lo, sr.z_hi);
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
    spk->uniform = smin == smax;
    *out = spk;
    return MI_OK;
}

// One proof over the ranks of the group (groth16.Prove, mt.go:496).  Inputs either in host memory (host = true: W is the WHOLE wire
// vector, a process reads only the ranges of its local ranks; a, b, c are read by the process that holds rank 0) or already on the
// devices (W_dev[i] = the wire range of local rank i on its device; a, b, c on rank 0's device).  c == null on the lead: c = a o b,
// formed on the device (mi_groth16_prove).
// mode 0: per-rank partial sums (option i); mode 1: bucket reduce-scatter before the reduce (option ii).
// Phases: [local: checks, workspaces, uploads, wire MSMs, computeH + Z on the lead] AGREE [h slices over the transport] [local: the other
// ranks' Z MSMs; mode 1: bucket views] (mode 1: AGREE [bucket slices of A, B1, B2, K] [bucket slices of Z]) [local: collect] ALL-GATHER
// of every rank's status and five partial sums.
// abc_sl (may be null): per LOCAL rank device pointers to its rows of a, b, c (abc_sl[2] == null: c = a o b): computeH runs over the ranks
// (compute_h_sharded); so it does for host arrays when the group asks for it (mi_group_set_sharded_compute_h) and the shape allows.
static int32_t prove_sharded_impl(mi_group *g, mi_pk_sharded *spk, bool host, const mi_fr *W_host, const mi_fr *const *W_dev, size_t n_wires,
                                  const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m,
                                  uint32_t mode, mi_proof_out *out, mi_stats *stats, const mi_fr *const *const *abc_sl = nullptr) {
    const MiRange range_fn("mi.group.prove");
    if (!g || !spk || !out) return MI_EINVAL;
    const int nl = g->n_local(), W = g->world;
    if ((nl != W && nl != 1) || (int)spk->part.size() != nl) G_FAIL(g, MI_EINVAL, "group: a process holds either all ranks or exactly one");
    static const int slots[5] = {0, 1, 2, 3, 4}, curves[5] = {1, 1, 2, 1, 1};   // A, B1, B2, K, Z: the same order on every rank
    const bool lead_here = g->rank0 == 0;   // global rank 0 runs computeH and owns a, b, c
    const size_t N = (size_t)1 << spk->log_n;
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t cb = n_constraints * sizeof(mi_fr);
    const bool defer = mode == 1;
    const bool use_sh = (abc_sl != nullptr || (host && g->sharded_h)) && sharded_h_possible(g, spk->log_n);
    u32 log_w = 0;
    while ((1 << log_w) < W) log_w++;
    // ---- local phase 1.  Nothing returns from here on without the group having agreed on it (a process that left alone would leave
    //      the others waiting in the next exchange): failures are noted and carried to the agreement.
    int32_t lrc = MI_OK;
    std::string lerr;
    auto note = [&](int32_t rc, const std::string &msg) { if (rc != MI_OK && lrc == MI_OK) { lrc = rc; lerr = msg; } };
    if (!r_m || !s_m || mode > 1 || (host ? (!W_host && n_wires) : !W_dev)) note(MI_EINVAL, "prove: null argument or unknown mode");
    if (abc_sl && !use_sh) note(MI_EINVAL, "prove: row slices of a, b, c need computeH over the ranks: 2, 4, 8 or 16 ranks and N >= ranks^2");
    if (abc_sl && (!abc_sl[0] || !abc_sl[1])) note(MI_EINVAL, "prove: null slice arrays");
    if (!abc_sl && (lead_here || use_sh) && (!a || !b) && n_constraints)
        note(MI_EINVAL, use_sh ? "prove: with computeH over the ranks every process passes a and b" : "prove: the process that holds rank 0 must pass a and b");
    if (use_sh && n_constraints > N) note(MI_EINVAL, "prove: witness size does not match the proving key");
    if (n_wires != spk->nb_wires || (lead_here && n_constraints > N)) note(MI_EINVAL, "prove: witness size does not match the proving key");
    if (mode == 1 && !spk->uniform) note(MI_EINVAL, "group: mode 1 needs every part of the key to use the same MSM plan");
    // (mi_group_wire_range answers with the group's CURRENT share: a caller of the _dev entry points would cut W unlike the key's parts)
    if (lead_share_of(g) != spk->lead_share) note(MI_EINVAL, "group: the lead's wire share (mi_group_set_lead_share) changed since this key was loaded: reload the key");
    // workspaces, each on its own device: W slice (+ a, b, c on the lead) for host inputs; h (whole on the lead, a slice elsewhere)
    for (int i = 0; i < nl && lrc == MI_OK; i++) {
        (void)hipSetDevice(g->dev[i]);
        mi_ctx *ctx = g->ctx[i];
        mi_pk *pk = spk->part[i];
        const bool lead = g->rank0 + i == 0;
        std::memset(&ctx->stats, 0, sizeof(ctx->stats));
        int32_t rc = MI_OK;
        if (host) rc = mi_reserve(ctx, ctx->ws[16], pk->nb_wires * sizeof(mi_fr) + (lead && !use_sh ? 3 * cb : 0) + 128);
        if (rc == MI_OK && !use_sh) rc = mi_reserve(ctx, ctx->ws[14], (lead ? N : pk->n_z_msm + 1) * sizeof(Fr));
        note(rc, mi_last_error(ctx));
    }
    // every local rank: its slice of W, its wire MSMs; the lead also a, b, c, computeH and its own Z MSM.  One host thread per rank:
    // enqueueing the wire MSMs waits once for the count pass of their sorts (msm.hip, MI_MSM_EXACT_SIZE)
    auto rank_main = [&](int i) -> int32_t {
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
        hipStream_t cps = nullptr;
        if (host) MI_TRY(mi_copy_stream(ctx, &cps));
        if (host && wb) {
            MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)Wd, W_host + pk->wire_lo, wb, hipMemcpyHostToDevice, cps));
            MI_CHECK_HIP(ctx, hipStreamSynchronize(cps));
        }
        MI_CHECK_HIP(ctx, hipEventRecord(ev[2], ctx->stream));
        MI_TRY(mi_prove_enqueue_wire_msms(ctx, pk, Wd, ev[2], defer));
        if (use_sh) {   // this rank's rows of a, b (, c) into the group's slice vectors; computeH itself is a collective and follows the agreement
            const u32 log_m = pk->log_n - log_w;
            MI_CHECK_HIP(ctx, hipEventRecord(ev[11], ctx->stream));
            for (int which = 0; which < 3; which++) {
                const mi_fr *src = abc_sl ? (abc_sl[which] ? abc_sl[which][i] : nullptr) : (which == 0 ? a : which == 1 ? b : c);
                const bool given = abc_sl ? abc_sl[which] != nullptr : src != nullptr;
                if (which == 2 && !given) { MI_TRY(mi_reserve(ctx, g->hx[2][i], sizeof(Fr) << log_m)); continue; }
                MI_TRY(load_h_slice(g, i, which, log_m, src, !abc_sl, n_constraints, abc_sl ? ctx->stream : cps));
            }
            if (!abc_sl) MI_CHECK_HIP(ctx, hipStreamSynchronize(cps));   // (grown buffers: hipFree inside mi_reserve has synchronised already)
            return compute_h_sharded_prepare(g, i, pk->log_n);   // the collective's own buffers too: a failure is agreed on below, before any rank enters it
        }
        if (!lead) return MI_OK;
        // lead: a, b, c arrive while the wire MSMs run; computeH; its own slice of h feeds its Z MSM straight away
        const mi_fr *da = a, *db = b, *dc = c;
        if (host) {
            char *base = (char *)ctx->ws[16].p;
            da = (mi_fr *)(base + wb); db = (mi_fr *)(base + wb + cb); dc = c ? (mi_fr *)(base + wb + 2 * cb) : nullptr;
            if (cb) {
                MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)da, a, cb, hipMemcpyHostToDevice, cps));
                MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)db, b, cb, hipMemcpyHostToDevice, cps));
                if (c) MI_CHECK_HIP(ctx, hipMemcpyAsync((void *)dc, c, cb, hipMemcpyHostToDevice, cps));
                MI_CHECK_HIP(ctx, hipStreamSynchronize(cps));
            }
            ctx->stats.h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_up).count();
        }
        MI_CHECK_HIP(ctx, hipEventRecord(ev[11], ctx->stream));
        Fr *h = (Fr *)ctx->ws[14].p;
        MI_TRY(mi_compute_h_dev_impl(ctx, pk->log_n, da, db, dc, n_constraints, (mi_fr *)h));
        MI_CHECK_HIP(ctx, hipEventRecord(ev[3], ctx->stream));
        return mi_prove_enqueue_z_msm(ctx, pk, (const mi_fr *)(h + pk->z_lo), ev[3], defer);
    };
    if (lrc == MI_OK) { std::string e; note(for_each_local_rank(g, &e, rank_main), e); }
    auto fail = [&](int32_t rc) { const std::string keep = g->err; drain_slots(g, slots, curves, 5); g->err = keep; return rc; };
    {
        const uint64_t check[5] = {(uint64_t)n_wires, (uint64_t)mode, (uint64_t)spk->log_n, use_sh ? 1u : 0u, use_sh ? (uint64_t)n_constraints : 0u};   // (without computeH over the ranks n_constraints is the lead's alone)
        const int32_t rc = group_agree(g, lrc, lerr, "before computeH / the exchange of the h slices", check, 5);
        if (rc != MI_OK) return fail(rc);
    }
    // ---- h: rank 0 hands every other rank its slice device to device, as one batch of the group's transport (grouped ncclSend / ncclRecv,
    // same-process copies or the shared-memory rings) on the exchange streams; the events that order the Z MSMs behind it are recorded by
    // each RECEIVER on its own stream (an event is recorded only on a stream of the device it was created on)
    if (use_sh) {
        // computeH over all ranks; every rank's h slice is born where its Z pairs live (front slot = the previous rank's last coefficient)
        const bool derive_c = abc_sl ? abc_sl[2] == nullptr : c == nullptr;
        const int32_t rc = compute_h_sharded(g, spk->log_n, derive_c);
        if (g->broken) return fail(rc);
        note(rc, g->err);
        if (lrc == MI_OK) {
            std::string e;
            note(for_each_local_rank(g, &e, [&](int i) -> int32_t {
                mi_ctx *ctx = g->ctx[i];
                MI_CHECK_HIP(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
                const Fr *hz = (const Fr *)g->hh[i].p + (g->rank0 + i == 0 ? 1 : 0);
                return mi_prove_enqueue_z_msm(ctx, spk->part[i], (const mi_fr *)hz, ctx->ev[3], defer);
            }), e);
        }
    } else if (W > 1) {
        std::vector<Xfer> list;
        for (int j = 1; j < W; j++) {
            u64 zlo, zhi;
            range_of(N - 1, W, j, zlo, zhi);
            Xfer x{0, j, nullptr, nullptr, (size_t)(zhi - zlo) * sizeof(Fr)};
            if (g->local(0)) x.sp = (const char *)g->ctx[0 - g->rank0]->ws[14].p + zlo * sizeof(Fr);
            if (g->local(j)) x.dp = g->ctx[j - g->rank0]->ws[14].p;
            list.push_back(x);
        }
        if (lead_here) {
            (void)hipSetDevice(g->dev[0]);
            const hipError_t e = hipStreamWaitEvent(g->xs[0], g->ctx[0]->ev[3], 0);
            if (e != hipSuccess) note(MI_EHIP, std::string("hipStreamWaitEvent (h ready): ") + hipGetErrorString(e));   // (still takes part in the batch)
        }
        const int32_t rc = run_xfers(g, list, g->xs);
        if (rc != MI_OK) return fail(rc);   // transport failure: the group is broken
        if (lrc == MI_OK) {
            std::string e;
            note(for_each_local_rank(g, &e, [&](int i) -> int32_t {
                if (g->rank0 + i == 0) return MI_OK;
                mi_ctx *ctx = g->ctx[i];
                MI_CHECK_HIP(ctx, hipEventRecord(g->ev_h[i], g->xs[i]));
                return mi_prove_enqueue_z_msm(ctx, spk->part[i], (const mi_fr *)ctx->ws[14].p, g->ev_h[i], defer);
            }), e);
        }
    }
    if (defer) {
        // the wire MSMs' slices go first (one batch), Z's -- the last to have its bucket sums -- second: their reduces then run under
        // Z's accumulation.  No local failure point lies between the agreement and the end of the second batch that a rank does not carry
        // THROUGH both batches.
        BucketExchange bx;
        std::vector<uint64_t> check;
        if (lrc == MI_OK) { const int32_t rc = exchange_prepare(g, bx, slots, curves, 5, check); note(rc, g->err); }
        const int32_t rc = group_agree(g, lrc, lerr, "before the exchange of the bucket sums", check.data(), (uint32_t)check.size());
        if (rc != MI_OK) return fail(rc);
        int32_t r1 = exchange_run(g, bx, 0, 4);
        if (g->broken) return fail(r1);
        note(r1, g->err);
        r1 = exchange_run(g, bx, 4, 1);
        if (g->broken) return fail(r1);
        note(r1, g->err);
    }
    ProofAssembler as;
    if (lrc == MI_OK) as.start(spk->part[0], r_m, s_m);
    // ---- collect: this process's partial results of the five MSMs, then ONE all-gather of (status, five partial sums) per rank: every
    // process ends with the same five sums, or every process returns an error
    struct Part { int32_t rc; uint32_t pad[3]; G1X a, b1, k, z; G2X b2; };
    std::vector<Part> mine((size_t)nl), all((size_t)W);
    for (int i = 0; i < nl; i++) {
        (void)hipSetDevice(g->dev[i]);
        std::memset(&mine[i], 0, sizeof(Part));
        if (lrc == MI_OK) {
            mi_ctx *ctx = g->ctx[i];
            int32_t rc = mi_msm_finish(ctx, 0, 1, &mine[i].a);
            if (rc == MI_OK) rc = mi_msm_finish(ctx, 1, 1, &mine[i].b1);
            if (rc == MI_OK) rc = mi_msm_finish(ctx, 3, 1, &mine[i].k);
            if (rc == MI_OK) rc = mi_msm_finish(ctx, 2, 2, &mine[i].b2);
            if (rc == MI_OK) rc = mi_msm_finish(ctx, 4, 1, &mine[i].z);
            if (rc == MI_OK && (hipStreamSynchronize(ctx->stream) != hipSuccess || hipStreamSynchronize(g->xs[i]) != hipSuccess)) { mi_set_err(ctx, "prove: stream synchronisation failed"); rc = MI_EHIP; }
            note(rc, mi_last_error(ctx));
        }
    }
    for (int i = 0; i < nl; i++) mine[i].rc = lrc;
    {
        const int32_t rc = group_allgather(g, mine.data(), sizeof(Part), all.data());
        if (rc != MI_OK) return fail(rc);
        for (int r = 0; r < W; r++)
            if (all[r].rc != MI_OK) {
                if (lrc != MI_OK) { g->err = lerr; return fail(lrc); }
                g->err = std::string("group: rank ") + std::to_string(r) + " failed in its part of the proof (status " + std::to_string(all[r].rc) + ")";
                return fail(all[r].rc);
            }
    }
    G1X sum_a = G1X::inf(), sum_b1 = G1X::inf(), sum_k = G1X::inf(), sum_z = G1X::inf();
    G2X sum_b2 = G2X::inf();
    for (const Part &p : all) { xyzz_add(sum_a, p.a); xyzz_add(sum_b1, p.b1); xyzz_add(sum_k, p.k); xyzz_add(sum_z, p.z); xyzz_add(sum_b2, p.b2); }
    const auto t_gpu_done = std::chrono::steady_clock::now();
    as.have_a_b1(sum_a, sum_b1);
    as.finish(sum_k, sum_b2, sum_z, out);
    const auto t_end = std::chrono::steady_clock::now();
    (void)hipSetDevice(g->dev[0]);
    mi_stats &st = g->ctx[0]->stats;
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<float, std::milli>(y - x).count(); };
    // (statistics are best effort: past the last all-gather nothing may fail on one rank alone)
    if ((lead_here || use_sh) && hipEventElapsedTime(&st.compute_h_ms, g->ctx[0]->ev[11], g->ctx[0]->ev[3]) != hipSuccess) { (void)hipGetLastError(); st.compute_h_ms = 0; }
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
int32_t mi_groth16_prove_sharded_slices_dev(mi_group *g, mi_pk_sharded *spk, const mi_fr *const *W_dev, size_t n_wires, const mi_fr *const *a_sl,
                                            const mi_fr *const *b_sl, const mi_fr *const *c_sl, size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m,
                                            uint32_t mode, mi_proof_out *out, mi_stats *stats) {
    G_ENTER(g);
    const mi_fr *const *abc[3] = {a_sl, b_sl, c_sl};
    return prove_sharded_impl(g, spk, false, nullptr, W_dev, n_wires, nullptr, nullptr, nullptr, n_constraints, r_m, s_m, mode, out, stats, abc);
}
int32_t mi_groth16_prove_sharded_dev(mi_group *g, mi_pk_sharded *spk, const mi_fr *const *W_dev, size_t n_wires, const mi_fr *a_dev, const mi_fr *b_dev,
                                     const mi_fr *c_dev, size_t n_constraints, const mi_fr *r_m, const mi_fr *s_m, uint32_t mode, mi_proof_out *out,
                                     mi_stats *stats) {
    G_ENTER(g);
    return prove_sharded_impl(g, spk, false, nullptr, W_dev, n_wires, a_dev, b_dev, c_dev, n_constraints, r_m, s_m, mode, out, stats);
}

}  // extern "C"
