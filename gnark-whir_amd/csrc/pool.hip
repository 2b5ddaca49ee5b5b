// Prover pool: keeps several proofs in flight on ONE device.
//
// A single mi_groth16_prove call already spreads its five MSMs and computeH over HIP streams, but every proof still has a
// serial head (the first NTT passes / digit extraction while the MSM streams wait for their inputs) and a serial tail (the
// last MSM's upper levels, bucket reduce, window combine on the host, blinding).  A prover service calls groth16.Prove
// from many goroutines (the reference has no batching of its own: mt.go:496 is one call per proof), so the drop-in keeps
// `in_flight` contexts -- each with its own streams, workspaces and a host worker thread -- and lets the GPU fill one
// proof's head and tail with the bulk of another.  The proving key is read-only during prove and is shared.
// The contexts' stream priorities are staggered (ctx.h MI_PRIO_*): the first context's proof runs nearly as if alone, the
// others fill what it leaves.  Measured at N = 2^23 (WHIR mix): DESIGN.md 4 / 5.
//
// Host inputs (mi_prover_submit, the cgo path: W, a, b, c live in Go memory, mt.go:494-496) go through an UPLOAD STAGE of
// their own: one uploader thread with its own copy stream moves the next job's four vectors into one of in_flight + 1
// device input sets while the compute workers are busy with earlier proofs, and only then queues the job for a worker,
// which runs the device-pointer path.  A copy from pageable memory holds the calling thread for its whole duration
// (20 ms per 1.07 GB proof at N = 2^23): done by the worker itself (round 1) it lengthened every context's cycle and cost
// 13 % of the throughput; on the uploader it overlaps the previous proofs' kernels.
#include <hip/hip_runtime.h>
#include "prove_internal.h"
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <future>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {
struct Job {
    uint64_t id = 0;
    bool host = false;
    mi_pk *pk = nullptr;
    const mi_fr *W = nullptr, *a = nullptr, *b = nullptr, *c = nullptr;
    size_t n_wires = 0, n_constraints = 0;
    mi_fr r, s;
    mi_proof_out *out = nullptr;
    mi_stats *stats = nullptr;
    int32_t rc = MI_OK;
    bool done = false, waited = false;
    int set = -1;          // device input set a host job was staged into (released when the proof is done)
    bool gated = false;    // handed to a worker before a, b, c had arrived: it waits for the set's abc_state
    float h2d_ms = 0;      // wall-clock of the staging copies
    std::string err;
    // BSB22 (mi_prover_submit_bsb22): the proof's CommitmentPok = sum_i challenge^i * ProveKnowledge_i(values_i), computed by the job
    std::vector<mi_bsb22_input> bsb;
    mi_fr challenge{};
    mi_g1_affine *pok_out = nullptr;
};
struct InputSet {          // W | a | b | c of one staged host job
    void *p = nullptr;
    size_t cap = 0;
    bool busy = false;
    int abc_state = 0;     // under the pool mutex: how many of a, b, c are resident (0..3; a job without c jumps from 1 to 3); -1 = their upload failed
    std::string abc_err;
};
}  // namespace

struct mi_prover {
    int dev = 0;
    std::vector<mi_ctx *> ctx;
    std::vector<std::thread> workers;
    std::thread uploader;
    hipStream_t copy_stream = nullptr;
    std::vector<InputSet> sets;                 // in_flight + 1: one being filled while in_flight are being consumed
    std::mutex m;
    std::condition_variable cv_work, cv_done, cv_up, cv_abc;
    std::deque<Job *> upq;                      // host jobs waiting for the upload stage
    std::deque<Job *> queue;                    // jobs whose inputs are in HBM
    std::unordered_map<uint64_t, Job *> jobs;   // submitted, not yet collected by mi_prover_wait
    uint64_t next_id = 1;
    bool stop = false, uploading = false;
    mi_ctx *commit_ctx = nullptr; // mi_prover_commit: the mid-solve Pedersen commitments of all callers, one at a time
    std::mutex commit_m;
    uint32_t busy = 0;            // workers inside a prove
    bool early_handover = true;   // a host job goes to a worker once W has arrived (false: only when W, a, b, c all have; measured equal on the job, 2 ms worse on one proof)
    std::string err;
};

static void finish_job(mi_prover *p, Job *j, int32_t rc, const char *err) {
    {
        std::lock_guard<std::mutex> lk(p->m);
        j->rc = rc;
        if (rc != MI_OK) { try { j->err = err ? err : ""; } catch (...) { } }
        if (j->set >= 0) { p->sets[j->set].busy = false; j->set = -1; }
        j->done = true;
    }
    p->cv_up.notify_all();
    p->cv_done.notify_all();
}

static void worker_main(mi_prover *p, mi_ctx *ctx) {
    (void)hipSetDevice(p->dev);   // the current device is per host thread
    for (;;) {
        Job *j;
        {
            std::unique_lock<std::mutex> lk(p->m);
            // leave only when the upload stage can hand over nothing more either
            p->cv_work.wait(lk, [&] { return !p->queue.empty() || (p->stop && p->upq.empty() && !p->uploading); });
            if (p->queue.empty()) return;
            j = p->queue.front();
            p->queue.pop_front();
            p->busy++;
        }
        const MiRange range_job("mi.pool.job");
        int32_t rc = MI_ENOMEM, rc_pok = MI_OK;
        std::string prove_err;
        try {   // (nothing may leave a worker thread: an allocation failure in the bookkeeping below is this job's MI_ENOMEM, not std::terminate)
        // BSB22: ProveKnowledge of every commitment but the last runs before the proof, the last one rides on slot 5 BESIDE the proof's
        // five MSMs and is collected after it (prove.go computes the PoK between the solve and computeH: same values, same points)
        const size_t nb = j->bsb.size();
        std::vector<mi_g1_affine> poks(nb);
        std::string pok_err;
        for (size_t i = 0; i + 1 < nb && rc_pok == MI_OK; i++) rc_pok = mi_pedersen_prove_knowledge(ctx, j->bsb[i].key, j->bsb[i].values, j->bsb[i].n, &poks[i]);
        // (enqueued from a helper thread while this one enqueues the proof: the PoK's sort waits once on the host for its count pass, and
        //  the values' pageable copy holds its thread -- neither should delay the proof's own kernels; slot 5 and ws[19] are the PoK's alone)
        bool pok_pending = false;
        std::future<int32_t> f_pok;
        std::string pok_enq_err;   // the helper's own error sink (ctx.h mi_err_sink): it works on ctx while this thread proves on it
        if (rc_pok != MI_OK) pok_err = mi_last_error(ctx);   // (of the synchronous ProveKnowledge calls above, before anything else runs on ctx)
        if (nb && rc_pok == MI_OK) {
            mi_pedersen_pk *key = j->bsb[nb - 1].key; const mi_fr *vals = j->bsb[nb - 1].values; const size_t nv = j->bsb[nb - 1].n; const int dev = p->dev;
            std::string *sink = &pok_enq_err;
            auto enq = [=]() -> int32_t {
                mi_err_sink = sink;
                int32_t r = MI_ENOMEM;
                try { (void)hipSetDevice(dev); r = mi_pedersen_pok_enqueue(ctx, key, vals, nv); } catch (...) { }
                mi_err_sink = nullptr;
                return r;
            };
            try { f_pok = std::async(std::launch::async, enq); } catch (...) { rc_pok = enq(); pok_pending = rc_pok == MI_OK; if (rc_pok != MI_OK) pok_err = pok_enq_err; }
        }
        if (j->gated) {
            InputSet &set = p->sets[j->set];
            const std::function<bool(int)> abc = [&](int k) -> bool {   // blocks until k of a, b, c are resident (or their upload has failed)
                std::unique_lock<std::mutex> lk(p->m);
                p->cv_abc.wait(lk, [&] { return set.abc_state < 0 || set.abc_state >= k; });
                return set.abc_state > 0;
            };
            bool arrived;
            {
                std::lock_guard<std::mutex> lk(p->m);
                arrived = set.abc_state == 3;   // the steady state: the uploader is a job ahead
            }
            rc = mi_groth16_prove_dev_gated(ctx, j->pk, j->W, j->n_wires, j->a, j->b, j->c, j->n_constraints, &j->r, &j->s, j->out, j->stats, abc, arrived);
            // Whatever the prove returned -- it can fail BEFORE it reaches the gate (witness size mismatch, a part of a sharded key, a
            // workspace that does not fit) -- the uploader may still be copying this job's a, b, c from the caller's buffers and will
            // still write j->h2d_ms: the job is not finished (its waiter may free the Job and the buffers) until the uploader is done with it.
            (void)abc(3);
        } else {
            rc = mi_groth16_prove_dev(ctx, j->pk, j->W, j->n_wires, j->a, j->b, j->c, j->n_constraints, &j->r, &j->s, j->out, j->stats);
        }
        prove_err = rc != MI_OK ? mi_last_error(ctx) : "";
        if (f_pok.valid()) { rc_pok = f_pok.get(); pok_pending = rc_pok == MI_OK; if (rc_pok != MI_OK) pok_err = "prover: enqueueing the ProveKnowledge MSM failed: " + pok_enq_err; }
        if (pok_pending) {   // collected whatever the proof did: slot 5 must be idle for the next job
            const int32_t r2 = mi_pedersen_pok_collect(ctx, &poks[nb - 1]);
            if (r2 != MI_OK && rc_pok == MI_OK) { rc_pok = r2; pok_err = mi_last_error(ctx); }
        }
        if (rc == MI_OK && rc_pok != MI_OK) { rc = rc_pok; prove_err = pok_err; }
        if (rc == MI_OK && nb && j->pok_out) rc = mi_pedersen_fold(poks.data(), nb, &j->challenge, j->pok_out);
        if (rc == MI_OK && j->host && j->stats) j->stats->h2d_ms = j->h2d_ms;
        } catch (...) {
            rc = MI_ENOMEM;
            if (j->gated) {   // the uploader may still be copying this job's vectors: wait for it as the regular path does
                std::unique_lock<std::mutex> lk(p->m);
                InputSet &set = p->sets[j->set];
                p->cv_abc.wait(lk, [&] { return set.abc_state < 0 || set.abc_state >= 3; });
            }
        }
        {
            std::lock_guard<std::mutex> lk(p->m);
            p->busy--;
        }
        finish_job(p, j, rc, rc != MI_OK ? prove_err.c_str() : nullptr);
    }
}

// upload stage: host job -> device input set -> compute queue
static void uploader_main(mi_prover *p) {
    (void)hipSetDevice(p->dev);
    for (;;) {
        Job *j;
        int si = -1;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->cv_up.wait(lk, [&] {
                if (p->upq.empty()) return p->stop;
                for (size_t i = 0; i < p->sets.size(); i++) if (!p->sets[i].busy) return true;
                return false;
            });
            if (p->upq.empty()) { lk.unlock(); p->cv_work.notify_all(); return; }
            j = p->upq.front();
            p->upq.pop_front();
            for (size_t i = 0; i < p->sets.size(); i++) if (!p->sets[i].busy) { si = (int)i; break; }
            p->sets[si].busy = true;
            p->uploading = true;
            j->set = si;
        }
        InputSet &set = p->sets[si];
        const MiRange range_up("mi.pool.upload");
        const size_t wb = j->n_wires * sizeof(mi_fr), cb = j->n_constraints * sizeof(mi_fr), need = wb + 3 * cb + 128;
        hipError_t e = hipSuccess;
        const char *what = "";
        if (need > set.cap) {   // grow-only, like every other workspace: no hipMalloc in steady state
            if (set.p) { (void)hipFree(set.p); set.p = nullptr; set.cap = 0; }
            e = hipMalloc(&set.p, need + need / 8);
            what = "prover: hipMalloc of a device input set";
            if (e == hipSuccess) set.cap = need + need / 8;
        }
        char *base = (char *)set.p;
        const auto t0 = std::chrono::steady_clock::now();
        // W first; the job can go to a worker as soon as W is resident (its wire MSMs start), a, b, c -- 3/4 of the bytes -- follow
        // while those MSMs run and the worker enqueues computeH once they are resident too.  In steady state the uploader is a whole job ahead and none of
        // this shows; it is the FIRST job of a burst (nothing to hide its upload behind) that gains.
        if (e == hipSuccess) {
            what = "prover: upload of W";
            // (no events on this stream: a marker between two pageable copies slowed the copies behind it -- the hand-overs are ordered
            //  by synchronising the stream on this thread instead)
            if (wb) e = hipMemcpyAsync(base, j->W, wb, hipMemcpyHostToDevice, p->copy_stream);
            if (e == hipSuccess && p->early_handover) e = hipStreamSynchronize(p->copy_stream);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            {
                std::lock_guard<std::mutex> lk(p->m);
                p->uploading = false;
            }
            finish_job(p, j, e == hipErrorOutOfMemory ? MI_ENOMEM : MI_EHIP, what);   // (a literal: nothing on this thread may throw)
            p->cv_work.notify_all();
            continue;
        }
        const mi_fr *ha = j->a, *hb = j->b, *hc = j->c;
        auto hand_over = [&](bool gated) {   // gated: a, b, c are still on their way (the worker waits for abc_state before computeH)
            {
                std::lock_guard<std::mutex> lk(p->m);
                j->W = (const mi_fr *)base; j->a = (const mi_fr *)(base + wb); j->b = (const mi_fr *)(base + wb + cb); j->c = hc ? (const mi_fr *)(base + wb + 2 * cb) : nullptr;
                j->gated = gated;
                p->queue.push_back(j);
            }
            p->cv_work.notify_all();
        };
        // The job goes to a worker as soon as ONE WOULD OTHERWISE SIT IDLE (nothing queued, a worker free) -- checked when W has
        // arrived and again after a and after b.  With work queued or every worker busy, a worker that took the job now would spend the
        // rest of the upload blocked on the host instead of proving something whose inputs are there (host and device jobs mixed measured
        // 22 against 32 proofs/s with an unconditional early hand-over).
        bool handed = false;
        {
            std::lock_guard<std::mutex> lk(p->m);
            set.abc_state = 0;
        }
        auto maybe_hand_over = [&] {
            if (handed || !p->early_handover) return;
            bool idle;
            {
                std::lock_guard<std::mutex> lk(p->m);
                idle = p->queue.empty() && p->busy < p->workers.size();
            }
            if (idle) { hand_over(true); handed = true; }
        };
        // a, b, c one at a time, each announced as soon as it is resident: a worker that holds the job already (gated) enqueues a's
        // transforms while b is still on the bus, b's while c is (prove.hip: mi_compute_h_part).  A pageable copy holds this thread for
        // its duration anyway, so synchronising the stream after each costs nothing (and no event ever sits between two copies).
        auto arrived = [&](int k) {
            { std::lock_guard<std::mutex> lk(p->m); set.abc_state = k; }
            p->cv_abc.notify_all();
        };
        maybe_hand_over();
        if (cb) { e = hipMemcpyAsync(base + wb, ha, cb, hipMemcpyHostToDevice, p->copy_stream); if (e == hipSuccess) e = hipStreamSynchronize(p->copy_stream); }
        if (e == hipSuccess) arrived(1);
        maybe_hand_over();
        if (e == hipSuccess && cb) { e = hipMemcpyAsync(base + wb + cb, hb, cb, hipMemcpyHostToDevice, p->copy_stream); if (e == hipSuccess) e = hipStreamSynchronize(p->copy_stream); }
        if (e == hipSuccess && hc) arrived(2);
        maybe_hand_over();
        if (e == hipSuccess && cb && hc) e = hipMemcpyAsync(base + wb + 2 * cb, hc, cb, hipMemcpyHostToDevice, p->copy_stream);   // hc == null: c = a o b on the device
        if (e == hipSuccess) e = hipStreamSynchronize(p->copy_stream);   // a, b, c are resident when abc_state says so
        if (e != hipSuccess) (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(p->m);
            j->h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            set.abc_state = e == hipSuccess ? 3 : -1;
            p->uploading = false;
        }
        // everything has arrived (the stream was synchronised above): the job runs the plain device-pointer path; a failed upload
        // still goes through the gate, which reports it
        if (!handed) hand_over(e != hipSuccess);
        p->cv_abc.notify_all();
        p->cv_work.notify_all();
    }
}

extern "C" {

int32_t mi_prover_create(int device_id, uint32_t in_flight, mi_prover **out) {
    if (!out || in_flight == 0 || in_flight > 16) return MI_EINVAL;
    *out = nullptr;
    mi_prover *p = new (std::nothrow) mi_prover();
    if (!p) return MI_ENOMEM;
    p->dev = device_id;
    for (uint32_t i = 0; i < in_flight; i++) {
        mi_ctx *c = nullptr;
        int32_t rc = mi_init_prio(device_id, i == 0 ? MI_PRIO_POOL_FIRST : i == 1 ? MI_PRIO_POOL_SECOND : MI_PRIO_POOL_REST, &c);
        if (rc != MI_OK) {
            for (mi_ctx *q : p->ctx) mi_shutdown(q);
            delete p;
            return rc;
        }
        p->ctx.push_back(c);
    }
    (void)hipSetDevice(device_id);
    if (hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        for (mi_ctx *q : p->ctx) mi_shutdown(q);
        delete p;
        return MI_EHIP;
    }
    if (mi_init_prio(device_id, MI_PRIO_POOL_FIRST, &p->commit_ctx) != MI_OK) {   // mid-solve commitments: a caller is blocked on each of them
        (void)hipStreamDestroy(p->copy_stream);
        for (mi_ctx *q : p->ctx) mi_shutdown(q);
        delete p;
        return MI_EHIP;
    }
    p->sets.resize(in_flight + 1);
    for (mi_ctx *c : p->ctx) p->workers.emplace_back(worker_main, p, c);
    p->uploader = std::thread(uploader_main, p);
    *out = p;
    return MI_OK;
}

int32_t mi_prover_destroy(mi_prover *p) {
    if (!p) return MI_EINVAL;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->stop = true;   // queued jobs still run: their callers may be blocked in mi_prover_wait
    }
    p->cv_up.notify_all();
    p->cv_work.notify_all();
    p->uploader.join();
    for (auto &t : p->workers) t.join();
    // every job has run by now and nothing references a caller's W/a/b/c/out/stats any more; tickets nobody waited on
    // are dropped here (waiting on a destroyed pool is the caller's bug, like any use after free)
    (void)hipSetDevice(p->dev);
    for (InputSet &s : p->sets) if (s.p) (void)hipFree(s.p);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    for (mi_ctx *c : p->ctx) mi_shutdown(c);
    if (p->commit_ctx) mi_shutdown(p->commit_ctx);
    for (auto &kv : p->jobs) delete kv.second;
    delete p;
    return MI_OK;
}

uint32_t mi_prover_in_flight(const mi_prover *p) { return p ? (uint32_t)p->ctx.size() : 0; }
mi_ctx *mi_prover_ctx(mi_prover *p, uint32_t i) { return p && i < p->ctx.size() ? p->ctx[i] : nullptr; }
const char *mi_prover_last_error(mi_prover *p) {
    if (!p) return "null prover";
    static thread_local std::string copy;   // the caller's own copy: p->err may change under another thread's wait
    std::lock_guard<std::mutex> lk(p->m);
    copy = p->err;
    return copy.c_str();   // valid until this thread's next mi_prover_last_error
}

static int32_t submit(mi_prover *p, bool host, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                      size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, uint64_t *ticket,
                      const mi_bsb22_input *bsb = nullptr, uint32_t n_bsb = 0, const mi_fr *challenge = nullptr, mi_g1_affine *pok_out = nullptr) {
    if (!p || !pk || !r || !s || !out || !ticket || (!W && n_wires) || ((!a || !b) && n_constraints)) return MI_EINVAL;   // c == null: c = a o b
    if (n_bsb && (!bsb || !challenge || !pok_out || n_bsb > MI_PK_RAW_MAX_COMMITMENTS)) return MI_EINVAL;
    for (uint32_t i = 0; i < n_bsb; i++) if (!bsb[i].key || (!bsb[i].values && bsb[i].n)) return MI_EINVAL;
    Job *j = new (std::nothrow) Job();
    if (!j) return MI_ENOMEM;
    try { j->bsb.assign(bsb, bsb + n_bsb); } catch (...) { delete j; return MI_ENOMEM; }
    if (n_bsb) { j->challenge = *challenge; j->pok_out = pok_out; }
    j->host = host; j->pk = pk; j->W = W; j->a = a; j->b = b; j->c = c;
    j->n_wires = n_wires; j->n_constraints = n_constraints;
    j->r = *r; j->s = *s;   // copied: the caller's r, s need not outlive the call
    j->out = out; j->stats = stats;
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (p->stop) { delete j; return MI_EINVAL; }
        j->id = p->next_id++;
        p->jobs[j->id] = j;
        (host ? p->upq : p->queue).push_back(j);
        *ticket = j->id;
    }
    if (host) p->cv_up.notify_one(); else p->cv_work.notify_one();
    return MI_OK;
}

int32_t mi_prover_submit(mi_prover *p, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                         size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, uint64_t *ticket) {
    return submit(p, true, pk, W, n_wires, a, b, c, n_constraints, r, s, out, stats, ticket);
}
int32_t mi_prover_submit_dev(mi_prover *p, mi_pk *pk, const mi_fr *W_dev, size_t n_wires, const mi_fr *a_dev, const mi_fr *b_dev,
                             const mi_fr *c_dev, size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats,
                             uint64_t *ticket) {
    return submit(p, false, pk, W_dev, n_wires, a_dev, b_dev, c_dev, n_constraints, r, s, out, stats, ticket);
}

int32_t mi_prover_submit_bsb22(mi_prover *p, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                               size_t n_constraints, const mi_fr *r, const mi_fr *s, const mi_bsb22_input *commitments, uint32_t n_commitments,
                               const mi_fr *challenge, mi_proof_out *out, mi_g1_affine *pok_out, mi_stats *stats, uint64_t *ticket) {
    return submit(p, true, pk, W, n_wires, a, b, c, n_constraints, r, s, out, stats, ticket, commitments, n_commitments, challenge, pok_out);
}
// Pedersen Commit inside the solve (gnark's BSB22 hint override): synchronous, from any thread; one commitment at a time on a context of
// its own whose streams rank with the pool's first context, so that a blocked solver waits for one small MSM, not for a proof
int32_t mi_prover_commit(mi_prover *p, mi_pedersen_pk *key, const mi_fr *values, size_t n, mi_g1_affine *commitment) {
    const MiRange range_fn("mi.pool.commit");
    if (!p || !key || !commitment || (!values && n)) return MI_EINVAL;
    std::lock_guard<std::mutex> lk(p->commit_m);
    (void)hipSetDevice(p->dev);
    const int32_t rc = mi_pedersen_commit(p->commit_ctx, key, values, n, commitment);
    if (rc != MI_OK) { std::lock_guard<std::mutex> lk2(p->m); p->err = mi_last_error(p->commit_ctx); }
    return rc;
}

// mi_ctx_trim for every context of an IDLE pool (nothing submitted and not yet waited for) plus its device input sets
int32_t mi_prover_trim(mi_prover *p) {
    if (!p) return MI_EINVAL;
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (!p->upq.empty() || !p->queue.empty() || p->busy || p->uploading) { p->err = "prover: trim needs an idle pool"; return MI_EINVAL; }
        for (InputSet &s : p->sets) if (s.busy) { p->err = "prover: trim needs an idle pool"; return MI_EINVAL; }
    }
    (void)hipSetDevice(p->dev);
    int32_t rc = MI_OK;
    for (mi_ctx *c : p->ctx) { const int32_t r = mi_ctx_trim(c); if (r != MI_OK && rc == MI_OK) rc = r; }
    if (p->commit_ctx) {   // a commit counts as activity: mi_prover_commit may be called mid-solve from any thread and is not in the idle check above
        std::unique_lock<std::mutex> lc(p->commit_m, std::try_to_lock);
        if (!lc.owns_lock()) { std::lock_guard<std::mutex> lk(p->m); p->err = "prover: trim needs an idle pool (a commit is running)"; return MI_EINVAL; }
        const int32_t r = mi_ctx_trim(p->commit_ctx);
        if (r != MI_OK && rc == MI_OK) rc = r;
    }
    (void)hipStreamSynchronize(p->copy_stream);
    for (InputSet &s : p->sets) if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
    return rc;
}

int32_t mi_prover_wait(mi_prover *p, uint64_t ticket) {
    if (!p) return MI_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) { p->err = "prover: unknown ticket"; return MI_EINVAL; }
    Job *j = it->second;
    if (j->waited) { p->err = "prover: ticket is already being waited on"; return MI_EINVAL; }   // each ticket: exactly one waiter
    j->waited = true;
    p->cv_done.wait(lk, [&] { return j->done; });
    int32_t rc = j->rc;
    if (rc != MI_OK) p->err = j->err;
    p->jobs.erase(it);
    delete j;
    return rc;
}

}  // extern "C"
