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
#include <hip/hip_runtime.h>
#include "ctx.h"
#include <condition_variable>
#include <deque>
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
    bool done = false;
    std::string err;
};
}  // namespace

struct mi_prover {
    int dev = 0;
    std::vector<mi_ctx *> ctx;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<Job *> queue;
    std::unordered_map<uint64_t, Job *> jobs;   // submitted, not yet collected by mi_prover_wait
    uint64_t next_id = 1;
    bool stop = false;
    std::string err;
};

static void worker_main(mi_prover *p, mi_ctx *ctx) {
    (void)hipSetDevice(p->dev);   // the current device is per host thread
    for (;;) {
        Job *j;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;   // stop requested and nothing left to run
            j = p->queue.front();
            p->queue.pop_front();
        }
        int32_t rc = j->host ? mi_groth16_prove(ctx, j->pk, j->W, j->n_wires, j->a, j->b, j->c, j->n_constraints, &j->r, &j->s, j->out, j->stats)
                             : mi_groth16_prove_dev(ctx, j->pk, j->W, j->n_wires, j->a, j->b, j->c, j->n_constraints, &j->r, &j->s, j->out, j->stats);
        {
            std::lock_guard<std::mutex> lk(p->m);
            j->rc = rc;
            if (rc != MI_OK) j->err = mi_last_error(ctx);
            j->done = true;
        }
        p->cv_done.notify_all();
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
    for (mi_ctx *c : p->ctx) p->workers.emplace_back(worker_main, p, c);
    *out = p;
    return MI_OK;
}

int32_t mi_prover_destroy(mi_prover *p) {
    if (!p) return MI_EINVAL;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->stop = true;   // queued jobs still run: their callers may be blocked in mi_prover_wait
    }
    p->cv_work.notify_all();
    for (auto &t : p->workers) t.join();
    (void)hipSetDevice(p->dev);
    for (mi_ctx *c : p->ctx) mi_shutdown(c);
    for (auto &kv : p->jobs) delete kv.second;
    delete p;
    return MI_OK;
}

uint32_t mi_prover_in_flight(const mi_prover *p) { return p ? (uint32_t)p->ctx.size() : 0; }
mi_ctx *mi_prover_ctx(mi_prover *p, uint32_t i) { return p && i < p->ctx.size() ? p->ctx[i] : nullptr; }
const char *mi_prover_last_error(mi_prover *p) {
    if (!p) return "null prover";
    std::lock_guard<std::mutex> lk(p->m);
    return p->err.c_str();   // stable until the next failing mi_prover_wait on this pool
}

static int32_t submit(mi_prover *p, bool host, mi_pk *pk, const mi_fr *W, size_t n_wires, const mi_fr *a, const mi_fr *b, const mi_fr *c,
                      size_t n_constraints, const mi_fr *r, const mi_fr *s, mi_proof_out *out, mi_stats *stats, uint64_t *ticket) {
    if (!p || !pk || !r || !s || !out || !ticket) return MI_EINVAL;
    Job *j = new (std::nothrow) Job();
    if (!j) return MI_ENOMEM;
    j->host = host; j->pk = pk; j->W = W; j->a = a; j->b = b; j->c = c;
    j->n_wires = n_wires; j->n_constraints = n_constraints;
    j->r = *r; j->s = *s;   // copied: the caller's r, s need not outlive the call
    j->out = out; j->stats = stats;
    {
        std::lock_guard<std::mutex> lk(p->m);
        if (p->stop) { delete j; return MI_EINVAL; }
        j->id = p->next_id++;
        p->jobs[j->id] = j;
        p->queue.push_back(j);
        *ticket = j->id;
    }
    p->cv_work.notify_one();
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

int32_t mi_prover_wait(mi_prover *p, uint64_t ticket) {
    if (!p) return MI_EINVAL;
    std::unique_lock<std::mutex> lk(p->m);
    auto it = p->jobs.find(ticket);
    if (it == p->jobs.end()) { p->err = "prover: unknown ticket"; return MI_EINVAL; }
    Job *j = it->second;
    p->cv_done.wait(lk, [&] { return j->done; });
    int32_t rc = j->rc;
    if (rc != MI_OK) p->err = j->err;
    p->jobs.erase(it);
    delete j;
    return rc;
}

}  // extern "C"
