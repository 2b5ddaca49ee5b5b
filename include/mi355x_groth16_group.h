/* mi355x_groth16_group.h -- device groups of the MI355X-native Groth16 prove path: one proof / one MSM point-sharded over the GPUs of a
 * node (BASELINE configs[4]; reference call site mt.go:496).  Part of libmi355x_groth16.so; conventions as in mi355x_groth16.h. */
#ifndef MI355X_GROTH16_GROUP_H
#define MI355X_GROTH16_GROUP_H
#include "mi355x_groth16.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device groups: one proof / one MSM point-sharded over several GPUs (BASELINE configs[4]).
 * The reference's single call groth16.Prove (mt.go:496) knows no devices; a Go caller that wants one proof spread over the
 * 8 MI355X of a node binds these.  pk points are static, so mi_pk_load_sharded cuts the wires
 * (and the N - 1 pairs of the Z MSM) into `world` contiguous ranges and keeps slice r of pk.G1.{A,B,K,Z} / pk.G2.B resident
 * on rank r; per proof only scalars move (W slices from the host, h slices device to device from the lead rank, which runs
 * computeH: NTT = replicas only).  EC addition is not an RCCL reduce op, so the exchange is byte-typed:
 *   mode 0  every rank finishes Pippenger locally; one partial sum per MSM is combined (host additions in one process,
 *           ncclAllGather(ncclUint8) with one rank per process);
 *   mode 1  "all-reduce of partial bucket sums": every rank stops at its bucket sums, rank r receives the keys it owns from
 *           every other rank (reduce-scatter as grouped ncclSend / ncclRecv, one hop on the xGMI mesh), adds them, reduces
 *           its slice; the per-rank results are combined as in mode 0.
 * Results are bit-identical to the unsharded entry points.  A group serves ONE call at a time: an entry point called while another
 * call on the same group is still running returns MI_EINVAL at once and touches nothing (the exchange streams and receive
 * buffers belong to the running call).  A process holds either ALL ranks of a group (mi_group_create) or exactly ONE
 * (mi_group_create_rank); with one rank per process every process makes the same calls in the same order (they are
 * collectives), each with its own rank's data.
 * Failures are collective too: before every exchange the ranks agree on their status (one small all-gather), so a call either succeeds
 * on every rank or returns an error on every rank -- the failing rank its own, the others "rank r failed" -- with every MSM slot
 * drained; nobody is left waiting in an exchange.  Only a failure INSIDE an exchange (a dead peer, an RCCL error) breaks the group:
 * later calls on it return MI_EHIP until it is destroyed and created anew. ---- */
typedef struct mi_group mi_group;
typedef struct mi_pk_sharded mi_pk_sharded;
/* all ranks in this process, one context per entry of dev_ids (SURVEY 8b proposed mi_init(dev_ids, n_dev, ...)).  Distinct
 * devices: RCCL communicator (ncclCommInitAll).  A device named twice (1-GPU rehearsal): same-process peer copies. */
int32_t mi_group_create(const int *dev_ids, int n_dev, mi_group **out);
/* one rank per process: id = mi_group_unique_id() from rank 0, handed to the others by the caller's own channel */
int32_t mi_group_unique_id(uint8_t id[128]);
int32_t mi_group_create_rank(int device_id, int rank, int world, const uint8_t id[128], mi_group **out);   /* = _ex(..., MI_GROUP_TRANSPORT_RCCL, ...) */
/* the same with the transport named.  MI_GROUP_TRANSPORT_HOST: the processes meet in a POSIX shared-memory segment named after the 128
 * id bytes (any 128 bytes all ranks share; mi_group_unique_id is not needed) and slices travel device -> segment -> device.  For ranks
 * RCCL cannot connect: two processes on ONE device (RCCL refuses two ranks per device; how a 1-GPU box rehearses this flow) or a box
 * without a working RCCL fabric.
 * THE DEAD-PEER CONTRACT, both transports: with one rank per process no call of this library waits for another rank without a deadline.
 * MI_GROUP_TIMEOUT_MS (environment, read when the group is created; default 60000) bounds the time a rank waits for its peers while
 * NOTHING completes -- joining the group, an exchange, an all-gather.  When it passes (a peer's process ended, a link went down), or
 * when RCCL reports an asynchronous error, the call returns MI_EHIP with a message that says "timeout", the group is broken (every
 * later call on it returns MI_EHIP at once) and must be destroyed; mi_group_destroy itself does not wait for anybody.
 *   RCCL transport: the per-rank communicator is non-blocking (ncclCommInitRankConfig, blocking = 0); joining, every group of sends /
 *     receives and every all-gather is polled with ncclCommGetAsyncError / hipStreamQuery against the deadline, and ncclCommAbort takes
 *     the communicator's kernels off the stream when it passes.  An exchange has completed on every rank that returns from it.
 *   host-staged transport: every wait on the shared segment has the deadline, and a rank that gives up poisons the segment so that the
 *     others stop waiting at once.  MI_GROUP_SHM_CHUNK_KB (default 1024, 4..65536): bytes per ring slot of the segment.
 * (Single-process groups, mi_group_create, have no peers in other processes: their communicators stay blocking.) */
#define MI_GROUP_TRANSPORT_RCCL 1
#define MI_GROUP_TRANSPORT_HOST 3
int32_t mi_group_create_rank_ex(int device_id, int rank, int world, const uint8_t id[128], int transport, mi_group **out);
int32_t mi_group_destroy(mi_group *g);
/* The lead's share of the WIRES of a sharded key.  Rank 0 also runs computeH (the NTT does not shard: SURVEY 8e) and no rank can start
 * its Z MSM before h exists, so a lead that carries an equal share of the wire MSMs lengthens the critical path of the proof.
 * permille = the fraction of an even share (nb_wires / world) that rank 0 takes, 0..1000; the other ranks split the rest evenly; the
 * N - 1 pairs of the Z MSM are always cut evenly.  1000 = the even cut.  MI_LEAD_SHARE_AUTO (the default): 1000 for one rank, 500 for
 * two, 0 from three ranks on -- from the measured ratio computeH : wire MSMs = 1 : 2 at N = 2^26; with computeH over the
 * ranks (mi_group_set_sharded_compute_h) the balanced cut is the even one: say 1000.  A key remembers the share it was cut with: a prove
 * under another one returns MI_EINVAL ("reload the key").  Set it -- to the same
 * value in every process -- BEFORE mi_pk_load_sharded*: the key's parts are cut by it (a disagreement fails that load on every rank),
 * and a caller that passes device slices (mi_pk_load_sharded_dev, mi_groth16_prove_sharded_dev) cuts its arrays by
 * mi_group_wire_range.  Same proofs whatever the share. */
#define MI_LEAD_SHARE_AUTO 0xffffffffu
int32_t mi_group_set_lead_share(mi_group *g, uint32_t permille);
/* wires [*lo, *hi) of global rank `rank` under the group's current lead share */
int32_t mi_group_wire_range(const mi_group *g, uint64_t nb_wires, int rank, uint64_t *lo, uint64_t *hi);
/* computeH OVER the ranks (2, 4, 8 or 16 of them, N >= ranks^2): every transform becomes a local size-N/ranks transform and one
 * cross-rank step between two all-to-alls over the group's transport (9 batches per computeH, each moving (ranks - 1) / ranks of a
 * slice per rank), and the h slices are born on the ranks whose Z pairs they multiply -- instead of rank 0 transforming alone while the
 * others wait for h (the cap of a proof sharded over 8 GPUs moves from ~3x to the MSMs' own 1 / ranks).  Same h, same
 * proof bytes.  on = 1: mi_groth16_prove_sharded (host arrays) then needs a and b (and c, or NULL) in EVERY process, not on rank 0's
 * alone, and takes its rank's rows from them; mi_groth16_prove_sharded_dev is unchanged (its a, b, c live on rank 0's device).
 * The same value in every process.  Default 0. */
int32_t mi_group_set_sharded_compute_h(mi_group *g, uint32_t on);
/* computeH alone, as a collective: local rank i passes device pointers to ITS rows of a, b (and c; c_sl == NULL: c = a o b on the
 * device) -- rows [r M, min((r + 1) M, n_constraints)) of global rank r, M = N / ranks -- and receives its M coefficients of h in
 * gnark's bit-reversed order (rank r: positions [r M, (r + 1) M) of what mi_compute_h_dev returns). */
int32_t mi_compute_h_sharded_dev(mi_group *g, uint32_t log_n, const mi_fr *const *a_sl, const mi_fr *const *b_sl, const mi_fr *const *c_sl,
                                 size_t n_constraints, mi_fr *const *h_sl);
int32_t mi_group_world(const mi_group *g);
int32_t mi_group_local(const mi_group *g);                 /* ranks held by this process */
mi_ctx *mi_group_ctx(mi_group *g, int local_rank);         /* for mi_dev_* / generators on that rank's device */
const char *mi_group_last_error(mi_group *g);
int32_t mi_group_rank(const mi_group *g);                  /* global rank of this process's first local rank */
int32_t mi_group_transport(const mi_group *g);             /* 1 = RCCL, 2 = copies inside one process, 3 = host-staged (shared memory) */
/* observed facts for a multi-GPU run's record: the rank count the RCCL communicator itself reports (ncclCommCount; 0 when the transport
 * is not RCCL) and the PCI bus id of a local rank's device ("0000:c1:00.0") -- N ranks on N distinct devices, or not */
int32_t mi_group_comm_ranks(const mi_group *g);
int32_t mi_group_device_pci(const mi_group *g, int local_rank, char out[32]);
/* transport check: every rank sends `bytes` patterned bytes to every rank (itself included) and verifies what it received */
int32_t mi_group_exchange_selftest(mi_group *g, size_t bytes);
/* desc: the same whole-key descriptor as mi_pk_load (host arrays); with one rank per process every process passes the whole
 * descriptor and keeps its own rank's slice.  The fixed-base table plan is agreed over the whole group (tightest device). */
int32_t mi_pk_load_sharded(mi_group *g, const mi_pk_desc *desc, mi_pk_sharded **out);
/* slice_descs[i], i < mi_group_local(): header (log_n, nb_public, nb_wires) and masks of the WHOLE key in host memory; the five
 * point arrays are DEVICE pointers on local rank i's device to that rank's slices (the points of wires [nb_wires r / world,
 * nb_wires (r+1) / world) and the Z pairs [(N-1) r / world, (N-1)(r+1) / world), r = global rank), counts = points of the slice.
 * Adopted by reference like mi_pk_load_dev (the caller keeps them alive). */
int32_t mi_pk_load_sharded_dev(mi_group *g, const mi_pk_desc *slice_descs, mi_pk_sharded **out);
int32_t mi_pk_sharded_free(mi_group *g, mi_pk_sharded *pk);
/* groth16.Prove after the solve (mt.go:496) over the ranks of the group; arguments as mi_groth16_prove.  W is the WHOLE wire
 * vector: a process reads only the wire ranges of its local ranks.  a, b, c are read by the process that holds rank 0 (which runs
 * computeH and hands every rank its slice of h over the group's transport); other processes may pass NULL.  Every process
 * receives the proof. */
int32_t mi_groth16_prove_sharded(mi_group *g, mi_pk_sharded *pk, const mi_fr *W, size_t n_wires,
                                 const mi_fr *a, const mi_fr *b, const mi_fr *c, size_t n_constraints,
                                 const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats_or_null);
/* the same with the inputs already in HBM: W_dev[i] = the wire range of local rank i on its device (n_wires = the WHOLE count);
 * a_dev, b_dev, c_dev on rank 0's device (NULL in the other processes) */
int32_t mi_groth16_prove_sharded_dev(mi_group *g, mi_pk_sharded *pk, const mi_fr *const *W_dev, size_t n_wires,
                                     const mi_fr *a_dev, const mi_fr *b_dev, const mi_fr *c_dev, size_t n_constraints,
                                     const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats_or_null);
/* the same with a, b, c as ROW SLICES per local rank (rows [r M, min((r + 1) M, n_constraints)) of global rank r on that rank's device,
 * M = N / ranks; c_sl == NULL: c = a o b): computeH runs over the ranks (mi_group_set_sharded_compute_h says what that means; here it
 * is the only way, whatever the group's setting).  2, 4, 8 or 16 ranks, N >= ranks^2. */
int32_t mi_groth16_prove_sharded_slices_dev(mi_group *g, mi_pk_sharded *pk, const mi_fr *const *W_dev, size_t n_wires,
                                            const mi_fr *const *a_sl, const mi_fr *const *b_sl, const mi_fr *const *c_sl, size_t n_constraints,
                                            const mi_fr *r, const mi_fr *s, uint32_t mode, mi_proof_out *out, mi_stats *stats);
/* one MSM over host arrays cut into contiguous slices (single-process groups) */
int32_t mi_msm_g1_sharded(mi_group *g, const mi_g1_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags,
                          uint32_t mode, mi_g1_jac *out);
int32_t mi_msm_g2_sharded(mi_group *g, const mi_g2_affine *pts, const mi_fr *scalars, size_t n, uint32_t flags,
                          uint32_t mode, mi_g2_jac *out);
/* one MSM whose pairs already sit on the ranks' devices: arrays of mi_group_local() device pointers / counts; n_total = pairs
 * over ALL ranks (every rank passes the same value: it fixes the common window width).  Every rank receives the result. */
int32_t mi_msm_g1_sharded_dev(mi_group *g, const mi_g1_affine *const *pts_dev, const mi_fr *const *scalars_dev,
                              const size_t *n_local, size_t n_total, uint32_t flags, uint32_t mode, mi_g1_jac *out);
int32_t mi_msm_g2_sharded_dev(mi_group *g, const mi_g2_affine *const *pts_dev, const mi_fr *const *scalars_dev,
                              const size_t *n_local, size_t n_total, uint32_t flags, uint32_t mode, mi_g2_jac *out);

/* ---- partial-sum combine for the point-sharded MSM (SURVEY section 8e option i): adds n
 * Jacobian partial results (e.g. all-gathered from the ranks), host side ---- */
int32_t mi_g1_sum(const mi_g1_jac *parts, size_t n, mi_g1_jac *out);
int32_t mi_g2_sum(const mi_g2_jac *parts, size_t n, mi_g2_jac *out);


#ifdef __cplusplus
}
#endif
#endif
