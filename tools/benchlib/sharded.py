"""BASELINE configs[4] through the C-ABI's device group: the legs bench.py runs in a helper process per rank."""
import json
import os
import sys
import time

from .common import _binding


def sharded_msm_section(B, g, rank, world, log_n_msm, steps):
    """BASELINE configs[4]: ONE G1 MSM of 2^log_n_msm pairs, bases point-sharded over the ranks (one per GPU), through the C-ABI's
    device group g (mi_group_create_rank + mi_msm_g1_sharded_dev, csrc/group.hip): mode 0 = all-gather of per-rank partial sums,
    mode 1 = reduce-scatter of bucket sums (grouped ncclSend / ncclRecv) before the bucket reduce.  Strong scaling: total work
    fixed.  Runs in the HELPER PROCESS (see main): no torch, no torch.distributed -- the ranks meet in the group's own collectives;
    returns this rank's seconds per mode, the caller takes the maximum over the ranks."""
    import numpy as np
    n = 1 << log_n_msm
    lo, hi = B.shard_range(n, world, rank)
    c = g.ctx(0)
    pts = c.gen_g1(hi - lo, 4242 + 17 * rank); sc = c.gen_scalars(hi - lo, 2424 + 17 * rank, 0)
    c.sync()
    out = {}
    for mode in (0, 1):
        ref = g.msm_dev([pts.ptr], [sc.ptr], [hi - lo], n, mode=mode)   # warm-up: sizes the workspaces, and lines the ranks up
        c.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            got = g.msm_dev([pts.ptr], [sc.ptr], [hi - lo], n, mode=mode)
        c.sync()
        dt = time.perf_counter() - t0
        assert np.array_equal(got, ref)
        out[mode] = (got, dt)
    res = {"steps": steps, "dt0": out[0][1], "dt1": out[1][1], "modes_agree": bool(np.array_equal(out[0][0], out[1][0]))}
    pts.free(); sc.free()
    return res


def sharded_prove_section(B, g, rank, world, log_n, steps):
    """BASELINE configs[4] as north_star states it: ONE groth16.Prove of an N = 2^log_n circuit over the ranks of the group
    (mi_pk_load_sharded_dev + mi_groth16_prove_sharded_dev, csrc/group.hip): every rank keeps its slice of pk.G1.{A,B,K,Z} / pk.G2.B
    (generated on its own device), rank 0 runs computeH and hands out h slices over the group's transport, the MSMs run point-sharded,
    mode 0 combines per-rank partial sums, mode 1 reduce-scatters bucket sums first.  Inputs resident in HBM.  Strong scaling.
    Validity: (1) a small key (N = 2^16, the SAME on every rank) proved sharded in both modes must give the bytes of the unsharded
    mi_groth16_prove on this rank's own device; (2) at N = 2^log_n both modes must give the same bytes, and with one rank those of the
    unsharded prove of the same key."""
    import numpy as np
    c = g.ctx(0)
    out = {}

    def masks(nb_wires, seed):
        rng = np.random.default_rng(seed)
        return (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8), (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)

    # ---- (1) small parity: whole key on every rank (device generators, same seeds), host arrays -> mi_pk_load_sharded
    ls = 16
    Ns = 1 << ls
    nw, npub, ncs = Ns - 50, 300, Ns - 10
    ia, ib = masks(nw, 99)
    na, nb, nk = int((ia == 0).sum()), int((ib == 0).sum()), nw - npub

    def pull(d, shape):
        o = d.download(shape); d.free(); return o
    small = pull(c.gen_g1(3, 206), (3, 8)); small2 = pull(c.gen_g2(2, 207), (2, 16))
    pk = {"log_n": ls, "nb_public": npub, "nb_wires": nw, "g1_a": pull(c.gen_g1(na, 201), (na, 8)), "g1_b": pull(c.gen_g1(nb, 202), (nb, 8)),
          "g1_k": pull(c.gen_g1(nk, 203), (nk, 8)), "g1_z": pull(c.gen_g1(Ns, 204), (Ns, 8)), "g2_b": pull(c.gen_g2(nb, 205), (nb, 16)),
          "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1], "infinity_a": ia, "infinity_b": ib}
    W = pull(c.gen_scalars(nw, 208, 1), (nw, 4)); a = pull(c.gen_scalars(ncs, 209, 1), (ncs, 4)); b = pull(c.gen_scalars(ncs, 210, 0), (ncs, 4))
    cc = c.field_op(0, 2, a, b)
    rs = pull(c.gen_scalars(2, 211, 0), (2, 4))
    pkh = c.pk_load(pk)
    want = B.proof_write(c.prove(pkh, W, a, b, cc, rs[0], rs[1])[0]["raw"])
    c.pk_free(pkh)
    spk = g.pk_load(pk)
    small_ok = all(B.proof_write(g.prove(spk, W, a if rank == 0 else None, b if rank == 0 else None, cc if rank == 0 else None, rs[0], rs[1], mode=m)[0]["raw"]) == want
                   for m in (0, 1))
    g.pk_free(spk)
    out["small_parity"] = {"log_n": ls, "sharded_equals_unsharded_both_modes": bool(small_ok)}
    if not small_ok:
        raise RuntimeError("sharded proof of the small key differs from the unsharded proof")

    # ---- (2) the big proof: every rank generates ITS slices on its device
    N = 1 << log_n
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    seed = 0x57484952 + 4
    ia, ib = masks(nb_wires, seed)
    lo, hi = g.wire_range(nb_wires, rank); zlo, zhi = B.shard_range(N - 1, world, rank)   # wires by the group's lead share (automatic), the Z pairs evenly
    na, nb = int((ia[lo:hi] == 0).sum()), int((ib[lo:hi] == 0).sum())
    nk = max(hi, nb_public) - max(lo, nb_public)
    rseed = seed + 1000 * rank
    arrs = {"g1_a": (c.gen_g1(na, rseed + 1), na), "g1_b": (c.gen_g1(nb, rseed + 2), nb), "g1_k": (c.gen_g1(nk, rseed + 3), nk),
            "g1_z": (c.gen_g1(zhi - zlo, rseed + 4), zhi - zlo), "g2_b": (c.gen_g2(nb, rseed + 5), nb)}
    hdr = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "alpha1": small[0], "beta1": small[1], "delta1": small[2],
           "beta2": small2[0], "delta2": small2[1], "infinity_a": ia, "infinity_b": ib}
    Wd = c.gen_scalars(hi - lo, rseed + 8, 1)
    da = db = dc = None
    over_ranks = world in (2, 4, 8, 16)   # computeH over the ranks (mi_groth16_prove_sharded_slices_dev): every rank then needs ITS rows of a and b
    if rank == 0 or over_ranks:           # (every rank generates the whole vectors -- same seeds -- and points into them: simple, and 2 x 2 GB at N = 2^26)
        da = c.gen_scalars(n_constraints, seed + 9, 1); db = c.gen_scalars(n_constraints, seed + 10, 0)
    if rank == 0:
        dc = c.alloc(32 * n_constraints)
        c.field_op_dev(0, 2, dc.ptr, da.ptr, db.ptr, n_constraints)
    c.sync()
    ptr = lambda d: None if d is None else d.ptr
    unsharded = None
    if world == 1:   # the same key through the unsharded entry points first (both keys at once would not fit at N = 2^26)
        full = dict(hdr); full.update({k: (v[0].ptr, v[1]) for k, v in arrs.items()})
        pkh = c.pk_load(full, device_points=True)
        unsharded = B.proof_write(c.prove(pkh, Wd.ptr, da.ptr, db.ptr, dc.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)[0]["raw"])
        c.pk_free(pkh)
    t0 = time.perf_counter()
    spk = g.pk_load_dev(hdr, [{k: (v[0].ptr, v[1]) for k, v in arrs.items()}])
    out["pk_load_sharded_s"] = time.perf_counter() - t0
    got = {}
    lead_abc = (ptr(da), ptr(db), ptr(dc)) if rank == 0 else (None, None, None)
    for mode in (0, 1):
        pr, _ = g.prove_dev(spk, [Wd.ptr], nb_wires, *lead_abc, n_constraints, rs[0], rs[1], mode=mode)   # warm-up: sizes the workspaces, lines the ranks up
        t0 = time.perf_counter()
        for _ in range(steps):
            pr, st = g.prove_dev(spk, [Wd.ptr], nb_wires, *lead_abc, n_constraints, rs[0], rs[1], mode=mode)
        got[mode] = (B.proof_write(pr["raw"]), time.perf_counter() - t0, st)
    if over_ranks:   # the same proof with computeH over the ranks: this rank's rows of a and b, c formed on the devices
        M = N // world
        row0 = min(rank * M, n_constraints)
        for mode in (0, 1):
            args = (spk, [Wd.ptr], nb_wires, [da.ptr + 32 * row0], [db.ptr + 32 * row0], None, n_constraints, rs[0], rs[1])
            pr, _ = g.prove_slices_dev(*args, mode=mode)
            t0 = time.perf_counter()
            for _ in range(steps):
                pr, st = g.prove_slices_dev(*args, mode=mode)
            got[2 + mode] = (B.proof_write(pr["raw"]), time.perf_counter() - t0, st)
        out.update({"compute_h_over_ranks": True, "dt2": got[2][1], "dt3": got[3][1], "over_ranks_agree": got[2][0] == got[0][0] and got[3][0] == got[0][0]})
        if not out["over_ranks_agree"]:
            raise RuntimeError("the proof with computeH over the ranks differs from the proof with computeH on the lead")
    g.pk_free(spk)
    for d in [v[0] for v in arrs.values()] + [Wd, da, db, dc]:
        if d is not None:
            d.free()
    out.update({"log_n": log_n, "steps": steps, "dt0": got[0][1], "dt1": got[1][1], "modes_agree": got[0][0] == got[1][0],
                "equals_unsharded": None if unsharded is None else bool(got[0][0] == unsharded),
                "compute_h_ms_on_rank0": got[0][2]["compute_h_ms"] if rank == 0 else None})
    if not out["modes_agree"] or out["equals_unsharded"] is False:
        raise RuntimeError("sharded proofs disagree (mode 0 vs mode 1, or sharded vs unsharded)")
    return out


def sharded_helper_main():
    """`bench.py --sharded-helper`: started by main() BEFORE the parent touches the GPU (a process that has initialised the GPU must not
    exec), idle until the parent writes one JSON line of parameters, then runs the multi-GPU legs on its own GPU context -- the
    transport self-test first (a broken communicator is diagnosed, not timed out), the point-sharded MSM, the point-sharded PROVE --
    and answers with one JSON line.  A fault or a stuck collective in these paths then costs the parent nothing but the `sharded_*`
    blocks of its line."""
    req = sys.stdin.readline()
    if not req.strip():
        return
    q = json.loads(req)
    if os.environ.get("MI_BENCH_HELPER_FAULT") == "abort":   # rehearsal of the failure this process exists for
        os.abort()
    if os.environ.get("MI_BENCH_HELPER_FAULT") == "hang":
        time.sleep(10000)
    res = {"ok": False, "selftest": "not run"}
    g = None
    try:
        B = _binding()
        g = B.Group.rank(q["local_rank"], q["rank"], q["world"], bytes.fromhex(q["uid"]), transport=q.get("transport", 1))
        try:
            g.exchange_selftest(1 << 20)
            res["selftest"] = "ok"
        except BaseException as e:
            res["selftest"] = f"FAILED: {e}"
            raise
        res["transport"] = g.transport()
        res["comm_ranks"] = g.comm_ranks()          # what ncclCommCount says (0: not RCCL)
        res["device_pci"] = g.device_pci(0)         # the GPU this rank's group context really sits on
        if q["log_n"]:
            res["msm"] = sharded_msm_section(B, g, q["rank"], q["world"], q["log_n"], q["steps"])
        res["ok"] = True   # the MSM block is valid from here on, whatever the prove leg does
        if q.get("prove_log_n"):
            try:
                res["prove"] = sharded_prove_section(B, g, q["rank"], q["world"], q["prove_log_n"], q["prove_steps"])
                res["prove"]["ok"] = True
            except BaseException as e:
                res["prove"] = {"ok": False, "error": f"{type(e).__name__}: {e}"}
    except BaseException as e:
        res["error"] = f"{type(e).__name__}: {e}"
    finally:
        if g is not None:
            g.close()
    print(json.dumps(res), flush=True)


def run_sharded_legs(helper, B, torch, dist, rank, local_rank, world, args):
    """configs[4] through the C-ABI's device group, in the helper process started at the top of main(): the transport self-test, one G1
    MSM point-sharded over the ranks, and ONE PROOF point-sharded over the ranks.  Transport: RCCL with one rank per GPU; the
    host-staged one (shared memory) with --rehearse-on-one-gpu, where every rank sits on device 0 and RCCL would refuse.  Bounded by a
    watchdog: a stuck collective or a fault there must not cost the run its proofs/s line.  Called after this process has released its
    own pool, key and buffers (an N = 2^26 proof wants most of a GPU)."""
    import threading
    sharded, sharded_prove = {"done": False}, {"done": False}
    transport = 3 if args.rehearse_on_one_gpu else 1
    dev = "cpu" if args.rehearse_on_one_gpu else torch.device("cuda", local_rank)
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = torch.tensor(list(os.urandom(128) if transport == 3 else B.Group.unique_id()), dtype=torch.uint8)
    if dist is not None:
        t = uid.to(dev); dist.broadcast(t, src=0); uid = t.cpu()
        dist.barrier()
    steps_msm, steps_prove = 3, 3
    answer = {}

    def ask():
        try:
            helper.stdin.write(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "uid": bytes(uid.tolist()).hex(), "transport": transport,
                                           "log_n": args.sharded_msm_log_n, "steps": steps_msm,
                                           "prove_log_n": args.sharded_prove_log_n, "prove_steps": steps_prove}) + "\n")
            helper.stdin.flush()
            while True:   # the answer is the first line that is a JSON object (anything a library prints before it is skipped)
                ln = helper.stdout.readline()
                if not ln or ln.lstrip().startswith("{"):
                    break
            answer["line"] = ln
        except BaseException as e:
            answer["line"] = json.dumps({"ok": False, "error": f"{type(e).__name__}: {e}"})
    th = threading.Thread(target=ask, daemon=True)
    th.start()
    watchdog_s = 360
    th.join(timeout=watchdog_s)
    res = {"ok": False, "error": f"timeout after {watchdog_s} s (collective stuck?)"}
    if th.is_alive():
        helper.kill()
    else:
        try:
            res = json.loads(answer.get("line") or "") if (answer.get("line") or "").strip() else {"ok": False, "error": "helper ended without an answer"}
        except ValueError:
            res = {"ok": False, "error": "helper answered garbage"}
    m, pv = res.get("msm") or {}, res.get("prove") or {}
    ok = 1.0 if res.get("ok") and m else 0.0
    okp = 1.0 if pv.get("ok") else 0.0
    v = [ok, float(m.get("dt0", 0.0)), float(m.get("dt1", 0.0)), 1.0 if m.get("modes_agree") else 0.0]
    w = [okp, float(pv.get("dt0", 0.0)), float(pv.get("dt1", 0.0)), float(pv.get("dt2", 0.0)), float(pv.get("dt3", 0.0))]
    if dist is not None:   # every rank takes part, whatever its helper did: all ok?  slowest rank's times; all agree?
        tmin = torch.tensor([v[0], v[3], w[0]], device=dev, dtype=torch.float64); dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        tmax = torch.tensor([v[1], v[2], w[1], w[2], w[3], w[4]], device=dev, dtype=torch.float64); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        v = [float(tmin[0]), float(tmax[0]), float(tmax[1]), float(tmin[1])]
        w = [float(tmin[2]), float(tmax[2]), float(tmax[3]), float(tmax[4]), float(tmax[5])]
    n_msm = 1 << args.sharded_msm_log_n
    sharded["selftest"] = sharded_prove["selftest"] = res.get("selftest", "not run")
    # OBSERVED, not derived from `world`: every rank's PCI bus id (from its helper's group context) and the rank count its RCCL
    # communicator reports, gathered over the ranks
    obs = [(res.get("device_pci"), res.get("comm_ranks"))]
    if dist is not None:
        allobs = [None] * world
        dist.all_gather_object(allobs, obs[0])
        obs = allobs
    pcis = [o[0] for o in obs if o and o[0]]
    counts = sorted({o[1] for o in obs if o and o[1] is not None})
    observed = {"distinct_devices": len(set(pcis)) if len(pcis) == world else None, "device_pci_by_rank": pcis,
                "rccl_ranks_seen": None if transport != 1 else (counts[0] if len(counts) == 1 else counts),
                "transport": res.get("transport"),
                "how": "device_pci: hipDeviceGetPCIBusId of each rank's group context (mi_group_device_pci); rccl_ranks_seen: ncclCommCount of each rank's communicator "
                       "(mi_group_comm_ranks; null under the host-staged rehearsal transport, which has no communicator)"}
    sharded["observed"] = sharded_prove["observed"] = observed
    devices = ("ONE device shared by all ranks (rehearsal: the multi-process code path, not a scaling measurement)" if args.rehearse_on_one_gpu
               else f"{observed['distinct_devices']} distinct device(s) observed for {world} rank(s)")
    if v[0] == 1.0 and v[1] > 0 and v[2] > 0:
        sharded.update({"workload": f"one G1 MSM, 2^{args.sharded_msm_log_n} uniform pairs, bases point-sharded over {world} rank(s) (BASELINE configs[4])",
                        "scaling": "strong", "transport": res.get("transport"), "devices": devices, "steps": steps_msm, "process": "helper process per rank (own GPU context)",
                        "mode0_partial_sums_pts_per_s": n_msm * steps_msm / v[1], "mode0_ms": v[1] / steps_msm * 1e3,
                        "mode1_bucket_exchange_pts_per_s": n_msm * steps_msm / v[2], "mode1_ms": v[2] / steps_msm * 1e3,
                        "modes_agree": v[3] == 1.0, "done": True})
    elif not args.sharded_msm_log_n:
        sharded["skipped"] = "--sharded-msm-log-n 0"
    else:
        sharded["error"] = res.get("error", "a rank's helper failed")
    if w[0] == 1.0 and w[1] > 0 and w[2] > 0:
        sharded_prove.update({"workload": f"ONE Groth16 proof, FFT domain N=2^{args.sharded_prove_log_n}, WHIR-verifier-shaped synthetic key point-sharded over {world} rank(s): "
                                          "slice r of pk.G1.{A,B,K,Z} / pk.G2.B resident on rank r, computeH on rank 0, h slices over the group's transport (BASELINE configs[4])",
                              "scaling": "strong", "transport": res.get("transport"), "devices": devices, "steps": steps_prove, "inputs": "resident in HBM (mi_groth16_prove_sharded_dev)",
                              "mode0_partial_sums_ms_per_proof": w[1] / steps_prove * 1e3, "mode0_proofs_per_s": steps_prove / w[1],
                              "mode1_bucket_exchange_ms_per_proof": w[2] / steps_prove * 1e3, "mode1_proofs_per_s": steps_prove / w[2],
                              "compute_h_over_ranks": None if not (w[3] > 0 and w[4] > 0) else {
                                  "what": "the same proof through mi_groth16_prove_sharded_slices_dev: computeH as local size-N/ranks transforms + cross-rank steps between all-to-alls, every rank's h slice born where its Z pairs live (DESIGN.md 6)",
                                  "mode0_ms_per_proof": w[3] / steps_prove * 1e3, "mode1_ms_per_proof": w[4] / steps_prove * 1e3, "bytes_equal_the_lead_computeH_proof": pv.get("over_ranks_agree")},
                              "modes_agree": pv.get("modes_agree"), "equals_unsharded_prove": pv.get("equals_unsharded"),
                              "small_parity": pv.get("small_parity"), "compute_h_ms_on_rank0": pv.get("compute_h_ms_on_rank0"),
                              "pk_load_sharded_s": pv.get("pk_load_sharded_s"),
                              "note": "NO SCALING CURVE EXISTS until this runs with n_gpus > 1 on distinct devices: with n_gpus = 1 this is the same code path over a world-1 RCCL "
                                      "communicator; the one-rank-per-process flow with world 2 and 3 is parity-tested on one GPU over the host-staged transport (tests/test_gpu_group_multiprocess.py)",
                              "done": True})
    elif not args.sharded_prove_log_n:
        sharded_prove["skipped"] = "--sharded-prove-log-n 0"
    else:
        sharded_prove["error"] = pv.get("error") or res.get("error", "a rank's helper failed")
    try:
        helper.stdin.close(); helper.wait(timeout=10)
    except BaseException:
        helper.kill()
    return sharded, sharded_prove
