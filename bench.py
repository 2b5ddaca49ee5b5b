"""bench.py --gpus N --steps K --warmup W

One step = the proof the WHIR-verifier circuit REALLY produces (reference mt.go:496 after gnark's solve), on the path a caller can
reach: W, a, b in HOST memory (Go slices, mt.go:494-496; c = a o b is formed on the device), one BSB22 commitment
(utilities/utilities.go:189 logderivlookup.New and mtUtilities.go:452 uints.New force it into every proof):
    pedersen Commit (synchronous, as inside the solve)  ->  groth16 prove: computeH (gnark's 7 NTTs of size 2^23, done with 6),
    4 G1 MSMs + 1 G2 MSM with the committed wires removed from K, the commitment's ProveKnowledge MSM beside them, fold  ->
    Proof.WriteTo = 196 bytes, compared with the oracle's bytes.
Workload: the synthetic WHIR-verifier-shaped key / witness of BASELINE.json configs[1] (2^20-variable multilinear -> FFT domain
N = 2^23).  The K steps are issued by in_flight + 1 caller threads through the prover pool (mi_prover_*: --in-flight
proofs overlap on the GPU, default 3 -- a prover service calling groth16.Prove from several goroutines); every step starts and
completes inside the timed region.  N > 1 GPUs: one independent pool per GPU (configs[3], no data-path collective) -> weak scaling,
value = proofs of all ranks / max-rank time.  `python bench.py --gpus N` launched PLAINLY starts its N rank processes itself
(children, before any GPU call); under torch.distributed.run it uses the ranks it is given.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` for the dominant kernel (G1 level-1 bucket
accumulate), `roofline_ntt`, `random_gather` / `valu` (the two ceilings that kernel runs against), `value_hbm_resident_inputs` (the
GPU-side rate: the same key, inputs already in HBM, no commitment), the single-proof latencies, `sharded_msm` / `sharded_prove`
(BASELINE configs[4] through the C-ABI's device group) and `cpu_baseline` (the oracle's C restatement on the host cores, rank 0,
N = 1 only: the whole step of the benchmarked workload itself -- Commit, ProveKnowledge, fold, prove -- 196 bytes compared).
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _binding():
    spec = importlib.util.spec_from_file_location("gnark_whir_amd_binding", os.path.join(ROOT, "gnark-whir_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["gnark_whir_amd_binding"] = mod
    spec.loader.exec_module(mod)
    return mod


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def cpu_baseline(pk_host, W, a, b, c, r, s, ped, log_n, gpu_proof_bytes):
    """Times the oracle (oracle/groth16_ref.c, OpenMP, all host cores) ON THE BENCHMARKED STEP ITSELF -- the same key, witness, solution
    vectors, (r, s), Pedersen key, committed values and challenge the GPU proofs above used, downloaded from the device -- and compares
    its 196 proof bytes with the GPU's (same run, identical inputs, byte for byte).  No extrapolation.
    The oracle is the checker and the reported baseline, never the product."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cref
    cref.WAIT_POLICY = None   # keep libgomp's default (spinning) waits for the timed sample: the CPU side at its best
    cref.NATIVE = True        # ... compiled for THIS machine (-march=native, on the spot); the portable library serves if that fails
    runs = []
    cpu_bytes = None
    for _ in range(2):   # the first call also spins up the OpenMP team; a second one only if the first was short
        t0 = time.perf_counter()
        if ped is not None:
            basis, sigma, values, challenge = ped
            cm = cref.pedersen_msm(basis, values)                                        # Commit (inside the solve in gnark)
            pok = cref.pedersen_fold(cref.pedersen_msm(sigma, values).reshape(1, 8), challenge)   # ProveKnowledge + fold
        want = cref.prove(pk_host, W, a, b, c, r, s)
        runs.append(time.perf_counter() - t0)
        cpu_bytes = cref.proof_write(want["raw"], cm.reshape(1, 8), pok) if ped is not None else cref.proof_write(want["raw"])
        if sum(runs) > 25.0:
            break
    n_vals = 0 if ped is None else ped[2].shape[0]
    if cpu_bytes != gpu_proof_bytes:
        raise SystemExit("bench.py: GPU proof bytes differ from the oracle's proof bytes on the same inputs")
    dt = min(runs)
    return {"value": 1.0 / dt, "unit": "proofs/s", "cores": cref.num_threads(),
            "kind": f"port (the oracle's portable C restatement, 4 x 64-bit CIOS with unsigned __int128, no assembly, no ADX/BMI2 intrinsics; gcc {cref.BUILD_FLAGS}; OpenMP): "
                    "a stated baseline, slower than gnark-crypto's assembly field arithmetic -- never gnark",
            "sample": f"N=2^{log_n}, measured: the whole benchmarked step itself (Commit + ProveKnowledge + fold over {n_vals} committed values, prove; same pk, W, a, b, c, r, s; "
                      f"best of {len(runs)}: {dt:.2f} s on {cref.num_threads()} threads); proof bytes equal the GPU proof's ({len(cpu_bytes)} B compared)",
            "proof_bytes_match": True, "proof_bytes_compared": len(cpu_bytes)}


def _kernel_short(name):
    """'void k_msm_accum_affine29<4, 3>(Affine<...> const*, ...)' -> 'k_msm_accum_affine29' (signature dropped; template arguments of the level-1 kernels too)"""
    k = name.split("(")[0].replace("void ", "")
    return k.split("<")[0] if k.startswith(("k_msm_accum_affine29", "k_msm_accum_affine_g2_29")) else k


class ClockSampler:
    """Reads the GPU's shader clock and package power (rocm-smi, read-only; a child process every ~0.2 s from a thread of rank 0) while the
    timed regions run: the roofline figures assume 2.4 GHz, the chip decides what it sustains under this instruction mix."""

    def __init__(self, device):
        import threading
        self.device, self.samples, self._stop = device, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _once(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.device), "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=5).stdout
            rows = [r for r in out.splitlines() if r.strip()]
            d = dict(zip(rows[0].split(","), rows[1].split(",")))
            sclk = int(re.sub(r"[^0-9]", "", d.get("sclk clock speed:", "")) or 0)
            pw = float(d.get("Current Socket Graphics Package Power (W)", "0") or 0)
            if sclk > 0:
                self.samples.append((sclk, pw))
        except Exception:
            pass

    def _run(self):
        while not self._stop.is_set():
            self._once()
            self._stop.wait(0.2)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(timeout=10)
        if not self.samples:
            return None
        sc = [x[0] for x in self.samples]; pw = [x[1] for x in self.samples]
        return {"sclk_mhz_mean": sum(sc) / len(sc), "sclk_mhz_min": min(sc), "sclk_mhz_max": max(sc), "package_power_w_mean": sum(pw) / len(pw), "samples": len(sc),
                "how": "rocm-smi --showclocks --showpower every ~0.2 s over the HBM-resident timed region (rank 0's GPU; the host is idle there)",
                "note": "every roofline / issue-floor figure of this line assumes 2.4 GHz; the chip sustains what its power management allows for the instruction mix"}


def live_pmc(script, script_args, counters, timeout_s=240):
    """Hardware counters MEASURED BY THIS RUN: one child `rocprofv3 --pmc <counter>` per counter (separate passes, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes; the program itself right after `--`) over tools/<script>.  Returns
    {counter: {kernel: {"launches", "total", "per_launch"}}, "seconds": s} (values as rocprofv3 reports them, summed over the counter's dimensions:
    KB for FETCH_SIZE / WRITE_SIZE) or {"error": ...}.  Children of a process that holds the GPU are started, never exec'ed into; the parent
    is idle meanwhile (called after the timed regions)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    t0 = time.time()
    out = {}
    tmp = tempfile.mkdtemp(prefix="live_pmc_", dir="/tmp")
    try:
        for counter in counters:
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", script)] + [str(x) for x in script_args]
            # (a session of its own: on a timeout the whole group goes -- the profiler AND the program under it -- not just the direct child)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                out_b, _ = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return {"error": f"{counter} pass over {script}: no answer within {timeout_s} s"}
            if pr.returncode != 0:
                return {"error": f"{counter} pass over {script}: rc {pr.returncode}: {out_b.decode(errors='replace')[-200:]}"}
            dbs = glob.glob(d + "/**/*_results.db", recursive=True)
            if not dbs:
                return {"error": f"{counter} pass over {script} wrote no rocpd database"}
            agg = defaultdict(lambda: [set(), 0.0])
            for k, did, v in sqlite3.connect(dbs[0]).execute("select kernel_name, dispatch_id, value from counters_collection where counter_name=?", (counter,)):
                k = _kernel_short(k); agg[k][0].add(did); agg[k][1] += v
            out[counter] = {k: {"launches": len(v[0]), "total": v[1], "per_launch": v[1] / len(v[0])} for k, v in agg.items()}
        out["seconds"] = time.time() - t0
        return out
    except Exception as e:   # a timeout, a refused profiler, an unreadable database: the line falls back to the committed passes and says so
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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
    devices = "ONE device shared by all ranks (rehearsal: the multi-process code path, not a scaling measurement)" if args.rehearse_on_one_gpu else f"{world} device(s)"
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


def bind_to_gpu_numa_node(torch, local_rank):
    """One rank per GPU on a multi-socket node: keep this rank's threads -- and, by first touch, the host buffers it is about to allocate,
    which the uploader reads at ~25 GB/s per rank -- on the NUMA node the GPU hangs off.  Best effort (sysfs may say -1 or be unreadable;
    the box may confine the process to other cores): returns a short description for the line, or None when nothing was changed."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        pick = cpus & allowed
        if len(pick) < 8:   # too few of that node's cores are ours: leave the affinity alone
            return None
        os.sched_setaffinity(0, pick)
        return f"GPU {bdf} on NUMA node {node}: {len(pick)} cores"
    except Exception:
        return None


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N rank processes as CHILDREN of this process -- which has not touched
    the GPU and never will -- with the environment torch.distributed.run would give them, relay rank 0's JSON line, and exit with the
    worst child status.  (A re-exec of this process would do as well here, but the rule is: children, before any GPU call.)"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line = None
    for ln in procs[0].stdout:   # rank 0 prints the line; anything else it writes to stdout goes to stderr here
        if ln.lstrip().startswith("{"):
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rcs = [p.wait() for p in procs]
    if line is not None:
        print(line, flush=True)
    bad = [rc for rc in rcs if rc != 0]
    if bad or line is None:
        sys.stderr.write(f"bench.py: rank exit codes {rcs}\n")
        sys.exit(bad[0] if bad else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)   # ~1 s of proofs: one rare runtime hiccup then moves the result by < 2 %
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=23, help="FFT domain (2^23 = BASELINE configs[1])")
    ap.add_argument("--dist", choices=["whir", "half", "uniform"], default="whir",
                    help="witness (W, a) distribution: the WHIR mix of SURVEY 8d (a documented guess), uniform Fr, or half of each (row by row)")
    ap.add_argument("--no-solo-legs", action="store_true", help="skip the solo MSM / computeH / probe legs after the proofs (PMC passes: their launches would mix into the per-kernel averages); the roofline line then falls back to the in-job launch")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the `sensitivity` legs (the same key proved with a half-uniform and a uniform witness)")
    ap.add_argument("--stream-plan", type=int, default=-1, help="tuning: mi_debug_set_stream_plan before the pool is created (0, 1, 2; -1 = the library's default)")
    ap.add_argument("--knobs", default="", help="tuning: name=value,... for mi_debug_set_knob on every context of the pool (include/mi355x_groth16.h lists the names)")
    ap.add_argument("--n-committed", type=int, default=-1,
                    help="private wires under the proof's ONE BSB22 commitment (lookup operands; default 2^18 = N / 32, a documented estimate like the infinity ratios; 0 = a circuit without lookups: 164-byte proofs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-samples", action="store_true", help="do not sample rocm-smi (shader clock, package power) during the timed regions")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure the roofline launch's HBM traffic in THIS run (then: the committed passes)")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the second timed region (inputs already in HBM, no commitment: the GPU-side rate)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="proofs kept in flight per GPU by the prover pool (mi_prover_*); 1 = strictly one proof at a time; "
                         "0 = 3 up to N=2^24, 1 above (a context's workspaces take about 1.2 KB x N of HBM)")
    ap.add_argument("--callers", type=int, default=0, help="caller threads that issue the steps (each: Commit, submit, wait, WriteTo); 0 = in_flight + 1")
    ap.add_argument("--msm-plan", default="", help="tuning: c,L1,L2,seg,G for mi_debug_set_msm_plan on every context (0 = automatic)")
    ap.add_argument("--fixed-base", default="", help="tuning: c_ak,c_b,c_z for mi_debug_set_prove_fixed_base before the key is loaded (0 = automatic, 1 = off)")
    ap.add_argument("--ntt-plan", default="", help="tuning: log_e,max_contig,max_strided[,threads] for mi_debug_set_ntt_plan / _threads on every context")
    ap.add_argument("--msm-group-bits", type=int, default=0, help="tuning: mi_debug_set_msm_group_bits on every context")
    ap.add_argument("--no-limb29", action="store_true", help="tuning: G1 level-1 accumulation in 8 x 32-bit limbs (mi_debug_set_msm_limb29(0)) instead of 9 x 29-bit")
    ap.add_argument("--g1-waves", type=int, default=3, choices=(2, 3), help="tuning: build of the G1 level-1 kernel (mi_debug_set_msm_l1_waves): 3 waves per SIMD (default) or 2")
    ap.add_argument("--bound-levels", action="store_true", help="tuning: mi_debug_set_msm_bound_levels(1): worst-case number of item levels per MSM instead of what the fullest bucket needs")
    ap.add_argument("--hold-accum", action="store_true", help="tuning: mi_debug_set_prove_schedule(1): the wire MSMs' bucket accumulations wait for computeH (measured: no gain)")
    ap.add_argument("--msm-chunk", type=int, default=0, help="tuning: mi_debug_set_msm_chunk on every context")
    ap.add_argument("--sharded-msm-log-n", type=int, default=26, help="configs[4]: size of the point-sharded G1 MSM run after the proofs (0 = skip)")
    ap.add_argument("--sharded-prove-log-n", type=int, default=26, help="configs[4]: FFT domain of the ONE proof point-sharded over the ranks, run after the proofs (0 = skip)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses device 0, torch.distributed runs over gloo and the device group over its "
                         "host-staged transport (size the run to fit: e.g. --log-n 20 --sharded-msm-log-n 22 --sharded-prove-log-n 20)")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, join them (gloo, no GPU) and print a one-line JSON summary: what the CPU test of the self-launch asserts")
    ap.add_argument("--sharded-helper", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.sharded_helper:
        return sharded_helper_main()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: either launch plainly (bench.py starts its ranks itself) or with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_check:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank)], dtype=torch.float64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": world, "sum_of_ranks": float(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    # the point-sharded legs run in a helper process of its own; it is started NOW, before anything here touches the GPU
    helper = None
    if args.sharded_msm_log_n or args.sharded_prove_log_n:
        import subprocess
        helper = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sharded-helper"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:   # one rank per GPU over RCCL (launched by torch.distributed.run, or by self_launch above)
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"
    numa = bind_to_gpu_numa_node(torch, local_rank) if world > 1 and not args.rehearse_on_one_gpu else None   # (N = 1 keeps every core: the CPU baseline wants them)

    B = _binding()
    if args.stream_plan >= 0:
        assert B.load().mi_debug_set_stream_plan(args.stream_plan) == 0
    # the prover pool: --in-flight contexts on this rank's GPU (own streams, workspaces, host worker thread), one shared key
    pool = B.Prover(local_rank, args.in_flight if args.in_flight > 0 else (3 if args.log_n <= 24 else 1))
    ctx = pool.ctx(0)
    for i in range(pool.in_flight):
        assert pool.lib.mi_debug_set_msm_limb29(pool.ctx(i).h, 0 if args.no_limb29 else 1) == 0
        assert pool.lib.mi_debug_set_msm_l1_waves(pool.ctx(i).h, args.g1_waves) == 0
        assert pool.lib.mi_debug_set_msm_group_bits(pool.ctx(i).h, args.msm_group_bits) == 0
        assert pool.lib.mi_debug_set_msm_chunk(pool.ctx(i).h, args.msm_chunk) == 0
        assert pool.lib.mi_debug_set_msm_bound_levels(pool.ctx(i).h, 1 if args.bound_levels else 0) == 0
        assert pool.lib.mi_debug_set_prove_schedule(pool.ctx(i).h, 1 if args.hold_accum else 0) == 0
    for part in [x for x in args.knobs.split(",") if x]:
        k_, _, v_ = part.partition("=")
        pool.set_knob(k_.strip(), int(v_))
    if args.ntt_plan:
        np_ = [int(x) for x in args.ntt_plan.split(",")]
        for i in range(pool.in_flight):
            assert pool.lib.mi_debug_set_ntt_plan(pool.ctx(i).h, *np_[:3]) == 0
            if len(np_) > 3:
                assert pool.lib.mi_debug_set_ntt_threads(pool.ctx(i).h, np_[3]) == 0
    if args.msm_plan:
        plan = [int(x) for x in args.msm_plan.split(",")]
        for i in range(pool.in_flight):
            assert pool.lib.mi_debug_set_msm_plan(pool.ctx(i).h, *plan) == 0

    # ---- synthetic workload, generated on the device: seed "WHIR" + config index
    import numpy as np
    log_n = args.log_n
    N = 1 << log_n
    seed = 0x57484952 + 1 + 1000 * rank
    nb_wires, nb_public, n_constraints = N - 1000, 4097, N - 100
    rng = np.random.default_rng(seed)
    inf_a = (rng.integers(0, 100, nb_wires) < 10).astype(np.uint8)
    inf_b = (rng.integers(0, 100, nb_wires) < 50).astype(np.uint8)
    # the ONE BSB22 commitment of the WHIR circuit: n_committed private wires (the lookup operands: STIR indices, every byte that passes a
    # range check) + the commitment wire itself leave the K MSM (gnark: PrivateCommitted + CommitmentIndex); the estimate's sensitivity
    # (2^16 / 2^18 / 2^20) is in profiles/
    n_committed = args.n_committed if args.n_committed >= 0 else max(1, N >> 5)
    committed_private = committed_wires = None
    if n_committed:
        committed_private = np.sort(np.random.default_rng(seed + 77).choice(nb_wires - 1 - nb_public, n_committed, replace=False).astype(np.uint32) + np.uint32(nb_public))
        committed_wires = np.concatenate([committed_private, np.array([nb_wires - 1], dtype=np.uint32)])   # the last wire plays CommitmentIndex
    na, nb = int((inf_a == 0).sum()), int((inf_b == 0).sum())
    nk = nb_wires - nb_public - (n_committed + 1 if n_committed else 0)
    dist_id = 0 if args.dist == "uniform" else 1
    g1a, g1b, g1k, g1z = ctx.gen_g1(na, seed + 1), ctx.gen_g1(nb, seed + 2), ctx.gen_g1(nk, seed + 3), ctx.gen_g1(N, seed + 4)
    g2b = ctx.gen_g2(nb, seed + 5)
    small = ctx.gen_g1(3, seed + 6).download((3, 8)); small2 = ctx.gen_g2(2, seed + 7).download((2, 16))
    pk = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires,
          "g1_a": (g1a.ptr, na), "g1_b": (g1b.ptr, nb), "g1_k": (g1k.ptr, nk), "g1_z": (g1z.ptr, N), "g2_b": (g2b.ptr, nb),
          "alpha1": small[0], "beta1": small[1], "delta1": small[2], "beta2": small2[0], "delta2": small2[1],
          "infinity_a": inf_a, "infinity_b": inf_b, "committed_wires": committed_wires}
    if args.fixed_base:
        assert pool.lib.mi_debug_set_prove_fixed_base(ctx.h, *[int(x) for x in args.fixed_base.split(",")]) == 0
    ctx.sync()
    t_load = time.perf_counter()
    pkh = ctx.pk_load(pk, device_points=True)   # includes building the fixed-base window tables (once per key)
    t_load = time.perf_counter() - t_load
    # The key took its own copies of every base it gathers from (per-wire expanded A / K, window tables, R'-form copies): the generated
    # source arrays of A, B, K, G2.B are dead weight from here on (16 GB at N = 2^26).  They leave the device -- to the host first when
    # the CPU baseline will want them; pk.G1.Z stays for the solo-MSM leg.  (With --no-limb29 the key references B / G2.B / Z in place.)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    key_host = {}
    if not args.no_limb29:
        for name, d, cnt, k in (("g1_a", g1a, na, 8), ("g1_b", g1b, nb, 8), ("g1_k", g1k, nk, 8), ("g2_b", g2b, nb, 16)):
            if want_cpu:
                key_host[name] = d.download((cnt, k))
            d.free()
    def gen_witness(dist):
        """W, a (the distribution under test), b (uniform), c = a o b on the device; 'half': every other row of W and a by a seeded coin uniform"""
        did = 0 if dist == "uniform" else 1
        W_ = ctx.gen_scalars(nb_wires, seed + 8, did); a_ = ctx.gen_scalars(n_constraints, seed + 9, did)
        if dist == "half":
            for arr, cnt, sd in ((W_, nb_wires, 8), (a_, n_constraints, 9)):
                u = ctx.gen_scalars(cnt, seed + sd, 0)
                hm, hu = arr.download((cnt, 4)), u.download((cnt, 4)); u.free()
                coin = np.random.default_rng(seed + 100 + sd).integers(0, 2, cnt).astype(bool)
                hm[coin] = hu[coin]
                arr.upload(hm)
        b_ = ctx.gen_scalars(n_constraints, seed + 10, 0)
        c_ = ctx.alloc(32 * n_constraints)
        ctx.field_op_dev(0, 2, c_.ptr, a_.ptr, b_.ptr, n_constraints)   # c = a*b so that (a, b, c) is a satisfied R1CS row set
        return W_, a_, b_, c_
    W, a, b, c = gen_witness(args.dist)
    rs = ctx.gen_scalars(3, seed + 11, 0).download((3, 4))       # r, s and the PoK fold challenge (gnark: fr.Hash of the commitment wire values, "G16-BSB22" -- Go's part)
    ctx.sync()
    # what the caller holds: host memory (the solver's output)
    Wh, ah, bh = W.download((nb_wires, 4)), a.download((n_constraints, 4)), b.download((n_constraints, 4))
    # Pedersen key of the commitment (pk.CommitmentKeys[0]: Basis, BasisExpSigma) and the private committed values the hint receives
    ped = ped_basis = ped_sigma = values = None
    if n_committed:
        ped_basis = ctx.gen_g1(n_committed, seed + 12).download((n_committed, 8)); ped_sigma = ctx.gen_g1(n_committed, seed + 13).download((n_committed, 8))
        ped = ctx.pedersen_pk_load(ped_basis, ped_sigma)
        values = np.ascontiguousarray(Wh[committed_private])

    def proof_bytes(proof, cm):
        return B.proof_write(proof["raw"]) if not n_committed else B.proof_write(proof["raw"], cm.reshape(1, 8), proof["pok"])

    def one_step(with_c=False):
        """the whole per-proof sequence of one caller (one goroutine of a prover service): Commit inside the solve, prove, WriteTo"""
        if not n_committed:
            proof, st = pool.wait(pool.submit(pkh, Wh, ah, bh, ch if with_c else None, rs[0], rs[1]))
            return proof_bytes(proof, None), st
        cm = pool.commit(ped, values)
        proof, st = pool.wait(pool.submit_bsb22(pkh, Wh, ah, bh, ch if with_c else None, rs[0], rs[1], [(ped, values)], rs[2]))
        return proof_bytes(proof, cm), st

    # untimed: size every context's workspaces (a pool job goes to whichever worker is free, so warm each one directly while
    # the workers are idle) and take the reference bytes from the plain entry points, one call at a time on context 0
    serial_ms = None
    body_bytes = None
    for i in range(pool.in_flight):
        ci = pool.ctx(i)
        for k in range(3 if i == 0 else 1):
            t1 = time.perf_counter()
            pr, _ = ci.prove(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
            if i == 0:
                serial_ms = (time.perf_counter() - t1) * 1e3
            if body_bytes is None:
                body_bytes = B.proof_write(pr["raw"])
            elif B.proof_write(pr["raw"]) != body_bytes:
                raise SystemExit("bench.py: proof bytes differ between contexts / repetitions on the same inputs")
    serial_bytes = body_bytes
    if n_committed:   # Commit, ProveKnowledge and fold through the plain one-context entry points: the reference the pool's proofs must equal
        cm0 = ctx.pedersen_commit(ped, values)
        pok0 = B.pedersen_fold(ctx.pedersen_commit(ped, values, knowledge=True).reshape(1, 8), rs[2])
        serial_bytes = B.proof_write(pr["raw"], cm0.reshape(1, 8), pok0)
    ch = None
    # single-proof latency on the caller's path (host inputs, nothing else on the GPU): Commit + submit -> wait, c formed on the device /
    # c uploaded as well
    lat_host = lat_host_with_c = lat_commit = None
    for k in range(3):
        t1 = time.perf_counter()
        bts, _ = one_step()
        lat_host = (time.perf_counter() - t1) * 1e3
        if bts != serial_bytes:
            raise SystemExit("bench.py: the pool's host-input proof differs from the one-context reference proof of the same inputs")
    if n_committed:
        t1 = time.perf_counter(); pool.commit(ped, values); lat_commit = (time.perf_counter() - t1) * 1e3
    ch = c.download((n_constraints, 4))
    for k in range(2):
        t1 = time.perf_counter()
        bts, _ = one_step(with_c=True)
        lat_host_with_c = (time.perf_counter() - t1) * 1e3
        if bts != serial_bytes:
            raise SystemExit("bench.py: the host-input proof with c uploaded differs from the reference proof")
    if not want_cpu:
        ch = None
    from concurrent.futures import ThreadPoolExecutor
    callers = args.callers if args.callers > 0 else pool.in_flight + 1
    ex = ThreadPoolExecutor(callers)
    list(ex.map(lambda _: one_step(), range(max(args.warmup, callers))))   # the W warm-up steps, through the same path

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the timed region: K steps
    fence()
    t0 = time.perf_counter()
    done_at = []
    def timed_step(_):
        r_ = one_step()
        done_at.append(time.perf_counter())   # (list.append is atomic under the GIL)
        return r_
    done = list(ex.map(timed_step, range(args.steps)))
    fence()
    dt = time.perf_counter() - t0
    # how evenly the steps completed: one stall of the host or the GPU inside a ~1 s region moves `value` by its whole length
    gaps = sorted((y - x) * 1e3 for x, y in zip([t0] + sorted(done_at)[:-1], sorted(done_at)))
    step_gaps = {"median_ms": gaps[len(gaps) // 2], "p90_ms": gaps[min(len(gaps) - 1, (len(gaps) * 9) // 10)], "max_ms": gaps[-1],
                 "note": "intervals between consecutive step completions inside the timed region (the first one includes the pipeline fill)"} if gaps else None
    if dist is not None:
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    accum_ms, accum_pairs, accum_launches, accum_entries, last = 0.0, 0, 0, 0, None
    h2d = []
    for bts, st in done:
        if bts != serial_bytes:   # every timed proof must be THE proof: same inputs, (r, s), committed values -> the reference bytes
            raise SystemExit("bench.py: a timed proof differs from the untimed reference proof of the same inputs")
        accum_ms += st["g1_accum_kernel_ms"]; accum_pairs += st["g1_accum_pairs"]; accum_launches += st["g1_accum_launches"]; accum_entries += st["g1_accum_entries"]; last = st
        h2d.append(st["h2d_ms"])
    h2d.sort()

    # second timed region: the same key with W, a, b, c ALREADY in HBM and no commitment (mi_prover_submit_dev, 164-byte proof body): the
    # GPU-side rate no caller of the reference can reach (its solver is CPU code) -- reported next to `value`, never as `value`
    dev_rate = dev_ms = clocks = None
    if not args.no_hbm_resident:
        sub = lambda: pool.submit(pkh, W.ptr, a.ptr, b.ptr, c.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
        for t in [sub() for _ in range(args.warmup)]:
            pool.wait(t)
        # (the clock samples are taken HERE, where the host is idle: a rocm-smi child every 0.2 s beside the caller threads and the upload
        #  stage of the first region cost it 4 % -- 32.4 against 33.7 proofs/s on one box -- and the kernels are the same)
        sampler = ClockSampler(local_rank).start() if rank == 0 and not args.no_clock_samples else None
        fence()
        t0d = time.perf_counter()
        dev_done = [pool.wait(t)[0]["raw"] for t in [sub() for _ in range(args.steps)]]
        fence()
        dtd = time.perf_counter() - t0d
        if dist is not None:
            t = torch.tensor([dtd], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtd = float(t.item())
        for raw in dev_done:
            if B.proof_write(raw) != body_bytes:
                raise SystemExit("bench.py: a device-input proof differs from the reference proof of the same inputs")
        clocks = sampler.stop() if sampler is not None else None
        dev_rate, dev_ms = args.steps * world / dtd, dtd / args.steps * 1e3

    # `sensitivity`: the headline rests on ONE guessed input, the witness distribution ("a documented guess").  The same key, the
    # same step, proved with half of the rows of W and a uniform and with all of them uniform (the floor: ~14 non-zero digits per wire
    # scalar instead of ~4): proofs/s on the caller's path, with the inputs in HBM, one proof alone, G1 level-1 additions per proof.
    # Rank 0 of a one-GPU run only; every proof of a leg is compared with that leg's own untimed reference proof.
    def sensitivity_leg(dist):
        W2, a2, b2, c2 = gen_witness(dist)
        ctx.sync()
        W2h, a2h, b2h = W2.download((nb_wires, 4)), a2.download((n_constraints, 4)), b2.download((n_constraints, 4))
        vals2 = np.ascontiguousarray(W2h[committed_private]) if n_committed else None
        lat = None
        for _ in range(2):
            t1 = time.perf_counter()
            pr, st1 = ctx.prove(pkh, W2.ptr, a2.ptr, b2.ptr, c2.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
            lat = (time.perf_counter() - t1) * 1e3
        body2 = B.proof_write(pr["raw"])
        ref2 = body2
        if n_committed:
            cm0 = ctx.pedersen_commit(ped, vals2)
            pok0 = B.pedersen_fold(ctx.pedersen_commit(ped, vals2, knowledge=True).reshape(1, 8), rs[2])
            ref2 = B.proof_write(pr["raw"], cm0.reshape(1, 8), pok0)

        def step2():
            if not n_committed:
                proof, st = pool.wait(pool.submit(pkh, W2h, a2h, b2h, None, rs[0], rs[1]))
                return B.proof_write(proof["raw"]), st
            cm = pool.commit(ped, vals2)
            proof, st = pool.wait(pool.submit_bsb22(pkh, W2h, a2h, b2h, None, rs[0], rs[1], [(ped, vals2)], rs[2]))
            return B.proof_write(proof["raw"], cm.reshape(1, 8), proof["pok"]), st
        list(ex.map(lambda _: step2(), range(max(args.warmup, callers))))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        got = list(ex.map(lambda _: step2(), range(args.steps)))
        torch.cuda.synchronize(); rate = args.steps / (time.perf_counter() - t1)
        if any(bts != ref2 for bts, _ in got):
            raise SystemExit(f"bench.py: a proof of the `{dist}` sensitivity leg differs from its reference proof")
        sub2 = lambda: pool.submit(pkh, W2.ptr, a2.ptr, b2.ptr, c2.ptr, rs[0], rs[1], device=True, n_wires=nb_wires, n_constraints=n_constraints)
        for t in [sub2() for _ in range(args.warmup)]:
            pool.wait(t)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        raws = [pool.wait(t)[0]["raw"] for t in [sub2() for _ in range(args.steps)]]
        torch.cuda.synchronize(); rate_dev = args.steps / (time.perf_counter() - t1)
        if any(B.proof_write(raw) != body2 for raw in raws):
            raise SystemExit(f"bench.py: an HBM-resident proof of the `{dist}` sensitivity leg differs from its reference proof")
        for d in (W2, a2, b2, c2):
            d.free()
        return {"value": rate, "value_hbm_resident_inputs": rate_dev, "single_proof_latency_ms": lat, "g1_level1_additions_per_proof": int(st1["g1_level1_additions"]),
                "proofs_validated": 2 * args.steps}
    sensitivity = None
    if rank == 0 and world == 1 and not args.no_sensitivity and args.dist == "whir":
        sensitivity = {"what": "the same key and step with other witness distributions (rows of W and a): `whir` = the headline's 45 % {0,1} / 25 % bytes / 5 % 64-bit / 25 % "
                               "uniform mix, `half_uniform` = every row uniform with probability 1/2, `uniform` = every row uniform (the floor)",
                       "half_uniform": sensitivity_leg("half"), "uniform": sensitivity_leg("uniform")}
    ex.shutdown()

    # HBM ledger of the PROVE path, taken right after the timed regions: key + tables, every context's workspaces, the pool's input sets,
    # bench.py's own inputs -- before the solo-MSM / solo-computeH / probe legs below grow the (grow-only) workspaces of context 0 for
    # their own shapes (a generic 2^26-pair MSM alone adds ~22 GB at N = 2^26)
    hbm_in_use_gb = (lambda fr_to: (fr_to[1] - fr_to[0]) / 1e9)(torch.cuda.mem_get_info())
    hbm_ledger = None
    if rank == 0:
        hbm_ledger = ctx.mem_ledger(pkh)   # the key + context 0; the other contexts of the pool add their own workspaces
        for i in range(1, pool.in_flight):
            for k, v in pool.ctx(i).mem_ledger().items():
                if k.startswith("ctx_"):
                    hbm_ledger[k] += v
        hbm_ledger["pool_contexts"] = pool.in_flight
        hbm_ledger["pool_input_sets_gb"] = (pool.in_flight + 1) * (nb_wires + 3 * n_constraints) * 32 / 1e9
        hbm_ledger["bench_inputs_gb"] = (nb_wires + 3 * n_constraints) * 32 / 1e9
        hbm_ledger["bench_key_source_arrays_gb"] = (N * 64 if not args.no_limb29 else (na + nb + nk + N) * 64 + nb * 128) / 1e9   # what bench.py itself still holds of the generated bases
    # the same kernel measured alone (no other stream competing for the CUs): one uniform-scalar G1 MSM over pk.G1.Z
    solo = {"skipped": "--no-solo-legs", "pairs": 0, "msm_total_ms": 0.0, "accum_launch_ms": 0.0, "accum_GBps_algorithmic": 0.0, "mixed_adds_per_s": 0.0, "msm_pts_per_s": 0.0}
    if rank == 0 and not args.no_solo_legs:
        ctx.msm_g1_dev(g1z.ptr, b.ptr, n_constraints)   # sizes the generic path's workspaces (the proofs above used the fixed-base tables)
        ctx.msm_g1_dev(g1z.ptr, b.ptr, n_constraints)
        st = ctx.stats()
        solo = {"pairs": n_constraints, "scalars": "uniform", "msm_total_ms": st["total_ms"], "accum_launch_ms": st["g1_accum_kernel_ms"],
                "accum_GBps_algorithmic": 96.0 * n_constraints / (st["g1_accum_kernel_ms"] * 1e-3) / 1e9,
                "mixed_adds_per_s": st["g1_accum_entries"] / (st["g1_accum_kernel_ms"] * 1e-3), "msm_pts_per_s": n_constraints / (st["total_ms"] * 1e-3)}
    # ... and the launch the roofline line is quoted on: the proof's LARGEST level-1 launch -- the Z MSM's (N - 1 uniform scalars against the
    # key's fixed-base window tables, here rebuilt through the public entry points with the key's own window width) -- ALONE on the GPU.
    # A solo launch is a basis a better schedule cannot lower: inside the job the same launch shares the CUs with whatever runs beside it,
    # and the more evenly it shares the longer it takes (rounds 1-4 quoted that in-job duration; it stays in the line as `in_job`).
    zsolo = None
    if rank == 0 and not args.no_solo_legs:
        try:
            cz = ctx.pk_table_plan(pkh)[2]
            n_z = N - 1
            nwin_z = (256 + cz - 1) // cz if cz else 0
            free_b = torch.cuda.mem_get_info()[0]
            if cz and free_b > 1.3 * nwin_z * n_z * 64:
                tab = ctx.msm_precompute(g1z.ptr, n_z, cz)
                ctx.msm_table_to_rprime(tab.ptr, nwin_z * n_z)
                hsc = ctx.gen_scalars(n_z, seed + 21, 0)
                ctx.msm_fixed_dev(tab.ptr, hsc.ptr, n_z, cz, flags=2)   # sizes the workspaces
                best = None
                for _ in range(3):
                    ctx.msm_fixed_dev(tab.ptr, hsc.ptr, n_z, cz, flags=2)
                    st = ctx.stats()
                    if best is None or st["g1_accum_kernel_ms"] < best["g1_accum_kernel_ms"]:
                        best = st
                tab.free(); hsc.free()
                zsolo = {"pairs": n_z, "scalars": "uniform", "window_bits": cz, "windows": nwin_z, "msm_total_ms": best["total_ms"], "accum_launch_ms": best["g1_accum_kernel_ms"],
                         "accum_GBps_algorithmic": 96.0 * n_z / (best["g1_accum_kernel_ms"] * 1e-3) / 1e9, "mixed_adds": int(best["g1_accum_entries"]),
                         "mixed_adds_per_s": best["g1_accum_entries"] / (best["g1_accum_kernel_ms"] * 1e-3), "msm_pts_per_s": n_z / (best["total_ms"] * 1e-3)}
        except B.MiError as e:
            zsolo = {"error": str(e)}
    # computeH alone on the GPU (6 transforms of size N, the pointwise steps fused into the last one's edges): the NTT's own roofline line
    ntt_solo = None
    if rank == 0:
        hbuf = ctx.alloc(32 * N)
        ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, n_constraints, hbuf.ptr)
        ms_h = min((ctx.compute_h_dev(log_n, a.ptr, b.ptr, c.ptr, n_constraints, hbuf.ptr), ctx.stats()["compute_h_ms"])[1] for _ in range(3))
        launches = ctx.stats()["ntt_launches"]
        hbuf.free()
        ntt_solo = {"compute_h_ms": ms_h, "transforms": 6, "pass_launches": launches, "ms_per_transform": ms_h / 6.0}
    # VALU context for the roofline line: the chip's measured 256-bit Montgomery product rate (dependent chains, all CUs)
    modmul_ms = min(ctx.bench_modmul(1, 256 * 4096, 256) for _ in range(3)) if rank == 0 else 0.0
    # ... and the memory system's ceiling for what that kernel asks of it: dependent random 64-byte gathers from a table far larger
    # than the 256 MB Infinity Cache (8.6 GB of scratch: the Z MSM's window tables are 7 GB at N = 2^23, A+K's 15 GB)
    gather_ms = 0.0
    if rank == 0:
        gtab = ctx.alloc(64 << 27)
        gather_ms = min(ctx.bench_gather(gtab.ptr, 1 << 27, 256 * 4 * 64 * 4, 128) for _ in range(3))
        gtab.free()
    # inputs of the CPU baseline leave the device before it is emptied for the sharded legs
    cpu_inputs = None
    if want_cpu:
        dl = lambda d, n, k: d.download((n, k))
        kh = lambda name, d, n, k: key_host[name] if name in key_host else dl(d, n, k)
        pk_host = {"log_n": log_n, "nb_public": nb_public, "nb_wires": nb_wires, "g1_a": kh("g1_a", g1a, na, 8), "g1_b": kh("g1_b", g1b, nb, 8),
                   "g1_k": kh("g1_k", g1k, nk, 8), "g1_z": dl(g1z, N, 8), "g2_b": kh("g2_b", g2b, nb, 16), "alpha1": small[0], "beta1": small[1],
                   "delta1": small[2], "beta2": small2[0], "delta2": small2[1], "infinity_a": inf_a, "infinity_b": inf_b, "committed_wires": committed_wires}
        cpu_inputs = (pk_host, Wh, ah, bh, ch)
    # configs[4]: this process gives its GPU memory back first (an N = 2^26 proof wants most of a GPU), then the helper runs the
    # transport self-test, the point-sharded MSM and the point-sharded PROVE over all ranks
    in_flight = pool.in_flight
    if ped is not None:
        ctx.pedersen_pk_free(ped)
    ctx.pk_free(pkh)
    for d in (g1a, g1b, g1k, g1z, g2b, W, a, b, c):
        d.free()   # (DevArray.free is idempotent)
    pool.close()
    # the roofline launch's HBM traffic, measured now that this process holds nothing on the GPU (rank 0 of a one-GPU run only)
    live_solo = live_proofs = None
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)   # never nest profilers
    if rank == 0 and world == 1 and not args.no_live_pmc and not under_profiler and not args.no_solo_legs and log_n <= 24 and zsolo and "accum_launch_ms" in zsolo:
        live_solo = live_pmc("solo_z_msm.py", [log_n, 2], ("FETCH_SIZE", "WRITE_SIZE"))
        if "error" not in live_solo and args.dist == "whir":   # (tools/prof_proof.py proves this workload with the WHIR mix: four proofs alone on one context)
            live_proofs = live_pmc("prof_proof.py", [log_n, 4], ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"))
    sharded, sharded_prove = {"done": False}, {"done": False}
    if helper is not None:
        sharded, sharded_prove = run_sharded_legs(helper, B, torch, dist, rank, local_rank, world, args)
    if rank == 0:
        proofs = args.steps * world
        # dominant kernel: G1 level-1 bucket accumulate; algorithmic bytes = 96 B per (point, scalar) pair
        per_launch_ms = accum_ms / max(accum_launches, 1)
        per_launch_bytes = 96.0 * accum_pairs / max(accum_launches, 1)
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        g1_pairs_per_proof = na + nb + nk + (N - 1)
        # HBM traffic per launch.  MEASURED BY THIS RUN when it can be (one GPU, no profiler around this process): child runs of
        # `rocprofv3 --pmc` (live_pmc above) over the solo Z-shaped launch and over four proofs of this workload alone on one context.
        # Otherwise -- and only for the profiled shape (N = 2^23, WHIR mix, automatic plans) with unchanged kernel sources -- the
        # committed passes of profiles/r05_pmc_bench_traffic.json (sha256 of the sources recorded in it): a file older than the kernels
        # reads as traffic: null, never as stale bytes.  The accumulate kernel gathers 64-B points, so its FETCH_SIZE is taken raw; the
        # NTT passes stream 16 B per lane, so theirs gets the guide's x2 correction.
        traffic = traffic_ntt = traffic_solo = None
        kname = "k_msm_accum_affine29"
        pmc_file = os.path.join("profiles", "r05_pmc_bench_traffic.json")
        csrc = os.path.join(ROOT, "gnark-whir_amd", "csrc")
        pmc = None
        fresh_msm = fresh_ntt = False
        live_err = "; ".join(f"live passes failed: {x['error']}" for x in (live_solo, live_proofs) if x and "error" in x)
        if live_proofs and "error" not in live_proofs:
            pmc = {c_: {k: {"kb_per_launch": v["per_launch"], "launches": v["launches"]} for k, v in live_proofs[c_].items()} for c_ in ("FETCH_SIZE", "WRITE_SIZE")}
            fresh_msm = fresh_ntt = True
            pmc_src = (f"THIS RUN: child runs `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc SQ_INSTS_VALU` (separate passes) over tools/prof_proof.py {log_n} 4 "
                       f"(four proofs of this workload alone on one context, {live_proofs['seconds']:.0f} s)")
        else:
            pmc_src = f"{pmc_file} (committed PMC passes of this workload, not this run; file sha256 {_sha16(os.path.join(ROOT, pmc_file))})" + (f"; {live_err}" if live_err else "")
            if log_n == 23 and args.dist == "whir" and not (args.msm_plan or args.fixed_base or args.ntt_plan or args.msm_group_bits or args.msm_chunk):
                try:
                    pmc = json.load(open(os.path.join(ROOT, pmc_file)))
                    src_now = {f: _sha16(os.path.join(csrc, f)) for f in pmc["_sources"]}
                    fresh_msm = all(src_now[f] == h for f, h in pmc["_sources"].items() if f.startswith(("msm", "curve29", "field29")))
                    fresh_ntt = all(src_now[f] == h for f, h in pmc["_sources"].items() if f.startswith(("ntt", "field.")))
                    if not fresh_msm:
                        pmc_src += "; STALE for the MSM kernels (their sources changed since the passes): traffic withheld"
                    if not fresh_ntt:
                        pmc_src += "; STALE for the NTT kernels: traffic withheld"
                except Exception:
                    pmc = None
        try:
            if pmc and fresh_msm:
                traffic = (pmc["FETCH_SIZE"][kname]["kb_per_launch"] + pmc["WRITE_SIZE"][kname]["kb_per_launch"]) * 1024.0
                if "solo_z" in pmc:   # the same two counters over the solo Z-shaped launch (tools/solo_z_msm.py under --pmc)
                    traffic_solo = (pmc["solo_z"]["FETCH_SIZE_kb"] + pmc["solo_z"]["WRITE_SIZE_kb"]) * 1024.0
            if pmc and fresh_ntt:
                # per pass launch (the fused contiguous pair -- two launches per computeH -- and the fused strided triple -- one -- counted
                # with their own figures); one transform = a sixth of computeH's traffic
                per = lambda kn: (2.0 * pmc["FETCH_SIZE"][kn]["kb_per_launch"] + pmc["WRITE_SIZE"][kn]["kb_per_launch"]) * 1024.0
                n_pair = 2 if "k_ntt_contig_pair" in pmc["FETCH_SIZE"] else 0
                n_triple = 1 if "k_ntt_strided_triple" in pmc["FETCH_SIZE"] else 0
                n_last = 1 if "k_ntt_contig_last_sub" in pmc["FETCH_SIZE"] else 0
                traffic_ntt = (per("k_ntt_pass_wave") * (ntt_solo["pass_launches"] - n_pair - n_triple - n_last) + (per("k_ntt_contig_pair") * n_pair if n_pair else 0.0) +
                               (per("k_ntt_strided_triple") if n_triple else 0.0) + (per("k_ntt_contig_last_sub") if n_last else 0.0)) / 6.0
        except Exception:
            traffic = traffic_ntt = None
        pmc_src_solo = pmc_src
        if live_solo and "error" not in live_solo and kname in live_solo["FETCH_SIZE"]:
            traffic_solo = (live_solo["FETCH_SIZE"][kname]["per_launch"] + live_solo["WRITE_SIZE"][kname]["per_launch"]) * 1024.0
            pmc_src_solo = (f"THIS RUN: child runs `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) over tools/solo_z_msm.py {log_n} 2 "
                            f"({live_solo['FETCH_SIZE'][kname]['launches']} launches each, {live_solo['seconds']:.0f} s)")
        # the level-1 kernel's own instruction-issue floor, from the committed ISA census of its code object (tools/isa_census.py ->
        # profiles/r05_isa_census_accum_affine29.json: instructions per loop iteration by class x the measured cycles per wave64
        # instruction of profiles/r02_probe_instr_rate.txt)
        issue_floor = census_src = None
        try:
            cen = json.load(open(os.path.join(ROOT, "profiles", "r05_isa_census_accum_affine29.json")))
            issue_floor = 2.4e9 / cen["cycles_per_addition"] * 64 * 1024
            census_src = f"profiles/r05_isa_census_accum_affine29.json: {cen['valu_per_addition']} vector instructions per mixed addition ({cen['mad_u64_u32_per_addition']} v_mad_u64_u32) = {cen['cycles_per_addition']:.0f} cycles per wave-addition"
        except Exception:
            pass
        # roofline of the dominant kernel: achieved = the algorithmic 96 B per pair of ONE launch / that launch's duration.
        # Basis: the Z-shaped launch alone on the GPU (zsolo above) when it ran; the job's average in-job launch otherwise (and always as `in_job`).
        in_job = {"launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": per_launch_bytes, "achieved": achieved, "frac": achieved / 8000.0,
                  "note": "average over the proofs' four G1 level-1 launches while three proofs share the GPU (HIP events on the launch's stream): what rounds 1-4 reported as "
                          "`achieved`; it falls when the launch shares the CUs more evenly with the other streams, i.e. when the job gets FASTER"}
        if zsolo and "accum_launch_ms" in zsolo:
            roofline = {"kernel": "k_msm_accum_affine29 (G1 level-1 bucket accumulate, 9 x 29-bit limbs)", "bound": "hbm",
                        "basis": f"solo launch: the proof's largest level-1 launch (Z MSM: {zsolo['pairs']} uniform scalars, {zsolo['windows']} windows of {zsolo['window_bits']} bits, fixed-base tables) alone on the GPU",
                        "achieved": zsolo["accum_GBps_algorithmic"], "peak": 8000.0, "unit": "GB/s", "frac": zsolo["accum_GBps_algorithmic"] / 8000.0,
                        "traffic": traffic_solo, "traffic_source": pmc_src_solo, "launch_ms": zsolo["accum_launch_ms"], "algorithmic_bytes_per_launch": 96.0 * zsolo["pairs"],
                        "in_job": dict(in_job, traffic=traffic, traffic_source=pmc_src)}
        else:
            roofline = {"kernel": "k_msm_accum_affine29 (G1 level-1 bucket accumulate, 9 x 29-bit limbs)", "bound": "hbm", "basis": "in-job average launch (the solo Z-shaped launch did not run)",
                        "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": pmc_src,
                        "launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": per_launch_bytes, "in_job": in_job}
        # vector-ALU utilisation of the job: wave-instructions per proof (SQ_INSTS_VALU over four proofs alone on one context,
        # tools/prof_proof.py; setup kernels excluded) / step time / the 6.4e11 wave-instructions per second the chip sustains on this
        # instruction mix (1024 SIMDs x 2.4 GHz / 3.84 cycles).  From this run's own pass when it ran; else from the committed pass, and
        # then only while the kernel sources are unchanged.
        valu_util = None
        try:
            setup = ("k_gen_", "k_xyzz_dbl_c", "k_xyzz_batch_to_affine", "k_xyzz_from_affine", "k_g1_to_rprime", "k_g2_to_rprime", "k_expand_points", "k_field_op", "k_pow_table",
                     "k_tw_layout", "k_sc_layout", "k_msm2_precompute", "__amd_rocclr_fillBuffer")
            if live_proofs and "error" not in live_proofs:
                per_kernel = {k: v["total"] / 4.0 for k, v in live_proofs["SQ_INSTS_VALU"].items() if not k.startswith(setup)}
                valu_src = pmc_src
            else:
                import csv
                per_kernel = {}
                for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r05_pmc_valu_proofs.csv"))):
                    k = _kernel_short(r["Kernel"])
                    if r["Counter"] == "SQ_INSTS_VALU" and not k.startswith(setup):
                        per_kernel[k] = per_kernel.get(k, 0.0) + float(r["Sum"]) / 4.0
                valu_src = "profiles/r05_pmc_valu_proofs.csv (committed pass, not this run)"
                if not (log_n == 23 and args.dist == "whir" and traffic is not None):
                    per_kernel = {}
            per_proof = sum(per_kernel.values())
            if per_proof > 0:
                share = lambda *names: sum(v for k, v in per_kernel.items() if k.startswith(names)) / per_proof
                valu_util = {"wave_instructions_per_proof": per_proof, "sustained_wave_instructions_per_s": 6.4e11, "source": valu_src,
                             "on_the_callers_path": per_proof / (dt / args.steps) / 6.4e11,
                             "hbm_resident_inputs": None if dev_ms is None else per_proof / (dev_ms * 1e-3) / 6.4e11,
                             # the same two at the shader clock the chip sustained during the HBM-resident region (1024 SIMDs x sclk / 3.84 cycles)
                             "at_measured_sclk": None if not clocks else {"sclk_mhz": clocks["sclk_mhz_mean"], "on_the_callers_path": per_proof / (dt / args.steps) / (1024 * clocks["sclk_mhz_mean"] * 1e6 / 3.84),
                                                                          "hbm_resident_inputs": None if dev_ms is None else per_proof / (dev_ms * 1e-3) / (1024 * clocks["sclk_mhz_mean"] * 1e6 / 3.84)},
                             "shares": {"g1_level1": share("k_msm_accum_affine29"), "g2_level1": share("k_msm_accum_affine_g2_29"), "ntt": share("k_ntt_"),
                                        "upper_levels_and_finisher": share("k_msm_accum_xyzz", "k_msm_finish"), "sorts": share("k_msm2_", "k_scan_"),
                                        "reduces": share("k_msm_bucket_reduce", "k_msm_sum_tree")},
                             "note": "an instruction-count figure: multiply-accumulate-heavy kernels (level 1: 4.43 cycles per instruction by the ISA census) weigh more than the 3.84-cycle average"}
        except Exception:
            valu_util = None
        line = {
            "metric": "Groth16 proofs/sec for WHIR-verifier circuit (2^20 poly); G1 MSM pts/sec",
            "value": proofs / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8 Montgomery (BN254 Fr/Fp, exact modular integers)", "data": "synthetic",
            "config": {"workload": f"full Groth16 prove of the WHIR-verifier-shaped synthetic key/witness, FFT domain N=2^{log_n} "
                                   f"(BASELINE {'configs[1]' if log_n == 23 else 'configs[2]' if log_n == 26 else 'non-baseline size'}; configs[3] = one such proof stream per GPU when n_gpus>1), "
                                   f"with its ONE BSB22 commitment over {n_committed} private wires (Commit inside the step, ProveKnowledge beside the proof's MSMs, fold; {196 if n_committed else 164}-byte proof), "
                                   "W, a, b handed over as HOST pointers (the cgo path; c = a o b formed on the device)",
                       "nb_wires": nb_wires, "nb_public": nb_public, "n_constraints": n_constraints, "n_committed": n_committed, "scalar_dist": args.dist,
                       "g1_msm_sizes": [na, nb, nk, N - 1], "g2_msm_size": nb, "pedersen_msm_sizes": [n_committed, n_committed], "proofs_in_flight_per_gpu": in_flight,
                       "caller_threads": callers, "inputs": "host memory (PCIe inside the step)"},
            "proof_bytes": len(serial_bytes), "rank0_numa_binding": numa,
            # the GPU-side rate: the same key and witness with W, a, b, c already in HBM and no commitment (164-byte body) -- what rounds 1-3
            # reported as `value`; no caller of the reference can reach it (gnark's solver is CPU code)
            "value_hbm_resident_inputs": dev_rate, "ms_per_step_hbm_resident_inputs": dev_ms,
            # latency of ONE proof with nothing else on the GPU -- the reference proves one circuit per run, so this is its own metric:
            # the caller's path (host W, a, b: Commit + submit -> wait through the pool); the same with c uploaded too; the GPU side alone
            # (inputs in HBM, plain mi_groth16_prove_dev, no commitment)
            "single_proof_latency_host_inputs_ms": lat_host, "single_proof_latency_host_inputs_with_c_uploaded_ms": lat_host_with_c,
            "single_proof_latency_ms": serial_ms, "pedersen_commit_latency_ms": lat_commit,
            # the upload stage's wall time per job (W, a, b = 0.8 GB at N = 2^23 from pageable host memory): when its median nears
            # ms_per_step the rate is bound by the PCIe / host-memory side of the box, not by the GPU
            "host_inputs_upload_ms": {"median": h2d[len(h2d) // 2], "max": h2d[-1]},
            # BASELINE configs[4] (one MSM point-sharded over the ranks, strong scaling); n_gpus = 1: the same code path with one rank
            "sharded_msm": sharded,
            # BASELINE configs[4] as north_star states it: ONE proof point-sharded over the ranks (strong scaling)
            "sharded_prove": sharded_prove,
            "multi_gpu_note": "no scaling curve exists until an 8-GPU node runs this command with --gpus 2/4/8; nothing here extrapolates one",
            "proofs_validated": f"{len(done)} timed proofs ({len(serial_bytes)} bytes each) + {0 if dev_rate is None else args.steps} HBM-resident-input proofs byte-equal to the untimed reference proofs",
            "pk_load_s": t_load,
            "hbm_in_use_gb": hbm_in_use_gb, "hbm_ledger_gb": hbm_ledger,
            # second half of BASELINE's metric: one G1 MSM of 2^23 uniform pairs alone on the GPU (standard MSM benchmark shape);
            # inside a proof the five MSMs overlap on five streams, so per-MSM spans there are not rates
            "g1_msm_pts_per_s": solo["msm_pts_per_s"], "g1_pairs_per_proof": g1_pairs_per_proof,
            "phase_ms": {k: last[k] for k in ("compute_h_ms", "msm_a_ms", "msm_b1_ms", "msm_b2_ms", "msm_k_ms", "msm_z_ms", "assemble_ms", "total_ms")},
            "step_completion_gaps": step_gaps,
            "clocks_under_load": clocks,
            "roofline": roofline,
            # second kernel: k_ntt_pass.  Algorithmic bytes 64 * N per size-N transform whatever the number of passes;
            # time = computeH alone on the GPU / its 6 transforms (gnark's 7th, the coset FFT of c, is never needed: DESIGN.md 4)
            "roofline_ntt": {"kernel": "k_ntt_pass_wave + k_ntt_contig_pair + k_ntt_strided_triple + k_ntt_contig_last_sub (all passes of one size-N transform)", "bound": "hbm",
                             "achieved": 64.0 * N / (ntt_solo["ms_per_transform"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": 64.0 * N / (ntt_solo["ms_per_transform"] * 1e-3) / 1e9 / 8000.0, "traffic": traffic_ntt,
                             "traffic_source": pmc_src + ": (2 x FETCH_SIZE + WRITE_SIZE) per pass launch x pass launches per transform",
                             "compute_h_solo_ms": ntt_solo["compute_h_ms"], "pass_launches_per_compute_h": ntt_solo["pass_launches"],
                             "algorithmic_bytes_per_transform": 64.0 * N},
            # why the HBM fraction is small: the kernel is bound by 256-bit modular products on the VALU (no MFMA form exists)
            "g1_msm_solo": solo, "g1_msm_z_shaped_solo": zsolo,
            # the headline's sensitivity to the witness distribution, and the floor next to the headline
            "sensitivity": sensitivity, "value_uniform_witness": None if not sensitivity else sensitivity["uniform"]["value"],
            # the level-1 accumulate gathers one 64-B point per mixed addition from tables of 7..16 GB: measured ceiling of the memory
            # system for that access pattern, the kernel's own gather rate alone on the GPU, and the job's aggregate rate
            "random_gather": {"ceiling_gathers_per_s": 256 * 4 * 64 * 4 * 128 / (gather_ms * 1e-3), "ceiling_GBps_useful": 256 * 4 * 64 * 4 * 128 * 64 / (gather_ms * 1e-3) / 1e9,
                              "kernel_alone_gathers_per_s": solo["mixed_adds_per_s"],
                              "frac_alone": solo["mixed_adds_per_s"] / (256 * 4 * 64 * 4 * 128 / (gather_ms * 1e-3)),
                              "job_g1_gathers_per_s": accum_entries / dt,
                              "note": "ceiling: dependent random 64-B reads from an 8.6 GB table, 4 waves per SIMD on every CU; G2 gathers (128 B) not counted"},
            "valu": {"modmul_ceiling_per_s": 256 * 4096 * 256 * 2 / (modmul_ms * 1e-3),
                     "kernel_mixed_adds_per_s": accum_entries / (accum_ms * 1e-3) if accum_ms > 0 else 0.0,
                     "kernel_modmul_per_s": 10.0 * accum_entries / (accum_ms * 1e-3) if accum_ms > 0 else 0.0,
                     "note": "one XYZZ mixed addition = 8M + 2S Fp products (+ ~7 add/sub); frac = kernel_modmul_per_s / modmul_ceiling_per_s",
                     # the kernel's OWN instruction mix allows a SIMD 2.4e9 / cycles_per_addition wave-additions/s: the kernel alone on the GPU against that floor
                     "issue_floor_adds_per_s": issue_floor, "issue_floor_source": census_src,
                     "kernel_alone_frac_of_issue_floor": None if issue_floor is None else solo["mixed_adds_per_s"] / issue_floor,
                     "z_shaped_launch_alone_frac_of_issue_floor": None if issue_floor is None or not zsolo or "mixed_adds_per_s" not in zsolo else zsolo["mixed_adds_per_s"] / issue_floor,
                     "job_utilisation": valu_util},
        }
        if cpu_inputs is not None:
            line["cpu_baseline"] = cpu_baseline(*cpu_inputs, rs[0], rs[1], (ped_basis, ped_sigma, values, rs[2]) if n_committed else None, log_n, serial_bytes)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
