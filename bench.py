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
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.common import _binding, _sha16, dist_id_of   # noqa: E402  (the tests and tools reach these through this module)
from benchlib.pmc import _kernel_short, live_pmc   # noqa: E402,F401
from benchlib.clocks import ClockSampler   # noqa: E402
from benchlib.sharded import sharded_helper_main, run_sharded_legs   # noqa: E402
from benchlib.launch import self_launch, bind_to_gpu_numa_node, require_gpus, observe_devices   # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)   # ~1 s of proofs: one rare runtime hiccup then moves the result by < 2 %
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-n", type=int, default=23, help="FFT domain (2^23 = BASELINE configs[1])")
    ap.add_argument("--dist", type=lambda v: v if v in ("whir", "half", "uniform", "census") or v.startswith("mix:") else ap.error(f"--dist {v}"), default="whir",
                    help="witness (W, a) distribution: the WHIR mix of SURVEY 8d / BASELINE.md 3 (a documented guess), uniform Fr, half of each (row by row), or "
                         "`census`: the midpoint mix tools/wire_census.py derives from the reference's circuit (profiles/r06_wire_census.txt), or mix:BIT,BYTE,U64 = per-mille "
                         "shares of one's own census (tools/wire_census.py --params FILE prints them); the rest full-width")
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
    if not args.launch_check:
        require_gpus(args.gpus, args.rehearse_on_one_gpu)   # N ranks on N distinct GPUs, or the declared rehearsal: said before any rank starts
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
    devices_observed = observe_devices(torch, dist, local_rank, world, args.rehearse_on_one_gpu)
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
        did = dist_id_of(B, "whir" if dist == "half" else dist)
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
                               "uniform mix, `half_uniform` = every row uniform with probability 1/2, `uniform` = every row uniform (the floor), `census` = the mix the reference's circuit implies",
                       "half_uniform": sensitivity_leg("half"), "uniform": sensitivity_leg("uniform"), "census": sensitivity_leg("census")}
        import wire_census   # (tools/ is on the path: pure Python, restated counting rules, no reference file is read)
        pm = wire_census.census_mix_permille()
        sensitivity["census"]["mix"] = (f"{pm[0] / 10:.1f} % {{0,1}} / {pm[1] / 10:.1f} % bytes / {pm[2] / 10:.1f} % 64-bit / {(1000 - sum(pm)) / 10:.1f} % full-width: the midpoint of the range "
                                        "tools/wire_census.py derives from the reference's circuit for configs[1] (profiles/r06_wire_census.txt: eq tables and matrix MLE, "
                                        "mtUtilities.go:494-532, are all full-width; Merkle / STIR terms are bytes and full-width in comparable numbers; bits 1-3 %)")
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
                # the shader clock this launch sustains (the floors below are cycles): ~0.5 s of the same MSM back to back under the sampler
                zclk = None
                if not args.no_clock_samples:
                    smp = ClockSampler(local_rank).start()
                    t_end = time.perf_counter() + 0.7
                    while time.perf_counter() < t_end:
                        ctx.msm_fixed_dev(tab.ptr, hsc.ptr, n_z, cz, flags=2)
                    zclk = smp.stop()
                tab.free(); hsc.free()
                zsolo = {"sclk_mhz_under_this_msm": None if not zclk else zclk["sclk_mhz_mean"], "pairs": n_z, "scalars": "uniform", "window_bits": cz, "windows": nwin_z, "msm_total_ms": best["total_ms"], "accum_launch_ms": best["g1_accum_kernel_ms"],
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
    # ... and the TRUE cycles of a v_mad_u64_u32 (the instruction 71 % of the dominant kernel's cycles go to): eight waves per SIMD of
    # independent multiply-accumulate chains for ~0.7 s under the clock sampler.  profiles/r02_probe_instr_rate.txt priced it at 5.28 cycles
    # ASSUMING 2.4 GHz; the chip clocks lower under such a load, so every cycle figure derived from that table is scaled by this ratio
    mad_probe = None
    if rank == 0 and not args.no_clock_samples and not args.no_solo_legs:
        thr, it = 256 * 4 * 8 * 64, 20000
        ck, rates = None, []
        try:   # (a probe: whatever goes wrong here must cost the line its `valu_frac`, not the line)
            ctx.bench_valu(0, thr, 256)
            smp = ClockSampler(local_rank).start()
            t_end = time.perf_counter() + 0.7
            while time.perf_counter() < t_end:
                rates.append((thr / 64) * 8 * it / (ctx.bench_valu(0, thr, it) * 1e-3) / 1024.0)
            ck = smp.stop()
        except B.MiError:
            ck = None
        if ck and rates:
            rate = max(rates)
            mad_probe = {"wave_instr_per_s_per_simd": rate, "sclk_mhz": ck["sclk_mhz_mean"], "cycles_per_mad": ck["sclk_mhz_mean"] * 1e6 / rate,
                         "cycles_per_mad_if_2400_mhz": 2.4e9 / rate, "table_value_at_assumed_2400_mhz": 5.28,
                         "how": "mi_bench_valu_dev kind 0 (eight independent v_mad_u64_u32 chains per lane, 8 waves per SIMD) back to back for 0.7 s; rocm-smi shader clock meanwhile"}
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
        # committed passes of profiles/r06_pmc_bench_traffic.json (sha256 of the sources recorded in it): a file older than the kernels
        # reads as traffic: null, never as stale bytes.  The accumulate kernel gathers 64-B points, so its FETCH_SIZE is taken raw; the
        # NTT passes stream 16 B per lane, so theirs gets the guide's x2 correction.
        traffic = traffic_ntt = traffic_solo = None
        kname = "k_msm_accum_affine29"
        pmc_file = os.path.join("profiles", "r06_pmc_bench_traffic.json")
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
        # profiles/r06_isa_census_accum_affine29.json: instructions per loop iteration by class x the measured cycles per wave64
        # instruction of profiles/r02_probe_instr_rate.txt)
        issue_floor = census_src = None
        try:
            cen = json.load(open(os.path.join(ROOT, "profiles", "r06_isa_census_accum_affine29.json")))
            issue_floor = 2.4e9 / cen["cycles_per_addition"] * 64 * 1024
            census_src = f"profiles/r06_isa_census_accum_affine29.json: {cen['valu_per_addition']} vector instructions per mixed addition ({cen['mad_u64_u32_per_addition']} v_mad_u64_u32) = {cen['cycles_per_addition']:.0f} cycles per wave-addition"
        except Exception:
            pass
        # roofline of the dominant kernel: achieved = the algorithmic 96 B per pair of ONE launch / that launch's duration.
        # Basis: the Z-shaped launch alone on the GPU (zsolo above) when it ran; the job's average in-job launch otherwise (and always as `in_job`).
        in_job = {"launch_ms": per_launch_ms, "algorithmic_bytes_per_launch": per_launch_bytes, "achieved": achieved, "frac": achieved / 8000.0,
                  "note": "average over the proofs' four G1 level-1 launches while three proofs share the GPU (HIP events on the launch's stream): what rounds 1-4 reported as "
                          "`achieved`; it falls when the launch shares the CUs more evenly with the other streams, i.e. when the job gets FASTER"}
        bound_note = ("the kernel is bound by vector-ALU instruction issue (exact 254-bit modular products: no MFMA form exists), not by HBM: `achieved` / `peak` / `frac` are the "
                      "HBM roofline BASELINE.json's north_star asks to report; `valu_frac` is the roofline that binds")
        if zsolo and "accum_launch_ms" in zsolo:
            zmhz = zsolo.get("sclk_mhz_under_this_msm") or (clocks or {}).get("sclk_mhz_mean")
            # the census's cycles per wave-addition come from a table that assumed 2.4 GHz: true cycles = table cycles x (true / table cycles of the mad)
            # (the table's rate for this instruction: 4.542e8 wave-instructions/s/SIMD; today's probe may sustain another rate and runs at a KNOWN clock.
            #  Floor under this MSM = census floor x today's rate / table rate x this MSM's clock / the probe's clock)
            cyc_scale = None if not mad_probe else (mad_probe["wave_instr_per_s_per_simd"] / 4.542e8) / (mad_probe["sclk_mhz"] / 2400.0)
            floor_meas = None if not (issue_floor and zmhz and cyc_scale) else issue_floor * (zmhz / 2400.0) * cyc_scale
            roofline = {"kernel": "k_msm_accum_affine29 (G1 level-1 bucket accumulate, 9 x 29-bit limbs)", "bound": "valu-issue", "bound_note": bound_note,
                        "valu_frac": None if not floor_meas else zsolo["mixed_adds_per_s"] / floor_meas,
                        "valu_frac_at_2400_mhz": None if not issue_floor else zsolo["mixed_adds_per_s"] / issue_floor,
                        "valu_frac_basis": None if not floor_meas else (f"{zsolo['mixed_adds_per_s'] / 1e9:.2f} G mixed additions/s / ({zmhz:.0f} MHz measured under this MSM x 65536 lanes / TRUE cycles per wave-addition: "
                                                                        f"the ISA census's {cen['cycles_per_addition']:.0f} (priced at an assumed 2.4 GHz) / {cyc_scale:.3f}: today's multiply-accumulate probe sustained {mad_probe['wave_instr_per_s_per_simd'] / 1e8:.3f}e8 "
                                                                        f"wave-instructions/s/SIMD at {mad_probe['sclk_mhz']:.0f} MHz = {mad_probe['cycles_per_mad']:.2f} true cycles against the table's 5.28)"),
                        "mad_probe": mad_probe,
                        "traffic_ratio": None if not traffic_solo else traffic_solo / (96.0 * zsolo["pairs"]),
                        "basis": f"solo launch: the proof's largest level-1 launch (Z MSM: {zsolo['pairs']} uniform scalars, {zsolo['windows']} windows of {zsolo['window_bits']} bits, fixed-base tables) alone on the GPU",
                        "achieved": zsolo["accum_GBps_algorithmic"], "peak": 8000.0, "unit": "GB/s", "frac": zsolo["accum_GBps_algorithmic"] / 8000.0,
                        "traffic": traffic_solo, "traffic_source": pmc_src_solo, "launch_ms": zsolo["accum_launch_ms"], "algorithmic_bytes_per_launch": 96.0 * zsolo["pairs"],
                        "in_job": dict(in_job, traffic=traffic, traffic_source=pmc_src)}
        else:
            roofline = {"kernel": "k_msm_accum_affine29 (G1 level-1 bucket accumulate, 9 x 29-bit limbs)", "bound": "valu-issue", "bound_note": bound_note, "valu_frac": None,
                        "traffic_ratio": None if not (traffic and per_launch_bytes) else traffic / per_launch_bytes, "basis": "in-job average launch (the solo Z-shaped launch did not run)",
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
                for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_pmc_valu_proofs.csv"))):
                    k = _kernel_short(r["Kernel"])
                    if r["Counter"] == "SQ_INSTS_VALU" and not k.startswith(setup):
                        per_kernel[k] = per_kernel.get(k, 0.0) + float(r["Sum"]) / 4.0
                valu_src = "profiles/r06_pmc_valu_proofs.csv (committed pass, not this run)"
                if not (log_n == 23 and args.dist == "whir" and traffic is not None):
                    per_kernel = {}
            per_proof = sum(per_kernel.values())
            if per_proof > 0:
                share = lambda *names: sum(v for k, v in per_kernel.items() if k.startswith(names)) / per_proof
                valu_util = {"wave_instructions_per_proof": per_proof, "sustained_wave_instructions_per_s": 6.4e11, "source": valu_src,
                             "on_the_callers_path": per_proof / (dt / args.steps) / 6.4e11,
                             "hbm_resident_inputs": None if dev_ms is None else per_proof / (dev_ms * 1e-3) / 6.4e11,
                             # the same two at the shader clock the chip sustained during the HBM-resident region (1024 SIMDs x sclk / 3.84 cycles)
                             # The 3.84 cycles come from a rate table that ASSUMED 2.4 GHz (profiles/r02_probe_instr_rate.txt) while such loads run at
                             # ~2.2 GHz: the table's cycles are too many by true / table cycles of its own multiply-accumulate row, which this run
                             # measures at a KNOWN clock (mad_probe).  Round 5's `at_measured_sclk` scaled the 6.4e11 by the job's clock without that
                             # correction -- counting the clock twice, 8-10 % too flattering; it stays as `..._r5_definition` for continuity.
                             "at_measured_sclk": None if not (clocks and mad_probe) else (lambda rate: {
                                 "sclk_mhz": clocks["sclk_mhz_mean"], "true_over_table_cycles": mad_probe["cycles_per_mad"] / 5.28, "sustained_wave_instructions_per_s": rate,
                                 "on_the_callers_path": per_proof / (dt / args.steps) / rate,
                                 "hbm_resident_inputs": None if dev_ms is None else per_proof / (dev_ms * 1e-3) / rate,
                                 "how": "1024 SIMDs x the job's shader clock / (3.84 table cycles x true / table cycles of v_mad_u64_u32 measured by this run's probe at a known clock)"})(
                                     1024 * clocks["sclk_mhz_mean"] * 1e6 / (3.84 * mad_probe["cycles_per_mad"] / 5.28)),
                             "at_measured_sclk_r5_definition": None if not clocks else {"sclk_mhz": clocks["sclk_mhz_mean"], "on_the_callers_path": per_proof / (dt / args.steps) / (1024 * clocks["sclk_mhz_mean"] * 1e6 / 3.84),
                                                                          "hbm_resident_inputs": None if dev_ms is None else per_proof / (dev_ms * 1e-3) / (1024 * clocks["sclk_mhz_mean"] * 1e6 / 3.84),
                                                                          "note": "counts the clock twice (see above): not comparable with a utilisation"},
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
            # observed, not derived from the arguments: the ranks' PCI bus ids and how many are distinct (a run with fewer distinct devices than
            # ranks is refused unless it is the declared one-GPU rehearsal); the device group's own view is in sharded_prove.observed
            "devices_observed": dict(devices_observed, rehearsal_on_one_gpu=bool(args.rehearse_on_one_gpu)),
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
            "roofline_ntt": {"kernel": "k_ntt_pass_wave + k_ntt_contig_pair + k_ntt_strided_triple + k_ntt_contig_last_sub (all passes of one size-N transform)", "bound": "valu-issue",
                             "bound_note": "77 modular products per element per computeH on the vector ALU; the HBM figures are what north_star asks for.  `achieved` counts the SIX transforms this "
                                           "library runs (64 N bytes each); `achieved_survey_8d_accounting` prices the same computeH at SURVEY 8d's 7 x 64 N = 448 N bytes",
                             "achieved_survey_8d_accounting": 448.0 * N / (ntt_solo["compute_h_ms"] * 1e-3) / 1e9,
                             "frac_survey_8d_accounting": 448.0 * N / (ntt_solo["compute_h_ms"] * 1e-3) / 1e9 / 8000.0,
                             "traffic_ratio": None if not traffic_ntt else traffic_ntt / (64.0 * N),
                             "achieved": 64.0 * N / (ntt_solo["ms_per_transform"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": 64.0 * N / (ntt_solo["ms_per_transform"] * 1e-3) / 1e9 / 8000.0, "traffic": traffic_ntt,
                             "traffic_source": pmc_src + ": (2 x FETCH_SIZE + WRITE_SIZE) per pass launch x pass launches per transform",
                             "compute_h_solo_ms": ntt_solo["compute_h_ms"], "pass_launches_per_compute_h": ntt_solo["pass_launches"],
                             "algorithmic_bytes_per_transform": 64.0 * N},
            # why the HBM fraction is small: the kernel is bound by 256-bit modular products on the VALU (no MFMA form exists)
            "g1_msm_solo": solo, "g1_msm_z_shaped_solo": zsolo,
            # the headline's sensitivity to the witness distribution, and the floor next to the headline
            "sensitivity": sensitivity, "value_uniform_witness": None if not sensitivity else sensitivity["uniform"]["value"],
            # the same step on the witness mix the reference's circuit implies (tools/wire_census.py): the number to hold beside `value`
            "value_census_mix": None if not sensitivity else sensitivity["census"]["value"],
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
