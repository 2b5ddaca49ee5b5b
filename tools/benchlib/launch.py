"""Starting the ranks of `bench.py --gpus N` and pinning a rank to its GPU's NUMA node."""
import os
import sys

from .common import BENCH_PY


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
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
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


def require_gpus(n, rehearse):
    """`--gpus N` means N ranks on N DISTINCT GPUs.  Before any rank touches a device: the node must show at least N (counting devices does not
    initialise the GPU).  The one-GPU rehearsal of the multi-process code path says so explicitly (--rehearse-on-one-gpu)."""
    if n <= 1 or rehearse:
        return
    import torch
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} needs {n} distinct GPUs and this node shows {have}: distinct_devices != n_gpus.  "
                         "(For the multi-process code path on one GPU: --rehearse-on-one-gpu; its line says what it is.)")


def observe_devices(torch, dist, local_rank, world, rehearse):
    """What the run OBSERVED about its devices, for the line: every rank's PCI bus id, gathered over the ranks, and how many are distinct.
    A multi-GPU run whose ranks do not sit on `world` distinct devices is refused here unless it is the declared rehearsal."""
    pr = torch.cuda.get_device_properties(local_rank)
    bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    all_bdf = [bdf]
    if dist is not None and world > 1:
        all_bdf = [None] * world
        dist.all_gather_object(all_bdf, bdf)
    distinct = len(set(all_bdf))
    if world > 1 and distinct != world and not rehearse:
        raise SystemExit(f"bench.py: {world} ranks sit on {distinct} distinct device(s) {sorted(set(all_bdf))}: distinct_devices != n_gpus (one rank per GPU is the contract; "
                         "--rehearse-on-one-gpu for the declared one-GPU rehearsal)")
    return {"distinct_devices": distinct, "device_pci_by_rank": all_bdf,
            "torch_distributed": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size()},
            "how": "torch.cuda.get_device_properties(local_rank) PCI ids of every rank, all-gathered; the device group's own view (hipDeviceGetPCIBusId, ncclCommCount) is in sharded_prove.observed"}
