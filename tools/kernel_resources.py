"""Resource table of every kernel in the shipped library: VGPRs (arch + acc), SGPRs, static LDS, scratch, the workgroup size the source
asks for (launch bounds) -- from the gfx950 assembly hipcc emits for each translation unit (no GPU needed).

    python tools/kernel_resources.py [--json profiles/rNN_kernel_resources.json] [file.hip ...]

Why it matters here ("Streams"): kernels of several proofs share the CUs.  A workgroup is placed only where EVERY one of
its waves finds registers at once: waves_per_simd(workgroup) x vgprs must fit what the SIMDs of one CU have free.  The columns
`waves/SIMD of a WG` and `VGPR x waves/SIMD` are that footprint (512 VGPRs per SIMD lane-row on gfx950, allocation granule 8)."""
import json
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gnark-whir_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels_of(src):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "x.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    res = []
    # the metadata block at the end of the file is YAML; parse the few keys we need per kernel by hand
    md = text[text.find("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count:", md)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, d=0: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, d])[1]
        res.append({
            "symbol": g("name", "?"),
            "vgpr": int(g("vgpr_count")), "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")),
            "lds_static": int(g("group_segment_fixed_size")), "scratch": int(g("private_segment_fixed_size")),
            "max_wg": int(g("max_flat_workgroup_size")),
            "file": os.path.basename(src),
        })
    return res


def scratch_blocks(src, needle):
    """WHERE a kernel's scratch is touched: per basic block of the kernel whose mangled name contains `needle`, the instruction lines, the
    v_mad_u64_u32 among them (the hot path of the curve arithmetic is where the multiplies are) and the scratch_load / scratch_store
    instructions.  A private segment that only cold blocks touch costs its allocation, not time.
        python tools/kernel_resources.py --scratch-blocks k_msm_accum_xyzz29 gnark-whir_amd/csrc/msm_g1.hip"""
    import bisect
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "x.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm], stderr=subprocess.DEVNULL)
        L = open(asm).read().split("\n")
    starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and needle in l.split(":")[0]]
    for st in starts:
        sym = L[st].split(":")[0]
        end = next(i for i in range(st, len(L)) if L[i].strip().startswith(".Lfunc_end"))
        body = L[st:end]
        labels = [(i, l.split(":")[0]) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
        idx = [i for i, _ in labels]
        stat = {}
        for i, l in enumerate(body):
            if not l.startswith("\t") or l.lstrip().startswith((";", ".")):
                continue
            k = bisect.bisect_right(idx, i) - 1
            name = labels[k][1] if k >= 0 else "entry"
            s = stat.setdefault(name, [0, 0, 0])
            s[0] += 1
            s[1] += "v_mad_u64_u32" in l
            s[2] += "scratch_" in l
        tot = [sum(v[j] for v in stat.values()) for j in range(3)]
        print(f"{demangle([sym])[sym][:110]}")
        print(f"  {len(stat)} basic blocks, {tot[0]} instructions, {tot[1]} v_mad_u64_u32, {tot[2]} scratch instructions; blocks with >= 100 multiplies or any scratch access:")
        print(f"  {'block':12s} {'instructions':>12s} {'v_mad_u64_u32':>14s} {'scratch':>8s}")
        for name in ["entry"] + [n for _, n in labels]:
            if name in stat and (stat[name][1] >= 100 or stat[name][2]):
                print(f"  {name:12s} {stat[name][0]:12d} {stat[name][1]:14d} {stat[name][2]:8d}")


def main():
    args = sys.argv[1:]
    if args and args[0] == "--scratch-blocks":
        return scratch_blocks(args[2], args[1])
    out_json = None
    if args and args[0] == "--json":
        out_json = args[1]
        args = args[2:]
    srcs = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with ThreadPoolExecutor(4) as ex:
        rows = [r for rs in ex.map(kernels_of, srcs) for r in rs]
    dm = demangle([r["symbol"] for r in rows])
    for r in rows:
        r["name"] = re.sub(r"\(.*", "", dm.get(r["symbol"], r["symbol"]))
        tot = r["vgpr"] + r["agpr"] if r["agpr"] else r["vgpr"]
        gran = (tot + 7) // 8 * 8
        r["vgpr_alloc"] = gran
        r["wg_waves"] = (r["max_wg"] + 63) // 64
        r["waves_per_simd_of_wg"] = (r["wg_waves"] + 3) // 4
        r["footprint_vgpr_per_simd"] = gran * r["waves_per_simd_of_wg"]
    rows.sort(key=lambda r: (r["file"], r["name"]))
    print(f"{'kernel':58s} {'file':16s} {'vgpr':>5s} {'agpr':>4s} {'sgpr':>4s} {'LDS':>7s} {'scr':>5s} {'maxWG':>5s} {'w/SIMD':>6s} {'VGPRxw':>6s}")
    for r in rows:
        print(f"{r['name'][:58]:58s} {r['file']:16s} {r['vgpr']:5d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds_static']:7d} {r['scratch']:5d} {r['max_wg']:5d} "
              f"{r['waves_per_simd_of_wg']:6d} {r['footprint_vgpr_per_simd']:6d}")
    if out_json:
        json.dump(rows, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
