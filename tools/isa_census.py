"""ISA census of the dominant kernel, k_msm_accum_affine29 (G1 level-1 bucket accumulate, csrc/msm_g1.hip): vector instructions per
mixed addition by class, from the gfx950 assembly hipcc emits for the shipped source, priced with the measured cycles per wave64
instruction of profiles/r02_probe_instr_rate.txt.  bench.py reads the result (profiles/r06_isa_census_accum_affine29.json) for the
kernel's instruction-issue floor (`valu.issue_floor_adds_per_s`) instead of a literal.

    python tools/isa_census.py            (no GPU needed: hipcc cross-compiles; ~15 s)

Method.  The kernel's main loop is the backward-branch loop with the most v_mad_u64_u32.  One iteration = one mixed addition per lane.
Inside it the compiler keeps three kinds of code:
  hot       executed by every iteration (unpack of the gathered point, sign, the 8M + 2S XYZZ mixed addition in nine 29-bit limbs,
            weak normalisations, loop control);
  fallback  a region skipped by s_cbranch_execz unless some lane of the wave meets P == 0 mod p (doubling / cancellation): the same
            addition in standard arithmetic, >= 1500 multiply-accumulates -- weight 0 (it needs two EQUAL points in one bucket);
  epilogue  a region skipped by s_cbranch_execz unless a lane has finished its item: the item's sum brought back to the standard form,
            exactly four 162-multiply products -- weight 1 / (entries per item) (an item holds <= 16 entries: 1/16 is the lower bound
            of its cost per addition, i.e. the floor stays a floor).
The regions are recognised by those signatures (multiply-accumulate counts), not by line numbers; if the compiler's output stops
matching them the script fails instead of writing a census of something else.  Every region is listed in the output with its counts
and the weight applied, so the cycle figure can be recomputed by hand."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gnark-whir_amd", "csrc", "msm_g1.hip")
KERNEL = "k_msm_accum_affine29"
OUT = os.path.join(ROOT, "profiles", "r06_isa_census_accum_affine29.json")
INSTANCE = "ILi4ELi3EE"   # <WG = 4 waves per workgroup, WPS = 3 waves per SIMD>: the default build (csrc/msm_g1.hip)


def sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def cycle_table():
    """cycles per wave64 instruction, measured (tools/bench_valu/instr_rate.hip, 8 waves per SIMD, independent chains)"""
    t = {}
    for ln in open(os.path.join(ROOT, "profiles", "r02_probe_instr_rate.txt")):
        m = re.match(r"^(\S+(?: \([vs],[vs]\))?(?: \+ \S+)?)\s+[\d.]+ ms.*= ([\d.]+) cycles", ln)
        if m:
            t[m.group(1)] = float(m.group(2))
    return t


def price(op, operands, tab):
    """cycles of one instruction: the measured class where there is one, else the nearest measured class (stated in the output)"""
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)   # encoding suffixes of the assembler's spelling
    if op == "v_mad_u64_u32":
        srcs = operands.split(",")[2:]
        return "v_mad_u64_u32 (v,s)" if any(s.strip().startswith("s") for s in srcs) else "v_mad_u64_u32 (v,v)"
    direct = {"v_lshrrev_b64": "v_lshrrev_b64", "v_lshl_add_u64": "v_lshl_add_u64", "v_mul_lo_u32": "v_mul_lo_u32", "v_mul_hi_u32": "v_mul_hi_u32",
              "v_add_u32": "v_add_u32", "v_add3_u32": "v_add3_u32", "v_alignbit_b32": "v_alignbit_b32", "v_and_b32": "v_and_b32",
              "v_lshl_add_u32": "v_lshl_add_u32", "v_mad_u32_u24": "v_mad_u32_u24", "v_mul_u32_u24": "v_mul_u32_u24"}
    if op in direct:
        return direct[op]
    if op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_subbrev_co_u32"):
        return "v_add_co + v_addc_co"
    if op in ("v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_cndmask_b32",
              "v_bfe_u32", "v_readfirstlane_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32") or op.startswith("v_cmp"):
        return "v_add_u32"          # plain 32-bit ALU class (2.3 - 2.5 cycles measured for add / and)
    return "v_add3_u32"             # anything else: priced as a 3-operand / 64-bit integer instruction (4.1 - 4.7 cycles measured)


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "msm_g1.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", SRC, "-o", asm],
                              stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\d+" + KERNEL + INSTANCE + r"v", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            a = labels[m.group(1)]
            loops.append((sum("v_mad_u64_u32" in x for x in body[a:i + 1]), a, i))
    _, la, lb = max(loops)
    # loop header: some loops are entered in the middle (rotated): the census takes the whole cyclic body la..lb
    tab = cycle_table()
    regions, hot = [], {}
    i = la

    def count(a, b):
        d = {}
        for x in body[a:b]:
            m = re.match(r"^\s+(v_[a-z_0-9]+)\s*(.*)", x)
            if m:
                cls = price(m.group(1), m.group(2), tab)
                d.setdefault(cls, {"n": 0, "ops": {}})
                d[cls]["n"] += 1
                d[cls]["ops"][m.group(1)] = d[cls]["ops"].get(m.group(1), 0) + 1
        return d

    def summ(d):
        return {"valu": sum(v["n"] for v in d.values()), "mad_u64_u32": sum(v["n"] for k, v in d.items() if k.startswith("v_mad_u64_u32")),
                "cycles": sum(v["n"] * tab[k] for k, v in d.items())}

    seg_start = la
    found = {"fallback": 0, "epilogue": 0}
    while i <= lb:
        m = re.match(r"^\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", body[i])
        if m and i < labels.get(m.group(1), -1) <= lb:
            tgt = labels[m.group(1)]
            d = count(i + 1, tgt)
            s = summ(d)
            kind = None
            if s["mad_u64_u32"] == 4 * 162:
                kind = "epilogue"
            elif s["mad_u64_u32"] >= 1500 and lb - tgt > 64:   # (guards that reach the loop's end are the "lane has an entry" masks around the whole body: walked into)
                kind = "fallback"
            if kind:
                for k, v in count(seg_start, i + 1).items():
                    hot.setdefault(k, {"n": 0, "ops": {}})
                    hot[k]["n"] += v["n"]
                    for o, c in v["ops"].items():
                        hot[k]["ops"][o] = hot[k]["ops"].get(o, 0) + c
                regions.append({"kind": kind, "guard": body[i].strip(), "asm_lines": tgt - i, **s,
                                "weight": 0.0 if kind == "fallback" else 1.0 / 16, "by_class": {k: v["n"] for k, v in d.items()}})
                found[kind] += 1
                i = tgt
                seg_start = tgt
                continue
        i += 1
    for k, v in count(seg_start, lb + 1).items():
        hot.setdefault(k, {"n": 0, "ops": {}})
        hot[k]["n"] += v["n"]
        for o, c in v["ops"].items():
            hot[k]["ops"][o] = hot[k]["ops"].get(o, 0) + c
    if found != {"fallback": 1, "epilogue": 1}:
        sys.exit(f"isa_census: the loop of {KERNEL} no longer has exactly one fallback and one epilogue region ({found}): look at the assembly and update the signatures")
    hs = summ(hot)
    epi = next(r for r in regions if r["kind"] == "epilogue")
    cycles = hs["cycles"] + epi["cycles"] * epi["weight"]
    out = {
        "kernel": KERNEL, "source": "gnark-whir_amd/csrc/msm_g1.hip", "source_sha256_16": sha16(SRC),
        "includes_sha256_16": {f: sha16(os.path.join(ROOT, "gnark-whir_amd", "csrc", f)) for f in ("curve29.cuh", "field29.cuh", "msm_core.cuh", "field.cuh")},
        "compiler": subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0],
        "method": "static count over `hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S`: main loop = backward-branch loop with the most v_mad_u64_u32; regions skipped by "
                  "s_cbranch_execz and recognised by their multiply counts are weighted (fallback 0, once-per-item epilogue 1/16); every other instruction of the loop counts once per addition",
        "cycles_per_instruction_source": "profiles/r02_probe_instr_rate.txt (tools/bench_valu/instr_rate.hip)",
        "cycles_per_instruction": tab,
        "hot_path": {"by_class": {k: {"n": v["n"], "cycles_each": tab[k], "opcodes": v["ops"]} for k, v in sorted(hot.items())}, **hs},
        "guarded_regions": regions,
        "valu_per_addition": hs["valu"], "mad_u64_u32_per_addition": hs["mad_u64_u32"],
        "cycles_per_addition": cycles,
        "issue_floor_adds_per_s": 2.4e9 / cycles * 64 * 1024,
        "issue_floor_formula": "2.4e9 Hz / cycles_per_addition x 64 lanes x 1024 SIMDs (256 CUs x 4)",
    }
    json.dump(out, open(OUT, "w"), indent=1)
    print(f"{OUT}: {hs['valu']} VALU / addition ({hs['mad_u64_u32']} v_mad_u64_u32), {cycles:.0f} cycles -> floor {out['issue_floor_adds_per_s'] / 1e9:.2f} G additions/s; "
          f"fallback region {regions[0]['valu'] if regions[0]['kind'] == 'fallback' else regions[1]['valu']} VALU (weight 0), epilogue {epi['valu']} VALU (weight 1/16)")


if __name__ == "__main__":
    main()
