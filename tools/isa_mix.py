#!/usr/bin/env python3
"""Instruction mix of a gfx950 kernel, per loop, from the compiler's assembly -- the per-class counts bench.py's issue roofline is
weighted with (a wave64 32-bit VALU instruction issues in 2 cycles, an f64 one in 4; tools/microbench/issue_rate.hip
measures the classes on the box, profiles/r03_issue_rate.txt).

    python tools/isa_mix.py minsum_regular.hip 'minsum_regular_kernel<6, 3, false, true, true, true>' [--loop N] [--json]

Compiles csrc/<file> to assembly for gfx950 with the product flags (hipcc --cuda-device-only -S, a few seconds, no GPU needed), finds
the kernel by its demangled name (substring match), splits the body into natural loops (a label that a later branch jumps back to) and
prints, for the whole kernel and for every loop, the static count of instructions per issue class.  The steady-state loop of a
persistent kernel is the innermost loop with the most VALU instructions; --loop picks another one.  Static counts weight every
basic block of the loop once: for the branch-free decoder loops that is the dynamic mix, for the OSD kernels it is a lower bound on
the share of the rare paths -- the PMC instruction totals (SQ_INSTS_VALU / SALU / LDS) stay the measured side of the roofline.
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qldpc-branched-off_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]

# ---- issue cost of an instruction: measured table first, rules for the mnemonics the microbenchmark does not cover ----
RATE_FILE = os.path.join(ROOT, "profiles", "r03_issue_rate.txt")
_MEASURED = None


def measured_cycles():
    """{mnemonic: cycles per wave64 instruction per SIMD at >= 2 waves per SIMD}, from the microbenchmark's output on the GPU box
    (profiles/r03_issue_rate.txt, column '4 waves per SIMD', wall-clock figure), rounded to the issue granularity of 2 cycles."""
    global _MEASURED
    if _MEASURED is None:
        _MEASURED = {}
        try:
            with open(RATE_FILE) as fh:
                for ln in fh:
                    mt = re.match(r"^(\w+)\s+([\d.]+) \[\s*[\d.]+\]\s+([\d.]+) \[\s*[\d.]+\]\s+([\d.]+) \[", ln)
                    if mt and not mt.group(1).startswith(("ds_", "mix_", "cmp_cndmask", "cmpf64")):
                        name = mt.group(1)
                        c = float(mt.group(4))
                        cyc = max(2, int(round(c / 2.0)) * 2)
                        _MEASURED[{"cndmask_b32": "v_cndmask_b32", "cndmask_vcc": "v_cndmask_b32_vcc", "readlane": "v_readlane_b32", "mov_dpp": "v_mov_b32_dpp"}.get(name, "v_" + name)] = cyc
        except OSError:
            pass
    return _MEASURED


def base_op(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def classify(op):
    """issue class of one instruction mnemonic"""
    if op.startswith(("s_", )):
        if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memrealtime", "s_atc")):
            return "smem"
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt", "s_trap", "s_code_end")):
            return "ctl"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_call", "s_swappc")):
            return "branch"
        return "salu"
    if op.startswith(("ds_", )):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return f"valu_{valu_cycles(op)}"
    return "other"


def valu_cycles(op):
    """cycles one wave64 VALU instruction holds a SIMD's issue port (two or more waves on the SIMD).  Measured classes on MI355X
    (profiles/r03_issue_rate.txt): 2 = v_add/sub_u32, and/or/xor/not, v_mov_b32, v_lshrrev / v_ashrrev, v_add / v_mul_f32;
    4 = every f64 / 64-bit instruction, every compare, v_cndmask, every three-operand (VOP3-only) integer op (v_bfi, v_bfe, v_and_or, v_or3,
    v_add3, v_lshl_add, v_perm, v_alignbit, v_xad), v_lshlrev_b32, v_min / v_max, v_mul_lo / hi, v_mul_u32_u24, conversions, lane moves
    (v_readlane, DPP); 8 = transcendentals; v_fma_f32 ~ 2.5."""
    b = base_op(op)
    tab = measured_cycles()
    if op.endswith("_dpp"):
        return tab.get("v_mov_b32_dpp", 4)
    if b in tab:
        return tab[b]
    if b.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane")):
        return tab.get("v_readlane_b32", 4)
    if b.startswith(("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")):
        return 8
    if re.search(r"_(f64|i64|u64|b64)$", b) or b.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_lshl_add_u64")):
        return tab.get("v_mad_u64_u32", 4) if b.startswith("v_mad_") else 4
    if b.startswith(("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_mul_i32", "v_cvt", "v_bfi", "v_bfe", "v_and_or", "v_or3", "v_xor3",
                     "v_add3", "v_lshl_add", "v_add_lshl", "v_lshl_or", "v_perm", "v_alignbit", "v_alignbyte", "v_xad", "v_bitop3", "v_mad_", "v_sad", "v_lshlrev_b32",
                     "v_bcnt", "v_mbcnt", "v_ffbl", "v_ffbh", "v_bfrev", "v_fma_f64", "v_ldexp", "v_frexp", "v_fract", "v_trunc", "v_floor", "v_ceil", "v_rndne")):
        return 4
    if b.startswith("v_fma_f32") or b.startswith("v_fmac_f32"):
        return 2          # 2.5 measured at 4 waves per SIMD: counted at the faster class
    return 2


CLASS_CYCLES = {"valu_2": 2, "valu_4": 4, "valu_8": 8, "valu_16": 16}


def compile_to_asm(src):
    out = "/tmp/isa_mix_" + os.path.basename(src) + ".s"
    cmd = [HIPCC] + FLAGS + [src, "-o", out]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("hipcc failed:\n" + r.stderr[-2000:])
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return r.stdout.split("\n")
    except OSError:
        return names


def kernels_of(asm_path):
    """{mangled: (body lines, metadata dict)}"""
    with open(asm_path) as fh:
        lines = fh.read().split("\n")
    out, cur, body = {}, None, []
    for ln in lines:
        mt = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if mt and cur is None:
            cur, body = mt.group(1), []
            continue
        if cur is not None:
            body.append(ln)
            if re.match(r"^\s*s_endpgm", ln):
                out[cur] = body
                cur = None
    meta = {}
    text = "\n".join(lines)
    for blk in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", text, re.S):
        b = blk.group(0)
        nm = re.search(r"\.name:\s*(\S+)", b)
        if nm:
            meta[nm.group(1)] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):\s*(\d+)", b)}
    return out, meta


def parse(body):
    """-> (instructions [(index, op, text)], labels {name: instruction index}, branches [(instruction index, target label)])"""
    ins, labels, branches = [], {}, []
    for ln in body:
        s = ln.split(";")[0].strip()
        if not s or s.startswith("."):
            mt = re.match(r"^(\.LBB\w+):", s)
            if mt:
                labels[mt.group(1)] = len(ins)
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            branches.append((len(ins), tgt))
        ins.append((len(ins), op, s))
    return ins, labels, branches


def cfg_loops(ins, labels, branches):
    """Natural loops of the kernel's control-flow graph: [(header block, sorted instruction indices)], via dominators and back edges.
    Basic blocks start at labels and behind branches; a conditional branch falls through, s_branch / s_endpgm / s_setpc do not."""
    n = len(ins)
    if n == 0:
        return []
    br = dict(branches)
    starts = {0} | set(labels.values()) | {i + 1 for i in br if i + 1 < n}
    starts = sorted(x for x in starts if x < n)
    blk_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = (starts[bi + 1] if bi + 1 < len(starts) else n) - 1
        blocks.append((st, en))
        for i in range(st, en + 1):
            blk_of[i] = bi
    succ = [[] for _ in blocks]
    for bi, (st, en) in enumerate(blocks):
        op = ins[en][1]
        if en in br and br[en] in labels and labels[br[en]] < n:
            succ[bi].append(blk_of[labels[br[en]]])
        if not op.startswith(("s_branch", "s_endpgm", "s_setpc")) and en + 1 < n:
            succ[bi].append(blk_of[en + 1])
    nb = len(blocks)
    pred = [[] for _ in blocks]
    for u in range(nb):
        for v in succ[u]:
            pred[v].append(u)
    # dominators (iterative bit sets)
    full = (1 << nb) - 1
    dom = [full] * nb
    dom[0] = 1
    changed = True
    while changed:
        changed = False
        for v in range(1, nb):
            d = full
            for u in pred[v]:
                d &= dom[u]
            d |= 1 << v
            if d != dom[v]:
                dom[v] = d
                changed = True
    loops = {}
    for u in range(nb):
        for v in succ[u]:
            if (dom[u] >> v) & 1:                      # back edge u -> v
                body = loops.setdefault(v, {v})
                stack = [u]
                while stack:
                    x = stack.pop()
                    if x in body:
                        continue
                    body.add(x)
                    stack.extend(pred[x])
    out = []
    for h, body in sorted(loops.items()):
        idx = sorted(i for bi in body for i in range(blocks[bi][0], blocks[bi][1] + 1))
        out.append((h, body, idx))
    return out


def mix_idx(ins, idx):
    cnt = {}
    for i in idx:
        c = classify(ins[i][1])
        cnt[c] = cnt.get(c, 0) + 1
    return cnt


def mix(ins, s, e):
    cnt = {}
    for i, op, _ in ins[s:e + 1]:
        c = classify(op)
        cnt[c] = cnt.get(c, 0) + 1
    return cnt


def weighted(cnt):
    return sum(int(c.split("_")[1]) * k for c, k in cnt.items() if c.startswith("valu_"))


def describe(cnt):
    valu = sum(k for c, k in cnt.items() if c.startswith("valu"))
    parts = [f"{c} {k}" for c, k in sorted(cnt.items())]
    return f"VALU {valu} (issue cycles {weighted(cnt)}), " + ", ".join(parts)


def analyse(src, pattern, loop_pick=None):
    asm = compile_to_asm(src)
    kernels, meta = kernels_of(asm)
    names = list(kernels)
    dem = demangle(names)
    hits = [(n, d) for n, d in zip(names, dem) if pattern in d or pattern in n]
    if len(hits) != 1:
        sys.exit(f"pattern {pattern!r} matches {len(hits)} kernels:\n  " + "\n  ".join(d for _, d in (hits or zip(names, dem))))
    name, dname = hits[0]
    ins, labels, branches = parse(kernels[name])
    loops = cfg_loops(ins, labels, branches)
    res = {"kernel": dname, "symbol": name, "meta": meta.get(name, {}), "whole": mix(ins, 0, len(ins) - 1), "loops": []}
    for (h, body, idx) in loops:
        c = mix_idx(ins, idx)
        ops = [ins[i][1] for i in idx]
        res["loops"].append({"header_block": h, "blocks": len(body), "instructions": len(idx), "first": idx[0], "last": idx[-1],
                             "innermost": not any(h2 != h and body2 < body for (h2, body2, _) in loops), "mix": c, "issue_cycles": weighted(c),
                             "valu": sum(k for cl, k in c.items() if cl.startswith("valu")),
                             "scratch": sum(1 for o in ops if o.startswith("scratch_")),
                             "xlane": sum(1 for o in ops if o.startswith(("v_readlane", "v_writelane"))),
                             "ops": {o: ops.count(o) for o in sorted(set(ops))}})
    if res["loops"]:
        inner = [l for l in res["loops"] if l["innermost"]]
        res["steady_state"] = res["loops"][loop_pick] if loop_pick is not None else max(inner, key=lambda l: l["valu"])
    return res


# the kernels bench.py prices: key -> (source file, kernel name pattern, which loop: None = the innermost loop with most VALU instructions,
# "whole" = the whole kernel body (kernels whose time is spread over many loops), files whose text fixes the instruction stream)
RECORDED = {
    "cc_bb144_fixed": ("minsum_regular.hip", "minsum_regular_kernel<6, 3, false, true, true, true>", None, ["minsum_regular.hip", "minsum_common.h", "minsum_f64.h", "mc_common.h"]),
    "cc_bb144_early_exit": ("mc_first.hip", "mc_first_kernel<8, 6, 3>", None, ["mc_first.hip", "mc_common.h"]),
    "circ144_bp": ("minsum_wg2.hip", "minsum_wg2_kernel<false>", "whole", ["minsum_wg2.hip", "minsum_common.h"]),
    "circ144_osd": ("osd_gj.hip", "osd0_gj_kernel<true>", "whole", ["osd_gj.hip", "osd_gj.h", "osd_common.h"]),
}


def digest(files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def record(path):
    out = {"rates": "profiles/r03_issue_rate.txt (tools/microbench/issue_rate.hip on the GPU box)", "entries": {}}
    for key, (src, pat, which, files) in RECORDED.items():
        r = analyse(src, pat)
        if which == "whole" or "steady_state" not in r:
            m, scope = r["whole"], "whole kernel (static)"
        else:
            m, scope = r["steady_state"]["mix"], f"steady-state loop [{r['steady_state']['first']}..{r['steady_state']['last']}] (static)"
        valu = sum(k for c, k in m.items() if c.startswith("valu_"))
        cyc = weighted(m)
        out["entries"][key] = {"kernel": r["kernel"], "scope": scope, "mix": m, "valu_instructions": valu, "valu_issue_cycles": cyc,
                               "cycles_per_valu_instruction": round(cyc / max(valu, 1), 4), "code_object": r["meta"],
                               "sources": files, "source_digest": digest(files)}
        print(f"{key:22s} {r['kernel'][:60]:60s} {scope:46s} VALU {valu:5d}  issue cycles {cyc:6d}  -> {cyc / max(valu, 1):.3f} cycles per instruction; {r['meta']}")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--record":
        record(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "isa_mix.json"))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("kernel")
    ap.add_argument("--loop", type=int, default=None)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--ops", action="store_true", help="mnemonic histogram of the steady-state loop")
    a = ap.parse_args()
    r = analyse(a.source, a.kernel, a.loop)
    if a.json:
        print(json.dumps(r))
        return
    print(r["kernel"])
    print("  code object:", r["meta"])
    print("  whole kernel:", describe(r["whole"]))
    for i, l in enumerate(r["loops"]):
        star = "*" if l is r.get("steady_state") else " "
        print(f" {star}loop {i} [{l['first']}..{l['last']}, {l['blocks']} blocks, {l['instructions']} instructions] {'innermost' if l['innermost'] else 'outer    '} "
              f"scratch ops {l['scratch']}, lane moves {l['xlane']}: {describe(l['mix'])}")
    if a.ops and r.get("steady_state"):
        print("  steady-state loop, instructions by mnemonic:")
        for o, k in sorted(r["steady_state"]["ops"].items(), key=lambda kv: -kv[1]):
            print(f"    {k:5d} {o}  [{classify(o)}]")


if __name__ == "__main__":
    main()
