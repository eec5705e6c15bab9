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

# issue classes (cycles per wave64 instruction per SIMD with >= 2 waves on it; profiles/r03_issue_rate.txt)
CLASS_CYCLES = {"valu_f64": 4, "valu_b32": 2, "valu_b32_slow": 4, "valu_trans": 8, "valu_xlane": 4}


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
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp")) or op.endswith("_dpp"):
            return "valu_xlane"
        if op.startswith(("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")):
            return "valu_trans"
        if re.search(r"_(f64|i64|u64|b64)(_e32|_e64)?$", op) or op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
            return "valu_f64"
        if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32")):
            return "valu_b32_slow"
        return "valu_b32"
    return "other"


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
    return sum(CLASS_CYCLES[c] * k for c, k in cnt.items() if c in CLASS_CYCLES)


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


def main():
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
