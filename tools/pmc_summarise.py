#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/pmc_passes.sh -> profiles/pmc.json (read by bench.py) + a readable table.

Per kernel the counted launches are the LAST ones of the workload's launch sequence (warm-ups come first).  Every entry records
the kernel symbol, the sha256 of the kernel's source files at collection time (bench.py refuses a record whose sources changed) and
the workload units one launch processed, so a count per unit carries over to other batch sizes of the same kernel.
HBM bytes: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of a wide coalesced read,
MI355X_MICROARCH.md "HBM") -- an upper estimate for narrow reads."""
import csv
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qldpc-branched-off_amd", "csrc")


def digest(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def regular_args(kernel_name):
    """template arguments of minsum_regular_kernel<CDEG, VDEG, DAMP, NANFREE, MC, FIXED> as strings, or None"""
    import re
    mt = re.search(r"minsum_regular_kernel<([^>]*)>", kernel_name)
    return [x.strip() for x in mt.group(1).split(",")] if mt else None


def load_pass(d):
    """{counter: [(dispatch_id, kernel_name, value), ...]} of one pass directory, in dispatch order."""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for k in out:
        out[k].sort()
    return out


def main():
    src = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(src.rstrip("/"))
    with open(os.path.join(src, "workload.json")) as fh:
        wl = json.load(fh)
    counters = {}
    for p in ("sq_a", "sq_b", "fetch", "write", "grbm"):
        counters.update(load_pass(os.path.join(src, p)))
    cl = wl["circuit_level"]
    regular = ["minsum_regular.hip", "minsum_common.h", "mc_common.h"]
    specs = {   # key: (kernel-name match, counted launches, units per launch, unit, sources)
        f"cc_{wl['code']}_fixed": (lambda k: (regular_args(k) or [""] * 6)[4:6] == ["true", "true"], 2,
                                   wl["cc_fixed"]["shots_per_launch"] * wl["cc_fixed"]["max_iter"], "shot_iteration", regular),
        f"cc_{wl['code']}_early_exit": (lambda k: (regular_args(k) or [""] * 6)[4:6] == ["true", "false"], 2,
                                        wl["cc_early_exit"]["shots_per_launch"], "shot", regular),
        f"{wl['circuit']}_bp": (lambda k: "minsum_wg_lean_kernel" in k, 2, (cl["iters_z"] + cl["iters_x"]) / 2.0, "decode_iteration",
                                ["minsum_wg.hip", "minsum_common.h"]),
        f"{wl['circuit']}_osd": (lambda k: "osd0_lds_kernel" in k or "osd0_fwd_kernel" in k, 2, (cl["osd_z"] + cl["osd_x"]) / 2.0, "osd_shot",
                                 ["gf2.hip", "osd_common.h", "osd_fwd.hip"]),
    }
    entries, lines = {}, []
    for key, (match, nl, units, unit, sources) in specs.items():
        e = {"unit": unit, "units_per_launch": units, "sources": sources, "source_digest": digest(sources), "counted_launches": nl,
             "source": f"profiles/{tag}_pmc.txt (rocprofv3 --pmc, tools/pmc_passes.sh, collected {time.strftime('%Y-%m-%d')})"}
        for cname, rows in counters.items():
            vals = [(did, k, v) for did, k, v in rows if match(k)]
            if not vals:
                continue
            e["kernel"] = vals[-1][1]
            last = vals[-nl:]
            e[cname] = sum(v for _, _, v in last) / len(last)
        if "SQ_INSTS_VALU" not in e:
            continue
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in e:
                    e[c + "_frac"] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT", 0.0) / e["SQ_LDS_IDX_ACTIVE"], 4)
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        # algorithmic floors (lane-operations per unit)
        if unit == "decode_iteration":
            fl = 0.0
            for dm in cl["dims"]:
                fl += dm["nnz"] * (11.0 + 14.0 / 6.0) + dm["nonempty_rows"] * 2.0 + dm["nnz"] + dm["n"]
            e["floor_lane_ops_per_unit"] = round(fl / len(cl["dims"]), 1)
        entries[key] = e
        per = e["SQ_INSTS_VALU"] / units
        lines.append(f"{key:26s} {e.get('kernel', '?')[:70]:70s} VALU/launch {e['SQ_INSTS_VALU']:.4g}  per {unit} {per:.3f}  "
                     f"SALU {e.get('SQ_INSTS_SALU', 0) / units:.3f}  LDS {e.get('SQ_INSTS_LDS', 0) / units:.3f}  wait_any {e.get('SQ_WAIT_ANY_frac')}  "
                     f"wait_inst {e.get('SQ_WAIT_INST_ANY_frac')}  active {e.get('SQ_ACTIVE_INST_ANY_frac')}  lds_conflict {e.get('lds_bank_conflict_frac')}  "
                     f"HBM B/launch {e.get('hbm_bytes_per_launch')}")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "pmc.json"), "w") as fh:
        json.dump({"collected": time.strftime("%Y-%m-%d"), "tag": tag, "workload": wl, "entries": entries}, fh, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.txt"), "w") as fh:
        fh.write(f"rocprofv3 --pmc passes of tools/pmc_workload.py (tools/pmc_passes.sh), per counted launch (average of the last launches of each kernel)\n")
        fh.write("workload: " + json.dumps(wl) + "\n\n")
        fh.write("\n".join(lines) + "\n\nraw per-launch averages:\n")
        for key, e in entries.items():
            fh.write(key + ": " + json.dumps({k: v for k, v in e.items() if k.isupper() or k.startswith("SQ_")}) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
