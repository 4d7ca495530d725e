#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of profiles/run_profile.sh (gpurun_out/prof_<tag>/) into
the small committed files profiles/<tag>_summary.md and profiles/<tag>_summary.json.

    python profiles/summarize.py r01a

Per SLFP kernel and per launch geometry (one MobileNetV1 layer = one grid size): average
duration from --kernel-trace --stats, and per-launch PMC counters from the separate --pmc
passes.  FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE
reads exactly half of a wide coalesced stream (MI355X_MICROARCH.md, section HBM), so
hbm_read_bytes = 2 * FETCH_SIZE * 1024 and hbm_write_bytes = WRITE_SIZE * 1024.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"slfp::(k_[a-z0-9_]+)(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    # kernels with a _Float16 parameter come back mangled: _ZN4slfp14k_dense_encodeILi0EEEv...
    m = re.search(r"_ZN4slfp\d+(k_[a-z0-9_]+?)I((?:Li\d+E)+)E", name)
    if m:
        return m.group(1) + "<" + ", ".join(re.findall(r"Li(\d+)E", m.group(2))) + ">"
    return None


def load_counters(d):
    """-> {(kernel, grid): {counter: [values per dispatch]}} and durations."""
    out = defaultdict(lambda: defaultdict(list))
    regs = {}
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")):
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            key = (k, int(r["Grid_Size"]))
            per_dispatch[(key, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            regs[key] = (int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Workgroup_Size"]))
        for (key, _), cs in per_dispatch.items():
            for c, v in cs.items():
                out[key][c].append(v)
    return out, regs


def load_trace(d):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur[(k, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return dur


def main():
    tag = sys.argv[1]
    if len(sys.argv) < 3:
        raise SystemExit("usage: summarize.py TAG \"what was profiled (command, net, batch)\"  -- the title is not guessed")
    what = sys.argv[2]
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dur = load_trace(os.path.join(base, "stats"))
    counters = {}
    regs = {}
    for sub in ("fetch", "write", "sq", "sq2"):
        c, r = load_counters(os.path.join(base, sub))
        regs.update(r)
        for key, cs in c.items():
            counters.setdefault(key, {}).update({k: sum(v) / len(v) for k, v in cs.items()})
    rows = []
    for key in sorted(dur, key=lambda k: (k[0], -k[1])):
        d = sorted(dur[key])
        d = d[: max(1, len(d))]
        avg_us = sum(d) / len(d) / 1e3
        c = counters.get(key, {})
        row = {"kernel": key[0], "grid_threads": key[1], "launches": len(d), "avg_us": round(avg_us, 2),
               "min_us": round(d[0] / 1e3, 2)}
        if key in regs:
            row.update(vgpr=regs[key][0], agpr=regs[key][1], lds=regs[key][2], wg=regs[key][3])
        if "FETCH_SIZE" in c:
            row["hbm_read_MB"] = round(2 * c["FETCH_SIZE"] * 1024 / 1e6, 2)
        if "WRITE_SIZE" in c:
            row["hbm_write_MB"] = round(c["WRITE_SIZE"] * 1024 / 1e6, 2)
        if "hbm_read_MB" in row and "hbm_write_MB" in row:
            row["hbm_GBps"] = round((row["hbm_read_MB"] + row["hbm_write_MB"]) / avg_us * 1e3 / 1e3, 1)
        for name in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
                     "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
                     "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if name in c:
                row[name] = c[name]
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            wc = c["SQ_WAVE_CYCLES"]
            for a, b in (("valu_active_frac", "SQ_ACTIVE_INST_VALU"), ("wait_any_frac", "SQ_WAIT_ANY"),
                         ("wait_inst_frac", "SQ_WAIT_INST_ANY")):
                if b in c:
                    row[a] = round(c[b] / wc, 3)
        if "SQ_LDS_IDX_ACTIVE" in c and c.get("SQ_LDS_BANK_CONFLICT") is not None and c["SQ_LDS_IDX_ACTIVE"] > 0:
            row["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            # matrix-core utilisation: MFMA-busy cycles summed over the 1024 SIMDs / (1024 x chip cycles); GRBM_GUI_ACTIVE
            # is the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            row["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"] > 0:
            row["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        rows.append(row)
    json.dump(rows, open(os.path.join(ROOT, "profiles", f"{tag}_summary.json"), "w"), indent=1)
    cols = ["kernel", "grid_threads", "launches", "avg_us", "vgpr", "agpr", "lds", "hbm_read_MB", "hbm_write_MB", "hbm_GBps",
            "valu_active_frac", "wait_any_frac", "lds_conflict_frac", "mfma_busy_frac", "valu_insts_per_wave"]
    with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` ({what}, 1x MI355X)\n\n")
        f.write("Source: `profiles/run_profile.sh` (kernel-trace --stats pass + separate --pmc passes). "
                "hbm_read_MB = 2 x FETCH_SIZE (gfx950 correction), hbm_write_MB = WRITE_SIZE.\n\n")
        f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
        for r in rows:
            f.write("| " + " | ".join(str(r.get(c, "")) for c in cols) + " |\n")
    tot = defaultdict(float)
    for r in rows:
        tot[r["kernel"]] += r["avg_us"]
    for r in rows:
        print(" ".join(f"{c}={r.get(c, '')}" for c in cols))
    print({k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
