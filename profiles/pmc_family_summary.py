#!/usr/bin/env python3
"""Per-dispatch PMC table from the passes of profiles/pmc_family.sh: one row per (kernel, grid, dispatch order),
counters averaged over repeats of the same layer (variants.py launches each layer 3 x rounds times in order)."""
import csv, glob, os, re, sys
from collections import defaultdict, OrderedDict

base = sys.argv[1]
rows = OrderedDict()
for d in sorted(glob.glob(os.path.join(base, "*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        seq = defaultdict(int)
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", k)
            if not m:
                continue
            per[r["Dispatch_Id"]]["_k"] = m.group(1) + (m.group(2) or "")
            per[r["Dispatch_Id"]]["_g"] = int(r["Grid_Size"])
            per[r["Dispatch_Id"]]["_lds"] = int(r["LDS_Block_Size"])
            per[r["Dispatch_Id"]]["_v"] = int(r["VGPR_Count"])
            per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        # dispatches come in runs of identical launches (same layer): group consecutive equal (kernel, grid, lds)
        run_id = 0
        prev = None
        layer_of_run = {}
        for did in sorted(per, key=int):
            e = per[did]
            key = (e["_k"], e["_g"], e["_lds"])
            if key != prev:
                run_id += 1
                prev = key
            # the same layer recurs every `rounds` passes: identify it by its ordinal among runs with the same key
            rows.setdefault((key, ), None)
            ent = rows.get(("acc", key))
            if ent is None:
                ent = rows[("acc", key)] = defaultdict(list)
            for c, v in e.items():
                if not c.startswith("_"):
                    ent[c].append(v)
            ent["_v"] = [e["_v"]]
print("kernel | grid | lds | vgpr | counters (mean per dispatch)")
for k, ent in rows.items():
    if k[0] != "acc":
        continue
    key = k[1]
    out = {c: sum(v) / len(v) for c, v in ent.items() if not c.startswith("_")}
    s = {}
    if "FETCH_SIZE" in out: s["hbm_read_MB"] = round(2 * out["FETCH_SIZE"] * 1024 / 1e6, 1)
    if "WRITE_SIZE" in out: s["hbm_write_MB"] = round(out["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in out: s["l2_hit"] = round(out["TCC_HIT_sum"] / max(1.0, out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 3)
    if "SQ_WAVE_CYCLES" in out and out["SQ_WAVE_CYCLES"] > 0:
        wc = out["SQ_WAVE_CYCLES"]
        s["valu_active"] = round(out.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
        s["wait_any"] = round(out.get("SQ_WAIT_ANY", 0) / wc, 3)
        s["wait_inst"] = round(out.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        s["valu_per_wave"] = round(out.get("SQ_INSTS_VALU", 0) / max(1.0, out.get("SQ_WAVES", 1)), 1)
    if "SQ_LDS_IDX_ACTIVE" in out and out["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in out:
        s["lds_conflict"] = round(out["SQ_LDS_BANK_CONFLICT"] / out["SQ_LDS_IDX_ACTIVE"], 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out and out["GRBM_GUI_ACTIVE"] > 0:
        # MFMA-busy cycles summed over the SIMDs / (1024 SIMDs x chip cycles; GRBM_GUI_ACTIVE sums the 8 XCDs)
        s["mfma_busy"] = round(out["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * out["GRBM_GUI_ACTIVE"] / 8), 3)
    print(key[0], "|", key[1], "|", key[2], "|", ent["_v"][0], "|", s)
