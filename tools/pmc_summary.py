#!/usr/bin/env python3
"""Summarise the per-pass CSVs of tools/pmc_profile.sh into one JSON (per kernel: mean counter values per launch)
and derive HBM traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB and were
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts a wide (16 B/lane) read at half its bytes, so the
read side is doubled; WRITE_SIZE is reported uncorrected (uncalibrated per the guide).

  python tools/pmc_summary.py gpurun_out/pmc3 profiles/r01_pmc_blend.json "note"
"""
import collections
import csv
import glob
import json
import os
import sys


def main(d, out, note=""):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"note": note, "kernels": {}}
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["launches_sampled"] = min(len(v) for v in cs.values())
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_read_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2.0
            m["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
            m["hbm_traffic_bytes"] = m["hbm_read_bytes_corrected"] + m["hbm_write_bytes"]
        if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
            # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU op occupies its SIMD ~4 cycles
            m["valu_issue_frac_est"] = (m["SQ_INSTS_VALU"] / 1024.0 * 4.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)
        res["kernels"][k] = m
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, m in res["kernels"].items():
        print(k, {c: round(m[c], 3) for c in ("hbm_traffic_bytes", "valu_issue_frac_est") if c in m})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
