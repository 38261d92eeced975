#!/usr/bin/env python3
"""Summarise the per-pass CSVs of tools/pmc_profile.sh into one JSON (per kernel: mean counter values per launch)
and derive HBM traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KiB and were
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts a wide (16 B/lane) read at half its bytes, so the
read side is doubled; WRITE_SIZE is reported uncorrected.
Round 4 calibrated both on this library's OWN access patterns (tools/microbench/gather_calib.hip -> profiles/r04_pmc_calib.json):
the x2 holds for 4-byte-per-lane coalesced reads and for the blend's gather of 48-byte records alike (the memory side is asked for
128-byte lines, each tallied at 64 B: a gathered record costs 1.25 lines = 160 B of fetch, 81 B counted), WRITE_SIZE is exact for
4- and 16-byte-per-lane streams, and the backward blend's float-atomic flush shows as 64 B written per record and no fetch.

  python tools/pmc_summary.py gpurun_out/pmc3 profiles/r01_pmc_blend.json "note"
"""
import collections
import csv
import glob
import json
import os
import sys


def calib():
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03_valu_calib.json")
    try:
        return json.load(open(f))["classes"]
    except Exception:
        return {"plain": 2.366, "double_pass": 4.296, "trans": 8.152}


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
            # 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Cost of a wave64 VALU instruction in SIMD cycles: CALIBRATED on
            # this chip (tools/microbench/valu_calib.hip -> profiles/r03_valu_calib.json): ~2.3 for a full-rate op, ~8.2 for a
            # transcendental (SQ_INSTS_VALU_TRANS).  Packed / DPP / f64 ops (~4.3) are not separable in the counters, so this is a
            # LOWER bound of the vector pipe's busy fraction.  (Rounds 1-2 multiplied by an uncalibrated 4.0.)
            cal = calib()
            trans = m.get("SQ_INSTS_VALU_TRANS_F32", m.get("SQ_INSTS_VALU_TRANS", 0.0))
            f64 = sum(m.get(c, 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"))
            cyc = (m["SQ_INSTS_VALU"] - trans - f64) * cal["plain"] + trans * cal["trans"] + f64 * cal["double_pass"]
            m["valu_trans_instructions"], m["valu_f64_instructions"] = trans, f64
            m["simd_cycles_available"] = m["GRBM_GUI_ACTIVE"] / 8.0
            m["valu_pipe_cycles_per_simd_lower_bound"] = cyc / 1024.0
            m["valu_issue_frac_est"] = (cyc / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)
            m["valu_cycles_per_instr_used"] = {"plain": cal["plain"], "trans": cal["trans"], "f64": cal["double_pass"]}
        res["kernels"][k] = m
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, m in res["kernels"].items():
        print(k, {c: round(m[c], 3) for c in ("hbm_traffic_bytes", "valu_issue_frac_est") if c in m})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
