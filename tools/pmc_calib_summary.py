#!/usr/bin/env python3
"""profiles/r04_pmc_calib.json from gpurun_out/pmc_calib (tools/microbench/run_gather_calib.sh): what FETCH_SIZE / WRITE_SIZE
(KiB, rocprofv3 on gfx950) report per launch of each access pattern against the bytes the pattern is known to move.

  python tools/pmc_calib_summary.py gpurun_out/pmc_calib profiles/r04_pmc_calib.json
"""
import collections
import csv
import glob
import json
import os
import sys


def main(d, out):
    exp = json.load(open(os.path.join(d, "expected.json")))
    got = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                if r["Counter_Name"] == c:
                    got[name][c].append(float(r["Counter_Value"]) * 1024.0)
    res = {"note": "bytes per launch: counted (FETCH_SIZE / WRITE_SIZE x 1024, uncorrected) vs known; separate --pmc passes; "
                   "tools/microbench/gather_calib.hip", "patterns": {}}
    for k, e in exp.items():
        g = {c: (sum(v) / len(v) if v else None) for c, v in got.get(k, {}).items()}
        res["patterns"][k] = {"known": e, "counted_fetch": g.get("FETCH_SIZE"), "counted_write": g.get("WRITE_SIZE")}
    P = res["patterns"]

    def ratio(a, b):
        return (a / b) if (a and b) else None
    f = {}
    f["stream_read16_true_over_counted"] = ratio(P["k_stream_read16"]["known"]["read"], P["k_stream_read16"]["counted_fetch"])
    f["stream_read4_true_over_counted"] = ratio(P["k_stream_read4"]["known"]["read"], P["k_stream_read4"]["counted_fetch"])
    g = P["k_gather48"]
    cf = g["counted_fetch"]
    if cf:
        idx = g["known"]["read_idx"]
        # the index stream is a 4-byte coalesced read: take it out at ITS factor, the rest is the gather
        idx_counted = idx / f["stream_read4_true_over_counted"] if f["stream_read4_true_over_counted"] else idx
        rec_counted = cf - idx_counted
        f["gather48_counted_bytes_per_record"] = rec_counted / (g["known"]["read_requested"] / 48.0)
        f["gather48_sectors64_over_counted"] = g["known"]["read_at_64B_sectors"] / rec_counted
        f["gather48_lines128_over_counted"] = g["known"]["read_at_128B_lines"] / rec_counted
    f["stream_write4_true_over_counted"] = ratio(P["k_stream_write4"]["known"]["write"], P["k_stream_write4"]["counted_write"])
    f["stream_write16_true_over_counted"] = ratio(P["k_stream_write16"]["known"]["write"], P["k_stream_write16"]["counted_write"])
    a = P["k_atomic_flush"]
    if a["counted_write"] is not None:
        f["atomic_flush_counted_write_bytes_per_record"] = a["counted_write"] / a["known"]["records"]
    if a["counted_fetch"] is not None:
        f["atomic_flush_counted_fetch_bytes_per_record"] = a["counted_fetch"] / a["known"]["records"]
    res["factors"] = f
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(f, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
