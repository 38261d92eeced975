#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db) into a per-kernel CSV + markdown table.

  python tools/rocprof_summary.py gpurun_out/prof1/r01_results.db profiles/r01_bench_kernel_stats
"""
import csv
import sqlite3
import sys


def main(db_path, out_prefix, note=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), avg(grid_x) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "vgpr", "sgpr", "lds_bytes",
                    "workgroup_x", "avg_grid_x"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100 * r[2] / tot, 3), r[6], r[7], r[8],
                        r[9], round(r[10], 1)])
    with open(out_prefix + ".md", "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\n{note}\n\nsource: `{db_path}` (not committed; "
                f"total kernel time {tot / 1e6:.3f} ms)\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | LDS B |\n|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:45]:
            name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
            f.write(f"| `{name}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | "
                    f"{100 * r[2] / tot:.2f} | {r[6]} | {r[8]} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
