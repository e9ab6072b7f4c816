#!/usr/bin/env python
"""Turn rocprofv3's rocpd SQLite outputs (gpurun_out/<run>/{stats,pmc_*}/*.db) into the text summaries
committed next to this script. Usage: python profiles/summarize_rocpd.py gpurun_out/r1 profiles/r1_v0"""
import glob
import sqlite3
import sys


def main(src, dst):
    out = []
    for db in sorted(glob.glob(f"{src}/**/*.db", recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        out.append(f"== {db}")
        try:
            rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            if rows:
                out.append("kernel-trace --stats (durations in us):")
                out.append(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
                for r in rows:
                    out.append(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:12.3f} {r[3]:10.3f} {r[4]:7.2f}")
        except sqlite3.Error as ex:
            out.append(f"(no top_kernels: {ex})")
        try:
            rows = cur.execute("select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), max(sgpr_count), max(grid_size), max(workgroup_size) "
                               "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
            if rows:
                out.append("pmc (average per dispatch):")
                for r in rows:
                    out.append(f"{r[0][:60]:60s} {r[1]:22s} {r[2]:18.3f}  n={r[3]} vgpr={r[4]} sgpr={r[5]} grid={r[6]} wg={r[7]}")
        except sqlite3.Error as ex:
            out.append(f"(no counters: {ex})")
    text = "\n".join(out) + "\n"
    open(dst + "_rocprof_summary.txt", "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
