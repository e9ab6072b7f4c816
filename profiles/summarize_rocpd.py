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
            # the same kernel name can be launched in several shapes by one command (bench.py's strong-scaling legs run k_main at
            # W, W/2, W/4, W/8): one line per (kernel, grid) so that every average belongs to ONE launch shape
            rows = cur.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(duration) / 1e3, min(duration) / 1e3, max(duration) / 1e3 "
                               "from kernels group by name, grid_x, grid_y, workgroup_x having count(*) >= 5 order by sum(duration) desc").fetchall()
            names = [r[0] for r in rows]
            if any(names.count(n) > 1 for n in names):
                out.append("per launch shape (kernels launched in more than one shape; durations in us):")
                out.append(f"{'kernel':58s} {'grid (work-items)':>18s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}")
                for r in rows:
                    if names.count(r[0]) > 1:
                        out.append(f"{r[0][:58]:58s} {str(r[1]) + ' x ' + str(r[2]):>18s} {r[3]:5d} {r[4]:6d} {r[5]:10.3f} {r[6]:10.3f} {r[7]:10.3f}")
        except sqlite3.Error as ex:
            out.append(f"(no kernels view: {ex})")
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
