#!/usr/bin/env python
"""One table of counter evidence per kernel from the rocpd databases tools/profile_kernels.sh (and tools/profile_round.sh) collected:
kernel, launch shape, average duration (kernel trace of the same pass), VALU instructions per row-wave, FP64 flops per evaluation,
fraction of the 78.6 TFLOP/s FP64 vector peak, real HBM bytes per launch (FETCH_SIZE doubled on gfx950 + WRITE_SIZE) and GB/s.
    python tools/make_kernel_table.py gpurun_out/<tag>_kernels [out.md] [gpurun_out/<round tag> for config 3]"""
import glob, sqlite3, sys
from pathlib import Path

PEAK_TF = 78.6
# workload -> (kernels of interest: substring -> (evaluations per launch, row-waves per launch, solves per evaluation))
E4 = 10_000
WORK = {
    "two_planet": {"k_main<2, true, true, 5, true, 4, false>": (5000 * 4096, 5000 * 64, 2), "k_finish<2, true, true, 5, false>": (5000 * 4096, None, 2)},
    "nuis": {"k_main<1, true, true, 1, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_finish<1, true, true, 1, false>": (E4 * E4, None, 1)},
    "fwd": {"k_main<1, false, false, 1, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_finish<1, false, false, 1, false>": (E4 * E4, None, 1)},
    "ofti": {"k_ofti_main": (E4 * E4, E4 * 157, 1), "k_ofti_finish": (E4 * E4, None, 1)},
    "logpost": {"k_model_fwd<true, false>": (E4 * E4, None, 1), "k_finish<1, true, false, 1, false>": (E4 * E4, None, 1)},
    "three_planet": {"k_main<3, true, true, 5, true, 4, false>": (6250 * 4096, 6250 * 64, 3), "k_finish<3, true, true, 5, false>": (6250 * 4096, None, 3)},
    # round 5: four planets and more on the planet-per-wave kernel (octo_mainp.h) — a block is P waves, so a tile-row is P row-waves
    "four_planet": {"k_mainp<true, true, 55, 4, 3>": (7500 * 4096, 7500 * 64 * 4, 4), "k_finish<4, true, true, 5, false>": (7500 * 4096, None, 4)},
    "five_planet": {"k_mainp<true, true, 55, 4, 3>": (8750 * 4096, 8750 * 64 * 5, 5), "k_finishp<true, true, 55>": (8750 * 4096, None, 5)},
    "eight_planet": {"k_mainp<true, true, 55, 2, 4>": (12500 * 4096, 12500 * 64 * 8, 8), "k_finishp<true, true, 55>": (12500 * 4096, None, 8)},
    # the per-GPU share of config 3 on 8 GPUs (1 250 walkers = 20 tiles): the eight-wave block of one-round launches
    "shard_1250": {"k_main<1, true, false, 1, true, 8, false>": (E4 * 1250, E4 * 20, 1), "k_finish<1, true, false, 1, false>": (E4 * 1250, None, 1)},
    "shard_2500": {"k_main<1, true, false, 1, true, 4, false>": (E4 * 2500, E4 * 40, 1), "k_finish<1, true, false, 1, false>": (E4 * 2500, None, 1)},
    "small_w1": {"k_small<1, true, false, 1, false>": (E4 * 1, E4 * 1 / 64.0, 1)},
    "small_w512": {"k_small<1, true, false, 1, false>": (E4 * 512, E4 * 512 / 64.0, 1)},
    # round 6: the non-uniform workloads (tests/synth.py: config_wide_prior, config_rv_gappy) and the tile sort that wide_prior turns on
    "wide_prior": {"k_main<1, true, false, 1, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_tile_sort": (E4 * E4, None, 1), "k_finish<1, true, false, 1, false>": (E4 * E4, None, 1)},
    "rv_gappy": {"k_main<1, true, false, 5, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_finish<1, true, false, 5, false>": (E4 * E4, None, 1)},
    "rv_gappy_nuis": {"k_main<1, true, true, 5, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_finish<1, true, true, 5, false>": (E4 * E4, None, 1)},
    "config3": {"k_main<1, true, false, 1, true, 4, false>": (E4 * E4, E4 * 157, 1), "k_finish<1, true, false, 1, false>": (E4 * E4, None, 1)},
}


def dominant_shape(con, sub):
    rows = con.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(duration) / 1e3 from kernels group by name, grid_x, grid_y").fetchall()
    rows = [r for r in rows if sub in r[0]]
    return max(rows, key=lambda r: r[4]) if rows else None


def counters(con, sub, grid):
    out = {}
    try:
        rows = con.execute("select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), grid_size from counters_collection "
                           "group by kernel_name, counter_name, grid_size").fetchall()
    except sqlite3.Error:
        return out
    for k, c, v, n, vg, gs in rows:
        if sub in k and (grid is None or gs == grid):
            out[c] = v; out["_vgpr"] = vg
    return out


def collect(wdir, subs):
    import re
    res = {s: {} for s in subs}
    for db in sorted(glob.glob(f"{wdir}/**/*.db", recursive=True)):
        if re.search(r"/w\d+/", db[len(str(wdir)):]):      # the round directory also holds the shard shapes' passes (w5000, w2500, w1250): not config 3
            continue
        con = sqlite3.connect(db)
        for s in subs:
            try:
                shp = dominant_shape(con, s)
            except sqlite3.Error:
                continue
            if not shp:
                continue
            grid = shp[1] * shp[2]
            c = counters(con, s, grid)
            r = res[s]
            r.setdefault("durs", []).append(shp[5]); r["grid"] = f"{shp[1]}x{shp[2]}"; r["wg"] = shp[3]; r["calls"] = shp[4]
            r.update(c)
    return res


def main():
    src = Path(sys.argv[1]); dst = sys.argv[2] if len(sys.argv) > 2 else None
    extra = sys.argv[3] if len(sys.argv) > 3 else None
    lines = ["| workload | kernel | grid x wg | avg µs (kernel trace, PMC passes) | VALU / row-wave | FP64 flops / evaluation | TFLOP/s | frac of 78.6 | HBM MB / launch (read + write) | HBM GB/s | of 8 TB/s |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    dirs = [(w, src / w) for w in WORK if (src / w).is_dir()]
    if extra:
        dirs.append(("config3", Path(extra)))
    for w, d in dirs:
        res = collect(d, WORK[w].keys())
        for sub, (evals, roww, _) in WORK[w].items():
            r = res[sub]
            if not r.get("durs"):
                continue
            us = sum(r["durs"]) / len(r["durs"])
            g = lambda k: r.get(k, 0.0)
            fl = (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64")) * 64.0
            tf = fl / (us * 1e-6) / 1e12
            rd, wr = g("FETCH_SIZE") * 1024.0 * 2.0, g("WRITE_SIZE") * 1024.0
            gbps = (rd + wr) / (us * 1e-6) / 1e9
            valu = f"{g('SQ_INSTS_VALU') / roww:.1f}" if roww else "-"
            lines.append(f"| {w} | `{sub}` | {r['grid']} x {r['wg']} | {us:.1f} | {valu} | {fl / evals:.1f} | {tf:.1f} | {tf / PEAK_TF:.3f} | "
                         f"{rd / 1e6:.2f} + {wr / 1e6:.2f} | {gbps:.0f} | {gbps / 8000:.4f} |")
    text = "\n".join(lines) + "\n"
    if dst:
        Path(dst).write_text(text)
    print(text)


if __name__ == "__main__":
    main()
