#!/usr/bin/env python
"""profiles/pmc_traffic.json from the rocpd databases tools/profile_round.sh collected (gpurun_out/<tag>/{stats,pmc_fetch,pmc_write,
pmc_mix,pmc_sq}/**/*.db): per-launch averages of the dominant kernel k_main<1, true, false, 1> on bench.py's default workload
(config 3), with the corrections MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE doubled; WRITE_SIZE checked against the
known partials byte count), the FP64 flop count per evaluation and the issue-time model. Records the sha256 of the kernel
sources so that bench.py can refuse figures collected for a different kernel.
    python tools/make_pmc_json.py gpurun_out/<tag> [profiles/pmc_traffic.json]
Round 5: the per-GPU launch shapes of a strong-scaled run (SURVEY 8d "Scaling runs": 1e4 walkers split over 2 / 4 / 8 GPUs) get their own
counter passes (gpurun_out/<tag>/w5000, w2500, w1250: the same bench command with --walkers W) and land under "shapes" keyed by the
walkers per launch; bench.py picks the entry of the shape it launched. The kernel of a shape is whichever k_main instantiation the
planner chose for it (the eight-wave block k_main<1, true, false, 1, true, 8, false> for one-round launches)."""
import glob, json, re, sqlite3, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import kernel_source_hash

KERNEL = "k_main<1, true, false, 1, true, 4, false>"      # (P, GRAD, NUIS, KM, FUSED, waves per block)
W, E = 10_000, 10_000
SHARD_WALKERS = (5_000, 2_500, 1_250)


def dbs(src, sub=""):
    """The rocpd databases of ONE launch shape: everything under src/sub except the shard-shape passes src/w<walkers>/ (their k_main may
    carry the same kernel name as the full batch's)."""
    base = f"{src}/{sub}" if sub else src
    return [db for db in sorted(glob.glob(f"{base}/**/*.db", recursive=True)) if not re.search(r"/w\d+/", db[len(src):])]


def dominant_kmain(src):
    """The k_main instantiation with the largest total duration in the kernel traces under src (the planner's choice for that launch shape)."""
    tot = {}
    for db in dbs(src):
        con = sqlite3.connect(db)
        try:
            for name, d in con.execute("select name, sum(duration) from kernels group by name").fetchall():
                if "k_main<" in name:
                    tot[name] = tot.get(name, 0.0) + d
        except sqlite3.Error:
            continue
    if not tot:
        return None
    name = max(tot, key=tot.get)
    return name[name.index("k_main<"):name.index(">") + 1]


def counters(src, KERNEL=KERNEL):
    out = {}
    for db in dbs(src):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            continue
        for k, c, v, n in rows:
            if KERNEL in k:
                out[c] = (v, n)
    return out


def kernel_avg_us(src, KERNEL=KERNEL, W=W, sub="stats"):
    """Average duration of the kernel in the kernel-trace run of the default bench command — of the launch shape the counters belong to
    (grid = ceil(W/64) tiles x tasks): the same command also launches k_main on W/2, W/4, W/8 walkers (strong_scaling_projection), and an
    average over all four shapes would describe none of them."""
    cols = (W + 63) // 64
    for db in dbs(src, sub):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select name, grid_x, grid_y, count(*), avg(duration) / 1e3 from kernels group by name, grid_x, grid_y").fetchall()
        except sqlite3.Error:
            continue
        wg = 64 * int(KERNEL.rstrip(">").split(",")[-2])      # (…, waves per block, FINF)
        rows = [r for r in rows if KERNEL in r[0] and r[1] == cols * wg]
        if rows:
            r = max(rows, key=lambda r: r[3])
            return r[4], r[3]
    return None, 0


def shape_entry(src, Wn):
    """Counter-derived figures of ONE launch shape (Wn walkers x E rows, fwd+grad): the passes under src/ are pmc_fetch, pmc_write, pmc_mix."""
    kern = dominant_kmain(src)
    if kern is None:
        return None
    c = counters(src, kern)
    g = lambda k: c[k][0] if k in c else 0.0
    if "SQ_INSTS_VALU_FMA_F64" not in c:
        return None
    evals = float(Wn) * E
    row_waves = E * ((Wn + 63) // 64)
    fetch_b, write_b = g("FETCH_SIZE") * 1024.0 * 2.0, g("WRITE_SIZE") * 1024.0
    fma, mul, add, trans = g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_ADD_F64"), g("SQ_INSTS_VALU_TRANS_F64")
    avg_us, calls = kernel_avg_us(src, kern, Wn, sub="pmc_mix")
    return {"kernel": "octo::" + kern, "walkers_per_launch": Wn, "rows": E,
            "workload": f"the per-GPU share of config 3 when 1e4 walkers are split over {W // Wn} GPUs: {Wn} walkers x 1e4 RA/Dec epochs, fwd+reverse-grad "
                        f"(bench.py --walkers {Wn})",
            "source": f"{src} (tools/profile_round.sh: --pmc FETCH_SIZE, --pmc WRITE_SIZE and instruction-mix passes, each in its own run with --kernel-trace)",
            "kernel_trace_avg_us": avg_us, "kernel_trace_calls": calls,
            "hbm_bytes_per_launch": int(round(fetch_b + write_b)), "hbm_read_bytes_per_launch": int(round(fetch_b)),
            "hbm_write_bytes_per_launch": int(round(write_b)),
            "algorithmic_bytes_per_launch": int(Wn * E * 40.0 + Wn * 136.0),
            "valu_instructions_per_row_per_wave": g("SQ_INSTS_VALU") / row_waves,
            "fp64_flops_per_eval": (2 * fma + mul + add + trans) * 64.0 / evals,
            "lanes_note": "flops are counted per WAVE instruction x 64 lanes over Wn x E evaluations: a partly filled last tile (1 250 = 19.53 tiles) counts its idle lanes as work issued",
            "instruction_mix_per_row": {"v_fma_f64": fma / row_waves, "v_mul_f64": mul / row_waves, "v_add_f64": add / row_waves,
                                        "v_rcp_f64 (TRANS_F64)": trans / row_waves, "fp32_fma": g("SQ_INSTS_VALU_FMA_F32") / row_waves,
                                        "fp32_transcendental": g("SQ_INSTS_VALU_TRANS_F32") / row_waves},
            "raw_counters_per_launch": {k: c[k][0] for k in ("GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_INSTS_VALU") if k in c}}


def main():
    src = sys.argv[1]
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "pmc_traffic.json"
    c = counters(src)
    need = ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"]
    missing = [k for k in need if k not in c]
    if missing:
        raise SystemExit(f"missing counters {missing} for {KERNEL} under {src}")
    g = lambda k: c[k][0] if k in c else 0.0
    evals = float(W) * E
    row_waves = E * ((W + 63) // 64)
    fetch_b = g("FETCH_SIZE") * 1024.0 * 2.0            # KB -> B, gfx950: 128-B requests tallied as 64 B
    write_b = g("WRITE_SIZE") * 1024.0
    fma, mul, add, trans = g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_ADD_F64"), g("SQ_INSTS_VALU_TRANS_F64")
    flops_per_eval = (2 * fma + mul + add + trans) * 64.0 / evals
    valu = g("SQ_INSTS_VALU")
    f32_fma, f32_tr = g("SQ_INSTS_VALU_FMA_F32"), g("SQ_INSTS_VALU_TRANS_F32")
    # issue-time model: per-class cost of one wave-instruction on one SIMD (tools/ubench.hip, MI355X): FP64 fma/mul/add/cvt 2.15 ns,
    # v_rcp_f64 6.9 ns, FP32/int 1.04 ns, FP32 transcendental 3.4 ns; FP64-rate converts/rounds are the part of VALU that is none of those
    per = lambda x: x / row_waves
    f64_alu = per(fma + mul + add)
    rcp = per(trans)
    f32_t = per(f32_tr)
    cvt64 = 4.0                                          # v_cvt_f32_f64, v_rndne_f64, v_cvt_f64_f32 x2 in the row body (ISA)
    f32_other = max(per(valu) - f64_alu - rcp - f32_t - cvt64, 0.0)
    ns_row = 2.15 * (f64_alu + cvt64) + 6.9 * rcp + 1.04 * f32_other + 3.4 * f32_t
    avg_us, calls = kernel_avg_us(src)
    out = {
        "kernel": "octo::" + KERNEL, "walkers_per_launch": W, "rows": E,
        "workload": "config3: 1 planet, 1e4 RA/Dec epochs x 1e4 walkers, fwd+reverse-grad (bench.py defaults)",
        "source": f"{src} (tools/profile_round.sh: rocprofv3 --kernel-trace --stats plus --pmc FETCH_SIZE, --pmc WRITE_SIZE, instruction-mix and SQ "
                  "passes, each in its own run; ROCm 7.2, MI355X); summary committed as profiles/<tag>_rocprof_summary.txt",
        "kernel_source_sha256": kernel_source_hash(),
        "kernel_trace_avg_us": avg_us, "kernel_trace_calls": calls,
        "FETCH_SIZE_KB_per_launch": g("FETCH_SIZE"), "WRITE_SIZE_KB_per_launch": g("WRITE_SIZE"),
        "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled (MI355X_MICROARCH.md, HBM section). WRITE_SIZE checked against a "
                      "known byte count: tasks x 10 sums x 10000 walkers x 8 B of partials + 0.72 MB of outputs.",
        "hbm_bytes_per_launch": int(round(fetch_b + write_b)),
        "hbm_read_bytes_per_launch": int(round(fetch_b)), "hbm_write_bytes_per_launch": int(round(write_b)),
        "algorithmic_bytes_per_launch": int(W * E * 40.0 + W * 136.0),
        "valu_instructions_per_row_per_wave": per(valu),
        "fp64_flops_per_eval": flops_per_eval,
        "instruction_mix_per_row": {"v_fma_f64": per(fma), "v_mul_f64": per(mul), "v_add_f64": per(add), "v_rcp_f64 (TRANS_F64)": rcp,
                                    "fp32_fma": per(f32_fma), "fp32_transcendental": f32_t, "other (fp32/int/cvt)": f32_other + cvt64},
        "issue_model": {"ns_per_wave_instruction (tools/ubench.hip, one SIMD)": {"fp64 fma/mul/add/cvt": 2.15, "v_rcp_f64": 6.9, "fp32/int": 1.04,
                                                                                "fp32 transcendental": 3.4},
                        "ns_per_row_per_wave": ns_row, "simds": 1024},
    }
    for k in ("GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS",
              "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"):
        if k in c:
            out.setdefault("raw_counters_per_launch", {})[k] = c[k][0]
    for Wn in SHARD_WALKERS:
        e = shape_entry(f"{src}/w{Wn}", Wn)
        if e is not None:
            out.setdefault("shapes", {})[str(Wn)] = e
    dst.write_text(json.dumps(out, indent=2) + "\n")
    print(json.dumps(out, indent=2))


if __name__ == "__main__":
    main()
