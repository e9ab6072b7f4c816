"""Prototype (NumPy) of a PREVIOUS-ROW warm start for the Kepler solve of k_main. Rows of a table are epoch-sorted
(src/likelihoods/relative-astrometry.jl:46-47) and a wave walks a contiguous slice of them for the same 64 walkers, so the previous
row's (sin E, cos E, 1/(1 − e cos E)) is a starter for the next one:

    x  = ΔM / D                                  ΔM = 2π Δt / P,  D = 1 − e cos E of the previous row
    dE = x (1 − ½ (e sin E / D) x [+ third order])
    (s1, c1) = rotation of (sin E, cos E) by dE   (the table rotation's polynomial: |dE| must stay ~1e-2)
    f0 = dE − ΔM − e (s1 − sin E)                 = E1 − e sin E1 − M without E or M as numbers
    δ5 of Markley's correction, rotation by δ5    (unchanged)

with a WAVE-UNIFORM fallback to the Markley starter whenever any lane of the wave fails a bound. This script measures, on config 3's
own walkers and epochs: (1) the accuracy of the warm chain against an 80-bit solve, including the drift of a chain that never sees E
or M as numbers; (2) the fallback rate per wave-row for a given bound, by eccentricity bin; (3) the issue-time model's estimate of the
step time. Development aid; not shipped, not imported by tests.
    python tools/kepler_warm_proto.py [rows_per_wave] [tol]"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import synth
from kepler_proto import starter32, refine_device, truth

TWO_PI = 2 * np.pi


def rcp23(x, rng):
    return (1.0 / x) * (1 + rng.uniform(-1, 1, x.shape) * 2.0**-23)


def correction(f0, s1, c1, e, rng):
    """octo_device.h: kepler_solve from f0 on (one crude reciprocal, δ4 / δ5 as quotient corrections, rotation by δ5, invD with one NR step)."""
    hf2 = 0.5 * e * s1; q24 = hf2 / 12; sf3 = e / 6 * c1; f1 = 1 - e * c1
    r3 = rcp23(f1 * f1 - f0 * hf2, rng)
    r4 = f1 * r3; d3 = -f0 * r4
    den4 = f1 + d3 * (hf2 + d3 * sf3)
    d4 = d3 - r4 * (den4 * d3 + f0)
    den5 = f1 + d4 * (hf2 + d4 * (sf3 - d4 * q24))
    d5 = d4 - r4 * (den5 * d4 + f0)
    dd = d5 * d5
    sd = d5 * (1 + dd * (-1 / 6)); cm1 = dd * (-0.5 + dd / 24)
    sE = s1 + (c1 * sd + s1 * cm1); cE = c1 + (-s1 * sd + c1 * cm1)
    D = 1 - e * cE
    r = rcp23(D, rng); r = r + r * (1 - D * r)
    return sE, cE, r, d5


def cold(frac, e, rng):
    M = TWO_PI * frac
    E1 = starter32(M, e)
    s1, c1 = np.sin(E1), np.cos(E1)
    f0 = (E1 - M) - e * s1
    return correction(f0, s1, c1, e, rng)


XMAX = 0.1


def warm(sE, cE, invD, dM, e, rng, order=2):
    x = dM * invD
    g = (0.5 * e * sE) * invD
    if order == 2:
        dE = x - (g * x) * x
    else:
        h = 2 * g * g - (e * cE * invD) / 6
        dE = x * (1 - x * (g - x * h))
    r2 = dE * dE
    # |dE| <= ~0.1: sin to r^9, cos - 1 to r^8
    sr = dE * (1 + r2 * (-1 / 6 + r2 * (1 / 120 + r2 * (-1 / 5040 + r2 / 362880)))); cm1 = r2 * (-0.5 + r2 * (1 / 24 + r2 * (-1 / 720 + r2 / 40320)))
    ds = sE * cm1 + cE * sr
    s1 = sE + ds; c1 = cE + (cE * cm1 - sE * sr)
    f0 = (dE - dM) - e * ds
    out = correction(f0, s1, c1, e, rng)
    return out + (dE, x)


def main():
    rows_per_wave = int(sys.argv[1]) if len(sys.argv) > 1 else 74
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
    order = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    n_rows = rows_per_wave * 12
    cfg = synth.config_astrom(n_epochs=10_000, n_walkers=10_000, cfg=3)
    el = cfg["elems"]; W = el.shape[1]
    a, e, tp, Mst = el[0], el[1], el[5], el[6]
    P = synth.K_YR * np.sqrt(a**3 / Mst)
    invP = 1.0 / P
    t = cfg["table"]["epoch"][:n_rows]
    rng = np.random.default_rng(5)
    tiles = (W + 63) // 64
    pad = tiles * 64 - W
    lane_ok = np.concatenate([np.ones(W, bool), np.zeros(pad, bool)])
    fallback_rows = 0; warm_rows = 0
    lane_fail_by_e = np.zeros(10); lane_rows_by_e = np.zeros(10)
    ebin = np.minimum((e * 10).astype(int), 9)
    err_w_max = 0.0; err_c_max = 0.0; chain_len_at_worst = 0
    errs = []
    state = None
    chain = np.zeros(W, int)
    for j, tj in enumerate(t):
        u = (np.longdouble(tj) - tp.astype(np.longdouble)) * invP.astype(np.longdouble)
        frac_t = (u - np.rint(u)).astype(np.longdouble)
        u64 = (tj - tp) * invP
        frac = u64 - np.rint(u64)
        sEc, cEc, invDc, _ = cold(frac, e, rng)
        # 80-bit Newton from the cold solution, on the exact M of the float64 inputs
        Ml = np.longdouble(TWO_PI) * frac_t + (np.longdouble(np.pi) * 2 - np.longdouble(TWO_PI)) * frac_t
        Et = np.arctan2(sEc, cEc).astype(np.longdouble); el_ = e.astype(np.longdouble)
        for _ in range(3):
            Et = Et - (Et - el_ * np.sin(Et) - Ml) / (1 - el_ * np.cos(Et))
        start = (j % rows_per_wave) == 0
        if start or state is None:
            sE, cE, invD = sEc, cEc, invDc
            chain[:] = 0
            used_warm = np.zeros(W, bool)
        else:
            dM = TWO_PI * (tj - t[j - 1]) * invP
            sEw, cEw, invDw, d5w, dE, x = warm(state[0], state[1], state[2], dM, e, rng, order)
            # a-priori bound, per lane: first-order step x = ΔM/D and the predictor's relative error scale
            bound = np.abs(x) ** 3 * state[2] ** 2 if order == 2 else np.abs(x) ** 4 * state[2] ** 3
            ok_lane = (np.abs(x) < XMAX) & (bound < tol)
            okp = np.concatenate([ok_lane, np.ones(pad, bool)]).reshape(tiles, 64)
            wave_ok = okp.all(axis=1)
            lane_wave_ok = np.repeat(wave_ok, 64)[:W]
            np.add.at(lane_fail_by_e, ebin, ~ok_lane); np.add.at(lane_rows_by_e, ebin, 1)
            fallback_rows += (~wave_ok).sum(); warm_rows += wave_ok.sum()
            sE = np.where(lane_wave_ok, sEw, sEc); cE = np.where(lane_wave_ok, cEw, cEc); invD = np.where(lane_wave_ok, invDw, invDc)
            chain = np.where(lane_wave_ok, chain + 1, 0)
            used_warm = lane_wave_ok
        state = (sE, cE, invD)
        D = 1 - e * np.cos(Et.astype(np.float64))
        es = np.abs(sE - np.sin(Et).astype(np.float64)); ec = np.abs(cE - np.cos(Et).astype(np.float64))
        ew = np.maximum(es, ec) * D
        ecold = np.maximum(np.abs(sEc - np.sin(Et).astype(np.float64)), np.abs(cEc - np.cos(Et).astype(np.float64))) * D
        k = np.argmax(ew)
        if ew[k] > err_w_max:
            err_w_max = ew[k]; chain_len_at_worst = chain[k]; worst = (e[k], a[k], j)
        err_c_max = max(err_c_max, ecold.max())
        errs.append(ew[used_warm].max() if used_warm.any() else 0.0)
    print(f"rows_per_wave {rows_per_wave}  tol {tol:g}  order {order}  rows {n_rows}")
    print(f"D-weighted max error of (sinE, cosE): warm chain {err_w_max:.3e} (chain length there {chain_len_at_worst}, e={worst[0]:.4f} a={worst[1]:.2f} row {worst[2]})"
          f" | cold {err_c_max:.3e}")
    tot = fallback_rows + warm_rows
    f = fallback_rows / tot
    print(f"wave-rows: {tot}  fallback {fallback_rows} = {f:.3f}  (+ 1/{rows_per_wave} cold chunk starts)")
    print("lane-level failure rate by eccentricity bin:", " ".join(f"{x:.4f}" for x in lane_fail_by_e / np.maximum(lane_rows_by_e, 1)))
    # issue model (profiles/pmc_traffic.json): per row 216.7 ns now; cold-only part: frac 4 + f0 2 FP64, starter 24 FP32 + 4 trans, table 4 FP32 + 10 FP64
    cold_part = 2.15 * (4 + 2 + 10) + 1.04 * 28 + 3.4 * 4
    warm_part = 2.15 * (1 + 5 + 14 + 4 + (4 if order == 3 else 0))
    check = 2.15 * 1 + 1.04 * 4          # bound: cvt + a few FP32 + compare/ballot
    fstart = 1.0 / rows_per_wave
    pw = (1 - f) * (1 - fstart)
    t_now = 216.7
    t_new = t_now - cold_part + check + pw * warm_part + (1 - pw) * (cold_part + 0.0)
    print(f"issue model: row {t_now:.1f} ns -> {t_new:.1f} ns ({100 * (t_new / t_now - 1):+.1f} %) at warm fraction {pw:.3f}; all-warm limit "
          f"{t_now - cold_part + check + warm_part:.1f} ns ({100 * ((t_now - cold_part + check + warm_part) / t_now - 1):+.1f} %)")


if __name__ == "__main__":
    main()
