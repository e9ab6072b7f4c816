"""Randomised sweeps of round 5's two new code paths against the oracle (stand-alone, like tests/stress_parity.py; run through gpurun):
  warm   single-planet systems on DENSE tables (cadence 0.2 … 2 days, 40 … 700 rows; RA/Dec, sep/PA, cor, absolute / relative RV; with and
         without per-walker nuisances): k_main's warm-started row loop with its wave-uniform fallback — periods drawn so that most waves pass
         the entry test, eccentricities to 0.98, periastron passages inside the table, now and then a walker too fast for the cadence (its wave
         runs cold), invalid walkers; and the same system with OCTO_WARM=0: warm and cold must agree to 1e-11 / 1e-9.
  many   systems of 4 … 8 planets (k_mainp, k_finishp): relative astrometry on some planets, absolute and relative RV, nuisances, ragged batches.
    python tests/stress_round5.py <n_systems> <seed>"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gpu_binding as gb
import oracle_binding as ob
import synth
import stress_parity as sp

ob.load_oracle()


def errs(res, ref):
    ll, g, gn = res; ll_o, g_o, gn_o = ref
    ok = np.isfinite(ll_o)
    same = np.array_equal(np.isfinite(ll), ok) and np.all(np.isneginf(ll[~ok])) and np.all(g[:, ~ok] == 0.0)
    if not ok.any():
        return same, 0.0, 0.0
    e_ll = np.max(np.abs(ll[ok] - ll_o[ok]) / np.maximum(1, np.abs(ll_o[ok])))
    G = np.concatenate([g] + ([gn] if gn is not None else [])); Go = np.concatenate([g_o] + ([gn_o] if gn_o is not None else []))
    scale = np.maximum(np.abs(Go[:, ok]).max(axis=1, keepdims=True), 1e-10 * np.abs(Go[:, ok]).max())
    return same, e_ll, np.max(np.abs(G[:, ok] - Go[:, ok]) / scale)


def warm_system(rng):
    W = int(rng.choice([64, 65, 130, 200, 333, 500]))
    cad = float(rng.choice([0.2, 0.5, 1.0, 2.0]))
    n = int(rng.integers(40, 700))
    t = 50000.0 + cad * np.arange(n) + (rng.uniform(0, 0.3 * cad, n) if rng.random() < 0.5 else 0.0)
    t = np.sort(t)
    a_lo = (2 * np.pi * 1.3 * cad / 0.0232 / 365.25) ** (2 / 3) * 1.3      # the shortest period the entry test admits for this cadence, with margin
    el = synth.draw_walkers(rng, W, a_lo, a_lo * 40, with_mass=True)
    el[1] = rng.uniform(0, 0.98, W)
    k = W // 5
    el[5, :k] = t[0] + rng.uniform(0, t[-1] - t[0], k)                        # periastron inside the table
    if rng.random() < 0.3:
        el[0, int(rng.integers(0, W))] = a_lo * 0.2                            # too fast for the cadence: its wave stays cold
    if rng.random() < 0.4:
        for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (6, -1.0), (5, np.nan))):
            el[row, w_bad] = val
    obs = []
    kinds = rng.choice(["radec", "cor", "seppa", "rvabs", "rvrel"], size=int(rng.integers(1, 4)), replace=False)
    for kd in kinds:
        tt = t if rng.random() < 0.7 else t[:: int(rng.integers(2, 4))]
        m = tt.size
        ra, dec = rng.normal(0, 300, m), rng.normal(0, 300, m)
        if kd == "radec": obs.append(dict(kind=0, planet=0, epoch=tt, y1=ra, y2=dec, s1=rng.uniform(3, 12, m), s2=rng.uniform(3, 12, m), cor=None))
        if kd == "cor": obs.append(dict(kind=0, planet=0, epoch=tt, y1=ra, y2=dec, s1=rng.uniform(3, 12, m), s2=rng.uniform(3, 12, m), cor=rng.uniform(-0.8, 0.8, m)))
        if kd == "seppa": obs.append(dict(kind=1, planet=0, epoch=tt, y1=np.arctan2(ra, dec), y2=np.hypot(ra, dec), s1=np.full(m, 0.03), s2=rng.uniform(3, 12, m), cor=None))
        if kd == "rvabs": obs.append(dict(kind=2, planet=-1, epoch=tt, y1=rng.normal(0, 30, m), y2=None, s1=rng.uniform(1, 8, m), s2=None, cor=None,
                                          extra=(tt - 50100.0) / 100.0 if rng.random() < 0.5 else None))
        if kd == "rvrel": obs.append(dict(kind=4, planet=0, epoch=tt, y1=rng.normal(0, 500, m), y2=None, s1=rng.uniform(20, 80, m), s2=None, cor=None))
    nuis = np.zeros((len(obs) * 3, W))
    for io, o in enumerate(obs):
        if o["kind"] in (0, 1):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.02, W)
            nuis[io * 3, : W // 5] = 0.0
        else:
            nuis[io * 3] = rng.normal(0, 10, W); nuis[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W)); nuis[io * 3 + 2] = rng.normal(0, 2, W)
    return obs, [dict(orbit_kind=0, has_mass=True)], el, (nuis if rng.random() < 0.5 else None)


def many_system(rng):
    P = int(rng.integers(4, 9))
    W = int(rng.choice([1, 7, 64, 65, 130, 200]))
    planets = [dict(orbit_kind=0, has_mass=True) for _ in range(P)]
    elems = np.concatenate([sp.planet_elems(rng, W, 0, 1.5 + 4 * i, 4.5 + 4 * i) for i in range(P)])
    obs = []
    for ip in range(P):
        if rng.random() < 0.6:
            n = int(rng.integers(1, 110)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
            seppa = rng.random() < 0.3
            ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
            obs.append(dict(kind=1 if seppa else 0, planet=ip, epoch=ep, y1=np.arctan2(ra, dec) if seppa else ra, y2=np.hypot(ra, dec) if seppa else dec,
                            s1=np.full(n, 0.03) if seppa else rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=rng.uniform(-0.8, 0.8, n) if (not seppa and rng.random() < 0.4) else None))
        if rng.random() < 0.25:
            n = int(rng.integers(1, 90)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
            obs.append(dict(kind=4, planet=ip, epoch=ep, y1=rng.normal(0, 500, n), y2=None, s1=rng.uniform(20, 80, n), s2=None, cor=None))
    if rng.random() < 0.5 or not obs:
        n = int(rng.integers(1, 150)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs.append(dict(kind=2, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None,
                        extra=(ep - 52000.0) / 1000.0 if rng.random() < 0.5 else None))
    if rng.random() < 0.3:      # round 6: marginalised RV on the planet-per-wave kernels (beyond four planets; four planets keep k_main<4> for it)
        n = int(rng.integers(2, 120)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs.append(dict(kind=3, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None,
                        extra=(ep - 52000.0) / 1000.0 if rng.random() < 0.5 else None))
    if rng.random() < 0.3:      # round 6, late: the O'Neil prior beyond four planets (four planets: k_main<4>)
        ip = int(rng.integers(0, P)); n = int(rng.integers(2, 60)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
        seppa = rng.random() < 0.4
        ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
        obs.append(dict(kind=6 if seppa else 5, planet=ip, epoch=ep, y1=np.arctan2(ra, dec) if seppa else ra, y2=np.hypot(ra, dec) if seppa else dec,
                        s1=np.full(n, 0.03) if seppa else rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=None))
    hgca = rng.random() < 0.3      # round 6, late: an HGCA table beyond four planets (k_hgcap -> k_finishp); four planets: k_hgca<4> -> k_finish<4>
    if hgca:
        N = int(rng.integers(1, 4))
        rows = []
        for d in np.linspace(-700, 700, N): rows += [(48348.0 + d, 0, 0), (48414.0 + d, 1, 0)]
        for d in np.linspace(-500, 500, N): rows += [(57408.0 + d, 0, 1), (57470.0 + d, 1, 1)]
        rows = np.array(rows)
        obs.append(dict(kind=7, planet=-1, epoch=rows[:, 0], y1=rows[:, 1], y2=rows[:, 2], s1=None, s2=None, cor=None, extra=sp.HG))
    nuis = np.zeros((len(obs) * 3, W))
    for io, o in enumerate(obs):
        if o["kind"] in (0, 1, 5, 6):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.02, W)
        elif o["kind"] == 7:
            nuis[io * 3] = rng.normal(4.3, 0.3, W); nuis[io * 3 + 1] = rng.normal(-2.0, 0.3, W)
        else:
            nuis[io * 3] = rng.normal(0, 10, W); nuis[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W)); nuis[io * 3 + 2] = rng.normal(0, 2, W)
    if W >= 7 and rng.random() < 0.5:
        for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (6, -1.0), (5, np.nan))):
            elems[int(rng.integers(0, P)) * 9 + row, w_bad] = val
    return obs, planets, elems, (nuis if (hgca or rng.random() < 0.6) else None)


def main():
    n_sys = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = dict(warm_ll=0.0, warm_g=0.0, wc_ll=0.0, wc_g=0.0, many_ll=0.0, many_g=0.0)
    fails = 0; n_warm_ran = 0
    for i in range(n_sys):
        obs, planets, el, nz = warm_system(rng)
        os.environ["OCTO_WARM"] = "1"; warm = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0)
        warm_f = gb.gpu_eval(obs, planets, el, nz, grad=False, small_batch=0)
        os.environ["OCTO_WARM"] = "0"; cold = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0)
        os.environ.pop("OCTO_WARM")
        ref = ob.oracle_eval(obs, planets, el, nz, grad=True, n_threads=0)
        same, e_ll, e_g = errs(warm, ref)
        _, c_ll, c_g = errs(warm, tuple(x if x is not None else None for x in cold)) if np.isfinite(cold[0]).any() else (True, 0.0, 0.0)
        n_warm_ran += int(not (np.array_equal(warm[0], cold[0]) and np.array_equal(warm[1], cold[1], equal_nan=True)))
        bad = (not same) or (not np.array_equal(warm[0], warm_f[0])) or e_ll > 1e-9 or e_g > 2e-8 or c_ll > 1e-11 or c_g > 1e-9
        fails += int(bad)
        if bad: print(f"FAIL warm system {i}: same={same} fwd==grad {np.array_equal(warm[0], warm_f[0])} vs oracle {e_ll:.2e} {e_g:.2e} vs cold {c_ll:.2e} {c_g:.2e}", flush=True)
        worst["warm_ll"] = max(worst["warm_ll"], e_ll); worst["warm_g"] = max(worst["warm_g"], e_g); worst["wc_ll"] = max(worst["wc_ll"], c_ll); worst["wc_g"] = max(worst["wc_g"], c_g)
        obs, planets, el, nz = many_system(rng)
        res = gb.gpu_eval(obs, planets, el, nz, grad=True)
        res_f = gb.gpu_eval(obs, planets, el, nz, grad=False)
        ref = ob.oracle_eval(obs, planets, el, nz, grad=True, n_threads=0)
        same, e_ll, e_g = errs(res, ref)
        bad = (not same) or (not np.array_equal(res[0], res_f[0])) or e_ll > 1e-9 or e_g > 2e-8
        fails += int(bad)
        if bad: print(f"FAIL many-planet system {i} (P = {len(planets)}, W = {el.shape[1]}): same={same} vs oracle {e_ll:.2e} {e_g:.2e}", flush=True)
        worst["many_ll"] = max(worst["many_ll"], e_ll); worst["many_g"] = max(worst["many_g"], e_g)
    print(f"{n_sys} dense single-planet systems (warm loop taken by some wave in {n_warm_ran}) + {n_sys} systems of 4-8 planets: {fails} failures; worst vs oracle: "
          f"warm ll {worst['warm_ll']:.2e} grad {worst['warm_g']:.2e} | warm vs cold ll {worst['wc_ll']:.2e} grad {worst['wc_g']:.2e} | "
          f"4-8 planets ll {worst['many_ll']:.2e} grad {worst['many_g']:.2e}")


if __name__ == "__main__":
    main()
