"""Randomised sweeps of round 6's new code paths against the oracle (stand-alone, like tests/stress_round5.py; run through gpurun):
  gaps   single-planet systems on tables WITH GAPS and mixed cadences (nightly runs of 1-6 exposures, 5-120 nights per season, random steps, now and then
         duplicate and unsorted epochs): the warm loop's per-wave step bound (the table's ladder) and per-row key, in the prefetching kernels and in the
         nuisance kernels with RV / sep-PA rows (plain loads); walkers from a WIDE prior (some too fast for every rung, some only for the long steps),
         eccentricities to 0.98, invalid walkers; each system four ways — as drawn, tiles sorted (OCTO_OPT_TILE_SORT = 1), cold (OCTO_OPT_WARM_START = 0)
         and batch-invariant — all against the oracle, warm against cold to 1e-11 / 1e-9, forward-only == the value returned with a gradient;
  two    two- and three-planet systems (RA/Dec or sep/PA on any planet, absolute / relative RV, nuisances) on dense tables with moderate outer eccentricities
         (the last-planet-always-warm loop) and as drawn, sorted and unsorted;
  short  one-table systems of 20-120 rows x 600-3000 walkers: the one-task launch that finishes inside k_main, bit-identical to OCTO_FIN_FUSED=0.
    python tests/stress_round6.py <n_systems> <seed>"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import gpu_binding as gb
import oracle_binding as ob
import synth
import stress_parity as sp
from stress_round5 import errs

capi = gb.capi
ob.load_oracle()


def gappy_table(rng):
    style = rng.choice(["runs", "random", "uniform+gap"])
    if style == "runs":
        t = synth.gappy_epochs(int(rng.integers(60, 900)), per_night=int(rng.integers(1, 7)), nights_per_season=int(rng.integers(5, 120)),
                               intra_night=float(rng.choice([0.01, 0.02, 0.1])), season=float(rng.choice([180.0, 365.25])), rng=rng)
    elif style == "random":
        t = np.sort(50000.0 + rng.uniform(0, rng.choice([300.0, 2000.0]), int(rng.integers(50, 700))))
    else:
        cad = float(rng.choice([0.25, 1.0, 3.0]))
        t = 50000.0 + cad * np.arange(int(rng.integers(80, 800)))
        t[t.size // 2:] += rng.uniform(50, 900)
    if rng.random() < 0.15:
        k = int(rng.integers(1, t.size - 3)); t[k + 1] = t[k]                     # a duplicate epoch
    if rng.random() < 0.1:
        k = int(rng.integers(1, t.size - 3)); t[k], t[k + 1] = t[k + 1], t[k]     # an unsorted pair
    return t


def gap_system(rng):
    W = int(rng.choice([64, 130, 333, 700, 1500]))
    t = gappy_table(rng)
    el = synth.draw_walkers(rng, W, float(rng.choice([0.05, 0.3, 1.0])), 60.0, with_mass=True)
    el[1] = rng.uniform(0, 0.98, W)
    k = W // 5
    el[5, :k] = t[0] + rng.uniform(0, t[-1] - t[0], k)
    if rng.random() < 0.4:
        for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (6, -1.0), (5, np.nan))):
            el[row, w_bad] = val
    obs = []
    for kd in rng.choice(["radec", "cor", "seppa", "rvabs", "rvrel"], size=int(rng.integers(1, 4)), replace=False):
        tt = t if rng.random() < 0.7 else gappy_table(rng)
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
        else:
            nuis[io * 3] = rng.normal(0, 10, W); nuis[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W)); nuis[io * 3 + 2] = rng.normal(0, 2, W)
    return obs, [dict(orbit_kind=0, has_mass=True)], el, (nuis if rng.random() < 0.5 else None)


def two_system(rng):
    W = int(rng.choice([64, 200, 700, 2300]))
    cad = float(rng.choice([0.5, 2.0, 4.0]))
    n = int(rng.integers(60, 500))
    t = 50000.0 + cad * np.arange(n)
    if rng.random() < 0.3:                        # gaps: rows whose own step exceeds the wave's bound start cold (slot 7 of the record)
        keep = np.ones(n, bool)
        for _ in range(int(rng.integers(1, 4))):
            g0 = int(rng.integers(5, n - 5)); keep[g0:g0 + int(rng.integers(3, 40))] = False
        t = t[keep]; n = t.size
    P = 3 if rng.random() < 0.4 else 2            # (three planets: the kind sets without sep/PA rows carry the last planet's warm start too)
    e1 = synth.draw_walkers(rng, W, 1.0, 5.0, with_mass=True); e2 = synth.draw_walkers(rng, W, 15.0 if P == 3 else 8.0, 40.0, with_mass=True)
    e2[6] = e1[6]; e2[7] = e1[7]
    em = None
    if P == 3:
        em = synth.draw_walkers(rng, W, 6.0, 12.0, with_mass=True); em[6] = e1[6]; em[7] = e1[7]
    if rng.random() < 0.6:
        e2[1] *= rng.uniform(0.3, 0.9)            # moderate outer eccentricities: (nearly) every row of the last planet warm; as drawn (e up to 0.95): cold rows near periastron
    el = np.concatenate([e1, e2] if P == 2 else [e1, em, e2])
    if rng.random() < 0.4:
        for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (9 * (P - 1) + 6, -1.0), (9 * (P - 1) + 5, np.nan))):
            el[row, w_bad] = val
    ip = int(rng.integers(0, P))
    ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
    obs = [dict(kind=0, planet=ip, epoch=t, y1=ra, y2=dec, s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=rng.uniform(-0.7, 0.7, n) if rng.random() < 0.3 else None)
           if rng.random() < 0.7 else dict(kind=1, planet=ip, epoch=t, y1=np.arctan2(ra, dec), y2=np.hypot(ra, dec), s1=np.full(n, 0.03), s2=rng.uniform(3, 12, n), cor=None)]
    if rng.random() < 0.7:
        obs.append(dict(kind=2, planet=-1, epoch=t + 0.3, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None))
    if rng.random() < 0.3:
        obs.append(dict(kind=4, planet=(ip + 1) % P, epoch=t[::2], y1=rng.normal(0, 500, t[::2].size), y2=None, s1=rng.uniform(20, 80, t[::2].size), s2=None, cor=None))
    nuis = np.zeros((len(obs) * 3, W))
    for io, o in enumerate(obs):
        if o["kind"] in (0, 1):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.02, W)
        else:
            nuis[io * 3] = rng.normal(0, 10, W); nuis[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    return obs, [dict(orbit_kind=0, has_mass=True)] * P, el, (nuis if rng.random() < 0.6 else None)


def short_system(rng):
    W = int(rng.integers(600, 3000)); n = int(rng.integers(20, 120))
    t = np.sort(50000.0 + rng.uniform(0, 3000, n))
    el = synth.draw_walkers(rng, W)
    for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (6, -1.0), (5, np.nan))):
        el[row, w_bad] = val
    kind = rng.choice(["radec", "cor", "rv"])
    if kind == "rv":
        el[8] = rng.uniform(1, 20, W)
        obs = [dict(kind=2, planet=-1, epoch=t, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None)]
        nuis = np.stack([rng.normal(0, 10, W), np.exp(rng.uniform(np.log(0.1), np.log(10), W)), np.zeros(W)])
    else:
        obs = [dict(kind=0, planet=0, epoch=t, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n),
                    cor=rng.uniform(-0.7, 0.7, n) if kind == "cor" else None)]
        nuis = np.stack([rng.uniform(0, 4, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)])
    return obs, [dict(orbit_kind=0, has_mass=kind == "rv")], el, (nuis if rng.random() < 0.5 else None)


def main():
    n_sys = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst = dict(gap_ll=0.0, gap_g=0.0, gc_ll=0.0, gc_g=0.0, two_ll=0.0, two_g=0.0, tc_ll=0.0, tc_g=0.0, short_ll=0.0, short_g=0.0)
    fails = 0; n_warm = 0; n_sorted_differs = 0; n_last = 0
    ev = lambda obs, pl, el, nz, grad, opts: gb.gpu_eval(obs, pl, el, nz, grad=grad, small_batch=0, options=opts)
    SORT, COLD, INV = {capi.OPT_TILE_SORT: 1, capi.OPT_TILE_MIN_WALKERS: 64}, {capi.OPT_WARM_START: 0, capi.OPT_TILE_SORT: 0}, {capi.OPT_BATCH_INVARIANT: 1}
    for i in range(n_sys):
        for name, make in (("gap", gap_system), ("two", two_system)):
            obs, planets, el, nz = make(rng)
            ref = ob.oracle_eval(obs, planets, el, nz, grad=True, n_threads=0)
            drawn = ev(obs, planets, el, nz, True, {capi.OPT_TILE_SORT: 0}); fwd = ev(obs, planets, el, nz, False, {capi.OPT_TILE_SORT: 0})
            srt = ev(obs, planets, el, nz, True, SORT); cold = ev(obs, planets, el, nz, True, COLD); inv = ev(obs, planets, el, nz, True, INV)
            ran = not (np.array_equal(drawn[0], cold[0]) and np.array_equal(drawn[1], cold[1], equal_nan=True))
            n_warm += int(ran and name == "gap"); n_last += int(ran and name == "two")
            n_sorted_differs += int(not (np.array_equal(srt[0], drawn[0]) and np.array_equal(srt[1], drawn[1], equal_nan=True)))
            bad = not np.array_equal(drawn[0], fwd[0])
            for tag, res in (("drawn", drawn), ("sorted", srt), ("cold", cold), ("invariant", inv)):
                same, e_ll, e_g = errs(res, ref)
                worst[f"{name}_ll"] = max(worst[f"{name}_ll"], e_ll); worst[f"{name}_g"] = max(worst[f"{name}_g"], e_g)
                bad = bad or (not same) or e_ll > 1e-9 or e_g > 2e-8
            for res in (drawn, srt):
                _, c_ll, c_g = errs(res, cold) if np.isfinite(cold[0]).any() else (True, 0.0, 0.0)
                key = "gc" if name == "gap" else "tc"
                worst[f"{key}_ll"] = max(worst[f"{key}_ll"], c_ll); worst[f"{key}_g"] = max(worst[f"{key}_g"], c_g)
                bad = bad or c_ll > 1e-11 or c_g > 1e-9
            fails += int(bad)
            if bad: print(f"FAIL {name} system {i}: W = {el.shape[1]}, tables {[o['kind'] for o in obs]}, nuis {nz is not None}", flush=True)
        obs, planets, el, nz = short_system(rng)
        outs = []
        for ff in ("1", "0"):
            os.environ["OCTO_FIN_FUSED"] = ff
            outs.append(gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0))
            os.environ.pop("OCTO_FIN_FUSED")
        ref = ob.oracle_eval(obs, planets, el, nz, grad=True, n_threads=0)
        same, e_ll, e_g = errs(outs[0], ref)
        worst["short_ll"] = max(worst["short_ll"], e_ll); worst["short_g"] = max(worst["short_g"], e_g)
        bad = (not same) or e_ll > 1e-9 or e_g > 2e-8 or not all((x is None and y is None) or np.array_equal(x, y, equal_nan=True) for x, y in zip(*outs))
        fails += int(bad)
        if bad: print(f"FAIL short system {i}: W = {el.shape[1]}, rows {obs[0]['epoch'].size}, nuis {nz is not None}", flush=True)
    print(f"{n_sys} gappy single-planet systems (warm loop taken by some wave in {n_warm}) + {n_sys} two- and three-planet systems (the last planet warm on some row in {n_last}) x "
          f"{{as drawn, sorted, cold, batch-invariant}} (the sort changed some wave in {n_sorted_differs} of {2 * n_sys}) + {n_sys} one-task systems: {fails} failures; worst vs oracle: "
          f"gaps ll {worst['gap_ll']:.2e} grad {worst['gap_g']:.2e} | warm vs cold ll {worst['gc_ll']:.2e} grad {worst['gc_g']:.2e} | two planets ll {worst['two_ll']:.2e} "
          f"grad {worst['two_g']:.2e} | vs cold ll {worst['tc_ll']:.2e} grad {worst['tc_g']:.2e} | one-task ll {worst['short_ll']:.2e} grad {worst['short_g']:.2e}")


if __name__ == "__main__":
    main()
