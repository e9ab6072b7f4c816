"""Randomised GPU-vs-oracle sweep over systems the unit tests do not enumerate: 1-3 planets on mixed bases (Campbell,
Thiele-Innes, radial-velocity-only), random subsets of every observation kind incl. the O'Neil wrapper and HGCA, with and
without the nuisance block, random table and batch sizes (ragged tiles). Run on a GPU box:
    python tests/stress_parity.py [n_systems] [seed]
Prints the worst errors; exits non-zero if a case breaks the bars of tests/test_gpu_parity.py. tests/test_sweeps_gpu.py runs a
fixed-seed slice of the same sweep under pytest -m gpu."""
import os
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob, gpu_binding as gb

HG = np.array([4.71, -1.86, 0.61, 0.49, 0.21, 4.352, -2.013, 0.031, 0.024, -0.12, 4.61, -1.72, 0.052, 0.041, 0.33])


def planet_elems(rng, W, kind, a_lo, a_hi):
    a, e = rng.uniform(a_lo, a_hi, W), rng.uniform(0, 0.85, W)
    inc, w, O = np.arccos(rng.uniform(-1, 1, W)), rng.uniform(-7, 7, W), rng.uniform(-7, 7, W)
    tp, M, plx, mass = 50000 + rng.uniform(-3000, 3000, W), rng.uniform(0.8, 1.6, W), rng.uniform(20, 60, W), rng.uniform(1, 40, W)
    if kind == 2:
        T = a * plx
        cO, sO, cw, sw, ci = np.cos(O), np.sin(O), np.cos(w), np.sin(w), np.cos(inc)
        return np.stack([T * (cO * cw - sO * sw * ci), e, T * (sO * cw + cO * sw * ci), T * (-cO * sw - sO * cw * ci), T * (-sO * sw + cO * cw * ci), tp, M, plx, mass])
    return np.stack([a, e, inc, w, O, tp, M, plx, mass])


SCALE = 1      # argv[3]: multiplies table sizes and batch sizes (partition planner at larger shapes)


def random_system(rng, invalid=True, P=None, W=None):
    P = int(rng.integers(1, 1 + int(os.environ.get("OCTO_TEST_MAX_P", "3")))) if P is None else P      # (OCTO_TEST_MAX_P=4: four-planet systems too)
    W = int(rng.choice([1, 7, 64, 65, 130, 200, 333])) * (1 if SCALE == 1 else int(rng.integers(1, SCALE + 1))) if W is None else W
    kinds_pl = [int(rng.choice([0, 0, 2, 1])) for _ in range(P)]
    has_rv_basis = any(k == 1 for k in kinds_pl)
    has_ti = any(k == 2 for k in kinds_pl)
    planets = [dict(orbit_kind=k, has_mass=True) for k in kinds_pl]
    elems = np.concatenate([planet_elems(rng, W, k, 2 + 6 * i, 6 + 6 * i) for i, k in enumerate(kinds_pl)])
    obs = []
    for ip, k in enumerate(kinds_pl):
        n = int(rng.integers(1, 120 * SCALE)) if rng.random() > 0.08 else 0      # now and then an empty table
        ep = np.sort(50000 + rng.uniform(0, 4000, n))
        if k != 1:
            for _ in range(int(rng.integers(0, 3))):
                seppa = rng.random() < 0.35
                cor = rng.uniform(-0.8, 0.8, n) if rng.random() < 0.4 else None
                ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
                kind = (1 if seppa else 0) + (5 if rng.random() < 0.3 else 0)
                y1, y2 = (np.arctan2(ra, dec), np.hypot(ra, dec)) if seppa else (ra, dec)
                s1 = np.full(n, 0.03) if seppa else rng.uniform(3, 12, n)
                obs.append(dict(kind=kind, planet=ip, epoch=ep, y1=y1, y2=y2, s1=s1, s2=rng.uniform(3, 12, n), cor=cor))
        if not has_ti and rng.random() < 0.5:
            obs.append(dict(kind=4, planet=ip, epoch=ep, y1=rng.normal(0, 500, n), y2=None, s1=rng.uniform(20, 80, n), s2=None, cor=None))
    if not has_ti:
        for kind in (2, 3):
            if rng.random() < 0.4:
                n = int(rng.integers(1, 150 * SCALE)); ep = np.sort(50000 + rng.uniform(0, 4000, n))
                obs.append(dict(kind=kind, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None))
    hgca = (not has_rv_basis or any(k != 1 for k in kinds_pl)) and rng.random() < 0.4
    if hgca:
        N = int(rng.integers(1, 4))
        rows = []
        for d in np.linspace(-700, 700, N): rows += [(48348.0 + d, 0, 0), (48414.0 + d, 1, 0)]
        for d in np.linspace(-500, 500, N): rows += [(57408.0 + d, 0, 1), (57470.0 + d, 1, 1)]
        rows = np.array(rows)
        obs.append(dict(kind=7, planet=-1, epoch=rows[:, 0], y1=rows[:, 1], y2=rows[:, 2], s1=None, s2=None, cor=None, extra=HG))
    if not obs or P * 9 + len(obs) * 3 > 64:      # (the checker carries at most 64 forward-mode partials: four planets with ten tables exceed it)
        return None
    nuis = np.zeros((len(obs) * 3, W))
    for io, o in enumerate(obs):
        if o["kind"] in (0, 1, 5, 6):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.02, W)
            nuis[io * 3, : W // 5] = 0.0
        elif o["kind"] == 7:
            nuis[io * 3] = rng.normal(4.3, 0.3, W); nuis[io * 3 + 1] = rng.normal(-2.0, 0.3, W)
        else:
            nuis[io * 3] = rng.normal(0, 10, W); nuis[io * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    if invalid and W >= 7 and rng.random() < 0.5:      # a few walkers outside the domain: -Inf, zero gradient, nothing else disturbed
        for w_bad, (row, val) in zip(rng.choice(W, 3, replace=False), ((1, 1.2), (6, -1.0), (5, np.nan))):
            elems[int(rng.integers(0, P)) * 9 + row, w_bad] = val
    use_nuis = hgca or rng.random() < 0.6
    # RV trend_function (include/octofitter_hip.h: OCTO_NU_RV_TREND): about half of the RV tables get a basis column (linear or quadratic
    # in the epoch) and every RV table a non-zero third nuisance row — ignored, with zero gradient, where there is no column. Drawn from a
    # generator of its own, seeded by the elements, so that the systems of a given seed are the ones earlier rounds swept.
    import zlib
    rng_t = np.random.default_rng(zlib.crc32(np.ascontiguousarray(elems).tobytes()))
    for io, o in enumerate(obs):
        if o["kind"] in (2, 3, 4):
            n = len(o["epoch"])
            if rng_t.random() < 0.5:
                o["extra"] = (o["epoch"] - 52000.0) / 100.0 if rng_t.random() < 0.7 else ((o["epoch"] - 52000.0) / 1000.0) ** 2
                if n == 0:
                    o["extra"] = None
            nuis[io * 3 + 2] = rng_t.normal(0, 2.0, W)
    return obs, planets, elems, (nuis if use_nuis else None)


def check_system(sysm):
    """GPU vs oracle for one random system: returns (ok, e_ll, e_grad, loose) with the bars of tests/test_gpu_parity.py."""
    obs, planets, elems, nuis = sysm
    ll, g, gn = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    llf, _, _ = gb.gpu_eval(obs, planets, elems, nuis, grad=False)
    ll_o, g_o, gn_o = ob.oracle_eval(obs, planets, elems, nuis, grad=True, n_threads=0)
    ok = np.isfinite(ll_o)
    same = np.array_equal(ll, llf) and np.array_equal(np.isfinite(ll), ok) and np.all(np.isneginf(ll[~ok])) and np.all(g[:, ~ok] == 0.0)
    e_ll = np.max(np.abs(ll[ok] - ll_o[ok]) / np.maximum(1, np.abs(ll_o[ok]))) if ok.any() else 0.0
    G = np.concatenate([g] + ([gn] if gn is not None else [])); Go = np.concatenate([g_o] + ([gn_o] if gn_o is not None else []))
    # per-input scale, floored: an input the likelihood does not depend on has a true gradient of 0 ± rounding noise
    scale = np.maximum(np.abs(Go[:, ok]).max(axis=1, keepdims=True), 1e-10 * np.abs(Go[:, ok]).max()) if ok.any() else 1.0
    e_g = np.max(np.abs(G[:, ok] - Go[:, ok]) / np.maximum(scale, 1e-300)) if ok.any() else 0.0
    marg = any(o["kind"] == 3 for o in obs)
    ti = any(p["orbit_kind"] == 2 for p in planets)
    # marginalised RV: cancellation in the reference's formula (rv-absolute-margin.jl:181). Thiele-Innes: a = α/plx with
    # α² = u + √((u+v)(u−v)) loses digits in u − v near face-on orbits IN THE REFERENCE'S ARITHMETIC, which the C oracle follows; the
    # device uses the cancellation-free form. A Thiele-Innes case beyond the bar is therefore re-judged against the
    # 60-digit oracle at its worst walker: the device must be within 1e-9 of it there.
    lim_ll, lim_g = (1e-9, 1e-8) if marg else ((1e-10, 1e-7) if ti else (1e-12, 1e-9))
    good = bool(same and e_ll < lim_ll and e_g < lim_g)
    if not good and ti and not marg and same and ok.any():
        err = np.abs(G - Go) / np.maximum(scale, 1e-300); err[:, ~ok] = 0.0
        r, w = np.unravel_index(np.argmax(err), err.shape)
        ell = np.abs(ll - ll_o) / np.maximum(1, np.abs(ll_o)); ell[~ok] = 0.0
        wl = int(np.argmax(ell))
        ll_m, gm = mp_value_and_gradient(obs, planets, elems, nuis, int(w))
        ll_ml = ll_m if wl == w else mp_value_and_gradient(obs, planets, elems, nuis, wl, grad=False)[0]
        e_dev, e_ora = abs(G[r, w] - gm[r]) / scale[r, 0], abs(Go[r, w] - gm[r]) / scale[r, 0]
        l_dev, l_ora = abs(ll[wl] - ll_ml) / max(1.0, abs(ll_ml)), abs(ll_o[wl] - ll_ml) / max(1.0, abs(ll_ml))
        print(f"[Thiele-Innes, against 60 digits: gradient (walker {w}, input {r}) device {e_dev:.1e}, reference-order oracle {e_ora:.1e}; "
              f"ll (walker {wl}) device {l_dev:.1e}, oracle {l_ora:.1e}] ", end="")
        good = bool(e_dev < 1e-9 and l_dev < 1e-12)
    return good, float(e_ll), float(e_g), bool(marg or ti)


def mp_value_and_gradient(obs, planets, elems, nuis, w, grad=True):
    """ll and ∂ll/∂(elements, nuisances) of walker w from the independent 60-digit oracle (oracle/mp_oracle.py)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import mpmath as mp, mp_oracle as mo
    KN = {0: "ASTROM_RADEC", 1: "ASTROM_SEPPA", 2: "RV_ABS", 3: "RV_ABS_MARG", 4: "RV_REL", 5: "ONEIL_RADEC", 6: "ONEIL_SEPPA", 7: "HGCA"}
    fl = lambda x: None if x is None else list(map(float, x))
    obs_m = [dict(kind=KN[o["kind"]], planet=o["planet"], epoch=fl(o["epoch"]), y1=fl(o["y1"]), y2=fl(o["y2"]), s1=fl(o["s1"]), s2=fl(o["s2"]),
                  cor=fl(o.get("cor")), extra=fl(o.get("extra"))) for o in obs]
    P = len(planets)
    el = [[mp.mpf(float(elems[p * 9 + k, w])) for k in range(9)] for p in range(P)]
    nu = None if nuis is None else [[mp.mpf(float(nuis[o * 3 + k, w])) for k in range(3)] for o in range(len(obs))]
    if not grad:
        return float(mo.ln_like(mo.DEFAULT_CONSTS, planets, obs_m, el, nu)), None
    f0, g_el, g_nu, _, _ = mo.ln_like_and_grad(mo.DEFAULT_CONSTS, planets, obs_m, el, nu, with_scale=True)
    out = [float(g_el[p][k]) for p in range(P) for k in range(9)]
    if nuis is not None:
        out += [float(g_nu[o][k]) for o in range(len(obs)) for k in range(3)]
    return float(f0), np.array(out)


def draw_system(rng, invalid=True, P=None, W=None):
    sysm = None
    while sysm is None:
        sysm = random_system(rng, invalid=invalid, P=P, W=W)
    return sysm


def describe(sysm):
    obs, planets, elems, nuis = sysm
    return (f"P={len(planets)} bases={[p['orbit_kind'] for p in planets]} kinds={[o['kind'] for o in obs]} "
            f"rows={[len(o['epoch']) for o in obs]} W={elems.shape[1]} nuis={nuis is not None}")


def main():
    global SCALE
    SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    n_sys = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst_ll = worst_g = 0.0
    bad = 0
    for k in range(n_sys):
        sysm = draw_system(rng)
        print(f"{k:3d} {describe(sysm)}: ", end="", flush=True)
        good, e_ll, e_g, loose = check_system(sysm)
        bad += not good
        if not loose:
            worst_ll, worst_g = max(worst_ll, e_ll), max(worst_g, e_g)
        print(f"ll {e_ll:.1e} grad/scale {e_g:.1e}{'' if good else '   <-- FAIL'}", flush=True)
    print(f"worst (no marginalised RV, no Thiele-Innes): ll {worst_ll:.2e} grad {worst_g:.2e}; failures {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
