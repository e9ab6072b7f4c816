"""
k_main's warm-started row loop (octo_device.h: KWarm): on tables whose cadence is dense against the walkers' periods
the previous row's (sin E, cos E, 1/(1 − e cos E)) replaces the Markley starter + table lookup, with a WAVE-UNIFORM fallback to the cold
starter. The root of Kepler's equation is unique (src/likelihoods/system.jl:250-269 solve every epoch independently), so the loop must
agree with the cold loop to rounding and with the oracle to the usual bars, on every single-planet kind set it is compiled for.

OCTO_WARM=0 in the environment of octo_ctx_create gives datasets that never take the warm loop: the same library, the cold loop.
"""
import os

import numpy as np
import pytest

import synth
from conftest import rel_err
from test_gpu_parity import _cmp_oracle, _gpu

pytestmark = pytest.mark.gpu


def _eval(gb, obs, planets, elems, nuis, grad, warm, env=None):
    old = {k: os.environ.get(k) for k in ("OCTO_WARM", *(env or {}))}
    os.environ["OCTO_WARM"] = "1" if warm else "0"
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        return gb.gpu_eval(obs, planets, elems, nuis, grad=grad, small_batch=0)      # the throughput kernels whatever the batch size
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same_bits(a, b):
    """Every output of two evaluations bit-identical: what a launch that took the cold loop looks like next to an OCTO_WARM=0 one. (The
    log-likelihood ALONE can coincide for a short table: the two loops differ by ~1e-16 per row, which the rounding of a sum of 1e6 may swallow;
    the 9+ gradient rows do not all coincide.)"""
    return all((x is None and y is None) or np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def _close(name, a, b, ll_tol=1e-12, g_tol=1e-10):
    ok = np.isfinite(b[0])
    assert np.array_equal(np.isfinite(a[0]), ok), name
    e = rel_err(a[0][ok], b[0][ok], 1.0)
    assert np.all(e < ll_tol), (name, "ll warm vs cold", e.max())
    for ga, gc in ((a[1], b[1]), (a[2], b[2])):
        if ga is None:
            continue
        sc = np.maximum(np.abs(gc[:, ok]).max(axis=1, keepdims=True), 1e-300)
        eg = np.abs(ga[:, ok] - gc[:, ok]) / sc
        assert np.all(eg < g_tol), (name, "gradient warm vs cold", eg.max())


def _dense_walkers(rng, W, a_lo, a_hi, e_hi=0.97):
    el = synth.draw_walkers(rng, W, a_lo, a_hi)
    el[1] = rng.uniform(0.0, e_hi, W)
    # a few walkers that pass periastron inside the table, some of them nearly parabolic
    k = min(W // 4, 40)
    el[5, :k] = 50000.0 + rng.uniform(5.0, 200.0, k)
    el[1, :k // 2] = 1.0 - 10.0 ** rng.uniform(-4, -1.3, k // 2)      # (conditioning cases for ANY solver: the oracle comparisons below use 1e-10 / 1e-8)
    return el


def test_warm_loop_matches_cold_loop_and_oracle_radec(oracle):
    """Config 3's shape in small: RA/Dec rows at half-day cadence, walkers whose fastest period still passes the wave's entry test, high
    eccentricities and periastron passages inside the table. Warm vs cold vs oracle; forward value == value with the gradient, bitwise."""
    gb = _gpu()
    rng = np.random.default_rng(51)
    n, W = 700, 333
    t = 50000.0 + 0.5 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = _dense_walkers(rng, W, 0.7, 60.0)
    warm = _eval(gb, obs, planets, el, None, True, True)
    cold = _eval(gb, obs, planets, el, None, True, False)
    assert not _same_bits(warm, cold), "the warm loop did not run: the two launches are bit-identical"
    _close("radec", warm, cold)
    warm_f = _eval(gb, obs, planets, el, None, False, True)
    assert np.array_equal(warm_f[0], warm[0]), "forward-only and gradient launches of the warm loop disagree"
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el, None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("radec warm", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    _cmp_oracle("radec cold", cold[0], cold[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    assert np.array_equal(_eval(gb, obs, planets, el, None, True, True)[0], warm[0]), "run-to-run determinism"


def test_warm_loop_every_single_planet_kind(oracle):
    """sep/PA + cor + per-walker nuisances, absolute / relative / marginalised RV with offsets, jitter and a trend: the warm loops of the
    astrometry AND the RV row bodies (kepler_solve_warm<1> / <2>), dense cadence."""
    gb = _gpu()
    rng = np.random.default_rng(52)
    n, W = 420, 200
    t = 50000.0 + 1.0 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    pa = np.arctan2(ra, dec); sep = np.hypot(ra, dec)
    el0 = dict(synth.TRUTH)
    rv = synth.truth_rv_star(t, el0, 8.0)
    obs = [
        dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 6.0), cor=rng.uniform(-0.6, 0.6, n)),
        dict(kind=1, planet=0, epoch=t + 0.25, y1=pa + rng.normal(0, 0.01, n), y2=sep + rng.normal(0, 4, n), s1=np.full(n, 0.01), s2=np.full(n, 4.0), cor=None),
        dict(kind=2, planet=-1, epoch=t, y1=rv + rng.normal(0, 3, n), y2=None, s1=np.full(n, 3.0), s2=None, cor=None, extra=(t - 50200.0) / 100.0),
        dict(kind=4, planet=0, epoch=t[::2], y1=-rv[::2] * 100 + rng.normal(0, 30, n // 2), y2=None, s1=np.full(n // 2, 30.0), s2=None, cor=None),
    ]
    planets = [dict(orbit_kind=0, has_mass=True)]
    el = _dense_walkers(rng, W, 1.2, 40.0, e_hi=0.9)
    el[8] = rng.uniform(1.0, 15.0, W)
    nuis = np.zeros((len(obs) * 3, W))
    for o in range(2):
        nuis[o * 3 + 0] = rng.uniform(0, 3, W); nuis[o * 3 + 1] = rng.normal(1, 0.01, W); nuis[o * 3 + 2] = rng.normal(0, 0.01, W)
    for o in range(2, 4):
        nuis[o * 3 + 0] = rng.normal(0, 5, W); nuis[o * 3 + 1] = rng.uniform(0.1, 4, W)
    nuis[2 * 3 + 2] = rng.normal(0, 2, W)      # trend coefficient of the absolute-RV table
    for nz in (nuis, None):
        warm = _eval(gb, obs, planets, el, nz, True, True)
        cold = _eval(gb, obs, planets, el, nz, True, False)
        # (round 6: the NUISANCE kernels of kind sets with sep/PA or RV rows carry a warm loop too — octo_kernels.h: main_warm_plain, rows by plain
        # scalar loads; round 5 left them cold, they run out of SGPRs with a second pair of prefetching loops)
        assert not _same_bits(warm, cold)
        _close("kinds", warm, cold, g_tol=1e-9)
        assert np.array_equal(_eval(gb, obs, planets, el, nz, False, True)[0], warm[0])
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el, nz, grad=True)
        # The contract's bar (1e-8) against the reference-order restatement wherever ITS arithmetic holds it — e <= 0.99; the ν = 2 atan(…tan(E/2))
        # route the reference takes loses digits towards the parabola (tests/test_oracle.py: 2e-6 in ∂/∂e at e = 0.999999). The near-parabolic
        # lanes are judged against the independent 60-digit oracle instead (round 5 loosened the bar to 2e-8 for them): the
        # worst of them, at the same 1e-8; all of them against the restatement at a sanity bound.
        lo = el[1] <= 0.99
        pick = lambda x, m: None if x is None else x[..., m]
        for tag, res in (("warm", warm), ("cold", cold)):
            _cmp_oracle(f"kinds {tag}, e <= 0.99", res[0][lo], res[1][:, lo], pick(res[2], lo), ll_o[lo], g_o[:, lo], pick(gn_o, lo), ll_rtol=1e-9, g_rtol=1e-8)
            _cmp_oracle(f"kinds {tag}, near-parabolic (sanity)", res[0][~lo], res[1][:, ~lo], pick(res[2], ~lo), ll_o[~lo], g_o[:, ~lo], pick(gn_o, ~lo), ll_rtol=1e-8, g_rtol=1e-5)
        sc = np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300)
        dev = (np.abs(warm[1] - g_o) / sc).max(axis=0); dev[lo] = 0.0
        w = int(np.argmax(dev))
        from stress_parity import mp_value_and_gradient
        ll_m, g_m = mp_value_and_gradient(obs, planets, el, nz, w)
        n_el = el.shape[0]
        for tag, res in (("warm", warm), ("cold", cold)):
            assert abs(res[0][w] - ll_m) < 1e-9 * max(1.0, abs(ll_m)), (tag, "ll against 60 digits", w, el[1, w])
            assert np.all(np.abs(res[1][:, w] - g_m[:n_el]) < 1e-8 * 10 * sc[:, 0] + 1e-13 * np.abs(g_m[:n_el]).max()), (tag, "gradient against 60 digits", w, el[1, w])
    # RA/Dec + cor with per-walker jitter / platescale / northangle: k_main<1, ·, true, RADEC|COR>, the nuisance kernel with a warm loop
    obs1, nuis1 = obs[:1], nuis[:3]
    planets1 = [dict(orbit_kind=0, has_mass=False)]
    warm = _eval(gb, obs1, planets1, el, nuis1, True, True)
    cold = _eval(gb, obs1, planets1, el, nuis1, True, False)
    assert not _same_bits(warm, cold)
    _close("radec+cor nuis", warm, cold, g_tol=1e-9)
    assert np.array_equal(_eval(gb, obs1, planets1, el, nuis1, False, True)[0], warm[0])
    ll_o, g_o, gn_o = oracle.oracle_eval(obs1, planets1, el, nuis1, grad=True, active=synth.active_mask(1, 1, mass=False))
    _cmp_oracle("radec+cor nuis warm", warm[0], warm[1], warm[2], ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)


def test_warm_entry_test_and_odd_tables(oracle):
    """What keeps a wave in the cold loop, and tables the predictor must survive: (a) one walker of the tile too fast for the cadence —
    its whole wave stays cold, bit-identical to OCTO_WARM=0; (b) a table with one long gap: round 6 — the gap's row alone is solved cold (a scalar
    branch on the row's key), the rest of the table stays on the warm loop (round 5 took the table's largest step for every row: all cold);
    (c) duplicate epochs (Δt = 0) and an UNSORTED table (negative steps) on the warm loop; (d) an invalid walker (e >= 1, NaN) among valid
    ones: −Inf for it, its neighbours against the oracle."""
    gb = _gpu()
    rng = np.random.default_rng(53)
    n, W = 300, 128
    t = 50000.0 + 0.5 * np.arange(n)
    mk = lambda tt: [dict(kind=0, planet=0, epoch=tt, y1=rng.normal(0, 300, tt.size), y2=rng.normal(0, 300, tt.size), s1=np.full(tt.size, 5.0), s2=np.full(tt.size, 7.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    act = synth.active_mask(1, 1, mass=False, nuis=False)
    el = _dense_walkers(rng, W, 1.0, 30.0)
    # (a) walker 70 (second tile) at a = 0.05 AU: ΔM per half day ~ 0.8 rad
    el_a = el.copy(); el_a[0, 70] = 0.05
    obs = mk(t)
    warm = _eval(gb, obs, planets, el_a, None, True, True); cold = _eval(gb, obs, planets, el_a, None, True, False)
    assert np.array_equal(warm[0][64:], cold[0][64:]) and np.array_equal(warm[1][:, 64:], cold[1][:, 64:]), "a vetoed wave must run the cold loop"
    assert not (np.array_equal(warm[0][:64], cold[0][:64]) and np.array_equal(warm[1][:, :64], cold[1][:, :64]))
    _close("veto", warm, cold)
    # (b) one long gap
    tg = t.copy(); tg[n // 2:] += 900.0
    obs = mk(tg)
    warm = _eval(gb, obs, planets, el, None, True, True); cold = _eval(gb, obs, planets, el, None, True, False)
    assert not _same_bits(warm, cold), "a table with one long gap must still take the warm loop"
    _close("one gap", warm, cold)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el, None, grad=True, active=act)
    _cmp_oracle("one gap warm", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    # (c) duplicates and an unsorted table
    td = t.copy(); td[10:13] = td[10]; td[100:140:2], td[101:141:2] = t[101:141:2], t[100:140:2]      # steps of 0, +1.0, -0.5 days
    obs = mk(td)
    el_c = el.copy(); el_c[0] = np.maximum(el_c[0], 1.6)      # (the largest step is 1.5 days now: periods the entry test still admits)
    warm = _eval(gb, obs, planets, el_c, None, True, True); cold = _eval(gb, obs, planets, el_c, None, True, False)
    assert not _same_bits(warm, cold)
    _close("unsorted", warm, cold)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el_c, None, grad=True, active=act)
    _cmp_oracle("unsorted warm", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    # (d) invalid walkers inside a tile
    el_d = el.copy(); el_d[1, 5] = 1.2; el_d[0, 17] = np.nan; el_d[6, 90] = -1.0
    obs = mk(t)
    warm = _eval(gb, obs, planets, el_d, None, True, True)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el_d, None, grad=True, active=act)
    assert np.isneginf(warm[0][[5, 17, 90]]).all() and np.isfinite(warm[0]).sum() == W - 3
    _cmp_oracle("invalid among valid", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)


def test_warm_chain_does_not_drift_over_a_long_chunk(oracle):
    """The warm chain never sees E or M as numbers, so its rounding accumulates until the next cold row. OCTO_CHUNK forces 2 500 rows per
    wave (the planner's own chunks are tens to a few hundred rows): slow, low-e walkers that never fall back — the longest chains there are —
    against the oracle at the usual bar and against the cold loop at 1e-11."""
    gb = _gpu()
    rng = np.random.default_rng(54)
    n, W = 10_000, 64
    t = 50000.0 + 1.0 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 10, n), y2=dec + rng.normal(0, 10, n), s1=np.full(n, 10.0), s2=np.full(n, 10.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = synth.draw_walkers(rng, W, 3.0, 100.0)
    el[1] = rng.uniform(0.0, 0.4, W)
    warm = _eval(gb, obs, planets, el, None, True, True, env={"OCTO_CHUNK": "2500"})
    cold = _eval(gb, obs, planets, el, None, True, False, env={"OCTO_CHUNK": "2500"})
    assert not _same_bits(warm, cold)
    _close("long chain", warm, cold, ll_tol=1e-11, g_tol=1e-9)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el, None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("long chain warm", warm[0], warm[1], None, ll_o, g_o, None)


def test_warm_step_device_routine_over_the_elliptic_domain(pkg):
    """kepler_solve_warm on its own (octo_debug_kepler_warm, a test hook): a cold solve at M, then ONE warm step by ΔM, over e up to
    1 − 1e-9, every phase of the orbit and ΔM from 1e-7 to 0.3 rad — against an 80-bit Newton solve at M + ΔM, error weighted by 1 − e cos E
    like test_kepler_device_solver. The inputs are grouped so that whole waves pass or fail the bound: where the wave took the warm path
    (`used`) the result must be as good as the cold routine's; and the bound must admit a useful share of the sample."""
    import ctypes as C
    capi = pkg.capi
    lib = capi.load_library()
    lib.octo_debug_kepler_warm.restype = C.c_int32
    lib.octo_debug_kepler_warm.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3
    rng = np.random.default_rng(77)
    n = 64 * 6000
    e = np.concatenate([rng.uniform(0, 1, n // 2), 1 - 10 ** rng.uniform(-9, -1, n // 2)])
    Ep = np.concatenate([rng.uniform(-np.pi, np.pi, n // 2), 10 ** rng.uniform(-6, 0.49, n // 2) * rng.choice([-1, 1], n // 2)])
    rng.shuffle(Ep)
    dM = 10 ** rng.uniform(-7, -0.5, n) * rng.choice([-1, 1], n)
    el = e.astype(np.longdouble); Epl = Ep.astype(np.longdouble)
    M = (Epl - el * np.sin(Epl)).astype(np.float64)
    M = np.clip(M, -np.pi, np.pi)
    # group by pass / fail of the a-priori bound so that a wave is (mostly) homogeneous
    D = 1 - e * np.cos(Ep)
    thr = (1e-3 / np.abs(dM) ** 3) ** 0.2      # WARM_TOL
    # (a bound below WARM_MIN_THR = 2 never starts warm, as in k_main: ΔM above 0.031 rad is the cold routine's)
    order = np.argsort(~(((1 / D) < 0.98 * thr) & (thr > 2.05)), kind="stable")
    M, dM, e = (np.ascontiguousarray(x[order]) for x in (M, dM, e))
    sE = np.empty(n); cE = np.empty(n); used = np.empty(n)
    ctx = C.c_void_p()
    assert lib.octo_ctx_create(C.byref(ctx), 0) == 0
    try:
        assert lib.octo_debug_kepler_warm(ctx, capi._dptr(M), capi._dptr(dM), capi._dptr(e), n, capi._dptr(sE), capi._dptr(cE), capi._dptr(used)) == 0
    finally:
        lib.octo_ctx_destroy(ctx)
    Mn = M.astype(np.longdouble) + dM.astype(np.longdouble)
    el = e.astype(np.longdouble)
    Et = np.arctan2(sE, cE).astype(np.longdouble)
    Et = Et + np.longdouble(2 * np.pi) * np.rint((Mn - Et) / np.longdouble(2 * np.pi))      # the branch of E that goes with M + ΔM
    for _ in range(60):
        Et = Et - (Et - el * np.sin(Et) - Mn) / (1 - el * np.cos(Et))
    conv = np.abs(Et - el * np.sin(Et) - Mn) < 1e-17
    cond = (1 - el * np.cos(Et)).astype(np.float64)
    err = np.maximum(np.abs(sE - np.sin(Et).astype(np.float64)), np.abs(cE - np.cos(Et).astype(np.float64))) * cond
    w = (used > 0) & conv
    assert w.mean() > 0.3, w.mean()
    assert err[w].max() < 2e-15, (err[w].max(), e[w][np.argmax(err[w])], dM[w][np.argmax(err[w])])
    c = (used == 0) & conv
    assert err[c].max() < 2e-15, err[c].max()                                                  # the fallback rows: the cold routine
    assert np.all(np.abs(sE[w] ** 2 + cE[w] ** 2 - 1) < 1e-14)


def test_dense_fixtures_at_60_digits_warm_and_cold():
    """tests/golden/dense.json F14 and gappy.json F16 (oracle/make_golden.py --dense-only / --gappy-only: the independent 60-digit oracle on dense tables and on tables with gaps — a Newton solve of Kepler's
    equation at 60 digits knows nothing of starters): the throughput kernels with the warm-started loop AND with OCTO_WARM=0 against the
    same numbers at the golden-vector bars (1e-12 on ll; gradients within 1e-9 of their 60-digit values + cancellation scale), and the warm
    loop really ran (the two launches differ in their last bits)."""
    import json
    from pathlib import Path
    from conftest import case_tables
    from test_gpu_parity import LL_RTOL, G_RTOL, G_CANCEL, grad_ok
    gb = _gpu()
    gdir = Path(__file__).resolve().parent / "golden"
    cases = [c for c in json.loads((gdir / "dense.json").read_text())["cases"] if c["name"].startswith("F14")]
    cases += [c for c in json.loads((gdir / "gappy.json").read_text())["cases"] if c["name"].startswith("F16")]      # round 6: tables with gaps
    assert len(cases) == 6
    for case in cases:
        obs, planets, elems, nuis = case_tables(case)
        res = {}
        for warm in (True, False):
            ll, g_el, g_nu = _eval(gb, obs, planets, elems, nuis, True, warm)
            err = rel_err(ll, np.asarray(case["ll"]), 1.0)
            assert np.all(err < LL_RTOL), (case["name"], warm, "ll", err.max())
            ok, worst = grad_ok(g_el, case["g_elems"], case["s_elems"], rtol=G_RTOL, cancel=G_CANCEL)
            assert ok, (case["name"], warm, "g_elems", worst)
            if nuis is not None:
                ok, worst = grad_ok(g_nu, case["g_nuis"], case["s_nuis"], rtol=G_RTOL, cancel=G_CANCEL)
                assert ok, (case["name"], warm, "g_nuis", worst)
            res[warm] = (ll, g_el, g_nu)
        assert not _same_bits(res[True], res[False]), (case["name"], "the warm loop did not run")


def test_warm_loop_on_gappy_and_multi_scale_tables(oracle):
    """Round 6: the step bound is per WAVE (the first entry of the table's ladder of step quantiles that none of the wave's lanes
    vetoes) and the veto per ROW. (a) an absolute-RV table of nightly runs with seasonal gaps (synth.gappy_epochs: steps of 0.02 d, ~1 d, ~240 d)
    without nuisances: warm, against cold and the oracle; (b) the same epochs as RA/Dec rows with walkers so fast that only the intra-night steps
    pass for some tiles (those tiles take a LOWER rung of the ladder, their night-to-night rows go cold) and too fast for any rung in one tile
    (cold, bit-identical to OCTO_WARM=0); (c) a table longer than WARM_RESTART = 256 rows per wave chunk at OCTO_CHUNK=700: the forced cold rows."""
    gb = _gpu()
    rng = np.random.default_rng(61)
    n, W = 1500, 192
    t = synth.gappy_epochs(n, per_night=4, nights_per_season=40, rng=rng)
    planets = [dict(orbit_kind=0, has_mass=True)]
    el = _dense_walkers(rng, W, 1.0, 40.0, e_hi=0.9)
    el[8] = rng.uniform(1.0, 15.0, W)
    rv = synth.truth_rv_star(t, dict(synth.TRUTH), 8.0)
    obs = [dict(kind=2, planet=-1, epoch=t, y1=rv + rng.normal(0, 3, n), y2=None, s1=np.full(n, 3.0), s2=None, cor=None)]
    warm = _eval(gb, obs, planets, el, None, True, True); cold = _eval(gb, obs, planets, el, None, True, False)
    assert not _same_bits(warm, cold), "the gappy RV table did not take the warm loop"
    _close("gappy rv", warm, cold, g_tol=1e-9)
    assert np.array_equal(_eval(gb, obs, planets, el, None, False, True)[0], warm[0])
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el, None, grad=True, active=synth.active_mask(1, 1, nuis=False))
    _cmp_oracle("gappy rv warm", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    # (b) RA/Dec rows at the same epochs; tile 1: periods of 20-60 days (ΔM of a night-to-night step 0.1-0.3 rad: only the 0.02-day rung passes),
    # tile 2: one walker at a = 0.02 AU (P ~ 1 day: no rung passes)
    planets1 = [dict(orbit_kind=0, has_mass=False)]
    ra, dec = synth.truth_radec(t)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=None)]
    el_b = el.copy(); el_b[8] = 0.0
    el_b[0, 64:128] = rng.uniform(0.16, 0.3, 64); el_b[1, 64:128] = rng.uniform(0.0, 0.6, 64)
    el_b[0, 130] = 0.02
    warm = _eval(gb, obs, planets1, el_b, None, True, True); cold = _eval(gb, obs, planets1, el_b, None, True, False)
    for lo, name in ((0, "slow tile"), (64, "fast tile: a lower rung")):
        assert not (np.array_equal(warm[0][lo:lo + 64], cold[0][lo:lo + 64]) and np.array_equal(warm[1][:, lo:lo + 64], cold[1][:, lo:lo + 64])), name
    assert np.array_equal(warm[0][128:], cold[0][128:]) and np.array_equal(warm[1][:, 128:], cold[1][:, 128:]), "a tile no rung admits must run the cold loop"
    _close("ladder", warm, cold, g_tol=1e-9)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets1, el_b, None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("ladder warm", warm[0], warm[1], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    # (c) long chunks: every 256th row of the table restarts the chain
    n2 = 2800
    t2 = 50000.0 + 0.5 * np.arange(n2)
    ra, dec = synth.truth_radec(t2)
    obs = [dict(kind=0, planet=0, epoch=t2, y1=ra + rng.normal(0, 5, n2), y2=dec + rng.normal(0, 5, n2), s1=np.full(n2, 5.0), s2=np.full(n2, 7.0), cor=None)]
    el_c = synth.draw_walkers(rng, 64, 3.0, 60.0); el_c[1] = rng.uniform(0, 0.4, 64)
    warm = _eval(gb, obs, planets1, el_c, None, True, True, env={"OCTO_CHUNK": "700"}); cold = _eval(gb, obs, planets1, el_c, None, True, False, env={"OCTO_CHUNK": "700"})
    assert not _same_bits(warm, cold)
    _close("restart", warm, cold, ll_tol=1e-12, g_tol=1e-10)
