"""
CPU tests (-m "not gpu"): the oracle (oracle/octo_oracle.c) against
  * the committed golden vectors (independent 50-digit mpmath oracle, oracle/make_golden.py),
  * Kepler's equation itself,
  * the tutorial astrometry table the reference ships in test/integration-tests.jl:8-15 (known-answer),
  * the self-consistency properties the reference's own tests assert for this path.
"""
from pathlib import Path

import numpy as np
import pytest

from conftest import case_tables, rel_err

LL_RTOL = 1e-12      # oracle (Float64, reference order) vs 50-digit truth
G_RTOL = 1e-9        # per gradient component ...
G_CANCEL = 1e-13     # ... plus this fraction of Σ_rows |∂ll_row/∂θ| (fixture field s_*, and its maximum over the
                     # fixture's walkers): the rounding floor of a sum whose terms cancel, and of residuals that are
                     # themselves differences of nearly equal numbers (a walker sitting on the likelihood maximum)


def grad_ok(g, ref, scale, rtol=G_RTOL, cancel=G_CANCEL):
    g, ref, scale = (np.asarray(v, dtype=np.float64) for v in (g, ref, scale))
    tol = rtol * np.abs(ref) + cancel * (scale + scale.max(axis=1, keepdims=True))
    bad = np.abs(g - ref) > tol
    worst = np.max(np.abs(g - ref) / np.where(tol > 0, tol, 1.0)) if g.size else 0.0
    return not bad.any(), worst


def test_golden_vectors(oracle, golden):
    assert len(golden["cases"]) >= 15
    for case in golden["cases"]:
        obs, planets, elems, nuis = case_tables(case)
        ll, g_el, g_nu = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
        ref_ll = np.asarray(case["ll"])
        # the reference's marginalised-RV formula (rv-absolute-margin.jl:181) subtracts B²/4A from C: two large,
        # nearly equal sums. That cancellation belongs to the reference arithmetic the oracle restates.
        has_marg = any(ob["kind"] == "RV_ABS_MARG" for ob in case["obs"])
        tol = 1e-10 if has_marg else LL_RTOL
        cancel = 1e-11 if has_marg else G_CANCEL
        if case["name"] == "F7_kepler_edges":
            # Reference order computes p = a(1−e²), ν_fact = √((1+e)/(1−e)) and r = p/(1+e cos ν). At e = 0.999999
            # the value loses log10(1/(1−e)) = 6 digits and the forward-mode ∂/∂e about twice that to cancellation —
            # in Julia exactly as here: conditioning of the reference arithmetic, not a bug. The last walker
            # (e = 0.999999) is therefore held to 1e-9 / 1e-5; the others (e <= 0.99) to the normal bars.
            sel = elems[1] <= 0.99
            assert np.all(rel_err(ll[sel], ref_ll[sel], 1.0) < tol)
            ok, worst = grad_ok(g_el[:, sel], np.asarray(case["g_elems"])[:, sel], np.asarray(case["s_elems"])[:, sel])
            assert ok, (case["name"], "g_elems e<=0.99", worst)
            assert np.all(rel_err(ll[~sel], ref_ll[~sel], 1.0) < 1e-9)
            ok, worst = grad_ok(g_el[:, ~sel], np.asarray(case["g_elems"])[:, ~sel], np.asarray(case["s_elems"])[:, ~sel], rtol=1e-5)
            assert ok, (case["name"], "g_elems e=0.999999", worst)
            continue
        assert np.all(rel_err(ll, ref_ll, 1.0) < tol), (case["name"], rel_err(ll, ref_ll, 1.0).max())
        ok, worst = grad_ok(g_el, case["g_elems"], case["s_elems"], cancel=cancel)
        assert ok, (case["name"], "g_elems", worst)
        if nuis is not None:
            ok, worst = grad_ok(g_nu, case["g_nuis"], case["s_nuis"], cancel=cancel)
            assert ok, (case["name"], "g_nuis", worst)


def test_value_path_equals_dual_path(oracle, golden):
    """ForwardDiff duals carry the same primal: forward-only and gradient evaluations agree bit for bit."""
    for case in golden["cases"][:6]:
        obs, planets, elems, nuis = case_tables(case)
        ll0, _, _ = oracle.oracle_eval(obs, planets, elems, nuis, grad=False)
        ll1, _, _ = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
        assert np.array_equal(ll0, ll1)


def test_markley_solves_keplers_equation(oracle):
    lib = oracle.load_oracle()
    rng = np.random.default_rng(1)
    M = rng.uniform(-np.pi, np.pi, 20000)
    e = np.concatenate([rng.uniform(0, 0.999, 15000), 1 - 10 ** rng.uniform(-6, -3, 5000)])
    worst = 0.0
    for m, ee in zip(M, e):
        E = lib.octo_oracle_kepler_markley(m, ee)
        worst = max(worst, abs(E - ee * np.sin(E) - m))
    assert worst < 2e-15, worst
    # reduction of the mean anomaly and the early-return branches
    assert lib.octo_oracle_kepler_markley(0.0, 0.3) == 0.0
    assert lib.octo_oracle_kepler_markley(1.25, 0.0) == 1.25
    assert abs(lib.octo_oracle_kepler_markley(40.0, 0.5) - lib.octo_oracle_kepler_markley(40.0 - 12 * np.pi, 0.5)) < 1e-14


TUT_EPOCH = np.array([50000, 50120, 50240, 50360, 50480, 50600, 50720, 50840], float)
TUT_RA = np.array([-505.7637580573554, -502.570356287689, -498.2089148883798, -492.67768482682357, -485.9770335870402,
                   -478.1095526888573, -469.0801731788123, -458.89628893460525])
TUT_DEC = np.array([-66.92982418533026, -37.47217527025044, -7.927548139010479, 21.63557115669823, 51.147204404903704,
                    80.53589069730698, 109.72870493064629, 138.65128697876773])


def test_tutorial_table_known_answer(oracle):
    """The 16 numbers of the reference's tutorial table (test/integration-tests.jl:8-15) were produced by
    PlanetOrbits.jl from a=12 AU, e=0.11, i=41°, ω=38°, Ω=16°, plx=50 mas — the only freedom left is the
    time scale (M, tp: the table predates the current year constant). With those two fitted, the
    restatement reproduces every entry to 1e-11 mas: this pins the Kepler solve, the projection and the
    angle conventions against numbers that came out of the reference's own dependency."""
    el = [12.0, 0.11, np.deg2rad(41), np.deg2rad(38), np.deg2rad(16), 41479.14852101943, 1.2000965847634995, 50.0, 0.0]
    for t, ra, dec in zip(TUT_EPOCH, TUT_RA, TUT_DEC):
        s = oracle.oracle_orbitsolve(el, t)
        assert abs(s["raoff"] - ra) < 1e-11 and abs(s["decoff"] - dec) < 1e-11, (t, s["raoff"] - ra, s["decoff"] - dec)


def test_tutorial_table_fit(oracle):
    """Where the two fitted numbers of the pin above come from: oracle/fit_tutorial_table.py, re-run here. 16 equations, 2 unknowns
    (total mass and periastron epoch; a, e, i, ω, Ω, plx held at the tutorial's round values): a restatement whose Kepler solve,
    projection or angle conventions differed from PlanetOrbits' could not reach 1e-11 mas."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fit_tutorial_table", Path(__file__).resolve().parent.parent / "oracle" / "fit_tutorial_table.py")
    ft = importlib.util.module_from_spec(spec); spec.loader.exec_module(ft)
    assert np.array_equal(ft.TUT_RA, TUT_RA) and np.array_equal(ft.TUT_DEC, TUT_DEC)
    M, tp, r = ft.fit(oracle)
    assert abs(M - 1.2000965847634995) < 1e-13 and abs(tp - 41479.14852101943) < 1e-8, (M, tp)
    assert np.abs(r).max() < 1e-11, np.abs(r).max()
    # and the fit is a genuine minimum, not a flat direction: moving either parameter by 1e-6 relative breaks the pin
    assert np.abs(ft.residuals(oracle, M * (1 + 1e-6), tp)).max() > 1e-6 and np.abs(ft.residuals(oracle, M, tp + 0.05)).max() > 1e-6


def test_reference_dump_if_present(oracle, golden):
    """When someone with Julia has run tools/julia_crosscheck.jl, tests/golden/reference_dump.json holds the REAL reference's
    numbers for the committed fixtures: hold the restatement to them (1e-10 relative on values, 1e-8 on gradients)."""
    dump_path = Path(__file__).resolve().parent / "golden" / "reference_dump.json"
    if not dump_path.exists():
        pytest.skip("tests/golden/reference_dump.json not generated (no Julia in the build image): parity stays pinned by the tutorial "
                    "table, the Thiele-Innes identities and the 60-digit oracle only")
    import json
    dump = json.loads(dump_path.read_text())
    checked = 0
    for case in golden["cases"]:
        for file in ("fixtures.json", "kep.json", "trend.json", "dense.json", "gappy.json"):
            r = dump.get(f"{file}/{case['name']}")
            if r is None:
                continue
            obs, planets, elems, nuis = case_tables(case)
            ll, _, _ = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
            assert np.all(rel_err(ll, np.asarray(r["ll"]), 1.0) < 1e-10), (case["name"], rel_err(ll, np.asarray(r["ll"]), 1.0).max())
            checked += 1
    assert checked > 0


def _northangle_tables(oracle):
    # test/unit/likelihoods.jl:38-58
    epochs = np.array([50000.0, 50300.0, 50600.0, 50900.0, 51200.0])
    el = np.array([15.0, 0.2, 0.6, 0.3, 1.1, 50000.0, 1.2, 50.0, 0.0])
    sols = [oracle.oracle_orbitsolve(el, t) for t in epochs]
    ra_m = np.array([s["raoff"] for s in sols]); dec_m = np.array([s["decoff"] for s in sols])
    pa_m = np.arctan2(ra_m, dec_m); sep_m = np.hypot(ra_m, dec_m)
    eps = 0.05
    pa_d = pa_m + eps
    ra_d = sep_m * np.sin(pa_d); dec_d = sep_m * np.cos(pa_d)
    n = len(epochs)
    seppa = dict(kind=1, planet=0, epoch=epochs, y1=pa_d, y2=sep_m, s1=np.full(n, 0.001), s2=np.full(n, 1.0), cor=None)
    radec = dict(kind=0, planet=0, epoch=epochs, y1=ra_d, y2=dec_d, s1=np.full(n, 1.0), s2=np.full(n, 1.0), cor=None)
    return el, eps, seppa, radec


def northangle_scan(eval_fn, tab, el, grid):
    elems = np.tile(el[:, None], (1, len(grid)))
    nuis = np.stack([np.zeros_like(grid), np.ones_like(grid), grid])
    return eval_fn([tab], [dict(orbit_kind=0, has_mass=False)], elems, nuis)


def test_northangle_sign_convention(oracle):
    """test/unit/likelihoods.jl:32-95 re-expressed: data rotated by +ε is undone by northangle = −ε in BOTH formats."""
    el, eps, seppa, radec = _northangle_tables(oracle)
    grid = np.linspace(-0.1, 0.1, 2001)
    f = lambda *a: oracle.oracle_eval(*a, grad=False)[0]
    best_s = grid[np.argmax(northangle_scan(f, seppa, el, grid))]
    best_r = grid[np.argmax(northangle_scan(f, radec, el, grid))]
    assert abs(best_s + eps) < 1e-3 and abs(best_r + eps) < 1e-3
    assert np.sign(best_s) == np.sign(best_r)
    z = np.array([0.0, -0.0])
    for tab in (seppa, radec):
        ll = northangle_scan(f, tab, el, z)
        assert ll[0] == ll[1]


def test_jitter_sensitivity(oracle):
    """test/unit/distributions.jl:102-152: Δll between jitter=300 and jitter=0.001 is > 1."""
    tab = dict(kind=0, planet=0, epoch=np.array([58000.0, 58200.0, 58400.0]), y1=np.array([100.0, 110.0, 120.0]),
               y2=np.array([100.0, 95.0, 90.0]), s1=np.full(3, 5.0), s2=np.full(3, 5.0), cor=None)
    elems = np.tile(np.array([10.0, 0.2, 0.5, 0.3, 0.4, 58000.0, 1.0, 50.0, 0.0])[:, None], (1, 2))
    nuis = np.array([[0.001, 300.0], [1.0, 1.0], [0.0, 0.0]])
    ll, _, _ = oracle.oracle_eval([tab], [dict(orbit_kind=0, has_mass=False)], elems, nuis, grad=False)
    assert ll[1] - ll[0] > 1


def test_gradient_matches_finite_differences(oracle):
    """test/integration/sampling.jl:136-192 re-expressed at the element level: AD gradient ≈ FiniteDiff gradient
    (atol=1e-3, rtol=1e-4 in the reference; central differences here are good to ~1e-6)."""
    tab = dict(kind=0, planet=0, epoch=np.array([50000.0, 50120.0, 50240.0, 50360.0]),
               y1=np.array([-505.76, -502.57, -498.21, -492.68]), y2=np.array([-66.93, -37.47, -7.93, 21.64]),
               s1=np.full(4, 10.0), s2=np.full(4, 10.0), cor=np.zeros(4))
    planets = [dict(orbit_kind=0, has_mass=False)]
    x0 = np.array([14.0, 0.21, 0.7, 0.5, 0.3, 41500.0, 1.25, 50.01, 0.0])
    _, g, _ = oracle.oracle_eval([tab], planets, x0[:, None], None, grad=True)
    for k in range(8):
        h = 1e-6 * max(1.0, abs(x0[k]))
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h; xm[k] -= h
        fp = oracle.oracle_eval([tab], planets, xp[:, None], None, grad=False)[0][0]
        fm = oracle.oracle_eval([tab], planets, xm[:, None], None, grad=False)[0][0]
        fd = (fp - fm) / (2 * h)
        assert abs(fd - g[k, 0]) <= 1e-3 + 1e-4 * abs(fd), (k, fd, g[k, 0])


def test_invalid_walkers_are_minus_inf(oracle):
    tab = dict(kind=0, planet=0, epoch=np.array([50000.0, 50100.0]), y1=np.array([1.0, 2.0]), y2=np.array([3.0, 4.0]),
               s1=np.ones(2), s2=np.ones(2), cor=None)
    good = np.array([10.0, 0.2, 0.5, 0.3, 0.4, 50000.0, 1.0, 50.0, 0.0])
    bad = np.tile(good[:, None], (1, 6))
    bad[1, 1] = 1.0; bad[1, 2] = -0.1; bad[0, 3] = np.nan; bad[0, 4] = -2.0; bad[6, 5] = 0.0
    ll, g, _ = oracle.oracle_eval([tab], [dict(orbit_kind=0, has_mass=False)], bad, None, grad=True)
    assert np.isfinite(ll[0]) and np.all(np.isneginf(ll[1:]))
    assert np.all(g[:, 1:] == 0.0)


def test_thiele_innes_semi_major_axis_known_answers(oracle):
    """The reference's own known answers for the Thiele-Innes angular semi-major axis (test/unit/nss.jl:3-27, function
    src/nss.jl:502-508 — the same u, v, α as src/parameterizations.jl:14-19): A=10, B=0, F=0, G=±10 -> α = 10; constants
    built from a_mas = 50, i = 1.2, Ω = 0.8, ω = 2.5 -> α = 50. The oracle's ThieleInnesOrbit derives a = α/plx and from it
    the mean motion, which is what these numbers pin."""
    c = oracle.oracle_consts()
    M, plx = 1.3, 40.0
    def alpha_from_oracle(A, B, F, G):
        el = np.array([A, 0.2, B, F, G, 50000.0, M, plx, 0.0])       # rows a, i, ω, Ω carry A, B, F, G
        n = oracle.oracle_orbitsolve(el, 50123.0, orbit_kind=2)["n"]    # rad / yr
        P_yr = 2 * np.pi / n
        a = (M * (P_yr * c.year2day_julian / c.kepler_year_to_julian_day) ** 2) ** (1 / 3)
        return a * plx
    assert abs(alpha_from_oracle(10.0, 0.0, 0.0, 10.0) - 10.0) < 1e-12
    assert abs(alpha_from_oracle(10.0, 0.0, 0.0, -10.0) - 10.0) < 1e-12
    a_mas, i, O, w = 50.0, 1.2, 0.8, 2.5
    A = a_mas * (np.cos(O) * np.cos(w) - np.sin(O) * np.sin(w) * np.cos(i))
    B = a_mas * (np.sin(O) * np.cos(w) + np.cos(O) * np.sin(w) * np.cos(i))
    F = a_mas * (-np.cos(O) * np.sin(w) - np.sin(O) * np.cos(w) * np.cos(i))
    G = a_mas * (-np.sin(O) * np.sin(w) + np.cos(O) * np.cos(w) * np.cos(i))
    assert abs(alpha_from_oracle(A, B, F, G) - a_mas) < 1e-8          # the reference's tolerance
    assert abs(alpha_from_oracle(A, B, F, G) - a_mas) < 1e-12


def test_thiele_innes_near_face_on_fixture(oracle):
    """Fixture F12 (oracle/make_ti_faceon.py, 60 digits): the reference-order restatement reproduces the value, and its gradient only up
    to the cancellation of the reference's own α² = u + √((u+v)(u−v)) near face-on (src/parameterizations.jl:15-18) — 1e-7 of scale here.
    The GPU test of the same fixture holds the kernels, which use the cancellation-free form, to the 60-digit gradient."""
    import json
    from conftest import case_tables, rel_err
    case = json.loads((Path(__file__).resolve().parent / "golden" / "ti_faceon.json").read_text())["cases"][0]
    obs, planets, elems, nuis = case_tables(case)
    ll, g_el, g_nu = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
    assert np.all(rel_err(ll, np.asarray(case["ll"]), 1.0) < 1e-11)
    ref = np.asarray(case["g_elems"]); sc = np.abs(ref).max(axis=1, keepdims=True)
    err = (np.abs(g_el - ref) / np.maximum(sc, 1e-300)).ravel()
    ti_rows = [0, 2, 3, 4]      # A, B, F, G of the Thiele-Innes planet (planet 0): where the cancellation enters
    assert 1e-7 < err[ti_rows].max() < 1e-3, err[ti_rows]
    assert np.delete(err, ti_rows).max() < 1e-8, err
