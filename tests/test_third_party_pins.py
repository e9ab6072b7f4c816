"""Pins of the oracle against THIRD-PARTY implementations that are in the image (scipy): not the reference — Julia cannot run here, DESIGN.md §1 —
but code and constants written by someone else, so that a slip in a recalled constant, in a density's normalisation or in the bivariate normal's
form cannot hide behind two in-house derivations that share the recollection. What these tests can NOT pin is a convention of the reference
itself (which bijector a bounded prior gets, the sign of an offset): that stays with tools/julia_crosscheck.jl.

  constants      scipy.constants (CODATA / IAU): au, the Julian year; the parsec only enters through rad2as / pc2au
  priors         scipy.stats logpdf of Uniform, LogUniform (reciprocal), Normal, truncated Normal and the sine law, + the log-Jacobian of the
                 bijector written out here (Bijectors' logit-type map of a two-sided support, the log map of a lower-bounded one: src/variables.jl:1449-1493)
  MvNormal       scipy.stats.multivariate_normal.logpdf of the row covariance [[σ1², ρσ1σ2], [ρσ1σ2, σ2²]] (src/likelihoods/relative-astrometry.jl:241-252)
  Kepler         scipy.optimize.brentq on M = E − e sin E against octo_oracle_orbitsolve's position (the solver row a4 of SURVEY.md §8)
  RV, sep/PA     scipy.stats.norm with offset / jitter / trend (rv-absolute.jl:172-204), PA = atan(ra, dec) with the wrapped residual (:192-202)
  barycentre     the reflex term of a planet strictly inside the observed one from per-planet solves (relative-astrometry.jl:104-142)
  OFTI           multivariate_normal(0, σ²DDᵀ + Σ) and a ridge solve (src/parameterizations.jl:318-405)
  reflex RV      relative RV = offset + radvel(sol); the star's RV = offset − (m·mjup2msol/M)·radvel(sol): momentum conservation with M the total mass
  UnitLength     a UniformCircular pair scaled by r: two Normal(0, 1) priors + logpdf(LogNormal(0, 0.1), r) by scipy.stats.lognorm (src/variables.jl:279-323)
  Thiele-Innes   the ThieleInnesOrbit planet = the Campbell orbit through the textbook constants A, B, F, G
  marginal RV    scipy.integrate.quad over the zero point: the reference's ll is 2·log(integral) − log 2π (rv-absolute-margin.jl:140-185)
  tperi          θ_at_epoch_to_tperi by its meaning — the position angle at the epoch IS θ — with brentq over the orbit, not by its formula
  O'Neil prior   2·log(Σ|t_j|·∛P/√(1−e²)) = 2·log(3·Σ|det ∂(x, y)/∂(P, e)|) with the determinant from 40-digit central differences (prior-observable.jl:78-137)
  HGCA           the model proper motions as time derivatives of the star's reflex offset, read off the likelihood's parabolas (hgca.jl:155-400)
  Each also runs against the HIP path itself under -m gpu (both kernel families), so the product is held to third-party numbers directly.
"""
import numpy as np
import pytest

scipy = pytest.importorskip("scipy")
import scipy.constants as sc
import scipy.optimize
import scipy.stats as ss


def test_constants_vs_scipy(oracle):
    c = oracle.oracle_consts()
    assert c.au2m == sc.au                                                   # IAU 2012 B2: exact
    assert abs(c.sec2year_julian * sc.Julian_year - 1.0) < 1e-15             # 1 / (365.25 · 86400)
    assert c.year2day_julian == sc.Julian_year / sc.day
    # Gaussian year from the IAU nominal GM☉ = 1.3271244e20 m³ s⁻² (Resolution B3 2015) and scipy's au: 2π√(au³/GM☉)
    gm_sun = 1.3271244e20
    assert abs(c.kepler_year_to_julian_day - 2 * np.pi * np.sqrt(sc.au ** 3 / gm_sun) / sc.day) < 1e-10
    # the parsec and the radian-to-arcsecond factor are both rounded to 206265 in PlanetOrbits; only their ratio reaches an observable (mas per AU per mas of parallax)
    assert c.rad2as / c.pc2au == 1.0
    assert abs(c.pc2au / (sc.parsec / sc.au) - 1.0) < 1e-6 and abs(c.rad2as / np.degrees(1.0) / 3600.0 - 1.0) < 1e-6
    # Jupiter mass in solar masses: ratio of the IAU 2015 nominal mass parameters
    assert abs(c.mjup2msol - 1.2668653e17 / gm_sun) < 1e-18


def _prior_only_lp(oracle, prior, theta_t):
    """lp of a model whose only term is ONE prior: a planet without observation tables, every element a constant but tp."""
    PR = dict(kind=prior[0], p0=prior[1], p1=prior[2], lo=prior[3], hi=prior[4])
    esrc = [dict(kind=0, i0=0, i1=0, flags=0, value=v) for v in (10.0, 0.1, 0.5, 0.3, 0.2, 50000.0, 1.0, 50.0, 0.0)]
    esrc[5] = dict(kind=1, i0=0, i1=0, flags=0, value=0.0)      # tp = θ[0]: any real number is a valid epoch of periastron
    th = np.asarray(theta_t, dtype=np.float64)[None, :]
    return oracle.oracle_model_logpost([], [dict(orbit_kind=0, has_mass=False)], oracle.make_priors([PR]), oracle.make_sources(esrc), None, th)


def _logistic(y):
    return 1.0 / (1.0 + np.exp(-y))


def test_prior_densities_vs_scipy(oracle):
    """logpdf_with_trans(prior, invlink(y), true) = logpdf(x) + log|dx/dy| (src/variables.jl:1205-1236, 1449-1493) for every prior kind of the C ABI,
    the density from scipy.stats, the Jacobian of the bijector spelled out, its derivative by central differences of the scipy value."""
    y = np.linspace(-6.0, 6.0, 25)
    two_sided = lambda lo, hi, y: (lo + (hi - lo) * _logistic(y), np.log(hi - lo) + np.log(_logistic(y)) + np.log1p(-_logistic(y)))

    def on(lo, hi, logpdf):      # y -> logpdf(x(y)) + log|dx/dy| under the bijector of the support (lo, hi)
        def f(y):
            if lo is not None and hi is not None: x, lj = two_sided(lo, hi, y)
            elif lo is not None: x, lj = lo + np.exp(y), y                  # lower bound only: log link
            elif hi is not None: x, lj = hi - np.exp(y), y                  # upper bound only
            else: x, lj = y, 0.0                                            # identity
            return logpdf(x) + lj
        return f
    cases = [((0, 2.0, 7.5, None, None), on(2.0, 7.5, ss.uniform(2.0, 5.5).logpdf)),
             ((1, 0.1, 300.0, None, None), on(0.1, 300.0, ss.loguniform(0.1, 300.0).logpdf)),
             ((2, 1.3, 0.4, None, None), on(None, None, ss.norm(1.3, 0.4).logpdf)),
             ((3, 1.2, 0.4, 0.1, None), on(0.1, None, ss.truncnorm((0.1 - 1.2) / 0.4, np.inf, 1.2, 0.4).logpdf)),
             ((3, 1.2, 0.4, 0.5, 2.0), on(0.5, 2.0, ss.truncnorm((0.5 - 1.2) / 0.4, (2.0 - 1.2) / 0.4, 1.2, 0.4).logpdf)),
             ((3, 1.2, 0.4, None, 3.0), on(None, 3.0, ss.truncnorm(-np.inf, (3.0 - 1.2) / 0.4, 1.2, 0.4).logpdf)),
             ((4, 0.0, 0.0, None, None), on(0.0, np.pi, lambda x: np.log(np.sin(x) / 2.0)))]      # Sine(): pdf sin(x)/2 on (0, π), src/distributions.jl:14-39
    for prior, f in cases:
        ref = f(y)
        lp, g = _prior_only_lp(oracle, prior, y)
        assert np.all(np.abs(lp - ref) <= 1e-12 * np.maximum(1.0, np.abs(ref))), (prior, np.max(np.abs(lp - ref)))
        h = 1e-5
        gnum = (f(y + h) - f(y - h)) / (2 * h)                              # central differences of the scipy value
        assert np.all(np.abs(g[0] - gnum) <= 1e-7 * np.maximum(1.0, np.abs(gnum))), (prior, np.max(np.abs(g[0] - gnum) / np.maximum(1.0, np.abs(gnum))))
    # the sine law integrates to one (the normalisation 1/2 is the only number in it)
    xs = np.linspace(0.0, np.pi, 20001)
    assert abs(np.trapezoid(np.sin(xs) / 2.0, xs) - 1.0) < 1e-8


def _mvnormal_case(oracle):
    """Three RA/Dec rows with correlations, and the scipy values of their log-likelihood without and with per-walker nuisances."""
    rng = np.random.default_rng(8)
    el = np.array([12.0, 0.3, 1.0, 0.5, 2.0, 50100.0, 1.2, 50.0, 0.0])
    t = np.array([50000.0, 50400.0, 51000.0])
    sol = [oracle.oracle_orbitsolve(el, tj) for tj in t]
    ra_m, dec_m = np.array([q["raoff"] for q in sol]), np.array([q["decoff"] for q in sol])      # [mas]
    s1, s2, cor = np.array([3.0, 5.0, 2.0]), np.array([4.0, 2.5, 6.0]), np.array([0.6, -0.35, 0.0])
    ra, dec = ra_m + rng.normal(0, 3, 3), dec_m + rng.normal(0, 3, 3)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra, y2=dec, s1=s1, s2=s2, cor=cor, extra=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    ref = sum(ss.multivariate_normal([0, 0], [[a * a, r * a * b], [r * a * b, b * b]]).logpdf([y1 - m1, y2 - m2])
              for a, b, r, y1, m1, y2, m2 in zip(s1, s2, cor, ra, ra_m, dec, dec_m))
    jit, ps, na = 1.7, 1.002, 0.01                                                # jitter, platescale, northangle (rad)
    nuis = np.array([[jit], [ps], [na]])
    ref_n = 0.0
    for a, b, r, y1, m1, y2, m2 in zip(s1, s2, cor, ra, ra_m, dec, dec_m):
        a2, b2 = np.hypot(a, jit), np.hypot(b, jit)
        u1 = ps * (y1 * np.cos(na) + y2 * np.sin(na)); u2 = ps * (y2 * np.cos(na) - y1 * np.sin(na))      # the data rotated / scaled, :210-215
        ref_n += ss.multivariate_normal([0, 0], [[a2 * a2, r * a2 * b2], [r * a2 * b2, b2 * b2]]).logpdf([u1 - m1, u2 - m2])
    return obs, planets, el[:, None], nuis, ref, ref_n


def test_row_density_vs_scipy_multivariate_normal(oracle):
    """One RA/Dec row with a correlation and a per-walker jitter: ll = logpdf(MvNormal([σ1² + j², ρσ1σ2; ρσ1σ2, σ2² + j²]), resid) with
    σ_k ← √(σ_k² + j²) BEFORE the off-diagonal is formed (relative-astrometry.jl:234-252); without nuisances and with cor = 0 the same call
    is two independent normals. scipy's multivariate_normal does the linear algebra; the model offset comes from octo_oracle_orbitsolve."""
    obs, planets, el, nuis, ref, ref_n = _mvnormal_case(oracle)
    ll, _, _ = oracle.oracle_eval(obs, planets, el, None, grad=False)
    assert abs(ll[0] - ref) < 1e-11 * abs(ref)
    ll_n, _, _ = oracle.oracle_eval(obs, planets, el, nuis, grad=False)
    assert abs(ll_n[0] - ref_n) < 1e-11 * abs(ref_n)


@pytest.mark.gpu
def test_gpu_row_density_vs_scipy_multivariate_normal(oracle):
    """The same rows through the HIP path (both kernel families): the product against scipy directly, not only against the oracle."""
    import gpu_binding
    obs, planets, el, nuis, ref, ref_n = _mvnormal_case(oracle)
    for small in (None, 0):
        W = 1 if small is None else 70
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), None, grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-11 * abs(ref))
        ll_n, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)
        assert np.all(np.abs(ll_n - ref_n) < 1e-11 * abs(ref_n))


def test_kepler_root_vs_scipy_brentq(oracle):
    """Position of a face-on, ω = Ω = 0 orbit from the oracle against E from scipy's bracketing root finder: x = a(cos E − e), y = a√(1−e²) sin E
    scaled by plx (mas per AU); eccentricities up to 0.999, mean anomalies over several revolutions and both signs."""
    c = oracle.oracle_consts()
    rng = np.random.default_rng(9)
    for e in (0.0, 0.1, 0.5, 0.9, 0.99, 0.999):
        for _ in range(12):
            a, M_tot, plx, tp = 3.0, 1.1, 40.0, 50000.0
            period_d = np.sqrt(a ** 3 / M_tot) * c.kepler_year_to_julian_day
            t = tp + period_d * rng.uniform(-2.5, 2.5)
            MA = 2 * np.pi * (t - tp) / period_d
            Mw = np.remainder(MA + np.pi, 2 * np.pi) - np.pi
            E = scipy.optimize.brentq(lambda E: E - e * np.sin(E) - Mw, -np.pi - 1e-9, np.pi + 1e-9, xtol=1e-15, rtol=1e-15)
            el = np.array([a, e, 0.0, 0.0, 0.0, tp, M_tot, plx, 0.0])
            out = oracle.oracle_orbitsolve(el, t)
            # i = 0, ω = Ω = 0: north (dec) carries cos ν, east (ra) sin ν  (PlanetOrbits: ra ∝ sin(ν+ω) at Ω = 0)
            x_mas = a * (np.cos(E) - e) * plx * (c.rad2as / c.pc2au); y_mas = a * np.sqrt(1 - e * e) * np.sin(E) * plx * (c.rad2as / c.pc2au)
            scale = a * plx * (1 + e)
            # near periastron of a very eccentric orbit the root itself is conditioned like 1/(1−e): the bracketing solver's 1e-15 on E is the bar
            tol = 1e-11 * scale / max(1.0 - e, 1e-3) ** 0.5
            assert abs(out["decoff"] - x_mas) < tol and abs(out["raoff"] - y_mas) < tol, (e, t, out["raoff"], out["decoff"], x_mas, y_mas)


def test_radial_velocity_is_the_time_derivative_of_the_line_of_sight_position(oracle):
    """K and the RV law K(cos(ν+ω) + e cos ω) [m/s] (the [PO] formulas the oracle restates from memory) against the central difference of
    z(t) = r sin(ν+ω) sin i [AU] in scipy's units (au, day): a slip in K's formula, in au2m or in the Julian-year factor would show here."""
    rng = np.random.default_rng(10)
    for _ in range(40):
        a, e, inc, w, O = rng.uniform(0.5, 20), rng.uniform(0, 0.9), rng.uniform(0.1, 3.0), rng.uniform(0, 6.28), rng.uniform(0, 6.28)
        el = np.array([a, e, inc, w, O, 50000.0 + rng.uniform(0, 3000), rng.uniform(0.5, 2.0), 25.0, 0.0])
        t = 50000.0 + rng.uniform(0, 5000)
        z = lambda tt: (lambda q: q["r"] * np.sin(q["nu"] + w) * np.sin(inc))(oracle.oracle_orbitsolve(el, tt))
        h = 1e-4 * np.sqrt(a ** 3 / el[6]) * 365.25 * (1 - e) ** 1.5      # a small fraction of the periastron passage time [d]
        vz = (z(t + h) - z(t - h)) / (2 * h) * sc.au / sc.day              # m/s
        rv = oracle.oracle_orbitsolve(el, t)["radvel"]
        K = oracle.oracle_orbitsolve(el, t)["K"]
        assert abs(rv - vz) < 2e-7 * abs(K), (rv, vz, K)


def _ofti_case(c):
    """A nine-epoch table with correlations, three (e, a, tp, M, plx) sets, and scipy's log-marginal / ridge mean for each."""
    import scipy.linalg
    rng = np.random.default_rng(12)
    N = 9
    t = np.sort(50000.0 + rng.uniform(0, 2500, N))
    ra, dec = rng.normal(0, 300, N), rng.normal(0, 300, N)
    s_ra, s_dec, cor = rng.uniform(2, 9, N), rng.uniform(2, 9, N), rng.uniform(-0.6, 0.6, N)
    sigma = 1500.0
    nl = np.array([[0.0, 0.35, 0.8], [6.0, 9.0, 14.0], [50100.0, 49800.0, 51234.5], [1.0, 1.3, 0.9], [40.0, 25.0, 60.0]])      # e, a, tp, M, plx
    refs, means = [], []
    for w in range(nl.shape[1]):
        e, a, tp, M, _ = nl[:, w]
        period_d = np.sqrt(a ** 3 / M) * c.kepler_year_to_julian_day
        D = np.zeros((2 * N, 4)); d = np.zeros(2 * N); S = np.zeros((2 * N, 2 * N))
        for j in range(N):
            Mw = np.remainder(2 * np.pi * (t[j] - tp) / period_d + np.pi, 2 * np.pi) - np.pi
            E = scipy.optimize.brentq(lambda E: E - e * np.sin(E) - Mw, -np.pi - 1e-9, np.pi + 1e-9, xtol=1e-15, rtol=1e-15)
            x, y = np.cos(E) - e, np.sin(E) * np.sqrt(1 - e * e)
            D[2 * j, 1] = x; D[2 * j, 3] = y; D[2 * j + 1, 0] = x; D[2 * j + 1, 2] = y      # ra = xB + yG, dec = xA + yF  (:347-351)
            d[2 * j], d[2 * j + 1] = ra[j], dec[j]
            S[2 * j, 2 * j] = s_ra[j] ** 2; S[2 * j + 1, 2 * j + 1] = s_dec[j] ** 2
            S[2 * j, 2 * j + 1] = S[2 * j + 1, 2 * j] = cor[j] * s_ra[j] * s_dec[j]
        refs.append(ss.multivariate_normal(np.zeros(2 * N), sigma ** 2 * D @ D.T + S).logpdf(d))
        Wt = np.linalg.inv(S)
        means.append(scipy.linalg.solve(D.T @ Wt @ D + np.eye(4) / sigma ** 2, D.T @ Wt @ d, assume_a="pos"))
    return (t, ra, dec, s_ra, s_dec, cor, sigma, nl), np.array(refs), np.array(means).T


def test_ofti_marginal_likelihood_vs_scipy(oracle):
    """ofti_linear_solve (src/parameterizations.jl:318-405) marginalises (A, B, F, G) ~ N(0, σ²I) out of a linear-Gaussian model, so its
    log-marginal is logpdf(MvNormal(0, σ² D Dᵀ + Σ_data), d) on the 2N-vector of data, and its (A, B, F, G) the ridge-regression mean. scipy does both
    from the design matrix built here with brentq's E: the completed-square form the oracle (and the kernels) restate must reproduce them."""
    (t, ra, dec, s_ra, s_dec, cor, sigma, nl), ref, mean = _ofti_case(oracle.oracle_consts())
    abfg, logml = oracle.oracle_ofti(t, ra, dec, s_ra, s_dec, cor, sigma, nl)
    assert np.all(np.abs(logml - ref) < 1e-10 * np.abs(ref)), (logml, ref)
    assert np.all(np.abs(abfg - mean) < 1e-9 * np.abs(mean).max(axis=0)), (abfg, mean)


@pytest.mark.gpu
def test_gpu_ofti_and_kepler_vs_scipy(pkg, oracle):
    """The device OFTI kernels (k_ofti_main / k_ofti_finish through the mirror's ofti_linear_solve) against scipy's multivariate normal and ridge
    solve, and the device Kepler routine (octo_kepler_solve, octo_kepler_solve_table) against scipy.optimize.brentq — directly, not via the oracle."""
    import ctypes as C
    (t, ra, dec, s_ra, s_dec, cor, sigma, nl), ref, mean = _ofti_case(oracle.oracle_consts())
    out = pkg.ofti_linear_solve(t, ra, dec, s_ra, s_dec, cor, sigma, *nl)
    assert np.all(np.abs(out["log_marginal_likelihood"] - ref) < 1e-10 * np.abs(ref))
    got = np.stack([out["A"], out["B"], out["F"], out["G"]])
    assert np.all(np.abs(got - mean) < 1e-9 * np.abs(mean).max(axis=0))
    rng = np.random.default_rng(14)
    n = 400
    e = np.concatenate([rng.uniform(0, 0.9, n // 2), 1 - 10 ** rng.uniform(-6, -1, n // 2)])
    M = rng.uniform(-np.pi, np.pi, n)
    Eb = np.array([scipy.optimize.brentq(lambda E: E - ei * np.sin(E) - Mi, -np.pi - 1e-9, np.pi + 1e-9, xtol=1e-16, rtol=1e-15) for Mi, ei in zip(M, e)])
    lib = pkg.capi.load_library()
    ctx = C.c_void_p()
    assert lib.octo_ctx_create(C.byref(ctx), 0) == 0
    try:
        for fn in (lib.octo_kepler_solve, lib.octo_kepler_solve_table):
            E = np.empty(n); sE = np.empty(n); cE = np.empty(n)
            dp = pkg.capi._dptr
            assert fn(ctx, dp(M), dp(e), n, dp(E), dp(sE), dp(cE)) == 0
            cond = 1 - e * np.cos(Eb)                                    # the root's conditioning: brentq's own 1e-15 on E is the bar
            assert np.all(np.abs(sE - np.sin(Eb)) * cond < 5e-15) and np.all(np.abs(cE - np.cos(Eb)) * cond < 5e-15)
    finally:
        lib.octo_ctx_destroy(ctx)


@pytest.mark.gpu
def test_gpu_model_priors_vs_scipy(pkg):
    """The standard parameterisation ON THE DEVICE (k_model_fwd / k_small<MODEL>: invlink, logpdf_with_trans, healing rule) for a model whose likelihood
    is a constant: one RA/Dec row far inside its error bar, every planet variable a prior of a different kind — the log-posterior's variation with θ_t is
    the priors' alone, held to scipy.stats + the bijectors' Jacobians (as test_prior_densities_vs_scipy holds the oracle), both kernel families."""
    table = dict(epoch=[50000.0], ra=[0.0], dec=[0.0], σ_ra=[1e9], σ_dec=[1e9])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="one_row")],
                   variables=pkg.variables(a=pkg.LogUniform(0.1, 300.0), e=pkg.Uniform(0.0, 0.5), i=pkg.Sine(), ω=pkg.Normal(1.3, 0.4), Ω=pkg.Uniform(2.0, 7.5),
                                           tp=pkg.Normal(50100.0, 30.0)))
    system = pkg.System(name="PriorsOnly", companions=[b], observations=[],
                        variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.4), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 4.0), lower=40.0, upper=60.0)))
    model = pkg.LogDensityModel(system)
    assert model.names == ["M", "plx", "b_a", "b_e", "b_i", "b_ω", "b_Ω", "b_tp"]
    rng = np.random.default_rng(21)

    def logistic(y): return 1.0 / (1.0 + np.exp(-y))

    def two_sided(lo, hi, y): return lo + (hi - lo) * logistic(y), np.log(hi - lo) + np.log(logistic(y)) + np.log1p(-logistic(y))

    def ref(th):
        out = np.zeros(th.shape[1])
        x, lj = 0.1 + np.exp(th[0]), th[0];          out += ss.truncnorm((0.1 - 1.2) / 0.4, np.inf, 1.2, 0.4).logpdf(x) + lj
        x, lj = two_sided(40.0, 60.0, th[1]);        out += ss.truncnorm(-2.5, 2.5, 50.0, 4.0).logpdf(x) + lj
        x, lj = two_sided(0.1, 300.0, th[2]);        out += ss.loguniform(0.1, 300.0).logpdf(x) + lj
        x, lj = two_sided(0.0, 0.5, th[3]);          out += ss.uniform(0.0, 0.5).logpdf(x) + lj
        x, lj = two_sided(0.0, np.pi, th[4]);        out += np.log(np.sin(x) / 2.0) + lj
        out += ss.norm(1.3, 0.4).logpdf(th[5])
        x, lj = two_sided(2.0, 7.5, th[6]);          out += ss.uniform(2.0, 5.5).logpdf(x) + lj
        out += ss.norm(50100.0, 30.0).logpdf(th[7])
        return out
    for W, small in ((5, None), (700, 0)):
        if small is not None:
            model.ln_like._check(model.ln_like.lib.octo_ctx_set_small_batch(model.ln_like._ctx, small), "set")
        th = rng.normal(0.0, 1.5, (model.D, W)); th[7] = rng.normal(50100.0, 30.0, W)
        lp, g = model.logdensity_and_gradient(th)
        r = ref(th)
        const = lp - r                                   # the likelihood of the one row: −log(2π σ²) up to 1e-18 of curvature
        assert np.all(np.abs(const - const[0]) < 1e-9), np.abs(const - const[0]).max()
        assert abs(const[0] + np.log(2 * np.pi * 1e18)) < 1e-9
        h = 1e-5
        for k in range(model.D):
            tp_, tm_ = th.copy(), th.copy(); tp_[k] += h; tm_[k] -= h
            gnum = (ref(tp_) - ref(tm_)) / (2 * h)
            assert np.all(np.abs(g[k] - gnum) <= 1e-6 * np.maximum(1.0, np.abs(gnum))), (k, np.abs(g[k] - gnum).max())
    model.close()


def _rv_case():
    """Absolute RVs of a star whose planet has no mass: the model is the offset (+ trend·basis) alone, the density scipy's normal with σ² + jitter²."""
    rng = np.random.default_rng(15)
    n = 40
    t = np.sort(50000.0 + rng.uniform(0, 900, n))
    rv, s = rng.normal(12.0, 6.0, n), rng.uniform(1.0, 5.0, n)
    basis = (t - 50450.0) / 365.25
    obs = [dict(kind=2, planet=-1, epoch=t, y1=rv, y2=None, s1=s, s2=None, cor=None, extra=basis)]
    planets = [dict(orbit_kind=0, has_mass=True)]
    el = np.array([5.0, 0.2, 1.0, 0.5, 2.0, 50100.0, 1.1, 30.0, 0.0])[:, None]       # mass = 0
    off, jit, trend = 11.0, 2.5, -0.7
    nuis = np.array([[off], [jit], [trend]])
    ref = ss.norm(off + trend * basis, np.hypot(s, jit)).logpdf(rv).sum()
    # exact derivatives of the normal: offset Σ r/v, trend Σ r·basis/v, jitter Σ j (r²/v² − 1/v)
    r = rv - off - trend * basis; v = s * s + jit * jit
    g = np.array([np.sum(r / v), np.sum(jit * (r * r / (v * v) - 1 / v)), np.sum(r * basis / v)])
    return obs, planets, el, nuis, ref, g


def test_rv_density_vs_scipy_normal(oracle):
    """rv-absolute.jl:172-204: logpdf(Normal(0, √(σ² + jitter²)), rv − offset − trend) summed over the rows — scipy's normal, its derivatives in closed form."""
    obs, planets, el, nuis, ref, g = _rv_case()
    ll, _, g_nu = oracle.oracle_eval(obs, planets, el, nuis, grad=True)
    assert abs(ll[0] - ref) < 1e-12 * abs(ref)
    assert np.all(np.abs(g_nu[:, 0] - g) < 1e-10 * np.abs(g).max())


@pytest.mark.gpu
def test_gpu_rv_density_vs_scipy_normal(oracle):
    import gpu_binding
    obs, planets, el, nuis, ref, g = _rv_case()
    for small, W in ((None, 1), (0, 70)):
        ll, _, g_nu = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-12 * abs(ref))
        assert np.all(np.abs(g_nu - g[:, None]) < 1e-10 * np.abs(g).max())


def _seppa_case(oracle):
    """Separation / position-angle rows (relative-astrometry.jl:192-202): PA = atan(ra, dec) east of north, its residual wrapped into (−π, π]."""
    rng = np.random.default_rng(16)
    el = np.array([9.0, 0.4, 0.9, 0.7, 2.5, 50050.0, 1.3, 45.0, 0.0])
    t = np.array([50000.0, 50300.0, 50800.0, 51500.0, 52400.0])
    sol = [oracle.oracle_orbitsolve(el, tj) for tj in t]
    ra_m, dec_m = np.array([q["raoff"] for q in sol]), np.array([q["decoff"] for q in sol])
    sep_m, pa_m = np.hypot(ra_m, dec_m), np.arctan2(ra_m, dec_m)
    s_pa, s_sep = rng.uniform(0.005, 0.03, 5), rng.uniform(2.0, 8.0, 5)
    pa = pa_m + rng.normal(0, 0.01, 5) + 2 * np.pi * np.array([0, 1, -1, 2, 0])      # data on other branches of the angle
    sep = sep_m + rng.normal(0, 4.0, 5)
    obs = [dict(kind=1, planet=0, epoch=t, y1=pa, y2=sep, s1=s_pa, s2=s_sep, cor=None, extra=None)]
    dpa = np.angle(np.exp(1j * (pa - pa_m)))
    ref = ss.norm(0, s_pa).logpdf(dpa).sum() + ss.norm(0, s_sep).logpdf(sep - sep_m).sum()
    return obs, [dict(orbit_kind=0, has_mass=False)], el[:, None], ref


def test_seppa_density_vs_scipy(oracle):
    obs, planets, el, ref = _seppa_case(oracle)
    ll, _, _ = oracle.oracle_eval(obs, planets, el, None, grad=False)
    assert abs(ll[0] - ref) < 1e-11 * abs(ref), (ll[0], ref)


@pytest.mark.gpu
def test_gpu_seppa_density_vs_scipy(oracle):
    import gpu_binding
    obs, planets, el, ref = _seppa_case(oracle)
    for small, W in ((None, 1), (0, 70)):
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), None, grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-11 * abs(ref))


def _barycentre_case(oracle):
    """RA/Dec of the OUTER of two planets: the model adds the reflex of the star about the inner planet, (m_in · mjup2msol / M) · (ra, dec)_inner, for
    planets strictly inside the observed one (relative-astrometry.jl:104-142); rows of the INNER planet get nothing from the outer one."""
    c = oracle.oracle_consts()
    rng = np.random.default_rng(17)
    inner = np.array([2.0, 0.1, 0.8, 1.0, 2.0, 50100.0, 1.2, 50.0, 8.0])
    outer = np.array([11.0, 0.3, 0.9, 0.4, 2.2, 50500.0, 1.2, 50.0, 3.0])
    t = np.array([50000.0, 50200.0, 50700.0, 51900.0])
    so = [oracle.oracle_orbitsolve(outer, tj) for tj in t]; si = [oracle.oracle_orbitsolve(inner, tj) for tj in t]
    f = inner[8] * c.mjup2msol / inner[6]
    ra_o = np.array([q["raoff"] for q in so]) + f * np.array([q["raoff"] for q in si])
    dec_o = np.array([q["decoff"] for q in so]) + f * np.array([q["decoff"] for q in si])
    ra_i, dec_i = np.array([q["raoff"] for q in si]), np.array([q["decoff"] for q in si])
    s = np.array([3.0, 4.0, 2.0, 5.0])
    d_o = (ra_o + rng.normal(0, 3, 4), dec_o + rng.normal(0, 3, 4)); d_i = (ra_i + rng.normal(0, 3, 4), dec_i + rng.normal(0, 3, 4))
    obs = [dict(kind=0, planet=1, epoch=t, y1=d_o[0], y2=d_o[1], s1=s, s2=s, cor=None, extra=None),
           dict(kind=0, planet=0, epoch=t, y1=d_i[0], y2=d_i[1], s1=s, s2=s, cor=None, extra=None)]
    ref = (ss.norm(ra_o, s).logpdf(d_o[0]) + ss.norm(dec_o, s).logpdf(d_o[1]) + ss.norm(ra_i, s).logpdf(d_i[0]) + ss.norm(dec_i, s).logpdf(d_i[1])).sum()
    return obs, [dict(orbit_kind=0, has_mass=True)] * 2, np.concatenate([inner, outer])[:, None], ref


def test_inner_barycentre_term_vs_per_planet_solves(oracle):
    obs, planets, el, ref = _barycentre_case(oracle)
    ll, _, _ = oracle.oracle_eval(obs, planets, el, None, grad=False)
    assert abs(ll[0] - ref) < 1e-11 * abs(ref), (ll[0], ref)


@pytest.mark.gpu
def test_gpu_inner_barycentre_term_vs_per_planet_solves(oracle):
    import gpu_binding
    obs, planets, el, ref = _barycentre_case(oracle)
    for small, W in ((None, 1), (0, 70)):
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), None, grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-11 * abs(ref))


def _relrv_case(oracle):
    """Relative RVs of a planet (OctofitterRadialVelocity/src/rv-relative.jl:128-160): offset + radvel(sol) of the planet itself, Normal(σ² + jitter²) —
    radvel from octo_oracle_orbitsolve (itself held to d(z)/dt above)."""
    rng = np.random.default_rng(18)
    el = np.array([7.0, 0.25, 1.1, 0.6, 2.1, 50080.0, 1.15, 35.0, 4.0])
    t = np.sort(50000.0 + rng.uniform(0, 3000, 12))
    rvm = np.array([oracle.oracle_orbitsolve(el, tj)["radvel"] for tj in t])
    s = rng.uniform(20.0, 80.0, 12)
    off, jit = 15.0, 30.0
    rv = rvm + off + rng.normal(0, 60.0, 12)
    obs = [dict(kind=4, planet=0, epoch=t, y1=rv, y2=None, s1=s, s2=None, cor=None, extra=None)]
    ref = ss.norm(rvm + off, np.hypot(s, jit)).logpdf(rv).sum()
    return obs, [dict(orbit_kind=0, has_mass=True)], el[:, None], np.array([[off], [jit], [0.0]]), ref


def test_relative_rv_vs_scipy(oracle):
    obs, planets, el, nuis, ref = _relrv_case(oracle)
    ll, _, _ = oracle.oracle_eval(obs, planets, el, nuis, grad=False)
    assert abs(ll[0] - ref) < 1e-11 * abs(ref), (ll[0], ref)


@pytest.mark.gpu
def test_gpu_relative_rv_vs_scipy(oracle):
    import gpu_binding
    obs, planets, el, nuis, ref = _relrv_case(oracle)
    for small, W in ((None, 1), (0, 70)):
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-11 * abs(ref))


def _absrv_planet_case(oracle):
    """Absolute RVs of the star with a massive planet: momentum conservation in the two-body problem whose M is the TOTAL mass gives
    v_star = −(m / M) · v_relative, i.e. offset − (m · mjup2msol / M) · radvel(sol) (OctofitterRadialVelocity/src/rv-absolute.jl:143-155)."""
    c = oracle.oracle_consts()
    rng = np.random.default_rng(19)
    el = np.array([3.0, 0.3, 1.2, 0.9, 1.0, 50200.0, 1.05, 40.0, 6.0])
    t = np.sort(50000.0 + rng.uniform(0, 2500, 15))
    v = np.array([oracle.oracle_orbitsolve(el, tj)["radvel"] for tj in t])
    f = el[8] * c.mjup2msol / el[6]
    s = rng.uniform(2.0, 6.0, 15)
    off, jit = -4.0, 1.5
    rv = off - f * v + rng.normal(0, 4.0, 15)
    obs = [dict(kind=2, planet=-1, epoch=t, y1=rv, y2=None, s1=s, s2=None, cor=None, extra=None)]
    ref = ss.norm(off - f * v, np.hypot(s, jit)).logpdf(rv).sum()
    return obs, [dict(orbit_kind=0, has_mass=True)], el[:, None], np.array([[off], [jit], [0.0]]), ref


def test_stellar_reflex_rv_vs_scipy(oracle):
    obs, planets, el, nuis, ref = _absrv_planet_case(oracle)
    ll, _, _ = oracle.oracle_eval(obs, planets, el, nuis, grad=False)
    assert abs(ll[0] - ref) < 1e-11 * abs(ref), (ll[0], ref)


@pytest.mark.gpu
def test_gpu_stellar_reflex_rv_vs_scipy(oracle):
    import gpu_binding
    obs, planets, el, nuis, ref = _absrv_planet_case(oracle)
    for small, W in ((None, 1), (0, 70)):
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-11 * abs(ref))


def _tperi_case(oracle):
    """θ_at_epoch_to_tperi(θ, epoch; M, e, a, i, ω, Ω) (src/parameterizations.jl:6-69) by its MEANING instead of its formula: the planet's position angle at
    `epoch` is θ. For the D = 11 reference test model (test/integration/sampling.jl:29-64; descriptors of tests/golden/model.json) a time t* with PA(t*) = θ
    is found on an orbit with an arbitrary tp by scipy.optimize.brentq over octo_oracle_orbitsolve; the position there is what the model must predict at the
    epoch. One RA/Dec row holds exactly that position (σ = 0.01 mas): its residual is zero iff tp means what it should, and moving the datum by (3σ, 4σ) must
    cost exactly 12.5 in log-posterior."""
    import json
    from pathlib import Path
    case = json.loads((Path(__file__).resolve().parent / "golden" / "model.json").read_text())["cases"][0]
    M, plx, a, e, inc, w, O, th, epoch = 1.2, 50.0, 12.0, 0.3, 0.9, 0.7, 2.0, 1.1, 50000.0
    el = np.array([a, e, inc, w, O, 50000.0, M, plx, 0.0])
    period = np.sqrt(a ** 3 / M) * oracle.oracle_consts().kepler_year_to_julian_day
    pa = lambda t: (lambda q: np.arctan2(q["raoff"], q["decoff"]))(oracle.oracle_orbitsolve(el, t))
    f = lambda t: np.angle(np.exp(1j * (pa(t) - th)))
    grid = np.linspace(50000.0, 50000.0 + period, 2001)
    fv = np.array([f(t) for t in grid])
    k = next(i for i in range(2000) if fv[i] * fv[i + 1] < 0 and abs(fv[i] - fv[i + 1]) < 1.0)      # a sign change that is not the branch cut
    ts = scipy.optimize.brentq(f, grid[k], grid[k + 1], xtol=1e-11)
    q = oracle.oracle_orbitsolve(el, ts)
    logit = lambda p: np.log(p / (1 - p))
    theta_t = np.array([np.log(M - 0.1), np.log(plx - 0.1), logit(a / 100.0), logit(e / 0.99), logit(inc / np.pi),
                        np.cos(w), np.sin(w), np.cos(O), np.sin(O), np.cos(th), np.sin(th)])[:, None]
    sig = 0.01
    def obs(dra, ddec):
        return [dict(kind=0, planet=0, epoch=np.array([epoch]), y1=np.array([q["raoff"] + dra]), y2=np.array([q["decoff"] + ddec]),
                     s1=np.array([sig]), s2=np.array([sig]), cor=None, extra=None)]
    return case, obs, theta_t, sig


def test_tperi_means_position_angle_at_epoch(oracle):
    case, obs, theta_t, sig = _tperi_case(oracle)
    pr, es = oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"])
    lp1, _ = oracle.oracle_model_logpost(obs(0.0, 0.0), case["planets"], pr, es, None, theta_t)
    lp2, _ = oracle.oracle_model_logpost(obs(3 * sig, 4 * sig), case["planets"], pr, es, None, theta_t)
    assert np.isfinite(lp1[0]) and abs((lp1[0] - lp2[0]) - 12.5) < 1e-5, (lp1, lp2)


def _d11_model(pkg, o):
    """The reference test model (test/integration/sampling.jl:29-64) over a one-row table, through the mirror."""
    table = dict(epoch=o[0]["epoch"], ra=o[0]["y1"], dec=o[0]["y2"], σ_ra=o[0]["s1"], σ_dec=o[0]["s2"], cor=[0.0])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromLikelihood(table, name="one_row")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    return pkg.LogDensityModel(pkg.System(name="TestSys", companions=[b], observations=[],
                               variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))


@pytest.mark.gpu
def test_gpu_tperi_means_position_angle_at_epoch(pkg, oracle):
    """The same through the device-side parameterisation (the mirror's LogDensityModel of the reference test model, one θ_t: k_small<MODEL>; 70: k_model_fwd)."""
    case, obs, theta_t, sig = _tperi_case(oracle)
    m1, m2 = _d11_model(pkg, obs(0.0, 0.0)), _d11_model(pkg, obs(3 * sig, 4 * sig))
    try:
        for W in (1, 70):
            th = np.repeat(theta_t, W, axis=1)
            if W > 1:
                for m in (m1, m2):
                    m.ln_like._check(m.ln_like.lib.octo_ctx_set_small_batch(m.ln_like._ctx, 0), "set")
            d = m1.logdensity(th) - m2.logdensity(th)
            assert np.all(np.isfinite(d)) and np.all(np.abs(d - 12.5) < 1e-5), d
    finally:
        m1.close(); m2.close()


def _marg_case():
    """MarginalizedStarAbsoluteRVObs (OctofitterRadialVelocity/src/rv-absolute-margin.jl:140-185, "math from the Orvara paper"): the zero point integrated out
    under a flat prior. Its ll = −Σ log(2π var) − (C − B²/4A) − log A is TWICE the log of that integral, minus log 2π (the reference keeps Orvara's −χ² scale:
    the 'missing ½' of DESIGN.md). scipy.integrate.quad does the integral."""
    import scipy.integrate
    rng = np.random.default_rng(20)
    n = 14
    t = np.sort(50000.0 + rng.uniform(0, 1200, n))
    s = rng.uniform(2.0, 6.0, n); jit = 1.8
    rv = rng.normal(7.0, 5.0, n)
    var = s * s + jit * jit
    mu_hat = np.sum(rv / var) / np.sum(1 / var); width = 1 / np.sqrt(np.sum(1 / var))
    f = lambda g: np.exp(np.sum(ss.norm(g, np.sqrt(var)).logpdf(rv)) - c0)
    c0 = np.sum(ss.norm(mu_hat, np.sqrt(var)).logpdf(rv))                   # scale out the peak so that quad sees O(1) numbers
    integral, err = scipy.integrate.quad(f, mu_hat - 12 * width, mu_hat + 12 * width, epsabs=0, epsrel=1e-13, limit=200)
    log_marginal = np.log(integral) + c0
    obs = [dict(kind=3, planet=-1, epoch=t, y1=rv, y2=None, s1=s, s2=None, cor=None, extra=None)]
    el = np.array([5.0, 0.2, 1.0, 0.5, 2.0, 50100.0, 1.1, 30.0, 0.0])[:, None]        # a massless planet: the model is the zero point alone
    return obs, [dict(orbit_kind=0, has_mass=True)], el, np.array([[0.0], [jit], [0.0]]), 2.0 * log_marginal - np.log(2 * np.pi)


def test_marginalised_rv_vs_scipy_quad(oracle):
    obs, planets, el, nuis, ref = _marg_case()
    ll, _, _ = oracle.oracle_eval(obs, planets, el, nuis, grad=False)
    assert abs(ll[0] - ref) < 1e-10 * abs(ref), (ll[0], ref)


@pytest.mark.gpu
def test_gpu_marginalised_rv_vs_scipy_quad(oracle):
    import gpu_binding
    obs, planets, el, nuis, ref = _marg_case()
    for small, W in ((None, 1), (0, 70)):
        ll, _, _ = gpu_binding.gpu_eval(obs, planets, np.repeat(el, W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)
        assert np.all(np.abs(ll - ref) < 1e-10 * abs(ref))


def _ti_case():
    """A Campbell orbit and the SAME orbit through the textbook Thiele-Innes constants (A, B, F, G in mas = a·plx·(…)): dec = A·X + F·Y, ra = B·X + G·Y with
    X = cos E − e, Y = √(1−e²) sin E. The ThieleInnesOrbit planet recovers a = α/plx from them (src/parameterizations.jl:14-19) for its period."""
    a, e, inc, w, O, tp, M, plx = 8.0, 0.35, 1.0, 0.8, 2.3, 50040.0, 1.25, 42.0
    ca, sa = np.cos, np.sin
    A = a * plx * (ca(w) * ca(O) - sa(w) * sa(O) * ca(inc)); B = a * plx * (ca(w) * sa(O) + sa(w) * ca(O) * ca(inc))
    F = a * plx * (-sa(w) * ca(O) - ca(w) * sa(O) * ca(inc)); G = a * plx * (-sa(w) * sa(O) + ca(w) * ca(O) * ca(inc))
    camp = np.array([a, e, inc, w, O, tp, M, plx, 0.0]); ti = np.array([A, e, B, F, G, tp, M, plx, 0.0])
    return camp, ti


def test_thiele_innes_orbit_is_the_campbell_orbit(oracle):
    camp, ti = _ti_case()
    for t in (50000.0, 50500.0, 52345.6):
        qc, qt = oracle.oracle_orbitsolve(camp, t, orbit_kind=0), oracle.oracle_orbitsolve(ti, t, orbit_kind=2)
        assert abs(qc["raoff"] - qt["raoff"]) < 1e-10 * abs(camp[0] * camp[7]) and abs(qc["decoff"] - qt["decoff"]) < 1e-10 * abs(camp[0] * camp[7])


@pytest.mark.gpu
def test_gpu_thiele_innes_orbit_is_the_campbell_orbit(oracle):
    import gpu_binding
    camp, ti = _ti_case()
    rng = np.random.default_rng(22)
    t = np.sort(50000.0 + rng.uniform(0, 4000, 20))
    obs = [dict(kind=0, planet=0, epoch=t, y1=rng.normal(0, 300, 20), y2=rng.normal(0, 300, 20), s1=np.full(20, 5.0), s2=np.full(20, 7.0), cor=rng.uniform(-0.5, 0.5, 20), extra=None)]
    for small, W in ((None, 1), (0, 70)):
        ll_c, _, _ = gpu_binding.gpu_eval(obs, [dict(orbit_kind=0, has_mass=False)], np.repeat(camp[:, None], W, axis=1), None, grad=True, small_batch=small)
        ll_t, _, _ = gpu_binding.gpu_eval(obs, [dict(orbit_kind=2, has_mass=False)], np.repeat(ti[:, None], W, axis=1), None, grad=True, small_batch=small)
        assert np.all(np.abs(ll_c - ll_t) < 1e-10 * np.abs(ll_c))


def _unit_length_ref(r):
    """What scaling ONE UniformCircular pair (x, y) = r·(cos, sin) changes in the log-posterior: its two Normal(0, 1) priors and its UnitLengthPrior
    logpdf(LogNormal(log 1, 0.1), √(x² + y²)) (src/variables.jl:279-323) — the angle, hence the orbit and the likelihood, stay what they were."""
    return (ss.lognorm(s=0.1, scale=1.0).logpdf(r) - ss.lognorm(s=0.1, scale=1.0).logpdf(1.0)) - 0.5 * (r * r - 1.0)


def test_unit_length_prior_vs_scipy_lognorm(oracle):
    case, obs, theta_t, sig = _tperi_case(oracle)
    pr, es = oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"])
    lp0, _ = oracle.oracle_model_logpost(obs(0.3, -0.2), case["planets"], pr, es, None, theta_t)
    for pair in ((5, 6), (7, 8), (9, 10)):                       # ω, Ω, θ (the one tp is derived from)
        for r in (0.8, 1.0, 1.17):
            th = theta_t.copy(); th[pair[0]] *= r; th[pair[1]] *= r
            lp, _ = oracle.oracle_model_logpost(obs(0.3, -0.2), case["planets"], pr, es, None, th)
            assert abs((lp[0] - lp0[0]) - _unit_length_ref(r)) < 1e-9 * max(1.0, abs(lp0[0])), (pair, r, lp[0] - lp0[0], _unit_length_ref(r))


@pytest.mark.gpu
def test_gpu_unit_length_prior_vs_scipy_lognorm(pkg, oracle):
    case, obs, theta_t, sig = _tperi_case(oracle)
    m = _d11_model(pkg, obs(0.3, -0.2))
    try:
        for W in (1, 70):
            if W > 1:
                m.ln_like._check(m.ln_like.lib.octo_ctx_set_small_batch(m.ln_like._ctx, 0), "set")
            lp0 = m.logdensity(np.repeat(theta_t, W, axis=1))
            for pair in ((5, 6), (7, 8), (9, 10)):
                for r in (0.8, 1.17):
                    th = np.repeat(theta_t, W, axis=1); th[pair[0]] *= r; th[pair[1]] *= r
                    d = m.logdensity(th) - lp0
                    assert np.all(np.abs(d - _unit_length_ref(r)) < 1e-9 * np.maximum(1.0, np.abs(lp0))), (pair, r, d[0], _unit_length_ref(r))
    finally:
        m.close()



# ---------------------------------------------------------------------------------------------------------------- round 6: the last two unpinned rules
def _oneil_case(oracle):
    """The O'Neil et al. (2019) observable-based prior (src/likelihoods/prior-observable.jl:78-137) BY ITS MEANING. The reference adds
    2·log(Σ_j |t_j| · ∛P / √(1−e²)) with t_j = 3M(e + cos E) + 2(−2 + e² + e cos E) sin E, P the period in Julian years. That expression is −3 times
    the Jacobian determinant ∂(x, y)/∂(P, e) of the orbital-plane position (x, y) = P^(2/3)·(cos E − e, √(1−e²) sin E) at the epoch — the
    observables against the parameters the prior is meant to flatten — so the term is 2·log(3·Σ_j |det J_j|). Here det J_j comes from central
    differences at 40 digits (mpmath) of (x, y) written out from Kepler's equation alone; epochs within half a period of tp, so that the
    reference's wrapped mean anomaly is the unwrapped one."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    c = oracle.oracle_consts()
    rng = np.random.default_rng(23)
    cases = []
    for _ in range(6):
        a, e, M_tot = rng.uniform(2, 30), rng.uniform(0.05, 0.85), rng.uniform(0.6, 2.0)
        period_d = np.sqrt(a ** 3 / M_tot) * c.kepler_year_to_julian_day
        tp = 50000.0 + rng.uniform(0, 500)
        t = tp + period_d * rng.uniform(-0.45, 0.45, 5)
        el = np.array([a, e, rng.uniform(0.2, 2.9), rng.uniform(0, 6.28), rng.uniform(0, 6.28), tp, M_tot, rng.uniform(20, 60), 0.0])

        def xy(P, ee, tau):
            Mm = 2 * mp.pi * tau / P
            E = Mm + ee * mp.sin(Mm)
            for _ in range(80):
                E = E - (E - ee * mp.sin(E) - Mm) / (1 - ee * mp.cos(E))
            s = P ** (mp.mpf(2) / 3)
            return s * (mp.cos(E) - ee), s * mp.sqrt(1 - ee * ee) * mp.sin(E)
        P_yr = mp.mpf(float(period_d)) / mp.mpf("365.25")
        h = mp.mpf("1e-18")
        tot = mp.mpf(0)
        for tj in t:
            tau = mp.mpf(float(tj - tp)) / mp.mpf("365.25")
            xp, yp = xy(P_yr + h, mp.mpf(float(e)), tau); xm, ym = xy(P_yr - h, mp.mpf(float(e)), tau)
            xe, ye = xy(P_yr, mp.mpf(float(e)) + h, tau); xf, yf = xy(P_yr, mp.mpf(float(e)) - h, tau)
            det = ((xp - xm) * (ye - yf) - (xe - xf) * (yp - ym)) / (4 * h * h)
            tot += abs(det)
        ref = float(2 * mp.log(3 * tot))
        n = t.size
        cols = dict(planet=0, epoch=t, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=None, extra=None)
        cases.append((el, cols, ref))
    return cases


def test_oneil_prior_is_the_jacobian_of_the_orbital_plane_position(oracle):
    planets = [dict(orbit_kind=0, has_mass=False)]
    for el, cols, ref in _oneil_case(oracle):
        ll5, _, _ = oracle.oracle_eval([dict(kind=5, **cols)], planets, el[:, None], None, grad=False)
        ll0, _, _ = oracle.oracle_eval([dict(kind=0, **cols)], planets, el[:, None], None, grad=False)
        assert abs((ll5[0] - ll0[0]) - ref) < 1e-9 * max(1.0, abs(ref)), (ll5[0] - ll0[0], ref)


@pytest.mark.gpu
def test_gpu_oneil_prior_is_the_jacobian_of_the_orbital_plane_position(oracle):
    import gpu_binding
    planets = [dict(orbit_kind=0, has_mass=False)]
    for el, cols, ref in _oneil_case(oracle):
        for small, W in ((None, 1), (0, 70)):
            ll5, _, _ = gpu_binding.gpu_eval([dict(kind=5, **cols)], planets, np.repeat(el[:, None], W, axis=1), None, grad=True, small_batch=small)
            ll0, _, _ = gpu_binding.gpu_eval([dict(kind=0, **cols)], planets, np.repeat(el[:, None], W, axis=1), None, grad=True, small_batch=small)
            assert np.all(np.abs((ll5 - ll0) - ref) < 1e-9 * max(1.0, abs(ref))), ((ll5 - ll0)[0], ref)


def _hgca_case(oracle):
    """HGCAInstantaneousObs (src/likelihoods/hgca.jl:155-400) BY ITS MEANING: the model's proper motions are those of the STAR — the reflex of the
    companion about the barycentre, −(m·mjup2msol/M)·(the companion's sky offset) — and `pmra` / `pmdec` are the TIME DERIVATIVES of that offset in
    mas per Julian year. With one RA and one Dec row per mission the three model values ln_like compares with the catalogue are
        hip: pm(t_h) + pm_sys,    hg: (R(t_g) − R(t_h)) / (t_g − t_h)·365.25 + pm_sys,    gaia: pm(t_g) + pm_sys      (hgca.jl:160-215, 288-376)
    with R(t) the star's offset [mas] (ln_like reads simulate's INDIVIDUAL values; the vectors μ_h, μ_hg, μ_g re-framed by the Gaia-epoch reflex motion,
    hgca.jl:380-390, are returned next to them and not used by it). R comes from the oracle's orbit solution (pinned against scipy's root finder and the reference's tutorial
    table), pm from central differences of it — not from the closed form the restatement recalls. The model values are read off the likelihood as
    the vertices of its parabolas in the catalogue values (three evaluations each, correlations 0)."""
    c = oracle.oracle_consts()
    rng = np.random.default_rng(29)
    out = []
    for _ in range(4):
        a, e = rng.uniform(4, 25), rng.uniform(0.0, 0.7)
        el = np.array([a, e, rng.uniform(0.2, 2.9), rng.uniform(0, 6.28), rng.uniform(0, 6.28), 50000.0 + rng.uniform(0, 9000), rng.uniform(0.7, 1.8), rng.uniform(15, 70),
                       rng.uniform(5, 60)])
        f = -el[8] * c.mjup2msol / el[6]
        t_h = np.array([48348.0 + rng.uniform(-40, 40), 48348.0 + rng.uniform(-40, 40)])      # RA row, Dec row of Hipparcos
        t_g = np.array([57388.0 + rng.uniform(-40, 40), 57388.0 + rng.uniform(-40, 40)])
        R = lambda tt, k: f * oracle.oracle_orbitsolve(el, tt)["raoff" if k == 0 else "decoff"]
        period_d = np.sqrt(a ** 3 / el[6]) * c.kepler_year_to_julian_day
        h = 2e-4 * period_d * (1 - e) ** 1.5
        pm = lambda tt, k: (R(tt + h, k) - R(tt - h, k)) / (2 * h) * 365.25                    # mas per Julian year
        pm_sys = np.array([rng.normal(0, 20), rng.normal(0, 20)])
        mu_h = np.array([pm(t_h[k], k) + pm_sys[k] for k in (0, 1)])
        mu_hg = np.array([(R(t_g[k], k) - R(t_h[k], k)) / (t_g[k] - t_h[k]) * 365.25 + pm_sys[k] for k in (0, 1)])
        mu_g = np.array([pm(t_g[k], k) + pm_sys[k] for k in (0, 1)])
        obs = dict(kind=7, planet=-1, epoch=np.array([t_h[0], t_h[1], t_g[0], t_g[1]]), y1=np.array([0., 1, 0, 1]), y2=np.array([0., 0, 1, 1]), s1=None, s2=None, cor=None)
        nuis = np.array([[pm_sys[0]], [pm_sys[1]], [0.0]])
        scale = max(np.abs(pm(t_h[0], 0)), np.abs(pm(t_h[1], 1)), 1e-3)
        out.append((el, obs, nuis, np.concatenate([mu_h, mu_hg, mu_g]), scale))
    return out


def _hgca_model_values(evaluate, obs, sig=0.7):
    """The six model values (μ_hip, μ_hg, μ_gaia: ra, dec each) from the likelihood: vertex of ll as a function of each catalogue value."""
    base = np.array([0.0, 0.0, sig, sig, 0.0] * 3)
    vals = []
    for k in range(3):
        for ax in (0, 1):
            f = []
            for d in (-1.0, 0.0, 1.0):
                ex = base.copy(); ex[5 * k + ax] = d
                f.append(evaluate(dict(obs, extra=ex)))
            vals.append(-(f[2] - f[0]) / (2 * (f[2] - 2 * f[1] + f[0])))
    return np.array(vals)


def test_hgca_proper_motions_are_time_derivatives_of_the_stars_reflex_position(oracle):
    planets = [dict(orbit_kind=0, has_mass=True)]
    for el, obs, nuis, ref, scale in _hgca_case(oracle):
        got = _hgca_model_values(lambda o: oracle.oracle_eval([o], planets, el[:, None], nuis, grad=False)[0][0], obs)
        assert np.all(np.abs(got - ref) < 2e-6 * scale + 1e-9 * np.abs(ref)), (got, ref)


@pytest.mark.gpu
def test_gpu_hgca_proper_motions_are_time_derivatives_of_the_stars_reflex_position(oracle):
    import gpu_binding
    planets = [dict(orbit_kind=0, has_mass=True)]
    for el, obs, nuis, ref, scale in _hgca_case(oracle):
        for small, W in ((None, 1), (0, 70)):
            ev = lambda o: gpu_binding.gpu_eval([o], planets, np.repeat(el[:, None], W, axis=1), np.repeat(nuis, W, axis=1), grad=True, small_batch=small)[0][W - 1]
            got = _hgca_model_values(ev, obs)
            assert np.all(np.abs(got - ref) < 2e-6 * scale + 1e-9 * np.abs(ref)), (small, got, ref)
