"""
GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
  * the committed golden vectors (50-digit oracle),
  * the CPU oracle (reference-order restatement) on the same seeded inputs,
  * size-independent properties at BASELINE.json's full sizes,
  * the reference tests' self-consistency properties, through the host mirror.

Tolerances. The north_star bar is "log-likelihood and gradient within 1e-8 relative"; the tests hold the HIP path
to 1000x tighter, close to what tests/parity_report.py measures on the GPU (ll ~5e-15, gradients ~5e-15 of scale):
  ll      |Δ| <= 1e-12 · max(|ll|, 1)          (1e-9 where the reference's marginalised-RV formula cancels)
  grad    |Δ| <= 1e-9 · |g| + 1e-13 · (S + max_w S),  S = Σ_rows |∂ll_row/∂θ|  (the rounding floor of a sum whose
          terms cancel; for oracle comparisons S is replaced by 10x the row maximum of |g| over the batch)
"""
import ctypes as C

import numpy as np
import pytest

import synth
from conftest import case_tables, rel_err
from test_oracle import grad_ok, northangle_scan, _northangle_tables

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[None, 0], ids=["auto", "throughput"])
def _kernel_family(request):
    """Every case of this module twice: with the library's own choice of kernel family (k_small for W·P <= 512) and forced onto
    the throughput kernels (lane = walker), so that mid-size batches keep checking both against the oracle."""
    import gpu_binding
    gpu_binding.DEFAULT_SMALL_BATCH = request.param
    yield
    gpu_binding.DEFAULT_SMALL_BATCH = None

LL_RTOL = 1e-12
G_RTOL = 1e-9
G_CANCEL = 1e-13


def _gpu():
    import gpu_binding
    return gpu_binding


def _cmp_oracle(name, ll, g_el, g_nu, ll_o, g_o, gn_o, ll_rtol=LL_RTOL, g_rtol=G_RTOL):
    ok_o = np.isfinite(ll_o)
    assert np.array_equal(np.isfinite(ll), ok_o), name
    assert np.all(np.isneginf(ll[~ok_o])), name
    err = rel_err(ll[ok_o], ll_o[ok_o], 1.0)
    assert np.all(err < ll_rtol), (name, "ll", err.max())
    if g_el is not None:
        assert np.all(g_el[:, ~ok_o] == 0.0), name
        scale = np.abs(g_o[:, ok_o]).max(axis=1, keepdims=True) * np.ones_like(g_o[:, ok_o]) * 10
        ok, worst = grad_ok(g_el[:, ok_o], g_o[:, ok_o], scale, rtol=g_rtol, cancel=G_CANCEL)
        assert ok, (name, "g_elems", worst)
    if g_nu is not None:
        scale = np.abs(gn_o[:, ok_o]).max(axis=1, keepdims=True) * np.ones_like(gn_o[:, ok_o]) * 10
        ok, worst = grad_ok(g_nu[:, ok_o], gn_o[:, ok_o], scale, rtol=g_rtol, cancel=G_CANCEL)
        assert ok, (name, "g_nuis", worst)


def test_golden_vectors_through_c_abi(golden):
    gb = _gpu()
    for case in golden["cases"]:
        obs, planets, elems, nuis = case_tables(case)
        ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
        ll_f, _, _ = gb.gpu_eval(obs, planets, elems, nuis, grad=False)
        assert np.array_equal(ll, ll_f), (case["name"], "forward-only and gradient launches disagree")
        ref_ll = np.asarray(case["ll"])
        has_marg = any(ob["kind"] == "RV_ABS_MARG" for ob in case["obs"])
        err = rel_err(ll, ref_ll, 1.0)
        assert np.all(err < (1e-9 if has_marg else LL_RTOL)), (case["name"], "ll", err.max())
        cancel = 1e-10 if has_marg else G_CANCEL
        rtol = G_RTOL
        if case["name"] == "F7_kepler_edges":
            rtol = 1e-9      # e = 0.999999: cond ~ 1/(1-e)^2 on d/de (the reference-order oracle is off by 2e-6 there,
            cancel = 1e-10   # the Thiele-Innes form of the kernel by 2e-11)
        ok, worst = grad_ok(g_el, case["g_elems"], case["s_elems"], rtol=rtol, cancel=cancel)
        assert ok, (case["name"], "g_elems", worst)
        if nuis is not None:
            ok, worst = grad_ok(g_nu, case["g_nuis"], case["s_nuis"], rtol=rtol, cancel=cancel)
            assert ok, (case["name"], "g_nuis", worst)


@pytest.mark.parametrize("n_epochs,n_walkers", [(1, 1), (7, 63), (96, 257), (513, 1000), (2048, 130)])
def test_astrometry_vs_oracle(oracle, n_epochs, n_walkers):
    gb = _gpu()
    cfg = synth.config_astrom(n_epochs=n_epochs, n_walkers=n_walkers, seed=100 + n_epochs)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    ll, g_el, _ = gb.gpu_eval(obs, planets, cfg["elems"], None, grad=True)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, cfg["elems"], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle(f"astrom {n_epochs}x{n_walkers}", ll, g_el, None, ll_o, g_o, None)


def test_all_kinds_two_planets_vs_oracle(oracle):
    gb = _gpu()
    cfg = synth.config_two_planet(n_astrom=300, n_rv=280, n_walkers=333, seed=5)
    a, r = cfg["astrom"], cfg["rv"]
    rng = np.random.default_rng(9)
    n = len(a["epoch"])
    pa = np.arctan2(a["ra"], a["dec"]); sep = np.hypot(a["ra"], a["dec"])
    obs = [
        dict(kind=0, planet=1, epoch=a["epoch"], y1=a["ra"], y2=a["dec"], s1=a["σ_ra"], s2=a["σ_dec"], cor=rng.uniform(-0.7, 0.7, n)),
        dict(kind=1, planet=1, epoch=a["epoch"] + 0.5, y1=pa, y2=sep, s1=np.full(n, 0.02), s2=a["σ_ra"], cor=None),
        dict(kind=0, planet=0, epoch=a["epoch"][:50], y1=a["ra"][:50] * 0.2, y2=a["dec"][:50] * 0.2, s1=a["σ_ra"][:50], s2=a["σ_dec"][:50], cor=None),
        dict(kind=4, planet=1, epoch=r["epoch"], y1=r["rv"] * 30, y2=None, s1=r["σ_rv"] * 10, s2=None, cor=None),
        dict(kind=2, planet=-1, epoch=r["epoch"], y1=r["rv"], y2=None, s1=r["σ_rv"], s2=None, cor=None),
        dict(kind=3, planet=-1, epoch=r["epoch"][::2], y1=r["rv"][::2] + 3.0, y2=None, s1=r["σ_rv"][::2] * 1.5, s2=None, cor=None),
    ]
    planets = [dict(orbit_kind=0, has_mass=True), dict(orbit_kind=0, has_mass=True)]
    W = cfg["n_walkers"]
    nuis = np.zeros((len(obs) * 3, W))
    for o in range(3):
        nuis[o * 3 + 0] = rng.uniform(0, 5, W); nuis[o * 3 + 1] = rng.normal(1, 0.01, W); nuis[o * 3 + 2] = rng.normal(0, 0.01, W)
    nuis[0, :40] = 0.0                      # jitter == 0 branch inside a nuisance batch
    for o in range(3, 6):
        nuis[o * 3 + 0] = rng.normal(10, 3, W); nuis[o * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    elems = cfg["elems"].copy()
    elems[9 + 0, :20] = elems[0, :20] * 0.5  # some walkers: "outer" planet inside the "inner" one
    for nz in (nuis, None):
        ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nz, grad=True)
        ll_f, _, _ = gb.gpu_eval(obs, planets, elems, nz, grad=False)
        assert np.array_equal(ll, ll_f)
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nz, grad=True)
        _cmp_oracle("all kinds", ll, g_el, g_nu, ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)   # marginalised-RV cancellation


def test_radvel_orbit_and_empty_table(oracle):
    gb = _gpu()
    rng = np.random.default_rng(3)
    W = 70
    ep = np.linspace(50000.0, 50200.0, 20)
    obs = [dict(kind=4, planet=0, epoch=ep, y1=rng.normal(0, 30, 20), y2=None, s1=np.full(20, 2.0), s2=None, cor=None),
           dict(kind=2, planet=-1, epoch=np.zeros(0), y1=np.zeros(0), y2=None, s1=np.zeros(0), s2=None, cor=None)]
    planets = [dict(orbit_kind=1, has_mass=True)]
    elems = np.stack([rng.uniform(0.5, 3, W), rng.uniform(0, 0.8, W), np.full(W, np.nan), rng.uniform(0, 6.28, W), np.full(W, np.nan),
                      50000 + rng.uniform(-100, 100, W), rng.normal(1, 0.05, W), np.full(W, np.nan), rng.uniform(0, 10, W)])
    nuis = np.stack([rng.normal(0, 5, W), rng.uniform(0.1, 3, W), np.zeros(W), rng.normal(0, 5, W), rng.uniform(0.1, 3, W), np.zeros(W)])
    ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
    assert np.all(np.isfinite(ll_o))
    _cmp_oracle("radvel", ll, g_el, g_nu, ll_o, g_o, gn_o)
    assert np.all(g_el[[2, 4, 7]] == 0.0)   # rows a RadialVelocityOrbit ignores carry no gradient (NaN inputs tolerated)


def test_plain_kep_orbit(pkg, oracle):
    """OCTO_ORBIT_KEP — the plain `KepOrbit` basis (src/likelihoods/system.jl:116-118 builds whatever basis the planet declares):
    K carries sin i, the KepOrbit invariants i mod π / Ω mod 2π apply, Ω and plx do not enter and carry no gradient, astrometry
    cannot be attached. Both kernel families (W = 70 and W = 9), inclinations outside [0, π) included."""
    gb = _gpu()
    rng = np.random.default_rng(13)
    ep = np.linspace(50000.0, 50700.0, 40)
    for W in (70, 9):
        obs = [dict(kind=4, planet=0, epoch=ep, y1=rng.normal(0, 900, 40), y2=None, s1=np.full(40, 50.0), s2=None, cor=None),
               dict(kind=2, planet=-1, epoch=ep + 2.0, y1=rng.normal(0, 30, 40), y2=None, s1=np.full(40, 2.0), s2=None, cor=None)]
        planets = [dict(orbit_kind=3, has_mass=True)]
        elems = np.stack([rng.uniform(0.5, 3, W), rng.uniform(0, 0.8, W), rng.uniform(-5.0, 8.0, W), rng.uniform(-7, 7, W), rng.uniform(-7, 7, W),
                          50000 + rng.uniform(-100, 100, W), rng.normal(1, 0.05, W), np.full(W, np.nan), rng.uniform(0.5, 10, W)])
        nuis = np.stack([rng.normal(0, 5, W), rng.uniform(0.1, 3, W), np.zeros(W), rng.normal(0, 5, W), rng.uniform(0.1, 3, W), np.zeros(W)])
        for nz in (nuis, None):
            ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nz, grad=True)
            ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nz, grad=True)
            assert np.all(np.isfinite(ll_o))
            _cmp_oracle("kep", ll, g_el, g_nu, ll_o, g_o, gn_o)
            assert np.all(g_el[[4, 7]] == 0.0) and np.any(g_el[2] != 0.0)
    # the mirror: basis="KepOrbit" takes RV tables, refuses astrometry (needs a distance)
    rvo = pkg.PlanetRelativeRVObs(dict(epoch=ep, rv=rng.normal(0, 900, 40), σ_rv=np.full(40, 50.0)), name="relrv")
    b = pkg.Planet(name="b", basis="KepOrbit", observations=[rvo])
    fn = pkg.make_ln_like(pkg.System(name="k", companions=[b]), dict(M=1.0, planets=dict(b=dict(a=1, e=0.1, i=1.0, ω=0.3, Ω=0.2, tp=5e4, mass=3.0))))
    θ = dict(M=1.05, planets=dict(b=dict(a=np.array([1.0, 1.3]), e=0.2, i=np.array([1.0, 1.0 + np.pi]), ω=0.4, Ω=1.0, tp=50010.0, mass=4.0)))
    ll2 = fn(θ)
    assert np.isfinite(ll2).all() and ll2[0] != ll2[1]
    fn.close()
    ast = pkg.PlanetRelAstromObs(dict(epoch=ep[:5], ra=np.ones(5), dec=np.ones(5), σ_ra=np.ones(5), σ_dec=np.ones(5)), name="ast")
    with pytest.raises(pkg.capi.OctoError):
        pkg.make_ln_like(pkg.System(name="k2", companions=[pkg.Planet(name="b", basis="KepOrbit", observations=[ast])]),
                         dict(M=1.0, planets=dict(b=dict(a=1, e=0.1, i=1.0, ω=0.3, Ω=0.2, tp=5e4))))


def test_dataset_validation(pkg):
    """octo_dataset_create refuses what would otherwise silently turn every walker into -Inf: σ <= 0, non-finite
    σ / epoch / measurement; octo_ofti_create likewise."""
    gb = _gpu()
    ep = np.linspace(50000.0, 50100.0, 5)
    ok = dict(kind=0, planet=0, epoch=ep, y1=np.ones(5), y2=np.ones(5), s1=np.ones(5), s2=np.ones(5), cor=None)
    planets = [dict(orbit_kind=0, has_mass=False)]
    gb.GpuPath([ok], planets).close()
    for col, val in (("s1", 0.0), ("s2", -1.0), ("s1", np.inf), ("epoch", np.nan), ("y2", np.inf)):
        bad = dict(ok); bad[col] = ok[col].copy(); bad[col][3] = val
        with pytest.raises(pkg.capi.OctoError) as ei:
            gb.GpuPath([bad], planets)
        assert ei.value.status == pkg.capi.OCTO_EINVAL and "row 3" in str(ei.value)
    with pytest.raises(pkg.capi.OctoError):
        pkg.OftiLinearSolver(ep, np.ones(5), np.ones(5), np.array([1, 1, 0.0, 1, 1]), np.ones(5), None, 100.0)


def test_invalid_walkers(oracle):
    gb = _gpu()
    cfg = synth.config_astrom(n_epochs=40, n_walkers=130, seed=11)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = cfg["elems"].copy()
    el[1, 3] = 1.0; el[1, 4] = -1e-3; el[0, 5] = 0.0; el[6, 6] = -1.0; el[7, 7] = 0.0; el[3, 8] = np.nan; el[5, 9] = np.inf; el[1, 64] = 1.5
    el[1, 70] = 1e30; el[0, 71] = np.inf; el[5, 72] = np.nan; el[1, 73] = -np.inf; el[6, 75] = 1e-30   # wild Kepler starters
    bad = [3, 4, 5, 6, 7, 8, 9, 64, 70, 71, 72, 73]
    ll, g_el, _ = gb.gpu_eval(obs, planets, el, None, grad=True)
    assert np.all(np.isneginf(ll[bad])) and np.all(g_el[:, bad] == 0.0)
    good = np.setdiff1d(np.arange(130), bad)
    assert np.all(np.isfinite(ll[good]))
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el, None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("invalid", ll, g_el, None, ll_o, g_o, None)


def test_full_size_properties(oracle):
    """BASELINE configs 2/3 (1 planet, 1e4 RA/Dec epochs × 1e4 walkers): determinism, additivity over a split of the
    table, and direct parity with the oracle for a seeded sample of walkers at the full epoch count."""
    gb = _gpu()
    cfg = synth.config_astrom()       # 1e4 × 1e4, rng 20260929+2
    t = cfg["table"]
    mk = lambda sl: dict(kind=0, planet=0, epoch=t["epoch"][sl], y1=t["ra"][sl], y2=t["dec"][sl], s1=t["σ_ra"][sl], s2=t["σ_dec"][sl], cor=None)
    planets = [dict(orbit_kind=0, has_mass=False)]
    full = gb.GpuPath([mk(slice(None))], planets)
    ll, g, _ = full.eval(cfg["elems"], None, grad=True)
    ll2, g2, _ = full.eval(cfg["elems"], None, grad=True)
    assert np.array_equal(ll, ll2) and np.array_equal(g, g2), "not deterministic"
    llf, _, _ = full.eval(cfg["elems"], None, grad=False)
    assert np.array_equal(ll, llf)
    full.close()
    assert np.all(np.isfinite(ll))
    # additivity: ln_like over the table == ln_like over two tables that split it (system.jl:93 sums observations)
    split = gb.GpuPath([mk(slice(0, 3777)), mk(slice(3777, None))], planets)
    ll_s, g_s, _ = split.eval(cfg["elems"], None, grad=True)
    split.close()
    assert np.all(rel_err(ll_s, ll, 1.0) < 1e-12)
    assert np.all(np.abs(g_s - g) <= 1e-11 * np.abs(g).max(axis=1, keepdims=True))
    # oracle parity for a seeded sample of walkers at full E
    idx = np.random.default_rng(0).choice(cfg["n_walkers"], 24, replace=False)
    ll_o, g_o, _ = oracle.oracle_eval([mk(slice(None))], planets, cfg["elems"][:, idx], None, grad=True,
                                      active=synth.active_mask(1, 1, mass=False, nuis=False), n_threads=0)
    _cmp_oracle("full size sample", ll[idx], g[:, idx], None, ll_o, g_o, None)


def test_full_size_nuisance_path(oracle):
    """The raw-σ path at full size (1e4 RA/Dec epochs × 4096 walkers with per-walker jitter, platescale, northangle,
    relative-astrometry.jl:234-252): the kernel sums the rows' log-variances as the log of a renormalised product, so
    check determinism, forward == gradient value, additivity over a split of the table (a different partition of that
    product) and the oracle on a seeded sample of walkers at the full epoch count."""
    gb = _gpu()
    cfg = synth.config_astrom(n_walkers=4096)
    W = 4096
    t = cfg["table"]
    mk = lambda sl: dict(kind=0, planet=0, epoch=t["epoch"][sl], y1=t["ra"][sl], y2=t["dec"][sl], s1=t["σ_ra"][sl], s2=t["σ_dec"][sl], cor=None)
    planets = [dict(orbit_kind=0, has_mass=False)]
    rng = np.random.default_rng(5)
    nu = np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)])
    nu[0, :7] = [0.0, 1e-12, 1e-3, 50.0, 1e3, 1e-300, 1e6]            # jitter extremes: product of 1e4 variances spans ±1e5 decades
    full = gb.GpuPath([mk(slice(None))], planets)
    ll, g, gn = full.eval(cfg["elems"], nu, grad=True)
    ll2, g2, gn2 = full.eval(cfg["elems"], nu, grad=True)
    assert np.array_equal(ll, ll2) and np.array_equal(g, g2) and np.array_equal(gn, gn2), "not deterministic"
    llf, _, _ = full.eval(cfg["elems"], nu, grad=False)
    assert np.array_equal(ll, llf)
    full.close()
    assert np.all(np.isfinite(ll))
    split = gb.GpuPath([mk(slice(0, 4321)), mk(slice(4321, None))], planets)
    nu2 = np.concatenate([nu, nu])
    ll_s, g_s, gn_s = split.eval(cfg["elems"], nu2, grad=True)
    split.close()
    assert np.all(rel_err(ll_s, ll, 1.0) < 1e-12)
    assert np.all(np.abs(g_s - g) <= 1e-11 * np.abs(g).max(axis=1, keepdims=True))
    assert np.all(np.abs(gn_s[:3] + gn_s[3:] - gn) <= 1e-11 * np.abs(gn).max(axis=1, keepdims=True))
    idx = np.concatenate([np.arange(7), np.random.default_rng(0).choice(np.arange(7, W), 17, replace=False)])
    ll_o, g_o, gn_o = oracle.oracle_eval([mk(slice(None))], planets, cfg["elems"][:, idx], nu[:, idx], grad=True,
                                         active=synth.active_mask(1, 1, mass=False, nuis=True), n_threads=0)
    _cmp_oracle("full size nuisance sample", ll[idx], g[:, idx], gn[:, idx], ll_o, g_o, gn_o)


def test_mirror_reference_properties(pkg, oracle):
    """test/unit/likelihoods.jl:32-95 and test/unit/distributions.jl:102-152 through the host mirror's
    PlanetRelAstromObs / Planet / System / make_ln_like surface, on the HIP path."""
    el, eps, seppa, radec = _northangle_tables(oracle)
    grid = np.linspace(-0.1, 0.1, 2001)
    best = {}
    for name, tab in (("seppa", dict(epoch=seppa["epoch"], sep=seppa["y2"], pa=seppa["y1"], σ_sep=seppa["s2"], σ_pa=seppa["s1"])),
                      ("radec", dict(epoch=radec["epoch"], ra=radec["y1"], dec=radec["y2"], σ_ra=radec["s1"], σ_dec=radec["s2"]))):
        obs = pkg.PlanetRelAstromObs(tab, name="inst")
        pl = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[obs])
        sys_ = pkg.System(name="northangle_test", companions=[pl], observations=[])
        θ = dict(plx=50.0, planets=dict(b=dict(M=1.2, a=15.0, e=0.2, i=0.6, ω=0.3, Ω=1.1, tp=50000.0, observations=dict(inst=dict(northangle=grid)))))
        ln_like = pkg.make_ln_like(sys_, θ)
        ll = ln_like(θ)
        best[name] = grid[np.argmax(ll)]
        θ0 = dict(θ); θ0["planets"] = dict(b=dict(θ["planets"]["b"], observations=dict(inst=dict(northangle=np.array([0.0, -0.0])))))
        z = ln_like(θ0)
        assert z[0] == z[1]
        ln_like.close()
    assert abs(best["seppa"] + eps) < 1e-3 and abs(best["radec"] + eps) < 1e-3
    # jitter sensitivity
    tbl = dict(epoch=[58000.0, 58200.0, 58400.0], ra=[100.0, 110.0, 120.0], dec=[100.0, 95.0, 90.0], σ_ra=[5.0] * 3, σ_dec=[5.0] * 3)
    obs = pkg.PlanetRelAstromLikelihood(tbl, name="d_radec")
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=(obs,))
    sys_ = pkg.System(name="jitter_prop_test", companions=(b,))
    θ = dict(M=1.0, plx=50.0, planets=dict(b=dict(a=10.0, e=0.2, i=0.5, ω=0.3, Ω=0.4, tp=58000.0, observations=dict(d_radec=dict(jitter=np.array([0.001, 300.0]))))))
    fn = pkg.make_ln_like(sys_, θ)
    ll, grads = fn.ln_like_and_grad(θ)
    assert ll[1] - ll[0] > 1
    assert grads["planets"]["b"]["observations"]["d_radec"]["jitter"].shape == (2,)
    fn.close()


def test_rv_trend_function_reference_test_model(pkg, oracle, golden):
    """The reference's "PlanetRelativeRV with offset and trend" test (OctofitterRadialVelocity/test/runtests.jl:168-231) through the
    mirror: PlanetRelativeRVObs(..., trend_function=(θ_obs, epoch) -> θ_obs.trend_slope * (epoch - ref_epoch)), a fixed circular
    RadialVelocityOrbit, free offset / jitter / trend_slope. The trend reaches the device as coefficient × basis column; checked against
    the 60-digit fixture F13 (the table is the fixture's), the oracle at a batch size of either kernel family, and the property the
    reference's test asserts (the posterior sits at offset ≈ 50, slope ≈ 0.1)."""
    case = next(c for c in golden["cases"] if c["name"] == "F13_trend_relative_reference_test")
    ob0 = case["obs"][0]
    ref_epoch = 50000.0
    rvlike = pkg.PlanetRelativeRVObs([dict(epoch=e, rv=r, σ_rv=1.0) for e, r in zip(ob0["epoch"], ob0["y1"])], name="RelRV",
                                     trend_function=lambda θ_obs, epoch: θ_obs.trend_slope * (epoch - ref_epoch),
                                     variables=pkg.variables(offset=pkg.Normal(0, 200), jitter=pkg.LogUniform(0.01, 50), trend_slope=pkg.Normal(0, 1)))
    b = pkg.Planet(name="b", basis="RadialVelocityOrbit", observations=[rvlike])
    sys_ = pkg.System(name="RelRVSys", companions=[b])
    el, nu = np.asarray(case["elems"]), np.asarray(case["nuis"])
    θ = dict(planets=dict(b=dict(M=el[6], e=el[1], ω=el[3], a=el[0], tp=el[5], mass=el[8],
                                 observations=dict(RelRV=dict(offset=nu[0], jitter=nu[1], trend_slope=nu[2])))))
    fn = pkg.make_ln_like(sys_, θ)
    assert rvlike.trend_coef == "trend_slope" and np.array_equal(fn.obs_tables[0]["extra"], np.asarray(ob0["extra"]))
    ll, g = fn.ln_like_and_grad(θ)
    assert np.all(rel_err(ll, case["ll"], 1.0) < LL_RTOL)
    gobs = g["planets"]["b"]["observations"]["RelRV"]
    for k, nm in enumerate(("offset", "jitter", "trend_slope")):
        ok, worst = grad_ok(gobs[nm][None, :], np.asarray(case["g_nuis"])[k][None, :], np.asarray(case["s_nuis"])[k][None, :])
        assert ok, (nm, worst)
    # a batch for the throughput kernels (W·P > 512) and one for k_small, against the oracle
    rng = np.random.default_rng(5)
    for W in (700, 40):
        elems = np.stack([el[0, 0] * rng.uniform(0.9, 1.1, W), rng.uniform(0, 0.5, W), np.zeros(W), rng.uniform(0, 6.28, W), np.zeros(W),
                          ref_epoch + rng.uniform(-20, 20, W), rng.normal(1, 0.05, W), np.zeros(W), np.zeros(W)])
        nuis = np.stack([rng.normal(50, 10, W), np.exp(rng.uniform(np.log(0.01), np.log(50), W)), rng.normal(0.1, 0.05, W)])
        llw, gel, gnu = fn.ln_like_arrays(elems, nuis, grad=True)
        ll_o, g_o, gn_o = oracle.oracle_eval(fn.obs_tables, fn.planet_desc, elems, nuis, grad=True)
        _cmp_oracle(f"trend W={W}", llw, gel, gnu, ll_o, g_o, gn_o)
        assert np.any(gnu[2] != 0.0)
    # the reference's assertions, as a likelihood scan on the fixed orbit: 30 < offset < 70, 0 < slope < 0.3
    off, slope = np.meshgrid(np.linspace(0, 100, 201), np.linspace(-0.2, 0.4, 241), indexing="ij")
    W = off.size
    elems = np.tile(el[:, :1], (1, W))
    lls = fn.ln_like_arrays(elems, np.stack([off.ravel(), np.ones(W), slope.ravel()]))
    k = int(np.argmax(lls))
    assert 45 < off.ravel()[k] < 55 and 0.09 < slope.ravel()[k] < 0.11, (off.ravel()[k], slope.ravel()[k])
    fn.close()
    # a closure the device cannot carry is refused, not dropped (VERDICT r2 "boundary defect")
    bad = pkg.PlanetRelativeRVObs(dict(epoch=ob0["epoch"], rv=ob0["y1"], σ_rv=np.ones(20)), name="RelRV",
                                  trend_function=lambda θ_obs, epoch: np.sin(θ_obs.trend_slope * epoch))
    with pytest.raises(NotImplementedError):
        pkg.make_ln_like(pkg.System(name="s", companions=[pkg.Planet(name="b", basis="RadialVelocityOrbit", observations=[bad])]), θ)


def test_rv_trend_basis_validation(pkg):
    gb = _gpu()
    ep = np.linspace(50000.0, 50100.0, 6)
    tab = dict(kind=2, planet=-1, epoch=ep, y1=np.zeros(6), y2=None, s1=np.ones(6), s2=None, cor=None)
    planets = [dict(orbit_kind=1, has_mass=True)]
    for extra in (np.ones(5), np.array([1, 2, 3, np.nan, 5, 6.0])):      # wrong length, non-finite
        with pytest.raises(pkg.capi.OctoError):
            gb.GpuPath([dict(tab, extra=extra)], planets)
    with pytest.raises(pkg.capi.OctoError):                              # astrometry tables take no `extra`
        gb.GpuPath([dict(kind=0, planet=0, epoch=ep, y1=np.zeros(6), y2=np.zeros(6), s1=np.ones(6), s2=np.ones(6), cor=None, extra=np.ones(6))],
                   [dict(orbit_kind=0, has_mass=False)])
    # without a nuisance block the trend coefficient is zero whatever the table carries (jitter 0, offset 0 defaults)
    rng = np.random.default_rng(2)
    W = 9
    elems = np.stack([rng.uniform(0.5, 3, W), rng.uniform(0, 0.8, W), np.zeros(W), rng.uniform(0, 6.28, W), np.zeros(W),
                      50000 + rng.uniform(-100, 100, W), rng.normal(1, 0.05, W), np.zeros(W), rng.uniform(0, 10, W)])
    a = gb.gpu_eval([dict(tab, extra=ep - 50000.0)], planets, elems, None, grad=False)[0]
    b = gb.gpu_eval([tab], planets, elems, None, grad=False)[0]
    assert np.array_equal(a, b)


def test_device_resident_api_matches_host_api(pkg):
    import torch
    cfg = synth.config_astrom(n_epochs=200, n_walkers=500, seed=21)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    ll_h, g_h, _ = fn.ln_like_arrays(cfg["elems"], None, grad=True)
    el_t = torch.tensor(cfg["elems"], device="cuda:0")
    ll_t, g_t, _ = fn.ln_like_device(el_t, None, grad=True)
    torch.cuda.synchronize()
    assert np.array_equal(ll_t.cpu().numpy(), ll_h) and np.array_equal(g_t.cpu().numpy(), g_h)
    fn.close()


def _pt_swap_reference(ll, beta, slot2rep, parity, seed, step):
    """NumPy restatement of k_pt_swap (octo_api.hip) for the test."""
    M64 = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9e3779b97f4a7c15) & M64
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
        return z ^ (z >> 31)
    n_temps, n_chains = ll.shape          # [replica][chain]
    out = slot2rep.copy()
    acc = np.zeros(n_temps, dtype=np.int32)
    for c in range(n_chains):
        for t in range(parity, n_temps - 1, 2):
            ri, rj = out[c, t], out[c, t + 1]
            li, lj = ll[ri, c], ll[rj, c]
            logA = (beta[t] - beta[t + 1]) * (lj - li)
            h = mix((mix((mix(seed ^ 0x6f63746f50545357) + step) & M64) + c * 0x100000001b3 + t) & M64)
            u = ((h >> 11) + 1.0) * (1.0 / 9007199254740992.0)
            if np.log(u) < logA:
                out[c, t], out[c, t + 1] = rj, ri
                acc[t] += 1
    return out, acc


def test_pt_swap_kernel(pkg):
    import torch
    lib = pkg.capi.load_library()
    ctx = C.c_void_p()
    assert lib.octo_ctx_create(C.byref(ctx), 0) == 0
    rng = np.random.default_rng(2)
    n_chains, n_temps = 37, 16
    ll = rng.normal(-100, 5, (n_temps, n_chains))
    beta = np.linspace(1.0, 0.0, n_temps) ** 2
    s2r = np.stack([rng.permutation(n_temps) for _ in range(n_chains)]).astype(np.int32)
    d_ll = torch.tensor(ll, device="cuda:0"); d_beta = torch.tensor(beta, device="cuda:0")
    d_s2r = torch.tensor(s2r, device="cuda:0"); d_acc = torch.zeros(n_temps, dtype=torch.int32, device="cuda:0")
    ref = s2r
    acc_ref = np.zeros(n_temps, dtype=np.int32)
    for step in range(6):
        st = lib.octo_pt_swap_device(ctx, d_ll.data_ptr(), d_beta.data_ptr(), d_s2r.data_ptr(), n_temps, n_chains, step % 2,
                                     1234, step, d_acc.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0
        ref, a = _pt_swap_reference(ll, beta, ref, step % 2, 1234, step)
        acc_ref += a
    torch.cuda.synchronize()
    assert np.array_equal(d_s2r.cpu().numpy(), ref)
    assert np.array_equal(d_acc.cpu().numpy(), acc_ref)
    assert np.all(np.sort(ref, axis=1) == np.arange(n_temps))
    assert acc_ref.sum() > 0
    lib.octo_ctx_destroy(ctx)


def _kepler_device(pkg, MA, e, table=False):
    lib = pkg.capi.load_library()
    ctx = C.c_void_p()
    assert lib.octo_ctx_create(C.byref(ctx), 0) == 0
    MA = np.ascontiguousarray(MA, dtype=np.float64); e = np.ascontiguousarray(e, dtype=np.float64)
    E = np.empty_like(MA); sE = np.empty_like(MA); cE = np.empty_like(MA)
    dp = pkg.capi._dptr
    fn = lib.octo_kepler_solve_table if table else lib.octo_kepler_solve
    assert fn(ctx, dp(MA), dp(e), MA.size, dp(E), dp(sE), dp(cE)) == 0
    lib.octo_ctx_destroy(ctx)
    return E, sE, cE


@pytest.mark.parametrize("table", [False, True], ids=["polynomial sincos (k_small)", "LDS table (k_main)"])
def test_kepler_device_solver(pkg, oracle, table):
    """The device Kepler routine (FP32 Markley starter + FP64 fifth-order correction) against Kepler's equation,
    an 80-bit Newton solve, and the reference algorithm in the oracle — over the whole elliptic domain, including
    e -> 1 − 1e-9, |M| -> 0 and |M| -> π. Error is weighted by 1 − e cos E (the conditioning of the root). Both variants of the
    routine: sin/cos of the starter from the half-angle polynomials (k_small, k_hgca) and from the table in LDS with the
    magic-number index and the exact FP32 remainder (k_main, k_ofti_main)."""
    import gpu_binding
    if gpu_binding.DEFAULT_SMALL_BATCH == 0:
        pytest.skip("the Kepler entry points do not depend on the likelihood kernels' family: checked once (under 'auto')")
    def kd(pkg_, M_, e_): return _kepler_device(pkg_, M_, e_, table=table)
    rng = np.random.default_rng(5)
    n = 400_000
    e = np.concatenate([rng.uniform(0, 1, n // 2), 1 - 10 ** rng.uniform(-9, -1, n // 2)])
    e[:8] = [0.0, 1e-12, 0.5, 0.9, 0.99, 0.999999, 0.0, 0.3]
    M = np.concatenate([rng.uniform(-np.pi, np.pi, n // 4), 10 ** rng.uniform(-17, 0.4, n // 4) * rng.choice([-1, 1], n // 4),
                        (np.pi - 10 ** rng.uniform(-16, 0, n // 4)) * rng.choice([-1, 1], n // 4), rng.uniform(-np.pi, np.pi, n - 3 * (n // 4))])
    M = np.clip(M, -np.pi, np.pi)
    rng.shuffle(M)
    M[:8] = [1.0, 1e-9, -1e-9, np.pi, -np.pi, 0.0, 0.0, 0.0]
    E, sE, cE = kd(pkg, M, e)
    assert np.all(np.isfinite(E))
    # 80-bit Newton truth
    Ml = M.astype(np.longdouble); el = e.astype(np.longdouble)
    Et = np.where(el < 0.8, Ml, np.sign(Ml) * np.longdouble(np.pi)); Et = np.where(Ml == 0, 0, Et)
    for _ in range(80):
        Et = Et - (Et - el * np.sin(Et) - Ml) / (1 - el * np.cos(Et))
    cond = (1 - el * np.cos(Et)).astype(np.float64)
    err = np.abs((E - Et).astype(np.float64)) * cond
    assert err.max() < 1.5e-15, err.max()
    assert np.abs(sE - np.sin(Et).astype(np.float64)).max() * 1.0 < 1e-11     # unweighted, dominated by e -> 1 conditioning
    assert (np.abs(sE - np.sin(Et).astype(np.float64)) * cond).max() < 2e-15
    assert (np.abs(cE - np.cos(Et).astype(np.float64)) * cond).max() < 2e-15
    assert E[5] == 0.0 and E[6] == 0.0 and abs(E[7]) == 0.0                    # M == 0 -> E == 0 exactly (early return)
    # reference algorithm (oracle) on a subsample: same root
    lib = oracle.load_oracle()
    idx = rng.choice(n, 5000, replace=False)
    Eo = np.array([lib.octo_oracle_kepler_markley(float(M[i]), float(e[i])) for i in idx])
    assert (np.abs(E[idx] - Eo) * cond[idx]).max() < 2e-15
    # large mean anomalies are reduced like rem2pi(·, RoundNearest)
    big = np.array([40.0, -1234.5, 6.0e3, 2 * np.pi * 7 + 0.25])
    Eb, _, _ = kd(pkg, big, np.full(4, 0.4))
    Eo = np.array([lib.octo_oracle_kepler_markley(float(m), 0.4) for m in big])
    assert np.abs(Eb - Eo).max() < 1e-12
    # invalid inputs
    En, _, _ = kd(pkg, np.array([1.0, 1.0, np.nan]), np.array([1.0, -0.1, 0.3]))
    assert np.all(np.isnan(En))


def test_c_abi_argument_checking_and_strides(pkg, oracle):
    """Status codes for API misuse (no exception crosses the boundary), and ld > W / W not a multiple of 64."""
    capi = pkg.capi
    lib = capi.load_library()
    ctx = C.c_void_p()
    assert lib.octo_ctx_create(C.byref(ctx), 0) == capi.OCTO_OK
    assert lib.octo_ctx_create(C.byref(C.c_void_p()), 99) == capi.OCTO_ENODEV
    cfg = synth.config_astrom(n_epochs=33, n_walkers=70, seed=2)
    t = cfg["table"]
    tab = dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)
    ds = C.c_void_p()
    # bad planet index, unknown kind, too many planets, |cor| too large, absolute RV without mass
    for bad in (dict(tab, planet=3), dict(tab, kind=9), dict(tab, cor=np.full(33, 0.999999))):
        arr, keep = capi.pack_obs([bad])
        assert lib.octo_dataset_create(ctx, arr, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=False)]), 1, C.byref(ds)) == capi.OCTO_EINVAL
        assert b"octo_dataset_create" in lib.octo_last_error(ctx)
    arr, keep = capi.pack_obs([tab])
    n_many = capi.MAX_PLANETS + 1      # (round 5: up to OCTO_MAX_PLANETS = 8 planets on the planet-per-wave kernels)
    assert lib.octo_dataset_create(ctx, arr, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=False)] * n_many), n_many, C.byref(ds)) == capi.OCTO_ENOTSUP      # (a VALID system that is not on the device path: round 6)
    rv = dict(kind=2, planet=-1, epoch=t["epoch"], y1=t["ra"], y2=None, s1=t["σ_ra"], s2=None, cor=None)
    arr2, keep2 = capi.pack_obs([rv])
    assert lib.octo_dataset_create(ctx, arr2, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=False)]), 1, C.byref(ds)) == capi.OCTO_EINVAL
    # RV tables with a ThieleInnesOrbit planet; HGCA without its catalogue numbers, with bad row codes, with a mass-less planet
    assert lib.octo_dataset_create(ctx, arr2, 1, capi.pack_planets([dict(orbit_kind=2, has_mass=True)]), 1, C.byref(ds)) == capi.OCTO_ENOTSUP
    hg = dict(kind=7, planet=-1, epoch=t["epoch"][:4], y1=np.array([0., 1, 0, 1]), y2=np.array([0., 0, 1, 1]), s1=None, s2=None, cor=None,
              extra=np.array([1., 1, .1, .1, 0] * 3))
    for bad in (dict(hg, extra=None), dict(hg, extra=np.ones(7)), dict(hg, y1=np.array([0., 2, 0, 1])), dict(hg, extra=np.array([1., 1, .1, .1, 1.5] * 3))):
        arr3, keep3 = capi.pack_obs([bad])
        assert lib.octo_dataset_create(ctx, arr3, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=True)]), 1, C.byref(ds)) == capi.OCTO_EINVAL
    arr3, keep3 = capi.pack_obs([hg])
    assert lib.octo_dataset_create(ctx, arr3, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=False)]), 1, C.byref(ds)) == capi.OCTO_EINVAL
    assert lib.octo_dataset_create(ctx, arr3, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=True)]), 1, C.byref(ds)) == capi.OCTO_OK
    el0 = np.ascontiguousarray(cfg["elems"]); ll0 = np.empty(70)
    assert lib.octo_eval(ctx, ds, capi._dptr(el0), None, 70, 70, capi._dptr(ll0), None, None) == capi.OCTO_EINVAL   # HGCA needs nuis (pmra, pmdec)
    assert b"pmra" in lib.octo_last_error(ctx)
    assert lib.octo_dataset_destroy(ds) == 0
    assert lib.octo_dataset_create(ctx, arr, 1, capi.pack_planets([dict(orbit_kind=0, has_mass=False)]), 1, C.byref(ds)) == capi.OCTO_OK
    assert lib.octo_dataset_n_rows(ds) == 33
    # strided buffers: ld = 96 > W = 70
    W, ld = 70, 96
    el = np.full((9, ld), np.nan); el[:, :W] = cfg["elems"]
    ll = np.full(ld, 7.0); g = np.full((9, ld), 7.0)
    dp = capi._dptr
    assert lib.octo_eval(ctx, ds, dp(el), None, ld, W, dp(ll), dp(g), None) == capi.OCTO_OK
    assert np.all(ll[W:] == 7.0) and np.all(g[:, W:] == 7.0)                   # nothing written past W
    ll_o, g_o, _ = oracle.oracle_eval([tab], [dict(orbit_kind=0, has_mass=False)], cfg["elems"], None, grad=True,
                                      active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("strided", ll[:W], g[:, :W], None, ll_o, g_o, None)
    # misuse
    assert lib.octo_eval(ctx, ds, dp(el), None, 10, W, dp(ll), None, None) == capi.OCTO_EINVAL      # ld < W
    assert lib.octo_eval(ctx, ds, None, None, ld, W, dp(ll), None, None) == capi.OCTO_EINVAL         # null elems
    assert lib.octo_eval(ctx, ds, dp(el), None, ld, W, dp(ll), None, dp(g)) == capi.OCTO_EINVAL      # g_nuis without nuis
    assert lib.octo_eval(ctx, ds, dp(el), None, ld, 0, dp(ll), None, None) == capi.OCTO_OK           # empty batch is fine
    c = capi.default_consts(); c.pc2au = -1.0
    assert lib.octo_consts_set(ctx, C.byref(c)) == capi.OCTO_EINVAL
    assert lib.octo_dataset_destroy(ds) == 0 and lib.octo_ctx_destroy(ctx) == 0
    assert lib.octo_dataset_destroy(None) == 0 and lib.octo_ctx_destroy(None) == 0


def test_custom_constants_reach_the_kernel(pkg, oracle):
    """octo_consts_set: the host's PlanetOrbits constants are the ones the kernels use (parity is a property of formulas,
    not of digits baked into a kernel)."""
    gb = _gpu()
    cfg = synth.config_astrom(n_epochs=20, n_walkers=65, seed=4)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    c = pkg.capi.default_consts()
    c.kepler_year_to_julian_day = 365.2422; c.pc2au = 206264.80624709636; c.rad2as = 206264.80624709636 * 1.0001
    ll, g, _ = gb.gpu_eval(obs, planets, cfg["elems"], None, grad=True, consts=c)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, cfg["elems"], None, grad=True, consts=c, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("custom consts", ll, g, None, ll_o, g_o, None)
    ll_d, _, _ = gb.gpu_eval(obs, planets, cfg["elems"], None, grad=False)
    assert np.max(np.abs(ll_d - ll)) > 1e-3            # and they do change the answer


def test_oneil_wrapper_receives_theta_obs(pkg):
    """test/unit/distributions.jl:102-152 ("wrapped likelihood receives θ_obs"): the sensitivity of ln_like to jitter is
    the same with and without the ObsPriorAstromONeil2019 wrapper (its Jacobian term does not depend on jitter), and > 1."""
    tbl = dict(epoch=[58000.0, 58200.0, 58400.0], ra=[100.0, 110.0, 120.0], dec=[100.0, 95.0, 90.0], σ_ra=[5.0] * 3, σ_dec=[5.0] * 3)

    def jitter_sensitivity(wrap):
        radec = pkg.PlanetRelAstromLikelihood(tbl, name="d_radec")
        obs = pkg.ObsPriorAstromONeil2019(radec) if wrap else radec
        b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=(obs,))
        sys_ = pkg.System(name="jitter_prop_test", companions=(b,))
        key = "obspri_d_radec" if wrap else "d_radec"
        θ = dict(M=1.0, plx=50.0, planets=dict(b=dict(a=10.0, e=0.2, i=0.5, ω=0.3, Ω=0.4, tp=58000.0,
                                                      observations={key: dict(jitter=np.array([0.001, 300.0]))})))
        fn = pkg.make_ln_like(sys_, θ)
        ll = fn(θ)
        fn.close()
        return ll[1] - ll[0], ll

    d_plain, ll_plain = jitter_sensitivity(False)
    d_wrapped, ll_wrapped = jitter_sensitivity(True)
    assert d_plain > 1
    assert abs(d_wrapped - d_plain) <= 1e-9 * abs(d_plain)
    assert abs((ll_wrapped[0] - ll_plain[0]) - (ll_wrapped[1] - ll_plain[1])) < 1e-9 and ll_wrapped[0] != ll_plain[0]


def test_hgca_mirror_vs_oracle(pkg, oracle):
    """HGCAInstantaneousObs through the mirror (System observation reading θ_system.pmra/.pmdec, hgca.jl:266-267) next to
    relative astrometry, against the reference-order oracle fed the same tables; gradient comes back under the
    reference's names."""
    from test_host import HGCA_ROW
    rng = np.random.default_rng(77)
    W = 70
    hg = pkg.HGCAInstantaneousObs(hgca=HGCA_ROW, N_ave=3)
    ep = 55000.0 + 300.0 * np.arange(5)
    ast = pkg.PlanetRelAstromObs(dict(epoch=ep, ra=rng.normal(0, 300, 5), dec=rng.normal(0, 300, 5), σ_ra=[5.0] * 5, σ_dec=[6.0] * 5), name="gpi")
    b = pkg.Planet(name="b", observations=(ast,))
    c = pkg.Planet(name="c", observations=())
    sys_ = pkg.System(name="hgca_sys", companions=(b, c), observations=(hg,))
    def planet(lo, hi, mlo, mhi):
        return dict(a=rng.uniform(lo, hi, W), e=rng.uniform(0, 0.6, W), i=np.arccos(rng.uniform(-1, 1, W)), ω=rng.uniform(0, 6.28, W),
                    Ω=rng.uniform(0, 6.28, W), tp=50000 + rng.uniform(0, 4000, W), mass=rng.uniform(mlo, mhi, W))
    θ = dict(M=1.2, plx=50.0, pmra=rng.normal(4.3, 0.2, W), pmdec=rng.normal(-2.0, 0.2, W),
             planets=dict(b=planet(8, 15, 10, 40), c=planet(2, 4, 1, 10)))
    θ["planets"]["b"]["observations"] = dict(gpi=dict(jitter=rng.uniform(0, 3, W)))
    fn = pkg.make_ln_like(sys_, θ)
    ll, g = fn.ln_like_and_grad(θ)
    ll_f = fn(θ)
    elems, nuis = fn.pack(θ)
    assert np.array_equal(ll, ll_f)
    ll_o, g_o, gn_o = oracle.oracle_eval(fn.obs_tables, fn.planet_desc, elems, nuis, grad=True)
    assert np.all(rel_err(ll, ll_o, 1.0) < LL_RTOL)
    assert np.all(rel_err(g["pmra"], gn_o[3], np.abs(gn_o[3]).max()) < 1e-10) and np.all(rel_err(g["pmdec"], gn_o[4], np.abs(gn_o[4]).max()) < 1e-10)
    for ip, nm in enumerate(("b", "c")):
        for k, key in enumerate(("a", "e", "i", "ω", "Ω", "tp", "M", "plx", "mass")):
            ref = g_o[ip * 9 + k]
            assert np.all(rel_err(g["planets"][nm][key], ref, np.abs(ref).max()) < 1e-9), (nm, key)
    # without the system proper motion the reference errors (θ_system.pmra); so does the mirror
    θ_bad = {k: v for k, v in θ.items() if k != "pmra"}
    with pytest.raises(KeyError):
        fn(θ_bad)
    fn.close()
    # a Visual planet without a mass cannot feed the reflex motion
    θ_nm = dict(θ, planets=dict(b=θ["planets"]["b"], c={k: v for k, v in θ["planets"]["c"].items() if k != "mass"}))
    with pytest.raises(KeyError):
        pkg.make_ln_like(sys_, θ_nm)


def test_hgca_next_to_oneil_and_marginalised_rv(oracle):
    """Regression (found by tests/stress_parity.py): an OCTO_HGCA table in a system that also selects the O'Neil /
    marginalised-RV kernel variant was taken for an O'Neil wrapper by k_finish (kind >= ONEIL) — ll = -Inf and an
    out-of-bounds planet index. Every kind together, against the oracle."""
    gb = _gpu()
    rng = np.random.default_rng(12)
    W, n = 130, 40
    ep = np.sort(50000 + rng.uniform(0, 4000, n))
    ra, dec = rng.normal(0, 300, n), rng.normal(0, 300, n)
    rows = np.array([(48348.0, 0, 0), (48414.0, 1, 0), (57408.0, 0, 1), (57470.0, 1, 1)])
    hg = np.array([4.71, -1.86, 0.61, 0.49, 0.21, 4.352, -2.013, 0.031, 0.024, -0.12, 4.61, -1.72, 0.052, 0.041, 0.33])
    obs = [dict(kind=5, planet=0, epoch=ep, y1=ra, y2=dec, s1=np.full(n, 8.0), s2=np.full(n, 9.0), cor=None),
           dict(kind=6, planet=1, epoch=ep, y1=np.arctan2(ra, dec), y2=np.hypot(ra, dec), s1=np.full(n, 0.03), s2=np.full(n, 9.0), cor=None),
           dict(kind=3, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=np.full(n, 4.0), s2=None, cor=None),
           dict(kind=7, planet=-1, epoch=rows[:, 0], y1=rows[:, 1], y2=rows[:, 2], s1=None, s2=None, cor=None, extra=hg)]
    planets = [dict(orbit_kind=0, has_mass=True)] * 2
    def pl(lo, hi):
        return np.stack([rng.uniform(lo, hi, W), rng.uniform(0, 0.6, W), np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W),
                         50000 + rng.uniform(0, 4000, W), np.full(W, 1.2), np.full(W, 50.0), rng.uniform(2, 30, W)])
    elems = np.concatenate([pl(3, 6), pl(9, 15)])
    nuis = np.zeros((12, W))
    nuis[0] = rng.uniform(0, 3, W); nuis[1] = 1.0; nuis[3] = rng.uniform(0, 3, W); nuis[4] = 1.0
    nuis[7] = rng.uniform(0.5, 5, W); nuis[9] = rng.normal(4.3, 0.3, W); nuis[10] = rng.normal(-2.0, 0.3, W)
    ll, g, gn = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    ll_f, _, _ = gb.gpu_eval(obs, planets, elems, nuis, grad=False)
    assert np.array_equal(ll, ll_f) and np.all(np.isfinite(ll))
    ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
    _cmp_oracle("hgca+oneil+marg", ll, g, gn, ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)


@pytest.mark.gpu
def test_registered_host_buffers_match_the_pageable_path(pkg):
    """octo_host_register: a big host-buffer batch whose arrays are all registered takes the zero-copy route (copy kernel in,
    results written in place); bit-identical to the pageable route and to a second call; partial registration falls back."""
    cfg = synth.config_astrom(n_epochs=96, n_walkers=9001, cfg=3, seed=77)      # 9001 x 19 doubles > 1 MiB: beyond the staged range
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="reg", companions=[planet], observations=[]), cfg["theta_example"])
    el = np.ascontiguousarray(cfg["elems"])
    ll0, g0, _ = fn.ln_like_arrays(el, None, grad=True)
    ll = np.full(el.shape[1], np.nan); g = np.full_like(el, np.nan)
    fn.host_register(el, ll)                       # the gradient array is not registered yet: pageable route
    fn.ln_like_into(el, None, ll, g)
    assert np.array_equal(ll, ll0) and np.array_equal(g, g0)
    fn.host_register(g)
    ll[:] = np.nan; g[:] = np.nan
    fn.ln_like_into(el, None, ll, g)
    assert np.array_equal(ll, ll0, equal_nan=True) and np.array_equal(g, g0, equal_nan=True)
    ll[:] = np.nan
    fn.ln_like_into(el, None, ll)                  # forward only
    assert np.array_equal(ll, ll0, equal_nan=True)
    # a view into a registered range is found too (interior pointers), and a leading dimension larger than W is honoured
    W2 = 8000
    fn._check(fn.lib.octo_eval(fn._ctx, fn._ds, pkg.capi._dptr(el), None, el.shape[1], W2, pkg.capi._dptr(ll), pkg.capi._dptr(g), None), "octo_eval")
    assert np.array_equal(ll[:W2], ll0[:W2], equal_nan=True) and np.array_equal(g[:, :W2], g0[:, :W2], equal_nan=True)
    with pytest.raises(pkg.capi.OctoError):
        fn.host_register(g)                        # twice
    fn.host_unregister(el, ll, g)
    with pytest.raises(pkg.capi.OctoError):
        fn.host_unregister(g)
    ll[:] = np.nan
    fn.ln_like_into(el, None, ll, g)               # pageable again
    assert np.array_equal(ll, ll0, equal_nan=True)
    fn.close()


def test_thiele_innes_near_face_on_vs_60_digits(oracle):
    """Fixture F12 (oracle/make_ti_faceon.py): the Thiele-Innes walker 2.4e-12 from face-on that the long random sweep found. The
    reference's α² = u + √((u+v)(u−v)) cancels there — the reference-order C oracle is 2e-5 off on ∂/∂G (1.7e-7 of the sweep's batch scale) — while the kernels'
    cancellation-free form must stay on the 60-digit value (deliberate deviations)."""
    import json
    from pathlib import Path
    case = json.loads((Path(__file__).resolve().parent / "golden" / "ti_faceon.json").read_text())["cases"][0]
    obs, planets, elems, nuis = case_tables(case)
    gb = _gpu()
    ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    assert np.all(rel_err(ll, np.asarray(case["ll"]), 1.0) < LL_RTOL)
    ok, worst = grad_ok(g_el, case["g_elems"], case["s_elems"], rtol=G_RTOL, cancel=G_CANCEL)
    assert ok, ("g_elems", worst)
    ok, worst = grad_ok(g_nu, case["g_nuis"], case["s_nuis"], rtol=G_RTOL, cancel=G_CANCEL)
    assert ok, ("g_nuis", worst)
    # the restatement of the reference's arithmetic is the one that is off (and not by more than its known cancellation)
    _, g_o, _ = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
    ref = np.asarray(case["g_elems"]); sc = np.abs(ref).max(axis=1, keepdims=True)
    e_dev = np.max(np.abs(g_el - ref) / np.maximum(sc, 1e-300)); e_ora = np.max(np.abs(g_o - ref) / np.maximum(sc, 1e-300))
    assert e_dev < 1e-9 and 100 * e_dev < e_ora < 1e-3, (e_dev, e_ora)


@pytest.mark.parametrize("P", [4, 5, 6, 7, 8])
def test_planet_per_wave_kernels_vs_oracle(oracle, P):
    """k_mainp / k_finishp (octo_mainp.h): ONE planet per wave, the number of planets a run-time block shape — four planets (k_mainp ->
    k_finish<4>) and five to OCTO_MAX_PLANETS = 8 (k_mainp -> k_finishp; the reference unrolls over any number, src/likelihoods/system.jl:116-118,
    156-170). A RA/Dec (+ cor) or sep/PA table on every planet, absolute and relative RV, with and without per-walker nuisances, a batch
    of 200 walkers (ragged last tile, short last chunk of rows) and one of 3 (no small-batch kernel beyond four planets): against the
    oracle, forward value == value with the gradient, some walkers with the planets' order swapped (the strictly-inner rule) and invalid ones."""
    gb = _gpu()
    rng = np.random.default_rng(100 + P)
    import stress_parity as sp
    for W in (200, 3):
        planets = [dict(orbit_kind=0, has_mass=True) for _ in range(P)]
        elems = np.concatenate([sp.planet_elems(rng, W, 0, 1.5 + 4 * i, 4.5 + 4 * i) for i in range(P)])
        for p in range(1, P):
            elems[p * 9 + 6] = elems[6]; elems[p * 9 + 7] = elems[7]      # shared system M, plx
        if W > 20:
            elems[0, :15] = elems[9 + 0, :15] * 1.7                       # planet 0 outside planet 1 for some walkers
            elems[1, 20] = 1.3; elems[9 * (P - 1) + 0, 21] = -2.0; elems[9 * 2 + 5, 22] = np.nan      # invalid walkers
        obs = []
        for ip in range(P):
            n = int(rng.integers(30, 90)) + (1 if ip == 0 else 0)
            ep = np.sort(50000 + rng.uniform(0, 4000, n))
            if ip % 3 == 1:
                obs.append(dict(kind=1, planet=ip, epoch=ep, y1=rng.uniform(-3, 3, n), y2=rng.uniform(100, 900, n), s1=rng.uniform(0.005, 0.03, n), s2=rng.uniform(3, 12, n), cor=None))
            else:
                obs.append(dict(kind=0, planet=ip, epoch=ep, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n),
                                cor=rng.uniform(-0.6, 0.6, n) if ip % 3 == 2 else None))
        n = 77; ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs.append(dict(kind=4, planet=P - 2, epoch=ep, y1=rng.normal(0, 900, n), y2=None, s1=rng.uniform(20, 60, n), s2=None, cor=None))
        n = 101; ep = np.sort(50000 + rng.uniform(0, 4000, n))
        obs.append(dict(kind=2, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None, extra=(ep - 52000.0) / 1000.0))
        nuis = np.zeros((len(obs) * 3, W))
        for io in range(P):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.01, W)
        nuis[0, : W // 4] = 0.0
        for io in (P, P + 1):
            nuis[io * 3] = rng.normal(0, 5, W); nuis[io * 3 + 1] = rng.uniform(0.1, 5, W)
        nuis[(P + 1) * 3 + 2] = rng.normal(0, 3, W)
        for nz in (nuis, None):
            ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nz, grad=True)
            ll_f, _, _ = gb.gpu_eval(obs, planets, elems, nz, grad=False)
            assert np.array_equal(ll, ll_f), (P, W, "forward-only and gradient launches disagree")
            ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nz, grad=True)
            if W > 20:
                assert np.isneginf(ll[[20, 21, 22]]).all()
            _cmp_oracle(f"{P} planets, W = {W}", ll, g_el, g_nu, ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)
        # only the astrometry tables: the kind set without RV (k_mainp<·, ·, RA/Dec | sep/PA | cor>)
        ll, g_el, _ = gb.gpu_eval(obs[:P], planets, elems, None, grad=True)
        ll_o, g_o, _ = oracle.oracle_eval(obs[:P], planets, elems, None, grad=True, active=synth.active_mask(P, P, nuis=False))
        _cmp_oracle(f"{P} planets astrometry only, W = {W}", ll, g_el, None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)


def test_more_than_four_planets_every_kind(pkg, oracle):
    """Beyond OCTO_MAX_PLANETS_ALL_KINDS (the limit of the templated kernels and of the small-batch family) the planet-per-wave kernels take every observation
    kind since round 6 — relative astrometry, absolute / MARGINALISED / relative RV, HGCA, the O'Neil prior; more than OCTO_MAX_PLANETS planets are refused at
    octo_dataset_create (OCTO_ENOTSUP: valid, not on the device path — the shim then keeps the system on the reference's path). The O'Neil prior beyond four
    planets (prior-observable.jl:78-137; the term by the wave of the planet the table is attached to, its closed form and adjoints in k_finishp): 5, 6 and 8 planets,
    RA/Dec and sep/PA wrappers on different planets next to marginalised RV, against the oracle. HGCA beyond four planets (k_hgcap -> `extra` -> k_finishp; hgca.jl:155-400): 5, 6 and 8 planets, an HGCA table next
    to relative astrometry and absolute RV, against the oracle, forward-only == the value returned with a gradient, a RadialVelocityOrbit planet among them (it does
    not contribute: hgca.jl:255-262). Marginalised RV on the planet-per-wave kernels (rv-absolute-margin.jl:140-185: the
    usual likelihood of a many-planet RV fit) is a parity case: 5, 6 and 8 planets, a marginalised-RV table with a trend next to relative astrometry and
    relative RV, with and without per-walker nuisances, forward-only == the value returned with a gradient, against the oracle."""
    gb = _gpu()
    capi = pkg.capi
    ep = np.linspace(50000.0, 50400.0, 12)
    rv = dict(kind=3, planet=-1, epoch=ep, y1=np.zeros(12), y2=None, s1=np.ones(12), s2=None, cor=None)
    on = dict(kind=5, planet=0, epoch=ep, y1=np.zeros(12), y2=np.zeros(12), s1=np.ones(12), s2=np.ones(12), cor=None)
    ok = dict(kind=0, planet=4, epoch=ep, y1=np.zeros(12), y2=np.zeros(12), s1=np.ones(12), s2=np.ones(12), cor=None)
    pl = lambda n: [dict(orbit_kind=0, has_mass=True) for _ in range(n)]
    with pytest.raises(capi.OctoError) as ei:      # more planets than OCTO_MAX_PLANETS: valid in the reference, not on the device path
        gb.GpuPath([ok], pl(capi.MAX_PLANETS + 1))
    assert ei.value.status == capi.OCTO_ENOTSUP
    gb.GpuPath([ok], pl(5)).close()
    gb.GpuPath([on], pl(5)).close()
    gb.GpuPath([rv, on], pl(4)).close()
    import stress_parity as sp
    rng = np.random.default_rng(88)
    for P, W in ((5, 70), (6, 257), (8, 3)):
        elems = np.concatenate([sp.planet_elems(rng, W, 0, 0.05 + 0.4 * i, 0.3 + 0.4 * i) for i in range(P)])      # a compact RV system: periods of days to months
        n1, n2 = 150, 40
        t1 = np.sort(50000 + rng.uniform(0, 900, n1)); t2 = np.sort(50000 + rng.uniform(0, 900, n2))
        obs = [dict(kind=3, planet=-1, epoch=t1, y1=rng.normal(0, 30, n1), y2=None, s1=rng.uniform(1, 8, n1), s2=None, cor=None, extra=(t1 - 50400.0) / 300.0),
               dict(kind=0, planet=P - 1, epoch=t2, y1=rng.normal(0, 30, n2), y2=rng.normal(0, 30, n2), s1=rng.uniform(3, 12, n2), s2=rng.uniform(3, 12, n2), cor=None),
               dict(kind=4, planet=1, epoch=t2, y1=rng.normal(0, 500, n2), y2=None, s1=rng.uniform(20, 80, n2), s2=None, cor=None),
               dict(kind=3, planet=-1, epoch=t2 + 0.5, y1=rng.normal(0, 30, n2), y2=None, s1=rng.uniform(1, 8, n2), s2=None, cor=None)]
        nuis = np.zeros((len(obs) * 3, W))
        nuis[0] = rng.normal(0, 10, W); nuis[1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W)); nuis[2] = rng.normal(0, 2, W)
        nuis[3] = rng.uniform(0, 4, W); nuis[4] = rng.normal(1, 0.01, W); nuis[5] = rng.normal(0, 0.02, W)
        nuis[6] = rng.normal(0, 10, W); nuis[7] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
        nuis[10] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
        if W >= 7:
            elems[9 + 1, 2] = 1.3; elems[6, 5] = np.nan
        for nz in (nuis, None):
            ll, g, gn = gb.gpu_eval(obs, pl(P), elems, nz, grad=True)
            llf, _, _ = gb.gpu_eval(obs, pl(P), elems, nz, grad=False)
            assert np.array_equal(ll, llf), (P, "forward-only and gradient launches disagree")
            ll_o, g_o, gn_o = oracle.oracle_eval(obs, pl(P), elems, nz, grad=True)
            _cmp_oracle(f"marginalised RV, {P} planets", ll, g, gn, ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)
    # the O'Neil prior beyond four planets
    for P, W in ((5, 70), (6, 130), (8, 5)):
        elems = np.concatenate([sp.planet_elems(rng, W, 0, 2 + 5 * i, 5 + 5 * i) for i in range(P)])
        n1, n2 = 40, 25
        t1 = np.sort(50000 + rng.uniform(0, 3000, n1)); t2 = np.sort(50000 + rng.uniform(0, 3000, n2))
        ra, dec = rng.normal(0, 300, n2), rng.normal(0, 300, n2)
        obs = [dict(kind=5, planet=P - 1, epoch=t1, y1=rng.normal(0, 300, n1), y2=rng.normal(0, 300, n1), s1=rng.uniform(3, 12, n1), s2=rng.uniform(3, 12, n1), cor=rng.uniform(-0.6, 0.6, n1)),
               dict(kind=6, planet=1, epoch=t2, y1=np.arctan2(ra, dec), y2=np.hypot(ra, dec), s1=np.full(n2, 0.03), s2=rng.uniform(3, 12, n2), cor=None),
               dict(kind=0, planet=2, epoch=t2 + 1.0, y1=rng.normal(0, 300, n2), y2=rng.normal(0, 300, n2), s1=rng.uniform(3, 12, n2), s2=rng.uniform(3, 12, n2), cor=None),
               dict(kind=3, planet=-1, epoch=t1 + 0.5, y1=rng.normal(0, 30, n1), y2=None, s1=rng.uniform(1, 8, n1), s2=None, cor=None)]
        nuis = np.zeros((len(obs) * 3, W))
        for io in range(3):
            nuis[io * 3] = rng.uniform(0, 4, W); nuis[io * 3 + 1] = rng.normal(1, 0.01, W); nuis[io * 3 + 2] = rng.normal(0, 0.02, W)
        nuis[10] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
        if W >= 7:
            elems[9 + 1, 2] = 1.3; elems[6, 5] = np.nan
        for nz in (nuis, None):
            ll, g, gn = gb.gpu_eval(obs, pl(P), elems, nz, grad=True)
            llf, _, _ = gb.gpu_eval(obs, pl(P), elems, nz, grad=False)
            assert np.array_equal(ll, llf), (P, "forward-only and gradient launches disagree")
            ll_o, g_o, gn_o = oracle.oracle_eval(obs, pl(P), elems, nz, grad=True)
            _cmp_oracle(f"O'Neil prior, {P} planets", ll, g, gn, ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)
    # HGCA beyond four planets
    rows = np.array([(48348.0, 0, 0), (48414.0, 1, 0), (48200.0, 0, 0), (57408.0, 0, 1), (57470.0, 1, 1), (57600.0, 1, 1)])
    hg = np.array([4.71, -1.86, 0.61, 0.49, 0.21, 4.352, -2.013, 0.031, 0.024, -0.12, 4.61, -1.72, 0.052, 0.041, 0.33])
    for P, W in ((5, 70), (6, 130), (8, 3)):
        elems = np.concatenate([sp.planet_elems(rng, W, 0, 2 + 5 * i, 5 + 5 * i) for i in range(P)])
        planets = pl(P)
        planets[2] = dict(orbit_kind=1, has_mass=True)      # a RadialVelocityOrbit among them
        n2 = 30
        t2 = np.sort(50000 + rng.uniform(0, 3000, n2))
        obs = [dict(kind=7, planet=-1, epoch=rows[:, 0], y1=rows[:, 1], y2=rows[:, 2], s1=None, s2=None, cor=None, extra=hg),
               dict(kind=0, planet=P - 1, epoch=t2, y1=rng.normal(0, 300, n2), y2=rng.normal(0, 300, n2), s1=rng.uniform(3, 12, n2), s2=rng.uniform(3, 12, n2), cor=None),
               dict(kind=2, planet=-1, epoch=t2 + 0.5, y1=rng.normal(0, 30, n2), y2=None, s1=rng.uniform(1, 8, n2), s2=None, cor=None)]
        nuis = np.zeros((len(obs) * 3, W))
        nuis[0] = rng.normal(4.3, 0.3, W); nuis[1] = rng.normal(-2.0, 0.3, W)
        nuis[3] = rng.uniform(0, 4, W); nuis[4] = rng.normal(1, 0.01, W); nuis[5] = rng.normal(0, 0.02, W)
        nuis[6] = rng.normal(0, 10, W); nuis[7] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
        if W >= 7:
            elems[9 + 1, 2] = 1.3; elems[6, 5] = np.nan
        ll, g, gn = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
        llf, _, _ = gb.gpu_eval(obs, planets, elems, nuis, grad=False)
        assert np.array_equal(ll, llf), (P, "forward-only and gradient launches disagree")
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nuis, grad=True)
        _cmp_oracle(f"HGCA, {P} planets", ll, g, gn, ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)
        with pytest.raises(capi.OctoError):      # an HGCA table needs pmra / pmdec
            gb.gpu_eval(obs, planets, elems, None, grad=False)
