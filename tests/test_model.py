"""
Standard parameterisation on the device: the whole ℓπcallback / ∇ℓπcallback
(src/logdensitymodel.jl:110-177) for models made of the reference's standard building blocks.
CPU: oracle restatement vs the 60-digit fixture. GPU: the HIP path vs fixture and oracle, and the reference's
own model-level tests re-expressed through the LogDensityModel mirror.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from conftest import KIND_IDS

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def model_golden():
    return (json.loads((ROOT / "tests" / "golden" / "model.json").read_text())["cases"]
            + json.loads((ROOT / "tests" / "golden" / "ti_model.json").read_text())["cases"]       # [2]: Thiele-Innes tutorial model
            + json.loads((ROOT / "tests" / "golden" / "trend_model.json").read_text())["cases"])   # [3]: relative RV with offset + trend (D = 3)


def _tables(case):
    obs = [dict(kind=KIND_IDS[o["kind"]], planet=o["planet"],
                **{k: (None if o.get(k) is None else np.asarray(o[k], dtype=np.float64)) for k in ("epoch", "y1", "y2", "s1", "s2", "cor", "extra")}) for o in case["obs"]]
    return obs, case["planets"]


def _check(lp, g, case, rtol_lp=1e-12, rtol_g=1e-9):
    ref_lp = np.asarray(case["lp"]); ref_g = np.asarray(case["grad"])
    assert np.all(np.abs(lp - ref_lp) <= rtol_lp * np.maximum(1, np.abs(ref_lp))), np.max(np.abs(lp - ref_lp) / np.maximum(1, np.abs(ref_lp)))
    scale = np.abs(ref_g).max(axis=1, keepdims=True)
    tol = rtol_g * np.abs(ref_g) + 1e-12 * scale
    assert np.all(np.abs(g - ref_g) <= tol), np.max(np.abs(g - ref_g) / tol)


def test_oracle_model_vs_golden(oracle, model_golden):
    for case in model_golden:
        obs, planets = _tables(case)
        lp, g = oracle.oracle_model_logpost(obs, planets, oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"]),
                                            None if case["nsrc"] is None else oracle.make_sources(case["nsrc"]), np.asarray(case["theta_t"]))
        _check(lp, g, case)
        lp0, _ = oracle.oracle_model_logpost(obs, planets, oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"]),
                                             None if case["nsrc"] is None else oracle.make_sources(case["nsrc"]), np.asarray(case["theta_t"]), grad=False)
        assert np.array_equal(lp0, lp)


def test_reference_model_dump_if_present(oracle, model_golden):
    """tools/julia_crosscheck.jl evaluates model.ℓπcallback / ∇ℓπcallback of the REAL reference for the committed θ_t of the D = 11
    test model and of config 1; when its output exists, the restatement of priors, bijectors, UniformCircular, θ_at_epoch_to_tperi and
    the likelihood is held to it."""
    import json
    dump_path = ROOT / "tests" / "golden" / "reference_dump.json"
    if not dump_path.exists():
        pytest.skip("tests/golden/reference_dump.json not generated (no Julia in the build image)")
    dump = json.loads(dump_path.read_text())
    cases = [("model.json/D11_reference_test_model", model_golden[0]),
             ("config1.json/config1_D11_50_epochs", json.loads((ROOT / "tests" / "golden" / "config1.json").read_text())["cases"][0])]
    for key, case in cases:
        r = dump[key]
        obs, planets = _tables(case)
        th = np.asarray(case["theta_t"])
        lp, g = oracle.oracle_model_logpost(obs, planets, oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"]), None, th)
        assert np.all(np.abs(lp - np.asarray(r["lp"])) <= 1e-10 * np.abs(lp)), key
        gref = np.asarray(r["grad"]).T
        assert np.all(np.abs(g - gref) <= 1e-8 * np.abs(gref).max(axis=0, keepdims=True)), key


def test_oracle_model_edges(oracle, model_golden):
    case = model_golden[0]
    obs, planets = _tables(case)
    th = np.asarray(case["theta_t"])[:, :3].copy()
    th[3, 0] = np.nan                    # non-finite θ_t -> -Inf   (logdensitymodel.jl:120-124)
    th[3, 1] = 800.0                     # logistic saturates: e == upper bound -> Jacobian term -Inf -> "healed" prior
    lp, g = oracle.oracle_model_logpost(obs, planets, oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"]), None, th)
    assert np.isneginf(lp[0]) and np.all(g[:, 0] == 0)
    assert lp[1] < -1e300 or np.isneginf(lp[1])
    assert np.isfinite(lp[2])


def _reference_test_model(pkg, table=None):
    """The model of test/integration/sampling.jl:29-64, through the mirror."""
    table = table or dict(epoch=[50000, 50120, 50240, 50360, 50480, 50600, 50720, 50840],
                          ra=[-505.76, -502.57, -498.21, -492.68, -485.98, -478.11, -469.08, -458.90],
                          dec=[-66.93, -37.47, -7.93, 21.64, 51.15, 80.54, 109.73, 138.65],
                          σ_ra=[10.0] * 8, σ_dec=[10.0] * 8, cor=[0.0] * 8)
    astrom_like = pkg.PlanetRelAstromLikelihood(table, name="sampling_test")
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[astrom_like],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    return pkg.System(name="TestSys", companions=[b], observations=[],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1)))


@pytest.mark.gpu
def test_gpu_model_vs_golden_and_oracle(pkg, oracle, model_golden):
    case = model_golden[0]
    model = pkg.LogDensityModel(_reference_test_model(pkg))
    assert model.D == 11                                                    # test/integration/sampling.jl:70
    assert model.names == ["M", "plx", "b_a", "b_e", "b_i", "b_ωx", "b_ωy", "b_Ωx", "b_Ωy", "b_θx", "b_θy"]
    # the mirror derives the same priors / sources as the hand-written fixture
    for k, p in enumerate(case["priors"]):
        assert model._c_priors[k].kind == p["kind"] and model._c_priors[k].p0 == p["p0"] and model._c_priors[k].p1 == p["p1"]
    for k, s in enumerate(case["esrc"]):
        c = model._c_esrc[k]
        assert (c.kind, c.i0, c.i1, c.flags, c.value) == (s["kind"], s["i0"], s["i1"], s["flags"], s["value"])
    th = np.asarray(case["theta_t"])
    lp, g = model.logdensity_and_gradient(th)
    _check(lp, g, case)
    assert np.array_equal(model.ℓπcallback(th), lp)
    lp1, g1 = model.logdensity_and_gradient(th[:, 0])                       # single θ_t, like the reference's callback
    assert lp1 == lp[0] and np.array_equal(g1, g[:, 0])
    obs, planets = _tables(case)
    lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th)
    assert np.all(np.abs(lp - lp_o) <= 1e-12 * np.abs(lp_o))
    model.close()


@pytest.mark.gpu
def test_gpu_model_healed_prior_keeps_the_likelihood_gradient(pkg, oracle, model_golden):
    """A prior that evaluates non-finite is "healed" to -floatmax (variables.jl:1229-1236): a CONSTANT, so the reference's ForwardDiff
    gradient is then the gradient of the likelihood alone — UnitLengthPrior terms included (variables.jl:309-323), not zero
    (ADVICE r1). Both the fused small-batch launch (W = 3) and the throughput kernels (forced, and W = 70)."""
    case = model_golden[0]
    obs, planets = _tables(case)
    base = np.asarray(case["theta_t"])
    for W, force_big in ((3, False), (3, True), (70, False)):
        th = np.tile(base, (1, W // base.shape[1] + 1))[:, :W].copy()
        th[3, 1] = 800.0                 # logistic saturates: e == upper bound -> log-Jacobian -Inf -> healed
        th[3, 0] = np.nan                # non-finite θ_t -> -Inf, zero gradient
        model = pkg.LogDensityModel(_reference_test_model(pkg))
        if force_big:
            model.ln_like._check(model.ln_like.lib.octo_ctx_set_small_batch(model.ln_like._ctx, 0), "set")
        lp, g = model.logdensity_and_gradient(th)
        lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th)
        model.close()
        assert np.isneginf(lp[0]) and np.all(g[:, 0] == 0.0)
        assert lp[1] < -1e300 and lp_o[1] < -1e300
        # e = upper bound exactly: the likelihood is -Inf there (e < 1 holds, 0.99, so it is finite): gradient = ∇ likelihood terms
        assert np.all(np.isfinite(g[:, 1])) and np.any(g[:, 1] != 0.0), (W, force_big, g[:, 1])
        sc = np.maximum(np.abs(g_o[:, 1]).max(), 1e-300)
        assert np.all(np.abs(g[:, 1] - g_o[:, 1]) <= 1e-9 * sc), (W, force_big, np.max(np.abs(g[:, 1] - g_o[:, 1]) / sc))
        ok = np.arange(W) >= 2
        assert np.all(np.abs(lp[ok] - lp_o[ok]) <= 1e-12 * np.abs(lp_o[ok]))


@pytest.mark.gpu
def test_gpu_model_two_planet_rv(pkg, oracle, model_golden):
    case = model_golden[1]
    o_a, o_r = case["obs"]
    astrom = pkg.PlanetRelAstromObs(dict(epoch=o_a["epoch"], ra=o_a["y1"], dec=o_a["y2"], σ_ra=o_a["s1"], σ_dec=o_a["s2"], cor=o_a["cor"]), name="GPI astrom",
                                    variables=pkg.variables(jitter=pkg.LogUniform(0.1, 20.0), northangle=pkg.Normal(0.0, 0.05)))
    rv = pkg.StarAbsoluteRVObs(dict(epoch=o_r["epoch"], rv=o_r["y1"], σ_rv=o_r["s1"]), name="HARPS",
                               variables=pkg.variables(offset=pkg.Normal(0.0, 30.0), jitter=pkg.LogUniform(0.1, 50.0)))
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[],
                   variables=pkg.variables(a=pkg.LogUniform(1.0, 5.0), e=pkg.Uniform(0.0, 0.9), i=pkg.Sine(), ω=pkg.UniformCircular(), Ω=pkg.UniformCircular(),
                                           tp=pkg.Uniform(49000.0, 51000.0), mass=pkg.LogUniform(0.5, 50.0)))
    c = pkg.Planet(name="c", basis="Visual{KepOrbit}", observations=[astrom],
                   variables=pkg.variables(a=pkg.LogUniform(8.0, 40.0), e=pkg.Uniform(0.0, 0.9), i=pkg.Sine(), ω=pkg.UniformCircular(), Ω=pkg.UniformCircular(),
                                           θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000.0), mass=pkg.Uniform(0.0, 30.0)))
    sys_ = pkg.System(name="two", companions=[b, c], observations=[rv],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.5, upper=2.0), plx=pkg.Normal(50.0, 0.5)))
    model = pkg.LogDensityModel(sys_)
    assert model.D == len(case["priors"]) == 25
    assert model.names[:4] == ["M", "plx", "HARPS_offset", "HARPS_jitter"] and model.names[-2:] == ["c_GPI_astrom_jitter", "c_GPI_astrom_northangle"]
    th = np.asarray(case["theta_t"])
    lp, g = model.logdensity_and_gradient(th)
    _check(lp, g, case, rtol_lp=1e-12, rtol_g=1e-9)
    model.close()


def _two_planet_model(pkg, case):
    o_a, o_r = case["obs"]
    astrom = pkg.PlanetRelAstromObs(dict(epoch=o_a["epoch"], ra=o_a["y1"], dec=o_a["y2"], σ_ra=o_a["s1"], σ_dec=o_a["s2"], cor=o_a["cor"]), name="GPI astrom",
                                    variables=pkg.variables(jitter=pkg.LogUniform(0.1, 20.0), northangle=pkg.Normal(0.0, 0.05)))
    rv = pkg.StarAbsoluteRVObs(dict(epoch=o_r["epoch"], rv=o_r["y1"], σ_rv=o_r["s1"]), name="HARPS",
                               variables=pkg.variables(offset=pkg.Normal(0.0, 30.0), jitter=pkg.LogUniform(0.1, 50.0)))
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[],
                   variables=pkg.variables(a=pkg.LogUniform(1.0, 5.0), e=pkg.Uniform(0.0, 0.9), i=pkg.Sine(), ω=pkg.UniformCircular(), Ω=pkg.UniformCircular(),
                                           tp=pkg.Uniform(49000.0, 51000.0), mass=pkg.LogUniform(0.5, 50.0)))
    c = pkg.Planet(name="c", basis="Visual{KepOrbit}", observations=[astrom],
                   variables=pkg.variables(a=pkg.LogUniform(8.0, 40.0), e=pkg.Uniform(0.0, 0.9), i=pkg.Sine(), ω=pkg.UniformCircular(), Ω=pkg.UniformCircular(),
                                           θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000.0), mass=pkg.Uniform(0.0, 30.0)))
    return pkg.System(name="two", companions=[b, c], observations=[rv],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.5, upper=2.0), plx=pkg.Normal(50.0, 0.5)))


@pytest.mark.gpu
def test_gpu_model_tail_inside_k_finish(pkg, oracle, model_golden):
    """Round 4: for batches on the throughput kernels the tail of the callback — lp = prior + ll with the callback's rules, ∇θ_t = Jᵀḡ +
    ∇prior (src/logdensitymodel.jl:110-146, 169-177) — runs inside k_finish (model_tail), for one planet (finish_tile) and for several
    (finish_tile_multi, with nuisance variables). Same θ_t through the fused small-batch launch and through the throughput route must
    agree to rounding and with the oracle; the forward-only callback (k_model_fwd<false, ·>) returns the value the gradient callback returns;
    a non-finite θ_t gives -Inf and a zero gradient on both routes."""
    rng = np.random.default_rng(44)
    for case, build in ((model_golden[0], lambda: _reference_test_model(pkg)), (model_golden[1], lambda: _two_planet_model(pkg, model_golden[1]))):
        base = np.asarray(case["theta_t"])
        D = base.shape[0]
        W = 150                                    # 3 walker tiles, the last one ragged
        th = base[:, rng.integers(0, base.shape[1], W)] + 0.02 * rng.normal(size=(D, W))
        th[2, 5] = np.inf                          # logdensitymodel.jl:120-124
        out = {}
        for route in ("small", "throughput"):
            model = pkg.LogDensityModel(build())
            if route == "throughput":
                model.ln_like._check(model.ln_like.lib.octo_ctx_set_small_batch(model.ln_like._ctx, 0), "set")
            lp, g = model.logdensity_and_gradient(th)
            lpf = model.ℓπcallback(th)
            out[route] = (lp, g, lpf)
            if route == "throughput":
                obs, planets = _tables(case)
                nsrc = getattr(model, "_c_nsrc", None)
                lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, nsrc, th)
            model.close()
        (lp_s, g_s, lpf_s), (lp_t, g_t, lpf_t) = out["small"], out["throughput"]
        assert np.isneginf(lp_t[5]) and np.isneginf(lp_s[5]) and np.all(g_t[:, 5] == 0.0) and np.all(g_s[:, 5] == 0.0)
        assert np.array_equal(lpf_t, lp_t), "forward-only callback != value returned with the gradient (throughput route)"
        ok = np.isfinite(lp_o)
        assert ok.sum() >= W - 1 and np.array_equal(np.isfinite(lp_t), ok)
        assert np.max(np.abs(lp_t[ok] - lp_s[ok]) / np.maximum(1.0, np.abs(lp_s[ok]))) < 1e-12
        assert np.max(np.abs(lp_t[ok] - lp_o[ok]) / np.maximum(1.0, np.abs(lp_o[ok]))) < 1e-11
        sc = np.maximum(np.abs(g_o[:, ok]).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(g_t[:, ok] - g_s[:, ok]) / sc) < 1e-10
        assert np.max(np.abs(g_t[:, ok] - g_o[:, ok]) / sc) < 1e-9


@pytest.mark.gpu
def test_gpu_model_closed_form_tperi_edges(pkg, oracle, model_golden):
    """Round 4: θ_at_epoch_to_tperi of a Campbell planet is evaluated in closed form on the device (octo_model.h: tperi_campbell — the
    reference inverts the Thiele-Innes matrix, src/parameterizations.jl:29-57; the device rotates (cos θ, sin θ) back through Ω, i, ω and
    carries the analytic gradient). The corners of that rewrite: retrograde orbits (cos i < 0, where the sign of the de-projected direction
    flips), nearly edge-on and nearly face-on orbits, e -> 0 and e -> 0.99, position angles on both sides of the ±π cut of θ − Ω — through
    the fused small-batch launch and the throughput kernels, against the oracle's reference-order duals."""
    case = model_golden[0]
    obs, planets = _tables(case)
    base = np.asarray(case["theta_t"])[:, 0]
    model = pkg.LogDensityModel(_reference_test_model(pkg))
    nat0 = model.invlink(base[:, None])[:, 0]
    names = model.names
    cols = []
    for inc in (0.02, 0.7, 1.5607, 1.5809, 2.4, 3.12):
        for ecc in (1e-7, 0.3, 0.985):
            for ang in (0.1, 3.1, -3.1, -1.4):
                n = nat0.copy()
                n[names.index("b_i")] = inc; n[names.index("b_e")] = ecc
                n[names.index("b_θx")], n[names.index("b_θy")] = np.cos(ang) * 1.02, np.sin(ang) * 1.02
                n[names.index("b_Ωx")], n[names.index("b_Ωy")] = np.cos(ang - 3.0) * 0.97, np.sin(ang - 3.0) * 0.97
                cols.append(n)
    th = model.link(np.stack(cols, axis=1))
    model.close()
    W = th.shape[1]
    lp_o, g_o = None, None
    for route in ("small", "throughput"):
        model = pkg.LogDensityModel(_reference_test_model(pkg))
        if route == "throughput":
            model.ln_like._check(model.ln_like.lib.octo_ctx_set_small_batch(model.ln_like._ctx, 0), "set")
        lp, g = model.logdensity_and_gradient(th)
        if lp_o is None:
            lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th)
        model.close()
        ok = np.isfinite(lp_o)
        assert ok.all() and np.all(np.isfinite(lp)), route
        assert np.max(np.abs(lp - lp_o) / np.maximum(1.0, np.abs(lp_o))) < 1e-10, route
        # per walker: every component against the walker's largest one (the gradients span 1e0 … 1e12 over this grid)
        sc = np.maximum(np.abs(g_o).max(axis=0, keepdims=True), 1e-300)
        assert np.max(np.abs(g - g_o) / sc) < 1e-8, (route, float(np.max(np.abs(g - g_o) / sc)))
    assert W == 72


@pytest.mark.gpu
def test_gpu_model_reference_style(pkg):
    """test/integration/sampling.jl:136-192 ("Autodiff Gradient Comparison") and :70-76 re-expressed: the device
    gradient w.r.t. θ_t equals a finite-difference gradient of the device value (atol=1e-3, rtol=1e-4 there), prior draws
    map through link/invlink, and a good starting point has ℓπ > -1000."""
    table = dict(epoch=[50000, 50120, 50240, 50360], ra=[-505.76, -502.57, -498.21, -492.68], dec=[-66.93, -37.47, -7.93, 21.64],
                 σ_ra=[10.0] * 4, σ_dec=[10.0] * 4, cor=[0.0] * 4)
    model = pkg.LogDensityModel(_reference_test_model(pkg, table))
    assert model.D == 11
    rng = np.random.default_rng(42)
    θ = model.sample_priors(rng, 64)
    θ_t = model.link(θ)
    assert np.allclose(model.invlink(θ_t), θ, rtol=1e-12, atol=1e-12)
    lp, grad = model.logdensity_and_gradient(θ_t)
    assert np.all(np.isfinite(lp))
    for w in range(4):
        for k in range(11):
            h = 1e-6
            tp, tm = θ_t[:, w].copy(), θ_t[:, w].copy()
            tp[k] += h; tm[k] -= h
            fd = (model.ℓπcallback(tp) - model.ℓπcallback(tm)) / (2 * h)
            assert abs(fd - grad[k, w]) <= 1e-3 + 1e-4 * abs(fd), (w, k, fd, grad[k, w])
    # batched starting-point search, as guess_starting_position does with prior draws (src/initialization.jl:14-66)
    big = model.link(model.sample_priors(rng, 200_000))
    best = np.max(model.ℓπcallback(big))
    assert best > -1000                                                        # test/integration/sampling.jl:76
    # host-side Derived variables agree with what the device resolved (elements fed to the likelihood kernel)
    elems, nuis = model.kernel_inputs(θ[:, :8])
    ll = model.ln_like.ln_like_arrays(elems, None)
    assert np.all(np.isfinite(ll))
    model.close()


@pytest.mark.gpu
def test_gpu_batched_callers(pkg):
    """guess_starting_position (src/initialization.jl:14-66) and octofit_rejection (src/sampling.jl:168-256)
    on the batch path. Data are simulated from a known orbit; the rejection posterior must recover it (the reference's
    statistical-recovery style, e.g. test/integration/multi_planet.jl:70)."""
    import synth
    rng = np.random.default_rng(5)
    t = 50000.0 + 150.0 * np.arange(6)
    ra, dec = synth.truth_radec(t)                      # a=10, e=0.3, i=1.0, ω=0.5, Ω=2.0, tp=50000, M=1.2, plx=50
    table = dict(epoch=t, ra=ra + rng.normal(0, 100.0, 6), dec=dec + rng.normal(0, 100.0, 6), σ_ra=np.full(6, 100.0), σ_dec=np.full(6, 100.0))
    obs = pkg.PlanetRelAstromObs(table, name="sim")
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[obs],
                   variables=pkg.variables(a=pkg.LogUniform(5, 20), e=pkg.Uniform(0.0, 0.6), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    sys_ = pkg.System(name="sim", companions=[b], variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.05), lower=0.1),
                                                                          plx=pkg.truncated(pkg.Normal(50.0, 0.1), lower=0.1)))
    model = pkg.LogDensityModel(sys_)
    best, best_lp = pkg.guess_starting_position(rng, model, N=400_000)
    assert best.shape == (model.D,) and best_lp > -200
    assert best_lp == model.ℓπcallback(model.link(best))                    # returned params reproduce the returned logpost
    chain = pkg.octofit_rejection(rng, model, draws=2_000_000)             #   (test/unit/initialization.jl:55)
    assert chain["n_accepted"] >= 8 and chain["samples"].shape == (model.D, chain["n_accepted"])
    assert np.all(np.isfinite(chain["logpost"])) and np.all(chain["loglike"] <= np.max(chain["loglike"]))
    a = chain["samples"][model.names.index("b_a")]
    assert 6.0 < np.median(a) < 15.0                                         # truth a = 10 (prior 5-20)
    ll = pkg.rejection_evaluate_likelihoods(model, chain["samples"])
    assert np.allclose(ll, chain["loglike"], rtol=1e-13, atol=0)      # another batch size may take the other kernel family: equal to rounding
    model.close()


@pytest.mark.gpu
def test_host_jacobian_route_equals_device_model_route(pkg):
    """The two ways the Julia side reaches the device must give the same ∇θ_t:
      route 1 (`accelerate(system)`, any model): the kernel returns ḡ = ∂ll/∂(elements, nuisances) and the HOST applies the chain rule
              through its own θ_t -> inputs map: ∇θ_t = Jᵀ ḡ + ∇(prior) — ForwardDiff in Julia, central differences of the NumPy mirror here;
      route 2 (`HIPLogDensityModel`): the whole callback on the device (octo_model_logpost).
    One θ_t per call (both take the fused small-batch launch) and a 70-walker batch (throughput kernels)."""
    from octofitter_jl_amd.host.callers import _unit_length_terms
    model = pkg.LogDensityModel(_reference_test_model(pkg))
    fn = model.ln_like
    rng = np.random.default_rng(23)
    for W in (1, 70):
        θt = model.link(model.sample_priors(rng, W))
        lp, g = model.logdensity_and_gradient(θt)
        assert np.all(np.isfinite(lp))

        def inputs(tt):                                   # θ_t -> (elements, nuisances) on the host, like arr2nt ∘ invlink
            el, nu = model.kernel_inputs(model.invlink(tt))
            return el

        def logpdf_with_trans(p, x, y):                   # a test-local NumPy statement of Distributions' logpdf + Bijectors' log-Jacobian
            from math import erf, isfinite, log, pi, sqrt
            a, b = p.bounds()
            if isfinite(a) and isfinite(b):
                sg = 1.0 / (1.0 + np.exp(-y)); ladj = log(b - a) + np.log(sg) + np.log1p(-sg)
            elif isfinite(a) or isfinite(b):
                ladj = y
            else:
                ladj = 0.0
            name = type(p).__name__
            if name == "Uniform":
                return -log(b - a) + ladj
            if name == "LogUniform":
                return -np.log(x) - log(log(b / a)) + ladj
            if name == "Sine":
                return np.log(np.sin(x) / 2) + ladj
            z = (x - p.μ) / p.σ
            lp = -0.5 * z * z - log(p.σ) - 0.5 * log(2 * pi)
            if name == "TruncatedNormal":
                Φ = lambda v: 0.5 * (1 + erf((v - p.μ) / p.σ / sqrt(2)))
                lp = lp - log((Φ(p.hi) if isfinite(p.hi) else 1.0) - (Φ(p.lo) if isfinite(p.lo) else 0.0))
            return lp + ladj

        def host_terms(tt):                               # prior with its log-Jacobian + the epoch-free UnitLengthPrior terms
            θ = model.invlink(tt)
            return sum(logpdf_with_trans(p, θ[k], tt[k]) for k, p in enumerate(model.priors)) + _unit_length_terms(model, θ)
        el = inputs(θt)
        ll, g_el, _ = fn.ln_like_arrays(el, None, grad=True)
        assert np.all(np.abs(ll + host_terms(θt) - lp) <= 1e-11 * np.abs(lp))
        grad1 = np.zeros_like(θt)
        for d in range(model.D):
            h = 1e-6 * max(1.0, np.abs(θt[d]).max())
            tp_, tm_ = θt.copy(), θt.copy(); tp_[d] += h; tm_[d] -= h
            J_d = (inputs(tp_) - inputs(tm_)) / (2 * h)                   # ∂inputs/∂θ_t[d], [9, W]
            grad1[d] = np.sum(J_d * g_el, axis=0) + (host_terms(tp_) - host_terms(tm_)) / (2 * h)
        sc = np.maximum(np.abs(g).max(axis=1, keepdims=True), 1e-300)
        assert np.all(np.abs(grad1 - g) <= 2e-6 * sc), (W, np.max(np.abs(grad1 - g) / sc))      # finite-difference accuracy
    model.close()


@pytest.mark.gpu
def test_gpu_batched_callers_vs_oracle(pkg, oracle):
    """SURVEY §8 f2 with a real parity check (the callers were only compared with themselves): IDENTICAL prior draws and
    uniforms go to the device callers and to the CPU restatement of the callback; the starting point (argmax), every log-posterior,
    the accept mask of the rejection sampler and the accepted chain must coincide. Also the Pigeons-style `model(Θ)` and
    `pointwise_like` (src/cross-validation.jl:17-46)."""
    import synth
    rng = np.random.default_rng(17)
    t = 50000.0 + 90.0 * np.arange(10)
    ra, dec = synth.truth_radec(t)
    table = dict(epoch=t, ra=ra + rng.normal(0, 60.0, 10), dec=dec + rng.normal(0, 60.0, 10), σ_ra=np.full(10, 60.0), σ_dec=np.full(10, 60.0))
    rvt = dict(epoch=t + 7.0, rv=rng.normal(0, 30, 10), σ_rv=np.full(10, 8.0))
    astrom = pkg.PlanetRelAstromObs(table, name="sim", variables=pkg.variables(jitter=pkg.LogUniform(0.1, 30.0)))
    rv = pkg.StarAbsoluteRVObs(rvt, name="rv", variables=pkg.variables(offset=pkg.Normal(0, 20), jitter=pkg.LogUniform(0.1, 20.0)))
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[astrom],
                   variables=pkg.variables(a=pkg.LogUniform(5, 20), e=pkg.Uniform(0.0, 0.6), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000),
                                           mass=pkg.LogUniform(1.0, 50.0)))
    sys_ = pkg.System(name="sim", companions=[b], observations=[rv],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.05), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.1), lower=0.1)))
    model = pkg.LogDensityModel(sys_)
    N = 30_000
    draws = model.sample_priors(rng, N)
    fn = model.ln_like
    # ---- guess_starting_position on the given draws vs the oracle's ℓπcallback on the same draws
    best, best_lp = pkg.guess_starting_position(rng, model, prior_samples=draws, batch=7_000)
    lp_o, _ = oracle.oracle_model_logpost(fn.obs_tables, fn.planet_desc, model._c_priors, model._c_esrc, model._c_nsrc, model.link(draws), grad=False, n_threads=0)
    k = int(np.argmax(lp_o))
    assert np.array_equal(best, draws[:, k]) and abs(best_lp - lp_o[k]) <= 1e-11 * abs(lp_o[k])
    lp_d = model(model.link(draws))                                         # Pigeons-style call on the whole batch
    assert np.all(np.abs(lp_d - lp_o) <= 1e-11 * np.abs(lp_o))
    # ---- octofit_rejection with the same draws and uniforms vs the oracle's likelihood
    u = rng.uniform(0, 1, N)
    chain = pkg.octofit_rejection(rng, model, prior_samples=draws, uniforms=u)
    elems, nuis = model.kernel_inputs(draws)
    ll_o, _, _ = oracle.oracle_eval(fn.obs_tables, fn.planet_desc, elems, nuis, grad=False, n_threads=0)
    from octofitter_jl_amd.host.callers import _unit_length_terms
    ll_o = ll_o + _unit_length_terms(model, draws)
    ll_o = np.where(np.isfinite(ll_o), ll_o, -np.inf)
    acc_o = (ll_o != -np.inf) & (u < np.exp(ll_o - ll_o.max()))
    assert np.all(np.abs(chain["all_loglike"] - ll_o) <= 1e-11 * np.maximum(1, np.abs(ll_o)))
    assert np.array_equal(chain["accept"], acc_o) and chain["n_accepted"] == int(acc_o.sum()) >= 1
    assert np.array_equal(chain["samples"], draws[:, acc_o])
    # ---- pointwise_like: per-observation columns sum to the total likelihood (without the epoch-free UnitLengthPrior terms)
    sub = draws[:, :500]
    LL, names = pkg.pointwise_like(model, sub)
    assert LL.shape == (500, 2) and names == ["sim", "rv"]
    el_s, nu_s = model.kernel_inputs(sub)
    tot = fn.ln_like_arrays(el_s, nu_s)
    assert np.all(np.abs(LL.sum(axis=1) - tot) <= 1e-11 * np.maximum(1, np.abs(tot)))
    for io in range(2):
        one_o, _, _ = oracle.oracle_eval([fn.obs_tables[io]], fn.planet_desc, el_s, nu_s[io * 3:(io + 1) * 3], grad=False, n_threads=0)
        assert np.all(np.abs(LL[:, io] - one_o) <= 1e-11 * np.maximum(1, np.abs(one_o))), names[io]
    model.close()


@pytest.mark.gpu
def test_gpu_model_hgca(pkg, oracle):
    """A joint fit in the reference's style (docs: astrometry + HGCAInstantaneousObs): system variables pmra, pmdec feed the
    HGCA term (hgca.jl:266-267), the planet's mass its reflex motion. Device ℓπ and ∇ℓπ w.r.t. θ_t against the oracle's
    restatement of the whole callback, and the gradient against finite differences (test/integration/sampling.jl:136-192)."""
    from test_host import HGCA_ROW
    table = dict(epoch=[50000, 50120, 50240, 50360], ra=[-505.76, -502.57, -498.21, -492.68], dec=[-66.93, -37.47, -7.93, 21.64],
                 σ_ra=[10.0] * 4, σ_dec=[10.0] * 4)
    astrom_like = pkg.PlanetRelAstromLikelihood(table, name="astrom")
    hg = pkg.HGCAInstantaneousObs(hgca=HGCA_ROW, N_ave=2)
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[astrom_like],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000),
                                           mass=pkg.LogUniform(1.0, 100.0)))
    sys_ = pkg.System(name="HGCASys", companions=[b], observations=[hg],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1),
                                              pmra=pkg.Normal(4.3, 1.0), pmdec=pkg.Normal(-2.0, 1.0)))
    model = pkg.LogDensityModel(sys_)
    assert model.D == 14 and model.names[:4] == ["M", "plx", "pmra", "pmdec"] and model.names[-1] == "b_mass"
    rng = np.random.default_rng(3)
    θ_t = model.link(model.sample_priors(rng, 96))
    lp, g = model.logdensity_and_gradient(θ_t)
    assert np.all(np.isfinite(lp))
    fn = model.ln_like
    lp_o, g_o = oracle.oracle_model_logpost(fn.obs_tables, fn.planet_desc, model._c_priors, model._c_esrc, model._c_nsrc, θ_t)
    assert np.all(np.abs(lp - lp_o) <= 1e-12 * np.maximum(1, np.abs(lp_o)))
    scale = np.abs(g_o).max(axis=1, keepdims=True)
    assert np.all(np.abs(g - g_o) <= 1e-9 * np.abs(g_o) + 1e-11 * scale), np.max(np.abs(g - g_o) / (np.abs(g_o) + scale))
    for k in (2, 3, 13):          # pmra, pmdec, mass: the inputs only the HGCA term sees
        h = 1e-6
        tp, tm = θ_t[:, 0].copy(), θ_t[:, 0].copy()
        tp[k] += h; tm[k] -= h
        fd = (model.ℓπcallback(tp) - model.ℓπcallback(tm)) / (2 * h)
        assert abs(fd - g[k, 0]) <= 1e-3 + 1e-4 * abs(fd), (k, fd, g[k, 0])
    model.close()


@pytest.mark.gpu
def test_gpu_model_thiele_innes_tutorial(pkg, oracle, model_golden):
    """docs/src/thiele-innes.md / test/unit/constructors.jl:124-153: a planet on the ThieleInnesOrbit basis — A, B, F, G ~
    Normal(0, 1000) mas, tp = θ_at_epoch_to_tperi(θ, 50000; plx, M, e, A, B, F, G). Whole callback against the 60-digit
    fixture and the oracle; and the same physical orbit gives the same likelihood in both bases."""
    case = model_golden[2]
    table = dict(epoch=case["obs"][0]["epoch"], ra=case["obs"][0]["y1"], dec=case["obs"][0]["y2"], σ_ra=case["obs"][0]["s1"],
                 σ_dec=case["obs"][0]["s2"], cor=case["obs"][0]["cor"])
    astrom_like = pkg.PlanetRelAstromObs(table, name="GPI")
    b = pkg.Planet(name="b", basis="ThieleInnesOrbit", observations=[astrom_like],
                   variables=pkg.variables(e=pkg.Uniform(0.0, 0.5), A=pkg.Normal(0, 1000), B=pkg.Normal(0, 1000), F=pkg.Normal(0, 1000),
                                           G=pkg.Normal(0, 1000), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000.0)))
    sys_ = pkg.System(name="TutoriaPrime", companions=[b], observations=[],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1)))
    model = pkg.LogDensityModel(sys_)
    assert model.D == 9 and model.names == ["M", "plx", "b_e", "b_A", "b_B", "b_F", "b_G", "b_θx", "b_θy"]
    for k, s_ in enumerate(case["esrc"]):
        c = model._c_esrc[k]
        assert (c.kind, c.i0, c.i1, c.flags, c.value) == (s_["kind"], s_["i0"], s_["i1"], s_["flags"], s_["value"])
    th = np.asarray(case["theta_t"])
    lp, g = model.logdensity_and_gradient(th)
    _check(lp, g, case)
    obs, planets = _tables(case)
    lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th)
    assert np.all(np.abs(lp - lp_o) <= 1e-12 * np.abs(lp_o))
    # same orbit, two bases: ln_like(Campbell a, i, ω, Ω) == ln_like(Thiele-Innes A, B, F, G = a·plx·R)
    rng = np.random.default_rng(8)
    W = 33
    a, e, inc, w, O = rng.uniform(5, 20, W), rng.uniform(0, 0.5, W), np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W)
    tp = 50000 + rng.uniform(0, 3000, W)
    T = a * 50.0
    cO, sO, cw, sw, ci = np.cos(O), np.sin(O), np.cos(w), np.sin(w), np.cos(inc)
    θ_ti = dict(M=1.2, plx=50.0, planets=dict(b=dict(e=e, tp=tp, A=T * (cO * cw - sO * sw * ci), B=T * (sO * cw + cO * sw * ci),
                                                      F=T * (-cO * sw - sO * cw * ci), G=T * (-sO * sw + cO * cw * ci))))
    ll_ti, g_ti = model.ln_like.ln_like_and_grad(θ_ti)
    bc = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="GPI")])
    θ_c = dict(M=1.2, plx=50.0, planets=dict(b=dict(a=a, e=e, i=inc, ω=w, Ω=O, tp=tp)))
    fc = pkg.make_ln_like(pkg.System(name="c", companions=[bc]), θ_c)
    ll_c, g_c = fc.ln_like_and_grad(θ_c)
    fc.close()
    assert np.all(np.abs(ll_ti - ll_c) <= 1e-12 * np.abs(ll_c))
    for key in ("e", "tp", "M"):
        ref = g_c["planets"]["b"][key]
        assert np.all(np.abs(g_ti["planets"]["b"][key] - ref) <= 1e-9 * np.abs(ref).max()), key
    assert set(g_ti["planets"]["b"]) >= {"A", "B", "F", "G"}
    model.close()


@pytest.mark.gpu
def test_gpu_model_relative_rv_trend(pkg, oracle, model_golden):
    """The model of OctofitterRadialVelocity/test/runtests.jl:168-231 through the whole callback: D = 3 (offset ~ Normal(0, 200),
    jitter ~ LogUniform(0.01, 50), trend_slope ~ Normal(0, 1)), fixed circular orbit, trend_function = θ_obs.trend_slope * (epoch − 50000).
    One θ_t per call (the fused k_small<MODEL> launch) and a batch, against the 60-digit fixture and the oracle."""
    case = model_golden[3]
    assert case["name"] == "D3_relative_rv_offset_trend"
    ob0 = case["obs"][0]
    rvlike = pkg.PlanetRelativeRVObs(dict(epoch=ob0["epoch"], rv=ob0["y1"], σ_rv=ob0["s1"]), name="RelRV",
                                     trend_function=lambda θ_obs, epoch: θ_obs.trend_slope * (epoch - 50000.0),
                                     variables=pkg.variables(offset=pkg.Normal(0, 200), jitter=pkg.LogUniform(0.01, 50), trend_slope=pkg.Normal(0, 1)))
    true_P, true_M = 80.0, 1.0
    b = pkg.Planet(name="b", basis="RadialVelocityOrbit", observations=[rvlike],
                   variables=pkg.variables(M=true_M, e=0.0, ω=0.0, a=float(np.cbrt(true_P ** 2 * true_M)), tp=50000.0, mass=0.0))
    model = pkg.LogDensityModel(pkg.System(name="RelRVSys", companions=[b]))
    assert model.D == 3 and model.names == ["b_RelRV_offset", "b_RelRV_jitter", "b_RelRV_trend_slope"]      # runtests.jl:229-230 reads chain[:b_RelRV_offset]
    for k, s_ in enumerate(case["nsrc"]):
        c = model._c_nsrc[k]
        assert (c.kind, c.i0, c.value) == (s_["kind"], s_["i0"], s_["value"])
    for k, s_ in enumerate(case["esrc"]):
        c = model._c_esrc[k]
        assert (c.kind, c.value) == (s_["kind"], s_["value"]), k
    th = np.asarray(case["theta_t"])
    lp, g = model.logdensity_and_gradient(th)
    _check(lp, g, case)
    for w in range(th.shape[1]):                                            # one θ_t per call
        lp1, g1 = model.logdensity_and_gradient(th[:, w])
        _check(np.array([lp1]), g1[:, None], dict(lp=[case["lp"][w]], grad=np.asarray(case["grad"])[:, w:w + 1].tolist()))
    rng = np.random.default_rng(3)
    thb = np.stack([rng.normal(50, 20, 2000), rng.normal(0, 1.5, 2000), rng.normal(0.1, 0.1, 2000)])      # the throughput kernels
    lpb, gb_ = model.logdensity_and_gradient(thb)
    obs, planets = _tables(case)
    lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, model._c_nsrc, thb, n_threads=0)
    assert np.all(np.abs(lpb - lp_o) <= 1e-12 * np.maximum(1, np.abs(lp_o)))
    assert np.all(np.abs(gb_ - g_o) <= 1e-9 * np.abs(g_o).max(axis=1, keepdims=True))
    model.close()


@pytest.mark.gpu
def test_gpu_model_many_observation_tables(pkg):
    """Seventeen RA/Dec tables on one planet against the same rows as ONE table: the fused one-θ launch stages the model's
    descriptors (priors, sources, one source triple per observation) as one block in LDS, so the number of observations is part of its
    shape. Same log-posterior and gradient to rounding (the split changes the order of the row sums only), one θ_t and a batch."""
    rng = np.random.default_rng(7)
    n_tab, per = 17, 3
    ep = np.sort(50000 + rng.uniform(0, 3000, n_tab * per))
    ra, dec = rng.normal(-480, 30, n_tab * per), rng.normal(40, 60, n_tab * per)
    one = dict(epoch=ep, ra=ra, dec=dec, σ_ra=np.full(ep.size, 10.0), σ_dec=np.full(ep.size, 12.0))
    def build(tables):
        likes = [pkg.PlanetRelAstromLikelihood(t, name=f"t{k}") for k, t in enumerate(tables)]
        b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=likes,
                       variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                               Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
        return pkg.LogDensityModel(pkg.System(name="many", companions=[b], observations=[],
                                   variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    split = [{k: v[j * per:(j + 1) * per] for k, v in one.items()} for j in range(n_tab)]
    m1, m17 = build([one]), build(split)
    θ_t = m1.link(m1.sample_priors(np.random.default_rng(3), 33))
    for W in (1, 33):
        lp1, g1 = m1.logdensity_and_gradient(θ_t[:, :W])
        lp17, g17 = m17.logdensity_and_gradient(θ_t[:, :W])
        assert np.all(np.isfinite(lp1)) and np.allclose(lp17, lp1, rtol=1e-12, atol=0)
        assert np.allclose(g17, g1, rtol=1e-9, atol=1e-9 * np.abs(g1).max())
    m1.close(); m17.close()


@pytest.mark.gpu
def test_gpu_model_five_planets(pkg, oracle):
    """The whole callback for a system of FIVE planets (more than the templated kernels are compiled for): θ_t -> elements (k_model_fwd) ->
    planet-per-wave epoch loop (k_mainp) -> finish with the model's tail (k_finishp: model_tail_n) — log-posterior and ∇θ_t of a batch of 70
    and of ONE θ_t (no fused small-batch launch beyond four planets) against the oracle's callback; the reference builds such a system with the
    same unrolled code as any other (src/likelihoods/system.jl:116-118, 156-170)."""
    import synth
    rng = np.random.default_rng(23)
    planets = []
    for k in range(5):
        t = np.sort(50000.0 + rng.uniform(0, 3000, 9))
        tab = dict(epoch=t, ra=rng.normal(0, 200, 9), dec=rng.normal(0, 200, 9), σ_ra=np.full(9, 30.0), σ_dec=np.full(9, 30.0))
        obs = [pkg.PlanetRelAstromObs(tab, name=f"astrom{k}", variables=pkg.variables(jitter=pkg.LogUniform(0.1, 30.0)) if k == 2 else None)]
        planets.append(pkg.Planet(name=f"p{k}", basis="Visual{KepOrbit}", observations=obs,
                                  variables=pkg.variables(a=pkg.LogUniform(1.0 + 3 * k, 3.0 + 3 * k), e=pkg.Uniform(0.0, 0.6), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                                          Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000),
                                                          mass=pkg.LogUniform(0.5, 30.0))))
    t = np.sort(50000.0 + rng.uniform(0, 3000, 14))
    rv = pkg.StarAbsoluteRVObs(dict(epoch=t, rv=rng.normal(0, 40, 14), σ_rv=np.full(14, 6.0)), name="rv",
                               variables=pkg.variables(offset=pkg.Normal(0, 20), jitter=pkg.LogUniform(0.1, 20.0)))
    sys_ = pkg.System(name="five", companions=planets, observations=[rv],
                      variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.05), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.1), lower=0.1)))
    model = pkg.LogDensityModel(sys_)
    fn = model.ln_like
    assert fn.n_planets == 5 and model.D == 2 + 2 + 1 + 5 * 10
    th = model.link(model.sample_priors(rng, 70))
    for sub in (th, th[:, :1]):
        lp, g = model.logdensity_and_gradient(sub)
        lp_o, g_o = oracle.oracle_model_logpost(fn.obs_tables, fn.planet_desc, model._c_priors, model._c_esrc, model._c_nsrc, sub, grad=True)
        ok = np.isfinite(lp_o)
        assert ok.all() and np.all(np.abs(lp - lp_o) <= 1e-11 * np.abs(lp_o))
        sc = np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300)
        assert np.all(np.abs(g - g_o) / sc < 1e-9), (np.abs(g - g_o) / sc).max()
        assert np.array_equal(model(sub), lp)                          # the forward-only callback returns the gradient callback's value
    model.close()


@pytest.mark.gpu
def test_gpu_model_small_batch_without_the_fused_launch(pkg, oracle):
    """A model whose observations outgrow the fused small-batch launch (70 tables, five of them with a jitter variable: 210 nuisance inputs against
    the 192 the one-θ launch shares through LDS — fused_ok is false) still takes SMALL batches through the small-batch likelihood kernel, with the
    model transform ahead of it (k_model_fwd) and its tail behind it (k_model_bwd): the route on which no k_finish launch carries the model's
    tail, i.e. the `tail applied == false` branch of octo_model_logpost_device (ADVICE r4 asked for a test of it). One θ_t, three, and a batch
    of 70 on the throughput kernels (tail inside k_finish) against the oracle's callback."""
    rng = np.random.default_rng(29)
    n_tab, per = 70, 2
    ep = np.sort(50000 + rng.uniform(0, 3000, n_tab * per))
    likes = []
    for k in range(n_tab):
        sl = slice(k * per, (k + 1) * per)
        tab = dict(epoch=ep[sl], ra=rng.normal(-480, 30, per), dec=rng.normal(40, 60, per), σ_ra=np.full(per, 10.0), σ_dec=np.full(per, 12.0))
        likes.append(pkg.PlanetRelAstromObs(tab, name=f"t{k}", variables=pkg.variables(jitter=pkg.LogUniform(0.1, 30.0)) if k % 14 == 3 else None))
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=likes,
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="many", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    fn = model.ln_like
    assert model.D == 11 + 5
    th = model.link(model.sample_priors(np.random.default_rng(3), 70))
    lp_o, g_o = oracle.oracle_model_logpost(fn.obs_tables, fn.planet_desc, model._c_priors, model._c_esrc, model._c_nsrc, th, grad=True)
    sc = np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300)
    for W in (1, 3, 70):
        sub = th[:, :W]
        lp, g = model.logdensity_and_gradient(sub)
        assert np.all(np.isfinite(lp)) and np.all(np.abs(lp - lp_o[:W]) <= 1e-11 * np.abs(lp_o[:W])), W
        assert np.all(np.abs(g - g_o[:, :W]) / sc < 1e-9), (W, (np.abs(g - g_o[:, :W]) / sc).max())
        assert np.array_equal(model(sub), lp)
    model.close()


@pytest.mark.gpu
def test_gpu_model_fused_launch_up_to_768_single_planet_callbacks(pkg, oracle, model_golden):
    """Round 5: single-planet whole-callback batches take the fused launch (k_small<MODEL>) up to 768 θ_t — three launches is what they would pay
    otherwise (tools/r5_midsize_small.py: 40 µs against 43 at 768, 46-48 against 45 at 1 024); likelihood-only batches keep the 512 limit. 700 θ_t
    through the library's default route and forced onto the throughput kernels: the same log-posterior and gradient to rounding, and the oracle's."""
    case = model_golden[0]
    model = pkg.LogDensityModel(_reference_test_model(pkg))
    fn = model.ln_like
    th = model.link(model.sample_priors(np.random.default_rng(41), 700))
    lp, g = model.logdensity_and_gradient(th)
    fn._check(fn.lib.octo_ctx_set_small_batch(fn._ctx, 0), "set")      # throughput kernels
    lp_t, g_t = model.logdensity_and_gradient(th)
    obs, planets = _tables(case)
    lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th, n_threads=0)
    ok = np.isfinite(lp_o)
    assert ok.sum() > 600 and np.array_equal(np.isfinite(lp), ok) and np.array_equal(np.isfinite(lp_t), ok)
    for a, ga in ((lp, g), (lp_t, g_t)):
        assert np.all(np.abs(a[ok] - lp_o[ok]) <= 1e-11 * np.abs(lp_o[ok]))
        sc = np.maximum(np.abs(g_o[:, ok]).max(axis=1, keepdims=True), 1e-300)
        assert np.all(np.abs(ga[:, ok] - g_o[:, ok]) / sc < 1e-9)
    assert not np.array_equal(lp, lp_t) or not np.array_equal(g, g_t), "both calls took the same route"
    model.close()
