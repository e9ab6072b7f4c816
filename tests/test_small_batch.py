"""
GPU parity of the small-batch path (-m gpu): octo_eval with W <= 32 takes the fused single-launch kernel k_small — epochs across
the 64 lanes, DPP/LDS tree reduction, last-block finish, inputs/outputs in mapped pinned memory (what a sampler that evaluates
ONE θ per call pays: src/logdensitymodel.jl:169-177, src/likelihoods/system.jl:257-269); the same kernel serves mid-size batches
(W·P <= 512: an ensemble sampler's walkers). Checked against the oracle at W ∈ {1, 2, 31, 32, 33, 128, 129, 512} and against the
throughput kernels on the same inputs (same math, different summation order: equal to rounding, not bitwise).
"""
import numpy as np
import pytest

import synth
from conftest import case_tables, rel_err
from test_gpu_parity import _cmp_oracle, _gpu
from test_oracle import grad_ok

pytestmark = pytest.mark.gpu


def _both(obs, planets, elems, nuis):
    gb = _gpu()
    small = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    small_f = gb.gpu_eval(obs, planets, elems, nuis, grad=False)
    big = gb.gpu_eval(obs, planets, elems, nuis, grad=True, small_batch=0)
    assert np.array_equal(small[0], small_f[0]), "forward-only and gradient launches disagree"
    return small, big


def _close(small, big, tol=2e-12, gtol=1e-10):
    ll, g, gn = small
    ll_b, g_b, gn_b = big
    fin = np.isfinite(ll_b)
    assert np.array_equal(np.isfinite(ll), fin)
    assert np.all(np.abs(ll[fin] - ll_b[fin]) <= tol * np.maximum(1.0, np.abs(ll_b[fin])))
    sc = np.maximum(np.abs(g_b).max(axis=1, keepdims=True), 1e-300)
    assert np.all(np.abs(g - g_b) <= gtol * sc), np.max(np.abs(g - g_b) / sc)
    if gn is not None:
        sc = np.maximum(np.abs(gn_b).max(axis=1, keepdims=True), 1e-300)
        assert np.all(np.abs(gn - gn_b) <= gtol * sc)


@pytest.mark.parametrize("n_walkers", [1, 2, 31, 32, 33, 128, 129, 512])      # 128/129: mapped pinned buffers -> one DMA each way
@pytest.mark.parametrize("n_epochs", [1, 50, 300, 10_000])
def test_astrometry_small_batches_vs_oracle(oracle, n_epochs, n_walkers):
    cfg = synth.config_astrom(n_epochs=n_epochs, n_walkers=n_walkers, seed=7000 + n_epochs + n_walkers)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    small, big = _both(obs, planets, cfg["elems"], None)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, cfg["elems"], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle(f"small {n_epochs}x{n_walkers}", small[0], small[1], None, ll_o, g_o, None)
    _close(small, big)
    # bit-reproducible run to run (fixed reduction tree, partials summed in task order whichever block finishes last)
    again = _gpu().gpu_eval(obs, planets, cfg["elems"], None, grad=True)
    assert np.array_equal(small[0], again[0]) and np.array_equal(small[1], again[1])


@pytest.mark.parametrize("n_walkers", [1, 5, 32])
def test_all_kinds_two_planets_small_batches(oracle, n_walkers):
    """Two planets, every epoch-loop kind (RA/Dec with cor, sep/PA, relative RV, absolute RV, marginalised RV; nuisances)."""
    cfg = synth.config_two_planet(n_astrom=300, n_rv=280, n_walkers=n_walkers, seed=5)
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
    ]
    planets = [dict(orbit_kind=0, has_mass=True), dict(orbit_kind=0, has_mass=True)]
    W = n_walkers
    nuis = np.zeros((len(obs) * 3, W))
    for o in range(3):
        nuis[o * 3 + 0] = rng.uniform(0, 5, W); nuis[o * 3 + 1] = rng.normal(1, 0.01, W); nuis[o * 3 + 2] = rng.normal(0, 0.01, W)
    nuis[0, : W // 2] = 0.0
    for o in range(3, 5):
        nuis[o * 3 + 0] = rng.normal(10, 3, W); nuis[o * 3 + 1] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    elems = cfg["elems"].copy()
    elems[9 + 0, : W // 3] = elems[0, : W // 3] * 0.5      # "outer" planet inside the "inner" one for some walkers
    for nz in (nuis, None):
        small, big = _both(obs, planets, elems, nz)
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, elems, nz, grad=True)
        _cmp_oracle("all kinds small", small[0], small[1], small[2], ll_o, g_o, gn_o, ll_rtol=1e-11, g_rtol=1e-8)
        _close(small, big, tol=1e-11)
    # marginalised RV (rv-absolute-margin.jl:171-181): k_small gives such a table to ONE block, which runs the forward rows for μ̂ first
    obs_m = obs + [dict(kind=3, planet=-1, epoch=r["epoch"][::2], y1=r["rv"][::2] + 3.0, y2=None, s1=r["σ_rv"][::2] * 1.5, s2=None, cor=None)]
    nuis_m = np.concatenate([nuis, np.stack([np.zeros(W), np.exp(rng.uniform(np.log(0.1), np.log(10), W)), np.zeros(W)])])
    for nz in (nuis_m, None):
        small, big = _both(obs_m, planets, elems, nz)
        ll_o, g_o, gn_o = oracle.oracle_eval(obs_m, planets, elems, nz, grad=True)
        _cmp_oracle("all kinds + marginalised RV, small", small[0], small[1], small[2], ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)
        _close(small, big, tol=1e-9, gtol=1e-8)      # the reference's −B²/(4A) + C cancels (rv-absolute-margin.jl:181)


@pytest.mark.parametrize("P,W", [(3, 1), (3, 32), (4, 1), (4, 7)])
def test_three_and_four_planets_small_batches(oracle, P, W):
    """k_small is compiled for every planet count the library takes (1-4): random systems on mixed bases with every epoch-loop
    kind, against the oracle (bars of test_gpu_parity) and against the throughput kernels on the same inputs."""
    import stress_parity as sp
    rng = np.random.default_rng(900 + 10 * P + W)
    took_small = 0
    for k in range(40):
        if took_small == 3:
            break
        sysm = sp.draw_system(rng, P=P, W=W)
        good, e_ll, e_g, loose = sp.check_system(sysm)
        assert good, (k, sp.describe(sysm), e_ll, e_g)
        obs, planets, elems, nuis = sysm
        took_small += 1
        small, big = _both(obs, planets, elems, nuis)
        marg = any(o["kind"] == 3 for o in obs)
        _close(small, big, tol=1e-9 if marg else 1e-10, gtol=1e-8 if marg else 1e-10)
    assert took_small == 3


@pytest.mark.parametrize("P,W", [(3, 1), (4, 1), (4, 9)])
def test_three_and_four_planet_models_fused_launch(pkg, oracle, P, W):
    """The whole callback in one launch (k_small<MODEL>) for 3- and 4-planet systems: random standard-parameterisation models
    against the oracle's forward-mode log-posterior."""
    import stress_model as sm
    rng = np.random.default_rng(700 + 10 * P + W)
    lib = pkg.capi.load_library()
    done = 0
    for k in range(60):
        if done == 3:
            break
        r = sm.check_model(rng, lib, P=P, W=W)
        if r is None:
            continue
        assert r[0], (k,) + r[1:]
        done += 1
    assert done == 3


def test_golden_vectors_small_and_throughput_paths(golden):
    """Every committed 50/60-digit fixture through BOTH kernel families (the fixtures hold 1-16 walkers, so the default route is
    k_small; small_batch=0 forces k_setup/k_main/k_finish)."""
    gb = _gpu()
    for case in golden["cases"]:
        obs, planets, elems, nuis = case_tables(case)
        has_marg = any(ob["kind"] == "RV_ABS_MARG" for ob in case["obs"])
        for sb in (None, 0):
            ll, g_el, g_nu = gb.gpu_eval(obs, planets, elems, nuis, grad=True, small_batch=sb)
            err = rel_err(ll, np.asarray(case["ll"]), 1.0)
            assert np.all(err < (1e-9 if has_marg else 1e-12)), (case["name"], sb, "ll", err.max())
            rtol, cancel = (1e-9, 1e-10) if (has_marg or case["name"] == "F7_kepler_edges") else (1e-9, 1e-13)
            ok, worst = grad_ok(g_el, case["g_elems"], case["s_elems"], rtol=rtol, cancel=cancel)
            assert ok, (case["name"], sb, "g_elems", worst)
            if nuis is not None:
                ok, worst = grad_ok(g_nu, case["g_nuis"], case["s_nuis"], rtol=rtol, cancel=cancel)
                assert ok, (case["name"], sb, "g_nuis", worst)


def test_invalid_walkers_small_batch():
    gb = _gpu()
    cfg = synth.config_astrom(n_epochs=700, n_walkers=8, seed=3)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = cfg["elems"].copy()
    ref, gref, _ = gb.gpu_eval(obs, planets, el, None, grad=True)
    bad = el.copy()
    bad[1, 0] = 1.0; bad[1, 1] = -0.1; bad[0, 2] = -3.0; bad[6, 3] = 0.0; bad[5, 4] = np.nan; bad[1, 5] = 1e30; bad[3, 6] = np.inf
    ll, g, _ = gb.gpu_eval(obs, planets, bad, None, grad=True)
    assert np.all(np.isneginf(ll[:7])) and np.all(g[:, :7] == 0.0)
    assert ll[7] == ref[7] and np.array_equal(g[:, 7], gref[:, 7])      # a neighbour's garbage does not leak


@pytest.mark.parametrize("W", [1, 2, 31, 32, 33])
def test_model_callback_fused_launch_vs_oracle(pkg, oracle, W):
    """The whole log-posterior callback in ONE launch (k_small<MODEL>: θ_t -> priors -> elements -> likelihood -> ∇θ_t, lane = partial)
    for the committed D = 11 and D = 25 (two planets, RV, nuisance priors) models at W ∈ {1, 2, 31, 32, 33} θ_t: against the oracle's
    restatement of the callback, against the 60-digit fixture where the committed θ_t are used, and against the throughput kernels."""
    import json
    from pathlib import Path
    import ctypes as C
    from test_model import _tables
    gold = json.loads((Path(__file__).resolve().parent / "golden" / "model.json").read_text())["cases"]
    capi = pkg.capi
    lib = capi.load_library()
    for case in gold:
        obs, planets = _tables(case)
        pr, es = oracle.make_priors(case["priors"]), oracle.make_sources(case["esrc"])
        ns = oracle.make_sources(case["nsrc"]) if case["nsrc"] else None
        base = np.asarray(case["theta_t"])
        D, W0 = base.shape
        rng = np.random.default_rng(100 + W)
        th = np.tile(base, (1, W // W0 + 1))[:, :W].copy()
        th[:, W0:] += 0.05 * rng.normal(size=(D, max(W - W0, 0)))         # beyond the committed θ_t: perturbed copies
        th = np.ascontiguousarray(th)
        lp_o, g_o = oracle.oracle_model_logpost(obs, planets, pr, es, ns, th, n_threads=0)
        res = {}
        for sb in (None, 0):
            path = _gpu().GpuPath(obs, planets, small_batch=sb)
            m = C.c_void_p()
            assert lib.octo_model_create(path.ctx, path.ds, pr, D, es, ns, C.byref(m)) == 0, lib.octo_last_error(path.ctx)
            lp = np.full(W, np.nan); g = np.full_like(th, np.nan); lp0 = np.full(W, np.nan)
            assert lib.octo_model_logpost(path.ctx, m, capi._dptr(th), W, W, capi._dptr(lp), capi._dptr(g)) == 0
            assert lib.octo_model_logpost(path.ctx, m, capi._dptr(th), W, W, capi._dptr(lp0), None) == 0
            lib.octo_model_destroy(m); path.close()
            assert np.array_equal(lp, lp0), (case["name"], sb, "value with and without gradient differ")
            res[sb] = (lp, g)
            assert np.all(np.abs(lp - lp_o) <= 1e-12 * np.abs(lp_o)), (case["name"], sb, np.max(np.abs(lp - lp_o) / np.abs(lp_o)))
            sc = np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300)
            assert np.all(np.abs(g - g_o) <= 1e-9 * sc), (case["name"], sb, np.max(np.abs(g - g_o) / sc))
        n0 = min(W, W0)                                                      # the committed θ_t: the 60-digit values
        assert np.all(np.abs(res[None][0][:n0] - np.asarray(case["lp"])[:n0]) <= 1e-12 * np.abs(np.asarray(case["lp"])[:n0]))
        gref = np.asarray(case["grad"])[:, :n0]
        assert np.all(np.abs(res[None][1][:, :n0] - gref) <= 1e-9 * np.abs(gref) + 1e-10 * np.abs(gref).max(axis=1, keepdims=True))
        assert np.all(np.abs(res[None][0] - res[0][0]) <= 1e-13 * np.abs(res[0][0]))


@pytest.mark.gpu
def test_one_task_launch_finishes_inside_k_main(pkg, oracle):
    """Round 6: a mid-size batch on a short table — 1 024 walkers x 50 epochs, a Pigeons round or a guess_starting_position chunk — is ONE
    task, and the k_main blocks finish their own tiles (octo_kernels.h: fin_in_main): no partials, no k_finish launch. Same routines on the same sums:
    BIT-identical to the two-launch route (OCTO_FIN_FUSED=0 at context creation), with and without nuisances, forward-only, with invalid walkers and a
    ragged last tile; against the oracle; and through the whole callback (octo_model_logpost_device: the model's tail runs in the same blocks)."""
    import os
    import gpu_binding as gb
    import synth
    from test_gpu_parity import _cmp_oracle
    rng = np.random.default_rng(91)
    n, W = 50, 1000
    t = 50000.0 + 20.0 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = synth.draw_walkers(rng, W)
    el[1, 5] = 1.2; el[0, 77] = np.nan; el[6, 999] = -1.0
    nuis = np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)])
    nuis[0, 300] = np.inf

    def both(obs, nz, grad):
        out = []
        for fused in ("1", "0"):
            os.environ["OCTO_FIN_FUSED"] = fused
            try:
                out.append(gb.gpu_eval(obs, planets, el, nz, grad=grad, small_batch=0))
            finally:
                os.environ.pop("OCTO_FIN_FUSED")
        return out
    for cor in (None, rng.uniform(-0.5, 0.5, n)):
        obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=cor)]
        for nz in (None, nuis):
            a, b = both(obs, nz, True)
            assert all((x is None and y is None) or np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b)), ("fused finish differs from k_finish", cor is None, nz is None)
            af, _ = both(obs, nz, False)
            assert np.array_equal(af[0], a[0])
            bad = [5, 77, 999] + ([300] if nz is not None else [])
            assert np.isneginf(a[0][bad]).all() and np.all(a[1][:, bad] == 0.0)
            ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el, nz, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=nz is not None))
            _cmp_oracle("fused finish", a[0], a[1], a[2], ll_o, g_o, gn_o)
    # the whole callback
    import torch
    tbl = dict(epoch=t, ra=ra, dec=dec, σ_ra=np.full(n, 5.0), σ_dec=np.full(n, 7.0))
    res = []
    for fused in ("1", "0"):
        os.environ["OCTO_FIN_FUSED"] = fused
        try:
            b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(tbl, name="astrom")],
                           variables=pkg.variables(a=pkg.LogUniform(1, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                                   Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
            model = pkg.LogDensityModel(pkg.System(name="mid", companions=[b], variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1),
                                                                                                         plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
        finally:
            os.environ.pop("OCTO_FIN_FUSED")
        th = model.link(model.sample_priors(np.random.default_rng(6), 1024))
        th[2, 9] = np.nan
        lp, g = model.logpost_device(torch.tensor(th, device="cuda"), grad=True)
        res.append((lp.cpu().numpy(), g.cpu().numpy()))
        model.close()
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and np.array_equal(res[0][1], res[1][1], equal_nan=True)
    assert np.isfinite(res[0][0]).sum() > 900
