"""
Walker tiles made homogeneous for the warm-started loop (octofitter.jl_amd/csrc/octo_tile.h: k_tile_sort) and the context
options around it (include/octofitter_hip.h: OCTO_OPT_*).

The sort changes WHICH 64 walkers share a wave, nothing else: inputs and outputs keep the caller's order, so a sorted evaluation must return every
walker's numbers at that walker's column and agree with the unsorted evaluation to rounding (the warm loop's wave-uniform fallback makes the last bits
a function of the wave's composition) and with the oracle at the usual bars. The reference evaluates one θ at a time
(src/likelihoods/system.jl:206-241): OCTO_OPT_BATCH_INVARIANT gives that property back bit for bit.
"""
import numpy as np
import pytest

import synth
from conftest import rel_err
from test_gpu_parity import _cmp_oracle, _gpu

pytestmark = pytest.mark.gpu


def _opts(capi, sort, min_w=64, **kw):
    o = {capi.OPT_TILE_SORT: sort, capi.OPT_TILE_MIN_WALKERS: min_w}
    o.update(kw)
    return o


def _close(name, a, b, ll_tol=1e-12, g_tol=1e-9):
    ok = np.isfinite(b[0])
    assert np.array_equal(np.isfinite(a[0]), ok), name
    assert np.all(rel_err(a[0][ok], b[0][ok], 1.0) < ll_tol), (name, "ll")
    for ga, gb_ in ((a[1], b[1]), (a[2], b[2])):
        if ga is None:
            continue
        sc = np.maximum(np.abs(gb_[:, ok]).max(axis=1, keepdims=True), 1e-300)
        assert np.all(np.abs(ga[:, ok] - gb_[:, ok]) / sc < g_tol), (name, "gradient")
        assert np.all(ga[:, ~ok] == 0.0), (name, "an invalid walker's gradient is zero")


def _wide_walkers(rng, W, a_lo=0.25, a_hi=80.0, mass=False):
    el = synth.draw_walkers(rng, W, a_lo, a_hi, with_mass=mass)
    el[1, :W // 8] = 1.0 - 10.0 ** rng.uniform(-3, -1, W // 8)        # some nearly parabolic ones
    el[5, : W // 4] = 50000.0 + rng.uniform(5.0, 400.0, W // 4)      # periastron inside the table
    bad = rng.choice(W, 7, replace=False)
    el[1, bad[0]] = 1.3; el[0, bad[1]] = np.nan; el[6, bad[2]] = -1.0; el[1, bad[3]] = -0.1; el[0, bad[4]] = 0.0; el[5, bad[5]] = np.inf; el[7, bad[6]] = np.nan
    return el, bad


def test_sorted_tiles_return_every_walker_at_its_own_column(pkg, oracle):
    """RA/Dec at a daily cadence, 5 000 walkers (full segments of TILE_SEG walkers and a ragged last one, the last tile of 8) with a ~ LogU(0.25, 80) AU —
    many lanes that veto the step bound — near-parabolic orbits, periastra inside the table and seven invalid walkers: sort forced on against sort
    off against the oracle; forward-only == the value returned with a gradient; run-to-run determinism."""
    gb = _gpu()
    capi = pkg.capi
    rng = np.random.default_rng(71)
    n, W = 600, 5000
    t = 50000.0 + 1.0 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el, bad = _wide_walkers(rng, W)
    with gb.GpuPath(obs, planets, small_batch=0, options=_opts(capi, 1)) as g:
        srt = g.eval(el, None, grad=True)
        srt_f = g.eval(el, None, grad=False)
        srt2 = g.eval(el, None, grad=True)
        n_sorted, _, _, _ = g.tile_state()
    assert n_sorted == 3, n_sorted
    uns = gb.gpu_eval(obs, planets, el, None, grad=True, small_batch=0, options=_opts(capi, 0))
    assert np.isneginf(srt[0][bad]).all() and np.isfinite(srt[0]).sum() == W - bad.size
    _close("sorted vs as drawn", srt, uns)
    assert not (np.array_equal(srt[0], uns[0]) and np.array_equal(srt[1], uns[1])), "the sort did not change a single wave: it did not run"
    assert np.array_equal(srt_f[0], srt[0]), "forward-only and gradient launches of the sorted batch disagree"
    assert np.array_equal(srt2[0], srt[0]) and np.array_equal(srt2[1], srt[1]), "run-to-run determinism"
    idx = np.concatenate([np.arange(0, W, 23), bad])
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el[:, idx], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("sorted vs oracle", srt[0][idx], srt[1][:, idx], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)


def test_sorted_tiles_with_nuisances_rv_and_the_model_tail(pkg, oracle):
    """The other routes the permutation threads through: per-walker nuisances (gathered by k_main, their adjoints scattered by k_finish), an RV table
    with offset / jitter / trend next to the astrometry, and the whole callback (octo_model_logpost_device: k_model_fwd writes the elements in the
    caller's order, k_finish's model tail reads the Jacobians at the WALKER, not at the tile position)."""
    gb = _gpu()
    capi = pkg.capi
    rng = np.random.default_rng(72)
    n, W = 400, 4200
    t = synth.gappy_epochs(n, per_night=2, nights_per_season=60, rng=rng)
    ra, dec = synth.truth_radec(t)
    rv = synth.truth_rv_star(t, dict(synth.TRUTH), 8.0)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 6.0), cor=rng.uniform(-0.5, 0.5, n)),
           dict(kind=2, planet=-1, epoch=t + 0.3, y1=rv + rng.normal(0, 3, n), y2=None, s1=np.full(n, 3.0), s2=None, cor=None, extra=(t - 50200.0) / 100.0)]
    planets = [dict(orbit_kind=0, has_mass=True)]
    el, bad = _wide_walkers(rng, W, 0.3, 60.0, mass=True)
    el[8] = rng.uniform(1.0, 15.0, W)
    nuis = np.zeros((6, W))
    nuis[0] = rng.uniform(0, 3, W); nuis[1] = rng.normal(1, 0.01, W); nuis[2] = rng.normal(0, 0.01, W)
    nuis[3] = rng.normal(0, 5, W); nuis[4] = rng.uniform(0.1, 4, W); nuis[5] = rng.normal(0, 2, W)
    nuis[4, 11] = np.nan
    for nz in (nuis, None):
        srt = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0, options=_opts(capi, 1))
        uns = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0, options=_opts(capi, 0))
        _close("kinds", srt, uns)
        assert not (np.array_equal(srt[0], uns[0]) and np.array_equal(srt[1], uns[1]))
        idx = np.arange(0, W, 37)
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el[:, idx], None if nz is None else nz[:, idx], grad=True)
        _cmp_oracle("kinds sorted vs oracle", srt[0][idx], srt[1][:, idx], None if nz is None else srt[2][:, idx], ll_o, g_o, gn_o, ll_rtol=1e-9, g_rtol=1e-8)
    # the whole callback on the device
    import torch
    tbl = dict(epoch=t, ra=obs[0]["y1"], dec=obs[0]["y2"], σ_ra=obs[0]["s1"], σ_dec=obs[0]["s2"])
    res = {}
    for mode in (1, 0):
        b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(tbl, name="astrom")],
                       variables=pkg.variables(a=pkg.LogUniform(0.3, 60), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                               Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
        model = pkg.LogDensityModel(pkg.System(name="tile", companions=[b], variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1),
                                                                                                      plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
        lib = model.ln_like.lib
        assert lib.octo_ctx_set_option(model.ln_like._ctx, capi.OPT_TILE_SORT, mode) == 0 and lib.octo_ctx_set_option(model.ln_like._ctx, capi.OPT_TILE_MIN_WALKERS, 64) == 0
        th = model.link(model.sample_priors(np.random.default_rng(5), W))
        th[3, 17] = np.nan
        lp, g = model.logpost_device(torch.tensor(th, device="cuda"), grad=True)
        res[mode] = (lp.cpu().numpy(), g.cpu().numpy())
        model.close()
    ok = np.isfinite(res[0][0])
    assert np.array_equal(np.isfinite(res[1][0]), ok) and ok.sum() > 0.9 * W
    assert np.all(rel_err(res[1][0][ok], res[0][0][ok], 1.0) < 1e-12)
    sc = np.maximum(np.abs(res[0][1][:, ok]).max(axis=1, keepdims=True), 1e-300)
    assert np.all(np.abs(res[1][1][:, ok] - res[0][1][:, ok]) / sc < 1e-9)
    assert not np.array_equal(res[1][0], res[0][0])


def test_auto_mode_sorts_only_where_it_pays(pkg):
    """OCTO_OPT_TILE_SORT = 2 (the default): the first eligible evaluation of a (dataset, batch size) prices the sort and reads the price at once
    (one host wait per shape), so the decision holds from that very evaluation: the same inputs give the same bits on every call. Config 3's prior (a ~ LogU(1, 100) AU at a daily cadence: no lane vetoes, 8 % of the wave-rows fall back as drawn, 4 % sorted) on a table of
    2 000 rows: the saving is below the cost of the launch -> off. The same table with a ~ LogU(0.3, 100) AU (14 % of the lanes veto: every wave cold
    as drawn) -> on. Deterministic: the same calls make the same decisions."""
    gb = _gpu()
    capi = pkg.capi
    for a_lo, expect_on in ((1.0, False), (0.3, True)):
        cfg = synth.config_wide_prior(n_epochs=2000, n_walkers=8192, a_lo=a_lo)
        t = cfg["table"]
        obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
        states = []
        for rep in range(2):
            with gb.GpuPath(obs, [dict(orbit_kind=0, has_mass=False)], small_batch=0) as g:
                outs = [g.eval(cfg["elems"], None, grad=True) for _ in range(4)]
                states.append(g.tile_state())
                # every evaluation of the same inputs returns the same bits, the first one included
                for o in outs[1:]:
                    assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
        n_sorted, probes, on, saving = states[0]
        assert states[0] == states[1], "the probe's decision must not depend on the run"
        assert probes == 1 and on == expect_on and n_sorted == (4 if expect_on else 0), (a_lo, states[0])
        assert (saving > 1.3 * 6.5) == expect_on, (a_lo, saving)


def test_batch_invariant_option_is_bitwise(pkg, oracle):
    """OCTO_OPT_BATCH_INVARIANT = 1: a walker's log-likelihood and gradient are the same bits whatever batch it is
    evaluated in — the full batch, a shard of it, a shuffled batch, one θ alone — on a DENSE table, where the default (warm loop, wave-uniform fallback,
    a row partition that follows the batch size) agrees only to rounding. And the default does differ there: the option is not a no-op."""
    gb = _gpu()
    capi = pkg.capi
    rng = np.random.default_rng(73)
    n, W = 900, 2600
    t = 50000.0 + 1.0 * np.arange(n)
    ra, dec = synth.truth_radec(t)
    obs = [dict(kind=0, planet=0, epoch=t, y1=ra + rng.normal(0, 5, n), y2=dec + rng.normal(0, 5, n), s1=np.full(n, 5.0), s2=np.full(n, 7.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = synth.draw_walkers(rng, W, 1.0, 60.0)
    perm = rng.permutation(W)
    res = {}
    for inv in (1, 0):
        with gb.GpuPath(obs, planets, options={capi.OPT_BATCH_INVARIANT: inv}) as g:
            full = g.eval(el, None, grad=True)
            shard = g.eval(el[:, 700:1400], None, grad=True)
            shuf = g.eval(el[:, perm], None, grad=True)
            one = g.eval(el[:, 1234:1235], None, grad=True)
            fwd = g.eval(el, None, grad=False)
        res[inv] = (full, shard, shuf, one)
        same = (np.array_equal(full[0][700:1400], shard[0]) and np.array_equal(full[1][:, 700:1400], shard[1]) and
                np.array_equal(full[0][perm], shuf[0]) and np.array_equal(full[1][:, perm], shuf[1]) and
                full[0][1234] == one[0][0] and np.array_equal(full[1][:, 1234], one[1][:, 0]))
        assert same == bool(inv), ("batch-invariant" if inv else "default", same)
        assert np.array_equal(fwd[0], full[0])
    # both modes against each other (rounding) and the invariant one against the oracle
    ok = np.isfinite(res[1][0][0])
    assert np.all(rel_err(res[0][0][0][ok], res[1][0][0][ok], 1.0) < 1e-12)
    idx = np.arange(0, W, 41)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el[:, idx], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False))
    _cmp_oracle("invariant vs oracle", res[1][0][0][idx], res[1][0][1][:, idx], None, ll_o, g_o, None, ll_rtol=1e-10, g_rtol=1e-8)
    # option plumbing: range checks and read-back
    with gb.GpuPath(obs, planets) as g:
        v = __import__("ctypes").c_int64(-1)
        assert g.lib.octo_ctx_get_option(g.ctx, capi.OPT_TILE_SORT, __import__("ctypes").byref(v)) == 0 and v.value == 2
        assert g.lib.octo_ctx_set_option(g.ctx, capi.OPT_TILE_SORT, 3) == capi.OCTO_EINVAL
        assert g.lib.octo_ctx_set_option(g.ctx, 99, 0) == capi.OCTO_EINVAL
        assert g.lib.octo_ctx_set_option(g.ctx, capi.OPT_TILE_MIN_WALKERS, 4096) == 0
        assert g.lib.octo_ctx_get_option(g.ctx, capi.OPT_TILE_MIN_WALKERS, __import__("ctypes").byref(v)) == 0 and v.value == 4096


def test_two_planets_last_planet_always_warm_and_sorted_tiles(pkg, oracle):
    """Round 6: the two-planet kernels carry a loop in which the LAST planet takes the unconditional warm step (no ballot, no branch:
    both planets' solves in one basic block) — taken by waves whose 64 lanes are provably safe on every row, 1/(1 − e) < thr — and the walker tiles are
    sorted by that planet's severity so that most tiles qualify. Config 4 in small (RA/Dec on the outer planet + absolute RV, nuisances, inner-barycentre
    term): (a) outer eccentricities below 0.7 — every wave qualifies as drawn: against the cold loop (OCTO_OPT_WARM_START = 0) and the oracle;
    (b) as drawn (e up to 0.95: no wave qualifies unsorted) with the sort forced on: equal to the unsorted, cold evaluation to rounding, every walker at
    its own column, invalid walkers included; forward-only == the value returned with a gradient."""
    gb = _gpu()
    capi = pkg.capi
    c4 = synth.config_two_planet(n_astrom=300, n_rv=260, n_walkers=2300, seed=77)
    obs = [dict(kind=0, planet=1, epoch=c4["astrom"]["epoch"], y1=c4["astrom"]["ra"], y2=c4["astrom"]["dec"], s1=c4["astrom"]["σ_ra"], s2=c4["astrom"]["σ_dec"], cor=None),
           dict(kind=2, planet=-1, epoch=c4["rv"]["epoch"], y1=c4["rv"]["rv"], y2=None, s1=c4["rv"]["σ_rv"], s2=None, cor=None)]
    planets = [dict(orbit_kind=0, has_mass=True)] * 2
    el, nuis = c4["elems"].copy(), c4["nuis"]
    W = el.shape[1]
    idx = np.arange(0, W, 31)
    for nz in (nuis, None):
        # (a) every wave safe as drawn
        el_a = el.copy(); el_a[9 + 1] *= 0.7 / 0.95
        warm = gb.gpu_eval(obs, planets, el_a, nz, grad=True, small_batch=0, options=_opts(capi, 0))
        cold = gb.gpu_eval(obs, planets, el_a, nz, grad=True, small_batch=0, options={capi.OPT_TILE_SORT: 0, capi.OPT_WARM_START: 0})
        assert not (np.array_equal(warm[0], cold[0]) and np.array_equal(warm[1], cold[1])), "the last-planet warm loop did not run"
        _close("last planet warm vs cold", warm, cold)
        fwd = gb.gpu_eval(obs, planets, el_a, nz, grad=False, small_batch=0, options=_opts(capi, 0))
        assert np.array_equal(fwd[0], warm[0])
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el_a[:, idx], None if nz is None else nz[:, idx], grad=True)
        _cmp_oracle("last planet warm vs oracle", warm[0][idx], warm[1][:, idx], None if nz is None else warm[2][:, idx], ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)
        # (b) as drawn + invalid walkers: (b0) unsorted — the per-row test sends the rows after a lane's periastron passage to the cold body —, (b1) sort forced on
        el_b = el.copy(); el_b[1, 7] = 1.1; el_b[9, 500] = np.nan; el_b[9 + 6, 2299] = -2.0
        drawn = gb.gpu_eval(obs, planets, el_b, nz, grad=True, small_batch=0, options=_opts(capi, 0))
        srt = gb.gpu_eval(obs, planets, el_b, nz, grad=True, small_batch=0, options=_opts(capi, 1))
        uns = gb.gpu_eval(obs, planets, el_b, nz, grad=True, small_batch=0, options={capi.OPT_TILE_SORT: 0, capi.OPT_WARM_START: 0})
        assert np.isneginf(srt[0][[7, 500, 2299]]).all() and np.isneginf(drawn[0][[7, 500, 2299]]).all()
        _close("two planets sorted vs as drawn", srt, uns)
        _close("two planets as drawn, warm vs cold", drawn, uns)
        assert not (np.array_equal(srt[0], uns[0]) and np.array_equal(srt[1], uns[1]))
        assert not (np.array_equal(drawn[0], uns[0]) and np.array_equal(drawn[1], uns[1])), "as drawn, no row of the last planet started warm"
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el_b[:, idx], None if nz is None else nz[:, idx], grad=True)
        _cmp_oracle("two planets sorted vs oracle", srt[0][idx], srt[1][:, idx], None if nz is None else srt[2][:, idx], ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)
        _cmp_oracle("two planets as drawn vs oracle", drawn[0][idx], drawn[1][:, idx], None if nz is None else drawn[2][:, idx], ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)


@pytest.mark.gpu
def test_three_planets_last_planet_warm(pkg, oracle):
    """Round 6, late: the three-planet kernels of the kind sets without sep/PA rows carry the last planet's warm start too (octo_kernels.h: main_warm_last) — RA/Dec
    on the outer planet + absolute RV on a dense cadence, outer eccentricities to 0.95 (rejected rows near periastron re-solve it cold), with and without per-walker
    nuisances, invalid walkers: against the cold loops (OCTO_OPT_WARM_START = 0; not the same bits: the loop ran) and the oracle; forward-only == the value returned
    with a gradient. A sep/PA table (a kind set that keeps the cold loop for three planets) gives the cold loop's bits."""
    gb = _gpu()
    capi = pkg.capi
    rng = np.random.default_rng(303)
    W, n = 700, 260
    t = 50000.0 + 2.0 * np.arange(n)
    e1 = synth.draw_walkers(rng, W, 1.0, 5.0, with_mass=True); em = synth.draw_walkers(rng, W, 6.0, 12.0, with_mass=True); e2 = synth.draw_walkers(rng, W, 15.0, 40.0, with_mass=True)
    for x in (em, e2): x[6] = e1[6]; x[7] = e1[7]
    el = np.concatenate([e1, em, e2])
    el[1, 7] = 1.1; el[9, 500] = np.nan; el[18 + 6, 699] = -2.0
    planets = [dict(orbit_kind=0, has_mass=True)] * 3
    radec = dict(kind=0, planet=2, epoch=t, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=None)
    rv = dict(kind=2, planet=-1, epoch=t + 0.3, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None)
    obs = [radec, rv]
    nuis = np.zeros((6, W))
    nuis[0] = rng.uniform(0, 4, W); nuis[1] = rng.normal(1, 0.01, W); nuis[2] = rng.normal(0, 0.02, W)
    nuis[3] = rng.normal(0, 10, W); nuis[4] = np.exp(rng.uniform(np.log(0.1), np.log(10), W))
    idx = np.arange(0, W, 23)
    COLD = {capi.OPT_TILE_SORT: 0, capi.OPT_WARM_START: 0}
    for nz in (nuis, None):
        warm = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0, options=_opts(capi, 0))
        cold = gb.gpu_eval(obs, planets, el, nz, grad=True, small_batch=0, options=COLD)
        assert np.isneginf(warm[0][[7, 500, 699]]).all()
        assert not (np.array_equal(warm[0], cold[0]) and np.array_equal(warm[1], cold[1], equal_nan=True)), "no row of the last planet started warm"
        _close("three planets, last planet warm vs cold", warm, cold)
        fwd = gb.gpu_eval(obs, planets, el, nz, grad=False, small_batch=0, options=_opts(capi, 0))
        assert np.array_equal(fwd[0], warm[0])
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, el[:, idx], None if nz is None else nz[:, idx], grad=True)
        _cmp_oracle("three planets vs oracle", warm[0][idx], warm[1][:, idx], None if nz is None else warm[2][:, idx], ll_o, g_o, gn_o, ll_rtol=1e-10, g_rtol=1e-8)
    seppa = dict(kind=1, planet=2, epoch=t, y1=np.arctan2(radec["y1"], radec["y2"]), y2=np.hypot(radec["y1"], radec["y2"]), s1=np.full(n, 0.03), s2=rng.uniform(3, 12, n), cor=None)
    a = gb.gpu_eval([seppa, rv], planets, el, None, grad=True, small_batch=0, options=_opts(capi, 0))
    b = gb.gpu_eval([seppa, rv], planets, el, None, grad=True, small_batch=0, options=COLD)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1], equal_nan=True)
