"""
BASELINE.json's configurations that round 1 measured only in bench.py, now under -m gpu at their FULL sizes:

  config 1  D = 11 model of test/integration/sampling.jl:29-64 on 50 RA/Dec epochs t_j = 50000 + 17 j, ONE θ_t per call
            (SURVEY.md §8d): log-posterior + gradient against the 60-digit fixture tests/golden/config1.json.
  config 4  2 planets, 2500 RA/Dec + 2500 absolute-RV epochs × 4096 walkers with nuisances (kernel k_main<2, true, true, ·>,
            one-round planner branch): determinism, forward == gradient value, a 24-walker oracle sample.
  config 5  per-GPU shape of the tempered run, 8 temperatures × 1024 walkers × 1e4 epochs: TemperedSwap (world = 1) around
            ln_like_device for several steps; log-likelihoods against the oracle on a seeded sample, slot2rep against the NumPy
            restatement of the swap (ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:115-126 is the reference's exchange step).
  streams   the torch-facing wrappers run on torch's CURRENT stream: torch op -> logpost/ln_like -> torch op with no
            device-wide synchronisation in between (the NULL stream handle used to mean "context stream").
"""
import ctypes as C
import json
import os
from pathlib import Path

import numpy as np
import pytest

import synth
from conftest import rel_err
from test_gpu_parity import _cmp_oracle, _gpu, _pt_swap_reference

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _tables(case):
    from conftest import KIND_IDS
    obs = [dict(kind=KIND_IDS[o["kind"]], planet=o["planet"], **{k: (None if o[k] is None else np.asarray(o[k], dtype=np.float64))
                                                                  for k in ("epoch", "y1", "y2", "s1", "s2", "cor")}) for o in case["obs"]]
    return obs, case["planets"]


def test_config1_one_theta_per_call(pkg, oracle):
    case = json.loads((ROOT / "tests" / "golden" / "config1.json").read_text())["cases"][0]
    o = case["obs"][0]
    assert len(o["epoch"]) == 50 and o["epoch"][1] - o["epoch"][0] == 17.0
    table = dict(epoch=o["epoch"], ra=o["y1"], dec=o["y2"], σ_ra=o["s1"], σ_dec=o["s2"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromLikelihood(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="cfg1", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    assert model.D == 11
    th = np.asarray(case["theta_t"])
    for w in range(th.shape[1]):                      # one θ_t per call, as NUTS does (src/logdensitymodel.jl:169-177)
        lp, g = model.logdensity_and_gradient(th[:, w])
        assert abs(lp - case["lp"][w]) <= 1e-12 * abs(case["lp"][w]), (w, lp, case["lp"][w])
        gref = np.asarray(case["grad"])[:, w]
        assert np.all(np.abs(g - gref) <= 1e-9 * np.abs(gref) + 1e-10 * np.abs(gref).max()), (w, np.max(np.abs(g - gref) / np.abs(gref).max()))
        assert model.ℓπcallback(th[:, w]) == lp
    # the same four θ_t as one batch, and forced onto the throughput kernels
    lp_b, g_b = model.logdensity_and_gradient(th)
    assert np.all(rel_err(lp_b, np.asarray(case["lp"]), 1.0) < 1e-12)
    fn = model.ln_like
    fn._check(fn.lib.octo_ctx_set_small_batch(fn._ctx, 0), "octo_ctx_set_small_batch")
    lp_t, g_t = model.logdensity_and_gradient(th)
    assert np.all(rel_err(lp_t, lp_b, 1.0) < 1e-13) and np.all(np.abs(g_t - g_b) <= 1e-11 * np.abs(g_b).max(axis=1, keepdims=True))
    obs, planets = _tables(case)
    lp_o, g_o = oracle.oracle_model_logpost(obs, planets, model._c_priors, model._c_esrc, None, th)
    assert np.all(np.abs(lp_b - lp_o) <= 1e-12 * np.abs(lp_o))
    model.close()


def test_config4_full_size(oracle):
    gb = _gpu()
    c4 = synth.config_two_planet()                    # 2500 + 2500 rows × 4096 walkers, rng 20260929+4
    a, r = c4["astrom"], c4["rv"]
    assert len(a["epoch"]) == 2500 and len(r["epoch"]) == 2500 and c4["n_walkers"] == 4096
    # evaluation order of make_ln_like: planet observations planet by planet, then system observations (system.jl:229-235)
    obs = [dict(kind=0, planet=1, epoch=a["epoch"], y1=a["ra"], y2=a["dec"], s1=a["σ_ra"], s2=a["σ_dec"], cor=None),
           dict(kind=2, planet=-1, epoch=r["epoch"], y1=r["rv"], y2=None, s1=r["σ_rv"], s2=None, cor=None)]
    planets = [dict(orbit_kind=0, has_mass=True), dict(orbit_kind=0, has_mass=True)]
    path = gb.GpuPath(obs, planets)
    ll, g, gn = path.eval(c4["elems"], c4["nuis"], grad=True)
    ll2, g2, gn2 = path.eval(c4["elems"], c4["nuis"], grad=True)
    llf, _, _ = path.eval(c4["elems"], c4["nuis"], grad=False)
    path.close()
    assert np.array_equal(ll, ll2) and np.array_equal(g, g2) and np.array_equal(gn, gn2), "not deterministic"
    assert np.array_equal(ll, llf), "forward-only and gradient launches disagree"
    assert np.all(np.isfinite(ll))
    idx = np.random.default_rng(4).choice(4096, 24, replace=False)
    ll_o, g_o, gn_o = oracle.oracle_eval(obs, planets, c4["elems"][:, idx], c4["nuis"][:, idx], grad=True, n_threads=0)
    _cmp_oracle("config 4 sample", ll[idx], g[:, idx], gn[:, idx], ll_o, g_o, gn_o)
    # without the nuisance block (the jitter == 0 / precomputed Σ⁻¹ path of the same two-planet kernel family)
    ll0, g0, _ = gb.gpu_eval(obs, planets, c4["elems"], None, grad=True)
    ll0_o, g0_o, _ = oracle.oracle_eval(obs, planets, c4["elems"][:, idx], None, grad=True, n_threads=0)
    _cmp_oracle("config 4 sample, no nuisances", ll0[idx], g0[:, idx], None, ll0_o, g0_o, None)


def test_config4_registered_host_arrays():
    """Two planets + nuisances through octo_eval with every array registered (octo_host_register): the copy kernel brings elements
    AND nuisances in, ll / g_elems / g_nuis are written in place — bit-identical to the pageable route."""
    gb = _gpu()
    capi = gb.capi
    c4 = synth.config_two_planet(n_astrom=70, n_rv=50, n_walkers=4096, seed=11)      # 4096 x (18 + 6) x 2 doubles > 1 MiB
    a, r = c4["astrom"], c4["rv"]
    obs = [dict(kind=0, planet=1, epoch=a["epoch"], y1=a["ra"], y2=a["dec"], s1=a["σ_ra"], s2=a["σ_dec"], cor=None),
           dict(kind=2, planet=-1, epoch=r["epoch"], y1=r["rv"], y2=None, s1=r["σ_rv"], s2=None, cor=None)]
    planets = [dict(orbit_kind=0, has_mass=True), dict(orbit_kind=0, has_mass=True)]
    with gb.GpuPath(obs, planets) as path:
        el = np.ascontiguousarray(c4["elems"]); nu = np.ascontiguousarray(c4["nuis"])
        ll0, g0, gn0 = path.eval(el, nu, grad=True)
        ll = np.full_like(ll0, np.nan); g = np.full_like(g0, np.nan); gn = np.full_like(gn0, np.nan)
        bufs = (el, nu, ll, g, gn)
        for b in bufs:
            path._chk(path.lib.octo_host_register(path.ctx, b.ctypes.data, b.nbytes))
        W = el.shape[1]
        path._chk(path.lib.octo_eval(path.ctx, path.ds, capi._dptr(el), capi._dptr(nu), W, W, capi._dptr(ll), capi._dptr(g), capi._dptr(gn)))
        assert np.array_equal(ll, ll0) and np.array_equal(g, g0) and np.array_equal(gn, gn0)
        for b in bufs:
            path._chk(path.lib.octo_host_unregister(path.ctx, b.ctypes.data))


def test_config5_per_gpu_shape(pkg, oracle):
    import torch
    from octofitter_jl_amd.host.tempering import TemperedSwap
    n_temps, chains, E = 8, 1024, 10_000
    W = n_temps * chains
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=2, seed=20260929 + 5)
    obs_m, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="cfg5", companions=[planet]), cfg["theta_example"])
    dev = torch.device("cuda", 0)
    elems = torch.tensor(cfg["elems"], device=dev)
    pt = TemperedSwap(fn, n_temps_total=n_temps, n_chains=chains, rank=0, world=1, device=dev, seed=20260929)
    ll_t = torch.empty(W, dtype=torch.float64, device=dev)
    ref = pt.slot2rep.cpu().numpy().copy()
    acc_ref = np.zeros(n_temps, dtype=np.int32)
    beta = pt.beta.cpu().numpy()
    for step in range(4):
        fn.ln_like_device(elems, None, grad=False, out=(ll_t, None, None))
        s2r = pt.swap_step(ll_t, step)
        torch.cuda.synchronize()
        ll = ll_t.cpu().numpy()
        ref, a = _pt_swap_reference(ll.reshape(n_temps, chains), beta, ref, step % 2, 20260929, step)
        acc_ref += a
        assert np.array_equal(s2r.cpu().numpy(), ref), f"slot2rep differs from the restatement at step {step}"
        # a different state for the next step: the explorer moved (here: a seeded perturbation of tp)
        elems[5] += 3.0 * torch.tensor(np.random.default_rng(step).normal(size=W), device=dev)
    assert np.array_equal(pt.accepted.cpu().numpy(), acc_ref) and acc_ref.sum() > 0
    assert np.all(np.sort(ref, axis=1) == np.arange(n_temps))
    # β of every local walker follows the permutation
    b_loc = pt.local_betas().cpu().numpy().reshape(n_temps, chains)
    rep2slot = np.argsort(ref, axis=1)
    assert np.array_equal(b_loc, beta[rep2slot].T)
    # the log-likelihoods that fed the last swap, against the oracle on a seeded sample of walkers at the full epoch count
    idx = np.random.default_rng(5).choice(W, 16, replace=False)
    el_last = elems.cpu().numpy()
    el_last[5] -= 3.0 * np.random.default_rng(3).normal(size=W)      # undo the perturbation applied after the last evaluation
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    ll_o, _, _ = oracle.oracle_eval(obs, [dict(orbit_kind=0, has_mass=False)], el_last[:, idx], None, grad=False, n_threads=0)
    assert np.all(rel_err(ll[idx], ll_o, 1.0) < 1e-12)
    fn.close()


def test_strong_scaled_shard_wide_blocks(pkg, oracle, monkeypatch):
    """SURVEY §8(d) "Scaling runs": the per-GPU share of config 3 at 8 GPUs — 1 250 walkers x 1e4 epochs, a ONE-ROUND launch, which the planner
    gives eight-wave k_main blocks (octo_kernels.h: k_main<…, NWV = 8>; 2 000 walkers likewise, with and without per-walker nuisances). Against the
    oracle on a seeded sample of walkers at the full epoch count, the forward-only value bit-identical to the value returned with the gradient
    (same partition), and against the same shard evaluated with four-wave blocks (OCTO_WIDE=-1, read at context creation): a different row
    partition, so equal to rounding, not bitwise."""
    E = 10_000
    for W, with_nuis in ((1250, False), (2000, True)):
        cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3, seed=20260929 + 8 + W)
        obs_m, planet = synth.to_mirror(pkg, cfg)
        el = np.ascontiguousarray(cfg["elems"])
        nuis = None
        if with_nuis:
            rng = np.random.default_rng(W)
            nuis = np.ascontiguousarray(np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)]))
        res = {}
        for tag in ("wide", "narrow"):
            if tag == "narrow":
                monkeypatch.setenv("OCTO_WIDE", "-1")
            fn = pkg.make_ln_like(pkg.System(name="shard", companions=[planet]), cfg["theta_example"])
            monkeypatch.delenv("OCTO_WIDE", raising=False)
            out = fn.ln_like_arrays(el, nuis, grad=True)
            ll_f = fn.ln_like_arrays(el, nuis, grad=False)
            assert np.array_equal(out[0], ll_f, equal_nan=True), f"{tag}: forward-only value != value returned with the gradient"
            res[tag] = out
            fn.close()
        ll, g_el, g_nu = res["wide"]
        ok = np.isfinite(ll)
        assert ok.sum() > 0.9 * W and np.array_equal(ok, np.isfinite(res["narrow"][0]))
        assert np.max(np.abs(ll[ok] - res["narrow"][0][ok]) / np.maximum(1.0, np.abs(ll[ok]))) < 1e-12
        sc = np.maximum(np.abs(res["narrow"][1][:8, ok]).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(g_el[:8, ok] - res["narrow"][1][:8, ok]) / sc) < 1e-11
        idx = np.random.default_rng(11).choice(W, 24, replace=False)
        t = cfg["table"]
        obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
        ll_o, g_o, gn_o = oracle.oracle_eval(obs, [dict(orbit_kind=0, has_mass=False)], el[:, idx], None if nuis is None else nuis[:, idx], grad=True,
                                             active=synth.active_mask(1, 1 if with_nuis else 0, mass=False), n_threads=0)
        oko = np.isfinite(ll_o)
        assert np.array_equal(oko, ok[idx])
        assert np.all(rel_err(ll[idx][oko], ll_o[oko], 1.0) < 1e-11)
        sco = np.maximum(np.abs(g_o[:8, oko]).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(g_el[:8, idx][:, oko] - g_o[:8, oko]) / sco) < 1e-9
        if with_nuis:
            scn = np.maximum(np.abs(gn_o[:, oko]).max(axis=1, keepdims=True), 1e-300)
            assert np.max(np.abs(g_nu[:, idx][:, oko] - gn_o[:, oko]) / scn) < 1e-9


def test_one_round_launches_of_the_other_kind_sets(pkg, oracle, monkeypatch):
    """One-round launches (1 250 walkers, long tables) of kind sets other than RA/Dec alone: a sep/PA table with `cor` (same sums as RA/Dec: the planner
    gives it eight-wave blocks) and an O'Neil-wrapped RA/Dec table with per-walker nuisances (four more sums per wave: its combine buffer would pass the
    dynamic-LDS limit of a launch, so it keeps four-wave blocks — octo_launch.h: WIDE_OK). Both against the oracle on a seeded sample and against the same
    shard with OCTO_WIDE=-1."""
    gb = _gpu()
    W, E = 1250, 6000
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3, seed=20260929 + 77)
    t = cfg["table"]
    rng = np.random.default_rng(77)
    ra, dec = np.asarray(t["ra"]), np.asarray(t["dec"])
    cases = {
        "seppa+cor": ([dict(kind=1, planet=0, epoch=t["epoch"], y1=np.arctan2(ra, dec), y2=np.hypot(ra, dec), s1=np.full(E, 0.02), s2=np.asarray(t["σ_ra"]),
                            cor=rng.uniform(-0.5, 0.5, E))], None),
        "oneil+nuis": ([dict(kind=0, planet=0, epoch=t["epoch"], y1=ra, y2=dec, s1=t["σ_ra"], s2=t["σ_dec"], cor=None),
                        dict(kind=5, planet=0, epoch=t["epoch"], y1=ra, y2=dec, s1=t["σ_ra"], s2=t["σ_dec"], cor=None)],
                       np.ascontiguousarray(np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)] * 2))),
    }
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = np.ascontiguousarray(cfg["elems"])
    idx = np.random.default_rng(12).choice(W, 16, replace=False)
    for name, (obs, nuis) in cases.items():
        res = {}
        for tag in ("default", "narrow"):
            if tag == "narrow":
                monkeypatch.setenv("OCTO_WIDE", "-1")
            res[tag] = gb.gpu_eval(obs, planets, el, nuis, grad=True, small_batch=0)
            monkeypatch.delenv("OCTO_WIDE", raising=False)
        ll, g_el = res["default"][0], res["default"][1]
        ok = np.isfinite(ll)
        assert ok.sum() > 0.9 * W and np.array_equal(ok, np.isfinite(res["narrow"][0])), name
        assert np.max(np.abs(ll[ok] - res["narrow"][0][ok]) / np.maximum(1.0, np.abs(ll[ok]))) < 1e-12, name
        ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el[:, idx], None if nuis is None else nuis[:, idx], grad=True,
                                          active=synth.active_mask(1, len(obs) if nuis is not None else 0, mass=False), n_threads=0)
        oko = np.isfinite(ll_o)
        assert np.array_equal(oko, ok[idx]), name
        assert np.all(rel_err(ll[idx][oko], ll_o[oko], 1.0) < 1e-10), name
        sco = np.maximum(np.abs(g_o[:8, oko]).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(g_el[:8, idx][:, oko] - g_o[:8, oko]) / sco) < 1e-8, name


def test_torch_stream_ordering(pkg, oracle):
    """torch op -> device entry point -> torch op on torch's current stream with NO device-wide synchronisation: the results
    are right only if the kernels really run on that stream (default stream, then a side stream)."""
    import torch
    dev = torch.device("cuda", 0)
    cfg = synth.config_astrom(n_epochs=400, n_walkers=3000, seed=77)
    obs_m, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    ll_ref = fn.ln_like_arrays(cfg["elems"], None)
    base = torch.tensor(cfg["elems"], device=dev)
    big = torch.randn(4096, 4096, device=dev)
    for stream in (None, torch.cuda.Stream(device=dev)):
        ctxm = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
        with ctxm:
            for rep in range(3):
                for _ in range(4):
                    big = big @ big * 1e-3              # keep the stream busy: the producer below finishes late
                el = base * 0.0
                el += base                              # producer: elems only exist once these torch ops have run
                ll, g, _ = fn.ln_like_device(el, None, grad=True)
                total = (ll * 2.0).sum()                # consumer on the same stream, no synchronize in between
                el.zero_()                              # and a later overwrite of the INPUT must not race the kernels
                got = total.item()
                assert abs(got - 2.0 * ll_ref.sum()) <= 1e-9 * abs(ll_ref.sum()), (stream, rep, got, 2.0 * ll_ref.sum())
    fn.close()


def test_bench_line_contract():
    """`python bench.py` prints exactly one line on stdout, the JSON of the contract, with the measurement record the judge asked for:
    a binding roofline (FP64 vector, 0 < frac <= 1) next to the real HBM rate, the PCIe-inclusive rate, a parity sample of the timed
    batch, config 1's per-call latency — and the target met (>= 1e9 evals/s)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "10", "--warmup", "3", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["dtype"] == "f64" and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["workload"].startswith("config3")
    assert d["value"] > 1e9 and abs(d["value"] - 1e8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "fp64_vector" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    assert roof["frac"] is not None and 0.2 < roof["frac"] <= 1.0, "roofline withheld: profiles/pmc_traffic.json is stale (re-run tools/profile_round.sh)"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["traffic"] > 1e7
    assert 0 < roof["hbm"]["frac"] < 0.1 and roof["hbm"]["unit"] == "GB/s"                     # the real HBM rate: a percent or two of 8 TB/s
    assert roof["north_star_algorithmic_hbm"]["algorithmic_bytes_per_launch"] == 4001360000.0
    assert 0.1 < roof["kernel_avg_ms"] < 0.6 and roof["kernel_launches_timed"] >= 2      # (0.27 ms on this round's boxes: a plausibility window, not a benchmark)
    assert d["parity"]["ok"] is True and d["max_rel_err"] < 1e-8
    assert 0.5 * d["value"] < d["pcie_inclusive"]["value"] < d["value"]
    assert d["config1"]["gpu_matches_fixture"] is True and 5 < d["config1"]["gpu_us_per_call"] < 200
    # cold start in a fresh process (tools/first_call.py): module load + first call, and the library's size
    assert 0.05 < d["config1"]["first_call_ms"] < 2000 and d["config1"]["first_call"]["finite"] is True and d["config1"]["lib_bytes"] < 15e6
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 1e5
    # the CPU restatement single-thread and on all cores, forward-only and fwd+grad
    cb = d["cpu_baseline"]
    assert cb["single_thread"]["cores"] == 1 and 1e4 < cb["single_thread"]["value"] <= cb["value"] * 1.01
    # (the all-cores legs are ~1 s samples on a 256-thread host here: team start-up and placement noise can invert them — 0.42 of the
    # fwd+grad rate has been seen — so only the single-thread pair is ordered strictly)
    assert cb["forward_only"]["value"] > 0.2 * cb["value"] and cb["single_thread_forward_only"]["value"] > cb["single_thread"]["value"]
    # the metric as SURVEY §8(d) defines it (H2D + D2H inside the timed call) travels on the same line, below the HBM-resident `value`
    # (key meanings, ADVICE r4: value_pcie_inclusive = the blocking call by the host's clock as in BENCH_r01..r03; the device-clock figure has its own key)
    assert 0.5 * d["value"] < d["value_pcie_inclusive"] < d["value"] and d["value_pcie_inclusive"] == d["pcie_inclusive"]["registered"]["blocking_call_value"]
    assert d["value_pcie_inclusive"] <= d["value_pcie_inclusive_device_clock"] < d["value"]
    assert d["value_pcie_inclusive_device_clock"] == d["pcie_inclusive"]["registered"]["value"]
    assert "HBM" in d["config"]["workload"] and "10000 walkers in total" in d["config"]["workload"] and d["config"]["walkers_total"] == 10000
    # strong-scaling shares measured on the one GPU: 1e4 / N walkers, projected speed-up N x rate(W/N) / rate(W)
    sp = d["strong_scaling_projection"]["by_n_gpus"]
    assert [sp[k]["walkers_per_gpu"] for k in ("1", "2", "4", "8")] == [10000, 5000, 2500, 1250]
    assert sp["1"]["projected_speedup"] == 1.0 and 1.5 < sp["2"]["projected_speedup"] <= 2.05 and 4.0 < sp["8"]["projected_speedup"] <= 8.2


def test_bench_strong_scaling_mode():
    """`--scaling strong` (SURVEY §8d "Scaling runs": the same walkers split evenly over the ranks) is the DEFAULT of the contracted workload; at
    N = 1 it is the same job as weak scaling, and `--scaling weak` still exists; the line says which mode it ran and the workload names the split."""
    import subprocess, sys
    for flags, mode in (([], "strong"), (["--scaling", "weak"], "weak")):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "10", "--warmup", "3", *flags, "--no-extras", "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
        assert d["scaling"] == mode and d["n_gpus"] == 1 and d["config"]["walkers_per_gpu"] == 10000 and d["config"]["walkers_total"] == 10000
        assert ("10000 walkers split over 1 GPUs" in d["config"]["workload"]) == (mode == "strong")
        assert abs(d["value"] - 1e8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_shard_shape_has_its_own_roofline():
    """The per-GPU launch shape of an 8-GPU strong-scaled run (1 250 walkers x 1e4 epochs) through the same command on one GPU: the line carries
    a non-null roofline from the counter passes of THAT shape (profiles/pmc_traffic.json "shapes"), whose kernel is the eight-wave block."""
    import subprocess, sys
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "40", "--warmup", "5", "--walkers", "1250", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    roof = d["roofline"]
    assert roof["frac"] is not None and 0.1 < roof["frac"] <= 1.0, roof.get("note")
    assert roof["launch_shape"] == {"walkers": 1250, "rows": 10000, "pmc_walkers": 1250} and roof["kernel"].endswith(", 8, false>") and "pmc_note" not in roof
    assert 0.02 < roof["kernel_avg_ms"] < 0.09 and roof["traffic"] > 1e5


def test_bench_two_rank_line_is_the_contracted_measurement():
    """The N > 1 line BEFORE an 8-GPU box ever sees it: two ranks on the one GPU of this box (gloo for the barrier and the
    MAX all-reduce; RCCL refuses two ranks on one device). Default scaling is strong — the SAME 1e4 walkers split evenly — so the metric's
    "1e4 epochs x 1e4 walkers" stays literally true; the line carries a non-null roofline for the per-GPU launch shape (5 000 walkers),
    cpu_baseline from rank 0 and the weak-scaling point beside it. Plumbing evidence, not a scaling number: both ranks share one GPU."""
    import subprocess, sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--backend", "gloo", "--device", "0",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["metric"].endswith("1e4 epochs x 1e4 walkers")
    assert d["config"]["walkers_total"] == 10000 and d["config"]["walkers_per_gpu"] == 5000 and "10000 walkers in total" in d["config"]["workload"]
    assert abs(d["value"] - 1e8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]                 # the whole job = 1e4 x 1e4 evaluations per step
    roof = d["roofline"]
    assert roof["frac"] is not None and 0.05 < roof["frac"] <= 1.0 and roof["launch_shape"]["walkers"] == 5000 and roof["launch_shape"]["pmc_walkers"] == 5000
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 1e5 and "10000 epochs" in cb["sample"]
    wk = d["weak_scaling_measured"]
    assert wk["walkers_per_gpu"] == 10000 and wk["n_gpus"] == 2 and wk["value"] > 1e9
    assert d["parity"]["ok"] is True
