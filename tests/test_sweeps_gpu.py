"""
The randomised sweeps of round 1 (tests/stress_*.py, tests/big_shapes.py — they found a real bug once, a6abd3d) as
parametrised -m gpu tests with fixed seeds: ~50 random systems over every observation kind and orbit basis, random
standard-parameterisation models on top, the OFTI solver, e -> 1, ~1e3-orbit phases, and one large shape. Every case is the HIP
path through the C ABI against the oracle (reference-order C restatement, or the 60-digit mpmath oracle where the
reference's own arithmetic loses digits). The scripts still run stand-alone for longer sweeps.
"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[None, 0], ids=["auto", "throughput"])
def _kernel_family(request):
    """Every case of this module twice: with the library's own choice of kernel family (k_small for W·P <= 512) and forced onto
    the throughput kernels (lane = walker), so that mid-size batches keep checking both against the oracle."""
    import gpu_binding
    gpu_binding.DEFAULT_SMALL_BATCH = request.param
    yield
    gpu_binding.DEFAULT_SMALL_BATCH = None


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_systems_vs_oracle(oracle, seed):
    import stress_parity as sp
    rng = np.random.default_rng(seed)
    fails = []
    for k in range(10):
        sysm = sp.draw_system(rng)
        good, e_ll, e_g, loose = sp.check_system(sysm)
        if not good:
            fails.append((k, sp.describe(sysm), e_ll, e_g))
    assert not fails, fails


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_models_vs_oracle(pkg, oracle, seed):
    import stress_model as sm
    rng = np.random.default_rng(seed)
    lib = pkg.capi.load_library()
    fails = []
    for k in range(8):
        r = sm.check_model(rng, lib)
        if r is not None and not r[0]:
            fails.append((k,) + r[1:])
    assert not fails, fails


@pytest.mark.parametrize("value", [float("nan"), 1e300])
def test_results_do_not_depend_on_stale_lds(pkg, oracle, value):
    """LDS keeps what the last block left there. With every word of every CU's LDS set to NaN (and to 1e300) ahead of EACH evaluation
    — `octo_debug_poison_lds`, a test hook of the library — random systems and random models still match the oracle, value with and
    without the gradient alike: no kernel reads an LDS word it has not written (the round-3 bug of the fused model launch did, and
    passed every fixture on stale zeros). Both kernel families (the module's fixture)."""
    import gpu_binding as gb
    import stress_parity as sp
    import stress_model as sm
    lib = pkg.capi.load_library()
    gb.POISON_LDS = value
    try:
        fails = []
        rng = np.random.default_rng(31)
        for k in range(12):
            sysm = sp.draw_system(rng)
            good, e_ll, e_g, loose = sp.check_system(sysm)
            if not good:
                fails.append(("system", k, sp.describe(sysm), e_ll, e_g))
        rng = np.random.default_rng(32)
        for k in range(12):
            r = sm.check_model(rng, lib)
            if r is not None and not r[0]:
                fails.append(("model", k) + r[1:])
        rng = np.random.default_rng(303)      # the two models of the long sweep that hit the uninitialised read
        for k in range(54):
            r = sm.check_model(rng, lib, evaluate=k in (32, 53))
            if r is not None and not r[0]:
                fails.append(("model 303", k) + r[1:])
    finally:
        gb.POISON_LDS = None
    assert not fails, fails


def test_model_cases_found_by_the_long_sweep(pkg, oracle):
    """Cases 32 and 53 of `stress_model.py 400 303` (round 3): models on an O'Neil-wrapped table WITHOUT rows and without nuisance
    variables. The fused model launch is then compiled without nuisances, and its finish read the observation's default nuisance
    values from an LDS array that only a nuisance launch fills — log-posterior −Inf with the gradient, right without. Both kernel
    families, value with and without the gradient, against the oracle."""
    import stress_model as sm
    rng = np.random.default_rng(303)
    lib = pkg.capi.load_library()
    seen = []
    for k in range(54):
        r = sm.check_model(rng, lib, evaluate=k in (32, 53))
        if r is not None:
            seen.append(k)
            assert r[0], (k,) + r[1:]
    assert seen == [32, 53]


@pytest.mark.parametrize("seed", [21, 22])
def test_random_ofti_vs_oracle(oracle, seed):
    import stress_ofti as so
    rng = np.random.default_rng(seed)
    fails = []
    for k in range(12):
        good, e_lm, e_ab, desc = so.check_case(rng)
        if not good:
            fails.append((k, desc, e_lm, e_ab))
    assert not fails, fails


@pytest.mark.parametrize("W,small_batch", [(40, None), (24, None), (24, 0)])
def test_high_eccentricity_vs_60_digits(W, small_batch):
    """e = 1 − 10^U(−6, −1): ll within 1e-14, gradients within 1e-14 of Σ|per-row terms| (throughput kernels at W = 40 and forced
    at W = 24, the small-batch kernel at W = 24) — the kernels' eccentric-anomaly / Thiele-Innes form has no 1/(1−e) conditioning."""
    import stress_high_e as sh
    e_ll, e_g, _ = sh.run(W=W, seed=1, small_batch=small_batch)
    assert e_ll < 1e-14 and e_g < 1e-14, (e_ll, e_g)


@pytest.mark.parametrize("small_batch", [None, 0])
def test_many_orbits_vs_60_digits(small_batch):
    """|t − tp|/P ~ 1e3: the phase's rounding is the floor for the kernel and for the reference-order restatement alike."""
    import stress_many_orbits as smo
    w = smo.run(W=16, seed=1, small_batch=small_batch)
    assert w[0] < 2e-11 and w[1] < 5e-11, w
    assert w[0] < 20 * max(w[2], 1e-13) and w[1] < 20 * max(w[3], 1e-13), w      # no worse than the reference's own order of operations


def test_big_shape_one_million_walkers(oracle):
    """1e6 walkers × 1e3 rows through the host-buffer entry point: finite, deterministic, forward == gradient value, and a
    sample of walkers against the oracle."""
    import gpu_binding as gb
    E, W = 1000, 1_000_000
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3, seed=123)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    path = gb.GpuPath(obs, planets)
    ll, g, _ = path.eval(cfg["elems"], None, grad=True)
    ll2, g2, _ = path.eval(cfg["elems"], None, grad=True)
    llf, _, _ = path.eval(cfg["elems"], None, grad=False)
    path.close()
    assert np.all(np.isfinite(ll)) and np.array_equal(ll, ll2) and np.array_equal(g, g2) and np.array_equal(ll, llf)
    idx = np.random.default_rng(0).choice(W, 16, replace=False)
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, cfg["elems"][:, idx], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False), n_threads=0)
    assert np.max(np.abs(ll[idx] - ll_o) / np.abs(ll_o)) < 1e-12
    assert np.max(np.abs(g[:, idx] - g_o) / np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300)) < 1e-9
