"""
OFTI marginal likelihood: `ofti_linear_solve`, src/parameterizations.jl:318-405.
CPU: the reference-order restatement in the oracle against the 60-digit fixture. GPU: the HIP path against both.
"""
import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def ofti_golden():
    return json.loads((ROOT / "tests" / "golden" / "ofti.json").read_text())["cases"]


def _cols(case):
    return [np.asarray(case[k]) for k in ("epochs", "ra", "dec", "s_ra", "s_dec")] + [None if case["cor"] is None else np.asarray(case["cor"])]


def _check(abfg, lm, case, rtol_lm, rtol_abfg):
    ref_lm = np.asarray(case["logml"]); ref = np.asarray(case["abfg"])
    assert np.all(np.abs(lm - ref_lm) <= rtol_lm * np.maximum(1.0, np.abs(ref_lm))), np.max(np.abs(lm - ref_lm) / np.maximum(1, np.abs(ref_lm)))
    scale = np.linalg.norm(ref, axis=0, keepdims=True)
    assert np.all(np.abs(abfg - ref) <= rtol_abfg * scale), np.max(np.abs(abfg - ref) / scale)


def test_oracle_ofti_vs_golden(oracle, ofti_golden):
    # the normal equations of an eccentric, poorly-sampled orbit are ill-conditioned (cond ~1e5 with σ_ABFG = 1000):
    # dense LU in Float64, as the reference does it, is good to ~1e-10 there
    for case in ofti_golden:
        abfg, lm = oracle.oracle_ofti(*_cols(case), case["sigma_abfg"], np.asarray(case["nl"]))
        _check(abfg, lm, case, 1e-9, 1e-8)


def test_oracle_ofti_invalid(oracle, ofti_golden):
    case = ofti_golden[0]
    nl = np.asarray(case["nl"])[:, :4].copy()
    nl[0, 0] = 1.0; nl[1, 1] = -2.0; nl[3, 2] = np.nan
    abfg, lm = oracle.oracle_ofti(*_cols(case), case["sigma_abfg"], nl)
    assert np.all(np.isneginf(lm[:3])) and np.isfinite(lm[3]) and np.all(np.isnan(abfg[:, :3]))


@pytest.mark.gpu
def test_gpu_ofti_vs_golden_and_oracle(pkg, oracle, ofti_golden):
    for case in ofti_golden:
        solver = pkg.OftiLinearSolver(*_cols(case), case["sigma_abfg"])
        nl = np.asarray(case["nl"])
        res = solver(*nl)
        abfg = np.stack([res[k] for k in "ABFG"]); lm = res["log_marginal_likelihood"]
        _check(abfg, lm, case, 1e-10, 1e-9)             # vs 60-digit truth (Cholesky on the device: tighter than the LU restatement)
        abfg_o, lm_o = oracle.oracle_ofti(*_cols(case), case["sigma_abfg"], nl)
        assert np.all(np.abs(lm - lm_o) <= 1e-9 * np.maximum(1, np.abs(lm_o)))
        solver.close()


@pytest.mark.gpu
def test_gpu_ofti_large_batch_and_edges(pkg, oracle, ofti_golden):
    """The example's rejection-sampling batch (examples/ofti_rejection_sampling.jl:108 draws from these priors):
    a seeded 20k-draw batch, sampled walkers checked against the oracle; invalid draws give -Inf; reference signature."""
    case = ofti_golden[1]
    rng = np.random.default_rng(8)
    W = 20_000
    M = np.abs(rng.normal(1.2, 0.1, W)) + 0.1; plx = rng.normal(50.0, 0.5, W)
    e = rng.uniform(0, 0.99, W); a = np.exp(rng.uniform(0, np.log(100.0), W))
    tp = 50000.0 + rng.uniform(0, 1, W) * np.sqrt(a ** 3 / M) * 365.2568983840419
    e[5] = 1.2; a[6] = 0.0; M[7] = -1.0; tp[8] = np.inf
    solver = pkg.OftiLinearSolver(*_cols(case), case["sigma_abfg"])
    res = solver(e, a, tp, M, plx)
    lm = res["log_marginal_likelihood"]
    assert np.all(np.isneginf(lm[5:9])) and np.all(np.isnan(res["A"][5:9]))
    good = np.setdiff1d(np.arange(W), [5, 6, 7, 8])
    assert np.all(np.isfinite(lm[good]))
    idx = rng.choice(good, 200, replace=False)
    abfg_o, lm_o = oracle.oracle_ofti(*_cols(case), case["sigma_abfg"], np.stack([e, a, tp, M, plx])[:, idx])
    assert np.all(np.abs(lm[idx] - lm_o) <= 1e-8 * np.maximum(1, np.abs(lm_o))), np.max(np.abs(lm[idx] - lm_o) / np.maximum(1, np.abs(lm_o)))
    res2 = solver(e, a, tp, M, plx)
    assert np.array_equal(res2["log_marginal_likelihood"], lm, equal_nan=True)      # deterministic
    solver.close()
    one = pkg.ofti_linear_solve(*_cols(case), case["sigma_abfg"], e[0], a[0], tp[0], M[0], plx[0])
    assert one["log_marginal_likelihood"].shape == (1,) and one["log_marginal_likelihood"][0] == lm[0]


@pytest.mark.gpu
def test_gpu_ofti_invalid_walkers_do_not_disturb_neighbours(pkg, ofti_golden):
    """k_ofti_main reads the starter's sin/cos from the LDS table with an UNCLAMPED index (octo_device.h: sincos_table): a walker
    with e >= 1 or non-finite elements may index anywhere. It must come back as -Inf / NaN and leave the other lanes of its
    wave bit-identical to a clean run (the same guarantee test_invalid_walkers checks for k_main)."""
    case = ofti_golden[1]
    rng = np.random.default_rng(12)
    W = 200
    M = np.abs(rng.normal(1.2, 0.1, W)) + 0.1; plx = rng.normal(50.0, 0.5, W)
    e = rng.uniform(0, 0.95, W); a = np.exp(rng.uniform(0, np.log(60.0), W))
    tp = 50000.0 + rng.uniform(0, 1, W) * np.sqrt(a ** 3 / M) * 365.2568983840419
    solver = pkg.OftiLinearSolver(*_cols(case), case["sigma_abfg"])
    clean = solver(e, a, tp, M, plx)
    eb, ab, tb, Mb = e.copy(), a.copy(), tp.copy(), M.copy()
    bad = [3, 64, 65, 127, 130, 131, 199]
    eb[3] = 1e30; eb[64] = np.nan; eb[65] = np.inf; eb[127] = -np.inf; eb[130] = 1.0; ab[131] = np.nan; tb[199] = np.nan
    res = solver(eb, ab, tb, Mb, plx)
    solver.close()
    lm, lm0 = res["log_marginal_likelihood"], clean["log_marginal_likelihood"]
    assert np.all(np.isneginf(lm[bad])) and all(np.all(np.isnan(res[q][bad])) for q in "ABFG")
    good = np.setdiff1d(np.arange(W), bad)
    assert np.array_equal(lm[good], lm0[good]) and all(np.array_equal(res[q][good], clean[q][good]) for q in "ABFG")
