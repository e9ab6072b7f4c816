"""Randomised OFTI sweep (octo_ofti_* vs the oracle's restatement of src/parameterizations.jl:318-405): random table sizes,
correlations, prior widths and batch sizes. Run on a GPU box: python tests/stress_ofti.py [n] [seed]."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob
from __graft_entry__ import load_package
pkg = load_package()

def check_case(rng, verbose=False):
    """One random OFTI table + batch: device vs the LU restatement. Returns (ok, e_logml, e_abfg, description)."""
    n = int(rng.choice([1, 2, 3, 8, 33, 200, 1500]))
    W = int(rng.choice([1, 7, 64, 65, 300, 1000]))
    ep = np.sort(50000 + rng.uniform(0, 5000, n))
    ra, dec = rng.normal(0, 400, n), rng.normal(0, 400, n)
    s_ra, s_dec = rng.uniform(2, 15, n), rng.uniform(2, 15, n)
    cor = rng.uniform(-0.8, 0.8, n) if rng.random() < 0.5 else None
    sig = float(10 ** rng.uniform(1, 4))
    nl = np.stack([rng.uniform(0, 0.9, W), rng.uniform(2, 30, W), 50000 + rng.uniform(-3000, 3000, W), rng.uniform(0.7, 1.8, W), rng.uniform(10, 80, W)])
    solver = pkg.OftiLinearSolver(ep, ra, dec, s_ra, s_dec, cor, sig)
    res = solver(nl[0], nl[1], nl[2], nl[3], nl[4])
    abfg = np.stack([np.atleast_1d(res[q]) for q in "ABFG"]); lm = np.atleast_1d(res["log_marginal_likelihood"])
    solver.close()
    abfg_o, lm_o = ob.oracle_ofti(ep, ra, dec, s_ra, s_dec, cor, sig, nl)
    e_lm = np.max(np.abs(lm - lm_o) / np.maximum(1, np.abs(lm_o)))
    e_ab = np.max(np.abs(np.asarray(abfg) - abfg_o) / np.maximum(np.abs(abfg_o).max(axis=0, keepdims=True), 1e-300))
    # n <= 2 epochs leave the 4 constants under-determined: the 4x4 system is held up by the prior alone (condition number
    # ~ σ_ABFG² × weights), and Julia's LU (restated by the oracle) and the device Cholesky both lose those digits
    # (checked against 60 digits: the device value is the accurate one there, e.g. 3e-11 vs the LU restatement's 3e-5)
    lim = (1e-4, 1e-6) if n <= 2 else (1e-9, 1e-8)
    good = bool(e_lm < lim[0] and e_ab < lim[1])
    if verbose and (not good or e_lm > 1e-8):      # who is off? the 60-digit value for the worst walker
        sys.path.insert(0, str(ROOT / "oracle"))
        import mpmath as mp, mp_oracle as mo
        wb = int(np.argmax(np.abs(lm - lm_o) / np.maximum(1, np.abs(lm_o))))
        f = lambda v: [mp.mpf(float(x)) for x in v]
        r = mo.ofti_linear_solve(mo.DEFAULT_CONSTS, f(ep), f(ra), f(dec), f(s_ra), f(s_dec), None if cor is None else f(cor), mp.mpf(sig),
                                 *[mp.mpf(float(v)) for v in nl[:, wb]])
        truth = float(r[-1] if not isinstance(r, dict) else r["log_marginal_likelihood"])
        print(f"\n     walker {wb}: 60-digit logml {truth:.15g}; device {lm[wb]:.15g} (err {abs(lm[wb]-truth):.1e}); LU oracle {lm_o[wb]:.15g} (err {abs(lm_o[wb]-truth):.1e})")
    return good, float(e_lm), float(e_ab), f"n={n} W={W} cor={cor is not None} sigma={sig:.3g}"


def main():
    n_sys = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0; worst = [0.0, 0.0]
    for k in range(n_sys):
        good, e_lm, e_ab, desc = check_case(rng, verbose=True)
        bad += not good; worst = [max(worst[0], e_lm), max(worst[1], e_ab)]
        print(f"{k:3d} {desc}: logml {e_lm:.1e} ABFG {e_ab:.1e}{'' if good else '   <-- FAIL'}", flush=True)
    print(f"worst: logml {worst[0]:.2e} ABFG {worst[1]:.2e}; failures {bad}")
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
