"""Short-period orbits observed over thousands of periods (hot-Jupiter RV, a ~ 0.03-0.1 AU over 15 yr) against the 60-digit
oracle: the mean anomaly is then a small remainder of a large phase, |t − tp|/P ~ 1e3, and its rounding (ours: the phase in
orbits minus its nearest integer; the reference's: n·Δt then rem2pi) is the error floor. Run on a GPU box;
tests/test_sweeps_gpu.py runs a fixed-seed slice under pytest."""
import sys
from pathlib import Path
import numpy as np, mpmath as mp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import gpu_binding as gb, mp_oracle as mo, oracle_binding as ob


def run(W=24, seed=1, verbose=False, small_batch=None):
    """Returns worst (GPU ll, GPU grad, reference-order oracle ll, reference-order oracle grad) errors against 60 digits."""
    rng = np.random.default_rng(seed)
    n = 12
    a = rng.uniform(0.03, 0.1, W); e = rng.uniform(0, 0.5, W); M = rng.uniform(0.8, 1.3, W)
    el = np.stack([a, e, rng.uniform(0.3, 2.8, W), rng.uniform(0, 6.28, W), np.zeros(W), 50000 + rng.uniform(-3, 3, W), M, np.full(W, 1.0), rng.uniform(0.3, 3, W)])
    ep = np.sort(50000 + rng.uniform(0, 5500, n))
    rv = rng.normal(0, 80, n)
    obs = [dict(kind="RV_ABS", planet=-1, epoch=ep.tolist(), y1=rv.tolist(), y2=None, s1=[3.0] * n, s2=None, cor=None)]
    obs_c = [dict(kind=2, planet=-1, epoch=ep, y1=rv, y2=None, s1=np.full(n, 3.0), s2=None, cor=None)]
    planets = [dict(orbit_kind=1, has_mass=True)]
    nu = np.stack([rng.normal(0, 5, W), rng.uniform(0.5, 4, W), np.zeros(W)])
    ll, g, gn = gb.gpu_eval(obs_c, planets, el, nu, grad=True, small_batch=small_batch)
    ll_o, g_o, gn_o = ob.oracle_eval(obs_c, planets, el, nu, grad=True)
    worst = np.zeros(4)
    for w in range(W):
        e_ = [[mp.mpf(float(v)) for v in el[:, w]]]; n_ = [[mp.mpf(float(v)) for v in nu[:, w]]]
        f0, g_el, g_nu, s_el, s_nu = mo.ln_like_and_grad(mo.DEFAULT_CONSTS, planets, obs, e_, n_, with_scale=True)
        gm = np.array([float(x) for x in g_el[0]]); sm = np.array([float(x) for x in s_el[0]]); sc = np.maximum(sm + sm.max(), 1e-300)
        act = [0, 1, 3, 5, 6, 8]
        errs = (float(abs(ll[w] - f0) / max(1, abs(f0))), float(np.max(np.abs(g[act, w] - gm[act]) / sc[act])),
                float(abs(ll_o[w] - f0) / max(1, abs(f0))), float(np.max(np.abs(g_o[act, w] - gm[act]) / sc[act])))
        worst = np.maximum(worst, errs)
        if verbose:
            norb = np.max(np.abs(ep - el[5, w])) / (365.2568983840419 * np.sqrt(a[w] ** 3 / M[w]))
            print(f"{w:3d} orbits {norb:7.0f}: GPU ll {errs[0]:.1e} grad {errs[1]:.1e} | reference-order oracle ll {errs[2]:.1e} grad {errs[3]:.1e}", flush=True)
    return worst


if __name__ == "__main__":
    worst = run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1, verbose=True)
    print(f"worst vs 60-digit: GPU ll {worst[0]:.2e} grad {worst[1]:.2e} | reference-order oracle ll {worst[2]:.2e} grad {worst[3]:.2e}")
