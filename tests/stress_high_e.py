"""High-eccentricity check against the 60-digit oracle (not the reference-order C restatement, whose ν-form loses digits as
e -> 1 exactly like the reference's): e = 1 − 10^U(−1, −6), epochs spread over the orbit incl. near periastron.
Run on a GPU box: python tests/stress_high_e.py [n_walkers] [seed]; tests/test_sweeps_gpu.py runs a fixed-seed slice under pytest."""
import sys
from pathlib import Path
import numpy as np, mpmath as mp
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import gpu_binding as gb, mp_oracle as mo


def run(W=40, seed=1, verbose=False, small_batch=None):
    """Returns (worst ll error, worst gradient error / Σ|per-row terms|, the same per unit condition number 1/(1−e))."""
    rng = np.random.default_rng(seed)
    n = 7
    e = 1 - 10 ** rng.uniform(-6, -1, W)
    a = rng.uniform(3, 20, W); M = rng.uniform(0.8, 1.5, W); plx = rng.uniform(20, 60, W)
    P = 365.2568983840419 * np.sqrt(a ** 3 / M)
    tp = 50000 + rng.uniform(-0.5, 0.5, W) * P
    el = np.stack([a, e, np.arccos(rng.uniform(-1, 1, W)), rng.uniform(0, 6.28, W), rng.uniform(0, 6.28, W), tp, M, plx, np.zeros(W)])
    # epochs: some within 1e-3 of a period from periastron of the MEDIAN walker, the rest anywhere
    ep = np.sort(np.concatenate([50000 + rng.uniform(-3000, 3000, n - 2), 50000 + rng.uniform(-2, 2, 2)]))
    obs = [dict(kind="ASTROM_RADEC", planet=0, epoch=ep.tolist(), y1=rng.normal(0, 200, n).tolist(), y2=rng.normal(0, 200, n).tolist(),
                s1=[5.0] * n, s2=[6.0] * n, cor=None)]
    obs_c = [dict(kind=0, planet=0, epoch=ep, y1=np.array(obs[0]["y1"]), y2=np.array(obs[0]["y2"]), s1=np.full(n, 5.0), s2=np.full(n, 6.0), cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    ll, g, _ = gb.gpu_eval(obs_c, planets, el, None, grad=True, small_batch=small_batch)
    worst = [0.0, 0.0, 0.0]
    for w in range(W):
        e_ = [[mp.mpf(float(v)) for v in el[:, w]]]
        f0, g_el, _, s_el, _ = mo.ln_like_and_grad(mo.DEFAULT_CONSTS, planets, obs, e_, None, with_scale=True)
        e_ll = float(abs(ll[w] - f0) / max(1, abs(f0)))
        gm = np.array([float(x) for x in g_el[0]]); sm = np.array([float(x) for x in s_el[0]])
        tol_scale = sm + sm.max()
        e_g = float(np.max(np.abs(g[:8, w] - gm[:8]) / np.maximum(tol_scale[:8], 1e-300)))
        cond = 1 / (1 - e[w])
        if verbose:
            print(f"{w:3d} 1-e={1-e[w]:.1e}: ll {e_ll:.1e}  grad/(sum|terms|) {e_g:.1e}   (x 1/(1-e): {e_g/cond:.1e} per unit condition)", flush=True)
        worst = [max(worst[0], e_ll), max(worst[1], e_g), max(worst[2], e_g / cond)]
    return tuple(worst)


if __name__ == "__main__":
    w = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1, verbose=True)
    print(f"worst: ll {w[0]:.2e}  grad {w[1]:.2e}  grad per unit condition 1/(1-e) {w[2]:.2e}")
