"""Large-shape sanity run (GPU box): 1e6 walkers x 1e3 rows and 2e4 walkers x 1e5 rows through the C ABI — finite, deterministic,
forward == gradient value, and a sample of walkers against the oracle. Not collected by pytest."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob, gpu_binding as gb, synth

for E, W in ((1000, 1_000_000), (100_000, 20_000)):      # tests/test_sweeps_gpu.py runs the first shape under pytest -m gpu
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3, seed=123)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    path = gb.GpuPath(obs, planets)
    t0 = time.perf_counter()
    ll, g, _ = path.eval(cfg["elems"], None, grad=True)
    dt = time.perf_counter() - t0
    ll2, g2, _ = path.eval(cfg["elems"], None, grad=True)
    llf, _, _ = path.eval(cfg["elems"], None, grad=False)
    path.close()
    assert np.all(np.isfinite(ll)) and np.array_equal(ll, ll2) and np.array_equal(g, g2) and np.array_equal(ll, llf)
    idx = np.random.default_rng(0).choice(W, 16, replace=False)
    ll_o, g_o, _ = ob.oracle_eval(obs, planets, cfg["elems"][:, idx], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False), n_threads=0)
    e_ll = np.max(np.abs(ll[idx] - ll_o) / np.abs(ll_o))
    e_g = np.max(np.abs(g[:, idx] - g_o) / np.maximum(np.abs(g_o).max(axis=1, keepdims=True), 1e-300))
    print(f"E={E} W={W}: first host-buffer eval {dt*1e3:.1f} ms; sample vs oracle ll {e_ll:.1e} grad/scale {e_g:.1e}", flush=True)
    assert e_ll < 1e-12 and e_g < 1e-9
print("ok")
