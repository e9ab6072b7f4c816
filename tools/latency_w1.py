"""Latency of ONE parameter set through the host-buffer entry point (what a single-chain sampler pays per gradient):
octo_eval with W = 1, for a 50-epoch and a 1e4-epoch table. Development aid."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package()
for E in (50, 10000):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    el = np.ascontiguousarray(cfg["elems"])
    for grad in (False, True):
        for _ in range(200): fn.ln_like_arrays(el, None, grad=grad)
        t0 = time.perf_counter(); n = 2000
        for _ in range(n): fn.ln_like_arrays(el, None, grad=grad)
        dt = (time.perf_counter() - t0) / n
        print(f"E={E:6d} W=1 grad={grad}: {dt*1e6:7.1f} us per call (Python + ctypes + H2D + 3 kernels + D2H + sync)", flush=True)
    fn.close()
