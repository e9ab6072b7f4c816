"""One small-batch configuration, N calls of octo_eval (fwd+grad) — a target for rocprofv3 --kernel-trace --stats, which then gives
k_small's device duration next to the end-to-end latency printed here.   python tools/small_batch_one.py E W [N]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
E, W = int(sys.argv[1]), int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
el = np.ascontiguousarray(cfg["elems"]); ll = np.empty(W); g = np.empty_like(el)
args = (fn._ctx, fn._ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None)
for _ in range(200): fn.lib.octo_eval(*args)
t0 = time.perf_counter()
for _ in range(N): fn.lib.octo_eval(*args)
print(f"E={E} W={W}: {(time.perf_counter() - t0) / N * 1e6:.1f} us per octo_eval (fwd+grad)")
fn.close()
