"""Measure the PCIe-inclusive rate of the hot path (octo_eval with HOST buffers: H2D elems, kernels, D2H ll+grad).
Reported in DESIGN.md next to the HBM-resident number; never used as bench.py's `value`."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
import synth
pkg = load_package()
cfg = synth.config_astrom(cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
for grad in (True, False):
    for _ in range(3):
        fn.ln_like_arrays(cfg["elems"], None, grad=grad)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        fn.ln_like_arrays(cfg["elems"], None, grad=grad)
    dt = (time.perf_counter() - t0) / n
    print(f"host buffers, grad={grad}: {dt*1e3:.3f} ms/eval -> {1e8/dt:.3e} evals/s")
