"""Step time at 1e4 epochs for a few batch sizes, no events in the stream; run under OCTOFITTER_HIP_LIB=<a build> to A/B two builds
in one gpurun call (boxes differ by a few % between calls). Development aid."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from __graft_entry__ import load_package
import synth
pkg = load_package()
tag = os.environ.get("OCTOFITTER_HIP_LIB", "default")
import numpy as np
for W, cfgid, grad, with_nuis in ((10000, 3, True, False), (10000, 3, False, False), (4096, 3, True, False), (65536, 3, True, False),
                                  (10000, 3, True, True), (10000, 3, False, True)):
    cfg = synth.config_astrom(n_epochs=10000, n_walkers=W, cfg=cfgid)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    el = torch.tensor(cfg["elems"], device="cuda")
    nu = None
    if with_nuis:      # jitter, platescale, northangle per walker: the raw-σ path of relative-astrometry.jl:234-252
        rng = np.random.default_rng(1)
        nu = torch.tensor(np.stack([rng.uniform(0, 3, W), rng.normal(1, 0.01, W), rng.normal(0, 0.02, W)]), device="cuda")
    out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el) if grad else None,
           torch.empty_like(nu) if (grad and nu is not None) else None)
    for _ in range(50): fn.ln_like_device(el, nu, grad=grad, out=out)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200 if W <= 10000 else 40
        for _ in range(n): fn.ln_like_device(el, nu, grad=grad, out=out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
    print(f"{tag[-40:]:>40} W={W:6d} grad={grad} nuis={with_nuis}: {best*1e6:8.1f} us/step  {W*1e4/best:.3e} evals/s", flush=True)
    fn.close()
