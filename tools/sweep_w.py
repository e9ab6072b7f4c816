"""Throughput of the fwd+grad path vs batch size (walkers), 1e4 RA/Dec epochs, inputs resident in HBM."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from __graft_entry__ import load_package
import synth
pkg = load_package()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for W in (1, 64, 256, 1024, 4096, 10000, 16384, 65536, 262144):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    el = torch.tensor(cfg["elems"], device="cuda")
    out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), None)
    n = 50 if W * E < 2e8 else 10
    for _ in range(5): fn.ln_like_device(el, None, grad=True, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn.ln_like_device(el, None, grad=True, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"W={W:7d} E={E}: {dt*1e6:9.1f} us/step  {W*E/dt:.3e} evals/s", flush=True)
    fn.close()
