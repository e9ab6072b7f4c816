"""Step time vs rows-per-wave (OCTO_CHUNK) at 1e4 epochs. Development aid."""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from __graft_entry__ import load_package
import synth
pkg = load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = synth.config_astrom(n_epochs=10000, n_walkers=W, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
el = torch.tensor(cfg["elems"], device="cuda")
out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), None)
for chunk in (0, 8, 16, 24, 32, 48, 64, 96, 128, 192):
    if chunk: os.environ["OCTO_CHUNK"] = str(chunk)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    for _ in range(50): fn.ln_like_device(el, None, grad=True, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for _ in range(n): fn.ln_like_device(el, None, grad=True, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"chunk={chunk or 'auto':>4} W={W}: {dt*1e6:8.1f} us/step  {W*1e4/dt:.3e} evals/s", flush=True)
    fn.close()
