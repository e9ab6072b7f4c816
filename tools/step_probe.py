"""Back-to-back step time and k_main time of config 3 at one batch size (device-resident, fwd+grad), for knob sweeps (OCTO_CHUNK, OCTO_ROUNDS,
OCTOFITTER_HIP_LIB) inside one gpurun call.   python tools/step_probe.py [W=1250] [E=10000]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package()
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
E = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = synth.config_astrom(n_epochs=E, n_walkers=10000, cfg=3)
obs, planet = synth.to_mirror(pkg, cfg)
fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
el = torch.tensor(np.ascontiguousarray(cfg["elems"][:, :W]), device="cuda")
out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), None)
for _ in range(200): fn.ln_like_device(el, None, grad=True, out=out)
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): fn.ln_like_device(el, None, grad=True, out=out)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 300)
fn.timing_enable(1)
for _ in range(100): fn.ln_like_device(el, None, grad=True, out=out)
torch.cuda.synchronize()
kmed = fn.timing_stats()[0]
print(f"W={W} E={E} step {best*1e6:7.2f} us  k_main {kmed*1e3:7.2f} us  knobs: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("OCTO")), flush=True)
fn.close()
