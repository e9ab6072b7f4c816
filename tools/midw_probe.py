"""Device-resident step time (back-to-back launches) of both kernel families at mid-size batches, E = 1e4 and 300: where is the
crossover between k_small (one launch, lane = epoch) and the throughput kernels (k_setup -> k_main -> k_finish, lane = walker)?
   python tools/midw_probe.py"""
import ctypes as C, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from __graft_entry__ import load_package
import synth
pkg = load_package(); capi = pkg.capi
tag = os.environ.get("OCTOFITTER_HIP_LIB", "default")[-30:]
for E in (10000, 300):
    cfg = synth.config_astrom(n_epochs=E, n_walkers=4096, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    for W in (128, 256, 512, 768, 1024, 2048, 4096):
        el = torch.tensor(np.ascontiguousarray(cfg["elems"][:, :W]), device="cuda")
        out = (torch.empty(W, dtype=torch.float64, device="cuda"), torch.empty_like(el), None)
        res = {}
        for sb in (1024, 0):
            if sb and W > 1024:
                res[sb] = float("nan"); continue
            fn._check(fn.lib.octo_ctx_set_small_batch(fn._ctx, sb), "set")
            for _ in range(100): fn.ln_like_device(el, None, grad=True, out=out)
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(200): fn.ln_like_device(el, None, grad=True, out=out)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
            res[sb] = best
        print(f"{tag:>30} E={E:5d} W={W:5d} fwd+grad device-resident: k_small {res[1024]*1e6:7.1f} us   throughput kernels {res[0]*1e6:7.1f} us", flush=True)
    fn.close()
