"""In-kernel timeline of k_main<1, true, false, 1, FUSED> (development build with -DOCTO_KMAIN_TRACE: tools/build_variant.py kmtrace
-DOCTO_KMAIN_TRACE): 100 MHz wall-clock stamps of thread 0 of four blocks (first / last tile x first / last task) at entry, elements loaded,
pieces written, pieces barrier passed, constants assembled, table filled (row loop starts), row loop done, combine done, partials stored.
    OCTOFITTER_HIP_LIB=octofitter.jl_amd/lib/variants/liboctofitter_hip_kmtrace.so python tools/kmain_trace.py [W=1250] [E=10000]"""
import ctypes as C, os, sys
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
for _ in range(50): fn.ln_like_device(el, None, grad=True, out=out)
torch.cuda.synchronize()
names = ["entry", "elements in", "pieces written", "pieces barrier", "assembled", "table filled", "rows done", "combined", "stored"]
acc = np.zeros((4, 9)); n = 0
buf = (C.c_ulonglong * 64)()
for it in range(50):
    fn.ln_like_device(el, None, grad=True, out=out); torch.cuda.synchronize()
    assert fn.lib.octo_debug_kmain_trace(buf) == 0
    t = np.array(list(buf), dtype=np.float64).reshape(4, 16)[:, :9]
    acc += (t - t[:, :1].min()) / 100.0; n += 1      # µs since the earliest entry of the four
acc /= n
for b, nm in enumerate(["tile 0 / task 0", "tile 0 / last task", "last tile / task 0", "last tile / last task"]):
    print(f"W={W} E={E} {nm:22s}: " + "  ".join(f"{names[k]} {acc[b, k]:6.2f}" for k in range(9)), flush=True)
fn.close()
