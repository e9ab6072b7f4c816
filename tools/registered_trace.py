"""Per-kernel timeline of octo_eval on registered host arrays (config 3), from a rocprofv3 kernel trace of this script:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/registered_trace.py run
    python tools/registered_trace.py report <dir>
Development aid: where the microseconds between the device-resident step and the PCIe-inclusive call go."""
import csv, glob, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if sys.argv[1] == "run":
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    from __graft_entry__ import load_package
    import synth
    pkg = load_package(); capi = pkg.capi
    W, E = 10000, 10000
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, cfg=3)
    obs, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    el_h = np.ascontiguousarray(cfg["elems"]); ll_h = np.empty(W); g_h = np.empty_like(el_h)
    a_ = (fn._ctx, fn._ds, capi._dptr(el_h), None, W, W, capi._dptr(ll_h), capi._dptr(g_h), None)
    fn.host_register(el_h, ll_h, g_h)
    for _ in range(40): fn.lib.octo_eval(*a_)
    fn.host_unregister(el_h, ll_h, g_h)
    fn.close()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-3 * 12:]
    t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{r['Kernel_Name'][:60]:60s} start {1e-3*(s-t0):9.1f} us  gap {1e-3*(s-prev_end):7.1f}  dur {1e-3*(e-s):7.1f}")
        prev_end = e
