"""Per-kernel timeline of a mid-size callback through the host-buffer entry point (octo_model_logpost, D = 11 test model, 50 epochs, W walkers),
from a rocprofv3 kernel trace of this script:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python tools/midsize_trace.py run 1024
    python tools/midsize_trace.py report <dir>
Development aid: what a Pigeons-sized batch pays per launch."""
import csv, glob, json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if sys.argv[1] == "run":
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import numpy as np
    from __graft_entry__ import load_package
    import synth
    pkg = load_package(); capi = pkg.capi
    W = int(sys.argv[2]); E = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    case = json.loads((ROOT / "tests" / "golden" / "model.json").read_text())["cases"][0]
    c0 = synth.config_astrom(n_epochs=E, n_walkers=1, cfg=3)["table"]
    table = dict(epoch=c0["epoch"], ra=c0["ra"], dec=c0["dec"], σ_ra=c0["σ_ra"], σ_dec=c0["σ_dec"])
    b = pkg.Planet(name="b", basis="Visual{KepOrbit}", observations=[pkg.PlanetRelAstromObs(table, name="astrom")],
                   variables=pkg.variables(a=pkg.Uniform(0, 100), e=pkg.Uniform(0.0, 0.99), i=pkg.Sine(), ω=pkg.UniformCircular(),
                                           Ω=pkg.UniformCircular(), θ=pkg.UniformCircular(), tp=pkg.θ_at_epoch_to_tperi("θ", 50000)))
    model = pkg.LogDensityModel(pkg.System(name="T", companions=[b], observations=[],
                                variables=pkg.variables(M=pkg.truncated(pkg.Normal(1.2, 0.1), lower=0.1), plx=pkg.truncated(pkg.Normal(50.0, 0.02), lower=0.1))))
    fn = model.ln_like
    rng = np.random.default_rng(3)
    th = np.ascontiguousarray(np.asarray(case["theta_t"])[:, :1] + 0.05 * rng.normal(size=(model.D, W))); lp = np.empty(W); g = np.empty_like(th)
    args = (fn._ctx, model._m, capi._dptr(th), W, W, capi._dptr(lp), capi._dptr(g))
    for _ in range(300): fn.lib.octo_model_logpost(*args)
    t0 = time.perf_counter()
    for _ in range(300): fn.lib.octo_model_logpost(*args)
    print(f"W={W} E={E}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us per call (under the tracer)")
    model.close()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-16:]
    t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{r['Kernel_Name'][:64]:64s} start {1e-3*(s-t0):8.1f} us  gap {1e-3*(s-prev_end):6.1f}  dur {1e-3*(e-s):6.1f}")
        prev_end = e
