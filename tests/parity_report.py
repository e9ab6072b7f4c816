"""Print the worst GPU-vs-oracle and GPU-vs-golden errors over the test scenarios (run on a GPU box).
Feeds the tolerances written in tests/test_gpu_parity.py and the parity table in DESIGN.md."""
import json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob, gpu_binding as gb, synth
from conftest import case_tables

def stats(name, ll, g, gn, ll_r, g_r, gn_r, s=None, sn=None):
    ok = np.isfinite(ll_r)
    e_ll = np.max(np.abs(ll[ok] - ll_r[ok]) / np.maximum(1, np.abs(ll_r[ok])))
    def gerr(a, b, sc):
        a, b = a[:, ok], b[:, ok]
        scale = (np.abs(b).max(axis=1, keepdims=True) if sc is None else np.asarray(sc)[:, ok] + np.asarray(sc)[:, ok].max(axis=1, keepdims=True))
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        floor = np.abs(a - b) / np.maximum(scale, 1e-300)
        return np.max(np.minimum(rel, 1e300)), np.max(floor), np.max(np.minimum(rel, floor / 1e-4))   # rel, vs scale, combined
    out = f"{name:34s} ll {e_ll:.2e}"
    if g is not None:
        r, f, c = gerr(g, g_r, s); out += f" | g_el rel {r:.2e} /scale {f:.2e}"
    if gn is not None:
        r, f, c = gerr(gn, gn_r, sn); out += f" | g_nu rel {r:.2e} /scale {f:.2e}"
    print(out, flush=True)

gold = json.load(open(ROOT / "tests/golden/fixtures.json"))
print("== GPU vs golden (50-digit)")
for case in gold["cases"]:
    obs, planets, elems, nuis = case_tables(case)
    ll, g, gn = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
    stats(case["name"], ll, g, gn, np.array(case["ll"]), np.array(case["g_elems"]), None if nuis is None else np.array(case["g_nuis"]),
          case["s_elems"], case["s_nuis"])
print("== oracle vs golden (50-digit)")
for case in gold["cases"]:
    obs, planets, elems, nuis = case_tables(case)
    ll, g, gn = ob.oracle_eval(obs, planets, elems, nuis, grad=True)
    stats(case["name"], ll, g, gn, np.array(case["ll"]), np.array(case["g_elems"]), None if nuis is None else np.array(case["g_nuis"]),
          case["s_elems"], case["s_nuis"])
print("== GPU vs oracle, seeded")
for E, W in [(96, 257), (2048, 130), (10000, 64)]:
    cfg = synth.config_astrom(n_epochs=E, n_walkers=W, seed=100 + E)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    pl = [dict(orbit_kind=0, has_mass=False)]
    ll, g, _ = gb.gpu_eval(obs, pl, cfg["elems"], None, grad=True)
    ll_o, g_o, _ = ob.oracle_eval(obs, pl, cfg["elems"], None, grad=True, active=synth.active_mask(1, 1, mass=False, nuis=False), n_threads=0)
    stats(f"astrom {E}x{W}", ll, g[:8], None, ll_o, g_o[:8], None)
    emax = cfg["elems"][1].max()
    print(f"   (e max in batch {emax:.3f})")
