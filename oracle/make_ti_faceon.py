#!/usr/bin/env python
"""Fixture F12 (tests/golden/ti_faceon.json): the Thiele-Innes near-face-on walker the long random sweep found
(tests/stress_parity.py seed 201, system 1375, walker 74: (u − |v|)/u = 2.4e-12). The reference's a = α/plx with
α² = u + √((u+v)(u−v)) (src/parameterizations.jl:15-18) cancels there; the kernels use the cancellation-free form.
Expected values: log-likelihood and gradient from the independent 60-digit oracle (oracle/mp_oracle.py). CPU only:
    python oracle/make_ti_faceon.py"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, mpmath as mp
import mp_oracle as mo

SEED, INDEX, WALKER = 201, 1375, 74


def main():
    import types
    sys.modules.setdefault("gpu_binding", types.ModuleType("gpu_binding"))      # stress_parity imports it; no GPU is used here
    import stress_parity as sp
    rng = np.random.default_rng(SEED)
    for _ in range(INDEX + 1):
        sysm = sp.draw_system(rng)
    obs, planets, elems, nuis = sysm
    w = WALKER
    KN = {0: "ASTROM_RADEC", 1: "ASTROM_SEPPA", 2: "RV_ABS", 3: "RV_ABS_MARG", 4: "RV_REL", 5: "ONEIL_RADEC", 6: "ONEIL_SEPPA", 7: "HGCA"}
    fl = lambda x: None if x is None else [float(v) for v in x]
    obs_m = [dict(kind=KN[o["kind"]], planet=int(o["planet"]), epoch=fl(o["epoch"]), y1=fl(o["y1"]), y2=fl(o["y2"]), s1=fl(o["s1"]), s2=fl(o["s2"]),
                  cor=fl(o.get("cor")), extra=fl(o.get("extra"))) for o in obs]
    P = len(planets)
    el = [[mp.mpf(float(elems[p * 9 + k, w])) for k in range(9)] for p in range(P)]
    nu = [[mp.mpf(float(nuis[o * 3 + k, w])) for k in range(3)] for o in range(len(obs))]
    ll, g_el, g_nu, s_el, s_nu = mo.ln_like_and_grad(mo.DEFAULT_CONSTS, planets, obs_m, el, nu, with_scale=True)
    A, B, F, G = [float(elems[k, w]) for k in (0, 2, 3, 4)]
    u = 0.5 * (A * A + B * B + F * F + G * G); v = A * G - B * F
    case = dict(name="F12_thiele_innes_near_face_on",
                note=f"stress_parity seed {SEED}, system {INDEX}, walker {w}; (u - |v|)/u = {(u - abs(v)) / u:.3e}; expected values at 60 digits (mp_oracle)",
                planets=planets, obs=obs_m, elems=[[float(elems[r, w])] for r in range(P * 9)], nuis=[[float(nuis[r, w])] for r in range(len(obs) * 3)],
                ll=[float(ll)], g_elems=[[float(g_el[p][k])] for p in range(P) for k in range(9)],
                g_nuis=[[float(g_nu[o][k])] for o in range(len(obs)) for k in range(3)],
                s_elems=[[float(s_el[p][k])] for p in range(P) for k in range(9)], s_nuis=[[float(s_nu[o][k])] for o in range(len(obs)) for k in range(3)])
    out = ROOT / "tests" / "golden" / "ti_faceon.json"
    out.write_text(json.dumps(dict(generator="oracle/make_ti_faceon.py", cases=[case])))
    print("wrote", out, out.stat().st_size, "bytes; ll =", float(ll))


if __name__ == "__main__":
    main()
