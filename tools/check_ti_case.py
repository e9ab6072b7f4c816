"""Who is off in a Thiele-Innes near-face-on case of the random sweep (tests/stress_parity.py; default seed 101, system 42)? The worst
walker's gradient from the device, from the reference-order C restatement and from the 60-digit oracle. Development aid.
    python tools/check_ti_case.py [seed] [system index] [size scale]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, mpmath as mp
import stress_parity as sp, oracle_binding as ob, gpu_binding as gb, mp_oracle as mo
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 101
INDEX = int(sys.argv[2]) if len(sys.argv) > 2 else 42
sp.SCALE = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(SEED)
for k in range(INDEX + 1):
    sysm = sp.draw_system(rng)
obs, planets, elems, nuis = sysm
print(sp.describe(sysm))
ll, g, gn = gb.gpu_eval(obs, planets, elems, nuis, grad=True)
ll_o, g_o, gn_o = ob.oracle_eval(obs, planets, elems, nuis, grad=True, n_threads=0)
ok = np.isfinite(ll_o)
G = np.concatenate([g, gn]); Go = np.concatenate([g_o, gn_o])
scale = np.maximum(np.abs(Go[:, ok]).max(axis=1, keepdims=True), 1e-10 * np.abs(Go[:, ok]).max())
err = np.abs(G - Go) / scale
err[:, ~ok] = 0
r, w = np.unravel_index(np.argmax(err), err.shape)
print("worst input row", r, "walker", w, "err/scale", err[r, w], "GPU", G[r, w], "C oracle", Go[r, w])
KN = {0: "ASTROM_RADEC", 1: "ASTROM_SEPPA", 2: "RV_ABS", 3: "RV_ABS_MARG", 4: "RV_REL", 5: "ONEIL_RADEC", 6: "ONEIL_SEPPA", 7: "HGCA"}
obs_m = [dict(kind=KN[o["kind"]], planet=o["planet"], epoch=list(map(float, o["epoch"])), y1=list(map(float, o["y1"])),
              y2=None if o["y2"] is None else list(map(float, o["y2"])), s1=None if o["s1"] is None else list(map(float, o["s1"])),
              s2=None if o["s2"] is None else list(map(float, o["s2"])), cor=None if o.get("cor") is None else list(map(float, o["cor"])),
              extra=None if o.get("extra") is None else list(map(float, o["extra"]))) for o in obs]
P = len(planets)
el = [[mp.mpf(float(elems[p * 9 + k, w])) for k in range(9)] for p in range(P)]
nu = [[mp.mpf(float(nuis[o * 3 + k, w])) for k in range(3)] for o in range(len(obs))]
f0, g_el, g_nu, s_el, s_nu = mo.ln_like_and_grad(mo.DEFAULT_CONSTS, planets, obs_m, el, nu, with_scale=True)
gm = np.array([float(g_el[p][k]) for p in range(P) for k in range(9)] + [float(g_nu[o][k]) for o in range(len(obs)) for k in range(3)])
ell = np.abs(ll - ll_o) / np.maximum(1, np.abs(ll_o)); ell[~ok] = 0; wl = int(np.argmax(ell))
if wl != w:
    el2 = [[mp.mpf(float(elems[p * 9 + k, wl])) for k in range(9)] for p in range(P)]
    nu2 = [[mp.mpf(float(nuis[o * 3 + k, wl])) for k in range(3)] for o in range(len(obs))]
    f0l = mo.ln_like(mo.DEFAULT_CONSTS, planets, obs_m, el2, nu2)
else:
    f0l = f0
print("worst ll walker", wl, "GPU", ll[wl], "C oracle", ll_o[wl], "60-digit", float(f0l), "| GPU err", abs(ll[wl] - float(f0l)) / max(1, abs(float(f0l))),
      "| C oracle err", abs(ll_o[wl] - float(f0l)) / max(1, abs(float(f0l))))
print("60-digit value", gm[r], "| GPU err", abs(G[r, w] - gm[r]) / scale[r, 0], "| C oracle err", abs(Go[r, w] - gm[r]) / scale[r, 0])
for ip, pl in enumerate(planets):
    if pl["orbit_kind"] != 2:
        continue
    A, B, F, Gc = [elems[9 * ip + k, w] for k in (0, 2, 3, 4)]
    u = 0.5 * (A * A + B * B + F * F + Gc * Gc); v = A * Gc - B * F
    print(f"Thiele-Innes planet {ip}: (u - |v|)/u =", (u - abs(v)) / u, " (0 = face-on: a = alpha/plx loses digits there)")
