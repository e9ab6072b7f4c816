#!/usr/bin/env python
"""Fallback rates of k_main's warm-started row loop, simulated on the CPU (NumPy) for a table and a batch of walkers — VERDICT r5 item 1:
"fallback-rate histogram by eccentricity bin", and what tiling the walkers by a severity key buys before it is built.

  python tools/warm_rates.py [config3|wide_prior] [--walkers N]

For every walker and every row: 1/D of the PREVIOUS row's solution against the lane's bound thr = (tol/ΔM³)^(1/5) (octo_device.h: KWarm);
a wave-row is cold when any of its 64 lanes fails, a wave whose fastest lane vetoes the step bound (ΔM > 0.0314) is cold altogether.
Orders compared: the batch as drawn; walkers sorted by the expected lane failure rate p (the share of the orbit, in mean anomaly, with
1/D >= thr — closed form) in segments of 4 096 (what k_tile_sort does) and over the whole batch."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import synth

TOL, VETO = 1e-3, 0.0314
which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "config3"
W = int(sys.argv[sys.argv.index("--walkers") + 1]) if "--walkers" in sys.argv else 4096
cfg = synth.config_wide_prior(n_epochs=10_000, n_walkers=W) if which == "wide_prior" else synth.config_astrom(n_epochs=10_000, n_walkers=W, cfg=3)
t = cfg["table"]["epoch"]
el = cfg["elems"]
a, e, tp, M = el[0], el[1], el[5], el[6]
P = synth.K_YR * np.sqrt(a ** 3 / M)
dm = 2 * np.pi * np.median(np.diff(t))
dM = dm / P
thr = (TOL / dM ** 3) ** 0.2
veto = dM > VETO


def lane_fail_share(e, thr):
    """share of the orbit (in mean anomaly) where 1/(1 - e cos E) >= thr, i.e. e cos E >= 1 - 1/thr"""
    g = 1.0 - 1.0 / thr
    c = np.clip(g / np.maximum(e, 1e-300), -1.0, 1.0)
    E0 = np.arccos(c)
    p = (E0 - e * np.sin(E0)) / np.pi
    return np.where(e <= g, 0.0, p)


p = np.where(veto, 1.0 + dM, lane_fail_share(e, thr))      # vetoing lanes last, fastest at the very end

# per-lane failure of every row (from the previous row's solution)
fail = np.zeros((W, t.size), dtype=bool)
for lo in range(0, W, 256):
    sl = slice(lo, min(lo + 256, W))
    Mm = 2 * np.pi * (t[None, :] - tp[sl, None]) / P[sl, None]
    E = synth._kepler(Mm, e[sl, None])
    invD = 1.0 / (1.0 - e[sl, None] * np.cos(E))
    fail[sl, 1:] = invD[:, :-1] >= thr[sl, None]
fail[veto, :] = True


def wave_rows_cold(order):
    n_tiles = W // 64
    f = fail[order[:n_tiles * 64]].reshape(n_tiles, 64, -1)
    return f.any(axis=1).mean()


ident = np.arange(W)
seg = np.concatenate([lo + np.argsort(p[lo:lo + 4096], kind="stable") for lo in range(0, W, 4096)])
for sz in (512, 1024, 2048):
    o = np.concatenate([lo + np.argsort(p[lo:lo + sz], kind="stable") for lo in range(0, W, sz)])
    print(f"  (segments of {sz}: {wave_rows_cold(o):.3%})")
glob = np.argsort(p, kind="stable")
print(f"{which}: {W} walkers x {t.size} rows, step {dm / (2 * np.pi):.3f} d; lanes vetoing the step bound: {veto.mean():.3%}; lane-rows failing: {fail[~veto].mean():.3%}")
print(f"  wave-rows cold   as drawn: {wave_rows_cold(ident):.3%}   sorted by p in segments of 4096: {wave_rows_cold(seg):.3%}   sorted over the batch: {wave_rows_cold(glob):.3%}")
print("  by eccentricity bin (lanes that do not veto): share of lane-rows failing | mean closed-form p")
bins = np.linspace(0, 1.0, 11)
for lo, hi in zip(bins[:-1], bins[1:]):
    m = (e >= lo) & (e < hi) & ~veto
    if m.any():
        print(f"    e in [{lo:.1f}, {hi:.1f}): {fail[m].mean():8.4%} | {p[m].mean():8.4%}   ({m.sum()} walkers)")
