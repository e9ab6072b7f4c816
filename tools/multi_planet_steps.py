"""Throughput-kernel steps of a P-planet system (one RA/Dec table of 1 250 rows per planet + one absolute-RV table of 2 500 rows, per-walker
nuisances, 4 096 walkers, fwd+grad, device-resident) — a rocprofv3 target for the 3- and 4-planet k_main / k_finish instantiations (VERDICT r3
item 3) and a same-box A/B probe (OCTOFITTER_HIP_LIB).   python tools/multi_planet_steps.py P [N steps]"""
import os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import ctypes as C
import numpy as np, torch
import gpu_binding as gb
import stress_parity as sp
capi = gb.capi
P = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W = 4096
rng = np.random.default_rng(11)
planets = [dict(orbit_kind=0, has_mass=True) for _ in range(P)]
elems = np.concatenate([sp.planet_elems(rng, W, 0, 2 + 6 * i, 6 + 6 * i) for i in range(P)])
obs = []
for ip in range(P):
    n = 1250; ep = np.sort(50000 + rng.uniform(0, 4000, n))
    obs.append(dict(kind=0, planet=ip, epoch=ep, y1=rng.normal(0, 300, n), y2=rng.normal(0, 300, n), s1=rng.uniform(3, 12, n), s2=rng.uniform(3, 12, n), cor=None))
n = 2500; ep = np.sort(50000 + rng.uniform(0, 4000, n))
obs.append(dict(kind=2, planet=-1, epoch=ep, y1=rng.normal(0, 30, n), y2=None, s1=rng.uniform(1, 8, n), s2=None, cor=None))
nuis = np.zeros((len(obs) * 3, W))
for io in range(P):
    nuis[io * 3] = 1.0; nuis[io * 3 + 1] = 1.0
nuis[P * 3] = 3.0; nuis[P * 3 + 1] = 2.0
rows = sum(len(o["epoch"]) for o in obs)
with gb.GpuPath(obs, planets, small_batch=0) as g:
    el = torch.tensor(elems, device="cuda"); nu = torch.tensor(nuis, device="cuda")
    ll = torch.empty(W, dtype=torch.float64, device="cuda"); ge = torch.empty_like(el); gn = torch.empty_like(nu)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = (g.ctx, g.ds, C.c_void_p(el.data_ptr()), C.c_void_p(nu.data_ptr()), W, W, C.c_void_p(ll.data_ptr()), C.c_void_p(ge.data_ptr()), C.c_void_p(gn.data_ptr()), st)
    for _ in range(30): assert g.lib.octo_eval_device(*args) == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): g.lib.octo_eval_device(*args)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
    print(f"{os.environ.get('OCTOFITTER_HIP_LIB', 'default')[-24:]:>24} P={P} W={W} rows={rows}: {dt * 1e6:8.1f} us per step  {W * rows / dt:.3e} evals/s  "
          f"({W * rows * P / dt:.3e} Kepler solves/s)  ll[0]={float(ll[0]):.6f}", flush=True)
