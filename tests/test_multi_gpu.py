"""
Multi-device entry points of the C ABI (-m gpu):
  * octo_eval_multi — one host batch split over several contexts from one host thread (run here with two contexts on device 0, and
    over every visible device when there is more than one);
  * octo_comm_* / octo_pt_step_device — the tempering swap step with the all-gather inside the library (RCCL bound at run time).
    On a one-GPU box RCCL is exercised with a one-rank communicator (dlopen, ncclCommInitRank, ncclAllGather all run); the
    two-rank test spawns one process per GPU and is skipped when fewer than two are visible.
"""
import ctypes as C
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

import synth
from test_gpu_parity import _gpu, _pt_swap_reference

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _device_count():
    import torch
    return torch.cuda.device_count()


def test_eval_multi_splits_a_host_batch(pkg, oracle):
    gb = _gpu()
    capi = pkg.capi
    lib = capi.load_library()
    cfg = synth.config_astrom(n_epochs=300, n_walkers=1001, seed=31)
    t = cfg["table"]
    # (40-day cadence: every wave runs k_main's cold row loop, where a walker's result does not depend on the walkers it shares a wave with. On a
    # table dense enough for the warm-started loop — tests/test_warm_start.py — the wave-uniform fallback makes the last bits of a walker's
    # result a function of its 63 neighbours: a split then agrees with the unsplit batch to rounding, not bitwise.)
    obs = [dict(kind=0, planet=0, epoch=50000.0 + 40.0 * (t["epoch"] - 50000.0), y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    ref = gb.gpu_eval(obs, planets, cfg["elems"], None, grad=True, small_batch=0)      # throughput kernels for the batch and for every slice
    n_vis = _device_count()
    # Round 6: the same split on a DENSE table — daily cadence, the warm-started loop — where a walker's last bits
    # depend on its wave's neighbours and on the batch's row partition: to rounding by default, BITWISE with OCTO_OPT_BATCH_INVARIANT on every context
    # (a 1-GPU against an N-GPU rerun of one chain, checkpoint / resume).
    obs_d = [dict(obs[0], epoch=t["epoch"])]
    for inv in (0, 1):
        opts = {capi.OPT_BATCH_INVARIANT: inv}
        ref_d = gb.gpu_eval(obs_d, planets, cfg["elems"], None, grad=True, small_batch=0, options=opts)
        paths = [gb.GpuPath(obs_d, planets, device=0, small_batch=0, options=opts) for _ in range(3)]
        ctxs = (C.c_void_p * 3)(*[p.ctx for p in paths]); dss = (C.c_void_p * 3)(*[p.ds for p in paths])
        el = np.ascontiguousarray(cfg["elems"]); W = el.shape[1]
        ll = np.full(W, np.nan); g = np.full_like(el, np.nan)
        assert lib.octo_eval_multi(ctxs, dss, 3, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None) == 0
        ok = np.isfinite(ref_d[0])
        if inv:
            assert np.array_equal(ll, ref_d[0]) and np.array_equal(g, ref_d[1]), "OCTO_OPT_BATCH_INVARIANT: split and unsplit must agree bit for bit"
        else:
            assert np.array_equal(np.isfinite(ll), ok)
            assert np.max(np.abs(ll[ok] - ref_d[0][ok]) / np.maximum(1.0, np.abs(ref_d[0][ok]))) < 1e-12
            sc = np.maximum(np.abs(ref_d[1][:8, ok]).max(axis=1, keepdims=True), 1e-300)
            assert np.max(np.abs(g[:8, ok] - ref_d[1][:8, ok]) / sc) < 1e-11
        for p in paths:
            p.close()
    for devices in ([0, 0], [0, 0, 0], list(range(n_vis)) if n_vis > 1 else [0]):
        paths = [gb.GpuPath(obs, planets, device=d, small_batch=0) for d in devices]
        n = len(paths)
        ctxs = (C.c_void_p * n)(*[p.ctx for p in paths]); dss = (C.c_void_p * n)(*[p.ds for p in paths])
        el = np.ascontiguousarray(cfg["elems"]); W = el.shape[1]
        ll = np.full(W, np.nan); g = np.full_like(el, np.nan)
        st = lib.octo_eval_multi(ctxs, dss, n, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None)
        assert st == 0, (st, lib.octo_last_error(paths[0].ctx))
        # walkers are independent and every slice is >= 64 walkers: the same kernels on the same inputs, so bit-identical
        assert np.array_equal(ll, ref[0]) and np.array_equal(g, ref[1]), devices
        # a batch smaller than the number of devices, and a forward-only call
        ll3 = np.full(2, np.nan)
        el3 = np.ascontiguousarray(el[:, :2])
        assert lib.octo_eval_multi(ctxs, dss, n, capi._dptr(el3), None, 2, 2, capi._dptr(ll3), None, None) == 0
        assert np.all(np.abs(ll3 - ref[0][:2]) <= 1e-12 * np.abs(ref[0][:2]))
        for p in paths:
            p.close()


def test_eval_multi_strong_split_shares(pkg, oracle):
    """SURVEY §8(d) "Scaling runs": config 3's 1e4 walkers split evenly over N = 2, 4, 8 devices (5 000 / 2 500 / 1 250 each) through
    octo_eval_multi — here N contexts on device 0 (and the visible devices when there are that many). Every share must be BIT-equal to the
    same walkers evaluated alone by one context (the shards are independent, system.jl:206-241: same kernels, same row partition for
    that batch size), and equal to the unsplit batch to rounding (its row partition differs, so the sums are ordered differently)."""
    gb = _gpu()
    capi = pkg.capi
    lib = capi.load_library()
    cfg = synth.config_astrom(n_epochs=10000, n_walkers=10000, cfg=3)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    el = np.ascontiguousarray(cfg["elems"]); W = el.shape[1]
    single = gb.GpuPath(obs, planets)
    ll_full, g_full, _ = single.eval(el, None, grad=True)
    ok = np.isfinite(ll_full)
    assert ok.sum() > 0.9 * W
    n_vis = _device_count()
    for n in (2, 4, 8):
        devices = list(range(n)) if n_vis >= n else [0] * n
        paths = [gb.GpuPath(obs, planets, device=d) for d in devices]
        ctxs = (C.c_void_p * n)(*[p.ctx for p in paths]); dss = (C.c_void_p * n)(*[p.ds for p in paths])
        ll = np.full(W, np.nan); g = np.full_like(el, np.nan)
        for rep in range(2):      # run-to-run determinism of the split call
            ll2 = np.full(W, np.nan); g2 = np.full_like(el, np.nan)
            assert lib.octo_eval_multi(ctxs, dss, n, capi._dptr(el), None, W, W, capi._dptr(ll2), capi._dptr(g2), None) == 0
            if rep == 0: ll, g = ll2, g2
            else: assert np.array_equal(ll, ll2) and np.array_equal(g, g2), n
        lo = 0
        for i in range(n):
            hi = lo + W // n + (1 if i < W % n else 0)
            ll_s, g_s, _ = single.eval(el[:, lo:hi], None, grad=True)
            assert np.array_equal(ll[lo:hi], ll_s) and np.array_equal(g[:, lo:hi], g_s), (n, i)
            lo = hi
        assert lo == W
        assert np.array_equal(np.isfinite(ll), ok)
        assert np.max(np.abs(ll[ok] - ll_full[ok]) / np.maximum(1.0, np.abs(ll_full[ok]))) < 1e-12
        scale = np.maximum(np.abs(g_full[:8, ok]).max(axis=1, keepdims=True), 1e-300)
        assert np.max(np.abs(g[:8, ok] - g_full[:8, ok]) / scale) < 1e-11
        for p in paths:
            p.close()
    # the same shares against the oracle on a sample of each
    ll_o, g_o, _ = oracle.oracle_eval(obs, planets, el[:, ::625], None, grad=True, active=synth.active_mask(1, 0, mass=False))
    oko = np.isfinite(ll_o)
    assert np.max(np.abs(ll_full[::625][oko] - ll_o[oko]) / np.maximum(1.0, np.abs(ll_o[oko]))) < 1e-9
    single.close()


def test_eval_begin_end_overlap_two_contexts(pkg):
    """octo_eval_begin on two contexts, then octo_eval_end on both: the two halves the multi-device split is made of."""
    gb = _gpu()
    capi = pkg.capi
    lib = capi.load_library()
    cfg = synth.config_astrom(n_epochs=200, n_walkers=640, seed=32)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    planets = [dict(orbit_kind=0, has_mass=False)]
    ref = gb.gpu_eval(obs, planets, cfg["elems"], None, grad=True)
    a, b = gb.GpuPath(obs, planets), gb.GpuPath(obs, planets)
    el = np.ascontiguousarray(cfg["elems"]); W = el.shape[1]
    outs = [(np.full(W, np.nan), np.full_like(el, np.nan)) for _ in range(2)]
    for p, (ll, g) in zip((a, b), outs):
        assert lib.octo_eval_begin(p.ctx, p.ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None) == 0
    assert lib.octo_eval_begin(a.ctx, a.ds, capi._dptr(el), None, W, W, capi._dptr(outs[0][0]), None, None) == capi.OCTO_EINVAL   # one outstanding begin per context
    # a blocking octo_eval in between is refused too — and must leave the outstanding evaluation intact (its error path
    # cleared `pending`, so the later octo_eval_end returned OK without waiting or copying the results out)
    scratch_ll = np.full(W, np.nan)
    assert lib.octo_eval(a.ctx, a.ds, capi._dptr(el), None, W, W, capi._dptr(scratch_ll), None, None) == capi.OCTO_EINVAL
    assert np.isnan(scratch_ll).all()
    for p in (a, b):
        assert lib.octo_eval_end(p.ctx) == 0
    for ll, g in outs:
        assert np.array_equal(ll, ref[0]) and np.array_equal(g, ref[1])
    a.close(); b.close()


def test_pt_step_with_rccl_one_rank(pkg):
    """The library's own RCCL path end to end on one GPU: octo_comm_unique_id -> octo_comm_create (one-rank communicator) ->
    octo_pt_step_device (ncclAllGather + swap kernel on torch's current stream) against the NumPy restatement."""
    import torch
    from octofitter_jl_amd.host.tempering import TemperedSwap
    cfg = synth.config_astrom(n_epochs=50, n_walkers=64, seed=33)
    obs_m, planet = synth.to_mirror(pkg, cfg)
    fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
    dev = torch.device("cuda", 0)
    n_temps, chains = 8, 41
    pt = TemperedSwap(fn, n_temps_total=n_temps, n_chains=chains, rank=0, world=1, device=dev, seed=7)
    assert pt.comm == "c_abi"
    pt.create_comm(force_rccl=True)
    rng = np.random.default_rng(3)
    ref = pt.slot2rep.cpu().numpy().copy(); acc = np.zeros(n_temps, dtype=np.int32)
    beta = pt.beta.cpu().numpy()
    for step in range(5):
        ll = rng.normal(-50, 4, (n_temps, chains))
        s2r = pt.swap_step(torch.tensor(ll.reshape(-1), device=dev), step)
        torch.cuda.synchronize()
        ref, a = _pt_swap_reference(ll, beta, ref, step % 2, 7, step); acc += a
        assert np.array_equal(s2r.cpu().numpy(), ref), step
        assert np.array_equal(pt._ll_all.cpu().numpy(), ll.reshape(-1)), "the gathered buffer is the local one on a single rank"
    assert np.array_equal(pt.accepted.cpu().numpy(), acc) and acc.sum() > 0
    assert fn.lib.octo_comm_destroy(fn._ctx) == 0
    fn.close()


def _rank_main(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    from __graft_entry__ import load_package
    pkg = load_package()
    from octofitter_jl_amd.host.tempering import TemperedSwap
    dist.init_process_group("gloo", rank=rank, world_size=world)      # only carries the 128-byte id and the final comparison
    try:
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        cfg = synth.config_astrom(n_epochs=400, n_walkers=4 * 64, seed=34)      # every rank draws the same 8 x 64 replicas, owns half
        obs_m, planet = synth.to_mirror(pkg, cfg)
        fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"], device=rank)
        n_temps, chains = 4 * world, 64
        full = synth.config_astrom(n_epochs=400, n_walkers=n_temps * chains, seed=35)["elems"]
        pt = TemperedSwap(fn, n_temps_total=n_temps, n_chains=chains, rank=rank, world=world, device=dev, seed=11)
        pt.create_comm()
        el = torch.tensor(np.ascontiguousarray(full[:, pt.lo * chains: pt.hi * chains]), device=dev)
        hist = []
        for step in range(4):
            ll = fn.ln_like_device(el, None, grad=False)
            hist.append(pt.swap_step(ll, step).clone())
            el[5] += 2.0
        torch.cuda.synchronize()
        flat = torch.stack(hist).to(torch.int64).cpu()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], x) for x in gathered), "ranks derived different permutations"
        assert int(pt.accepted.sum()) > 0
        fn.lib.octo_comm_destroy(fn._ctx); fn.close()
        Path(tmp, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif("_device_count() < 2", reason="needs two GPUs: one process per GPU over RCCL")
def test_pt_step_two_ranks_over_rccl(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_host_threads_with_their_own_contexts_share_one_dataset():
    """include/octofitter_hip.h: a dataset may be shared, without locking, by any number of contexts and host threads on the same
    device (the Julia shim keeps a pool of contexts for concurrent callbacks). Four threads, one context each, one dataset: one-θ calls
    (inputs inside the kernel arguments), a mid-size batch (mapped staging) and a big batch (copies), interleaved, each bit-identical to
    the result computed alone beforehand."""
    import threading
    import gpu_binding as gb
    capi = gb.capi
    cfg = synth.config_astrom(n_epochs=300, n_walkers=9000, cfg=3, seed=5)
    t = cfg["table"]
    obs = [dict(kind=0, planet=0, epoch=t["epoch"], y1=t["ra"], y2=t["dec"], s1=t["σ_ra"], s2=t["σ_dec"], cor=None)]
    with gb.GpuPath(obs, [dict(orbit_kind=0, has_mass=False)]) as path:
        lib = path.lib
        shapes = [(0, 1), (1, 1), (100, 300), (0, 9000), (2, 1), (500, 40)]      # (first walker, batch size)
        el_all = np.ascontiguousarray(cfg["elems"])

        def run(ctx, w0, W):
            el = np.ascontiguousarray(el_all[:, w0:w0 + W]); ll = np.empty(W); g = np.empty_like(el)
            st = lib.octo_eval(ctx, path.ds, capi._dptr(el), None, W, W, capi._dptr(ll), capi._dptr(g), None)
            assert st == 0, (lib.octo_last_error(ctx) or b"").decode()
            return ll, g

        ref = [run(path.ctx, w0, W) for w0, W in shapes]
        ctxs = []
        for _ in range(4):
            c = C.c_void_p()
            assert lib.octo_ctx_create(C.byref(c), 0) == 0
            ctxs.append(c)
        errors = []

        def worker(k):
            try:
                for it in range(40):
                    j = (it + k) % len(shapes)
                    ll, g = run(ctxs[k], *shapes[j])
                    if not (np.array_equal(ll, ref[j][0]) and np.array_equal(g, ref[j][1])):
                        errors.append((k, it, j))
            except Exception as ex:      # noqa: BLE001 - reported below
                errors.append((k, repr(ex)))

        th = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for x in th: x.start()
        for x in th: x.join()
        for c in ctxs:
            lib.octo_ctx_destroy(c)
        assert not errors, errors[:5]


def test_lifecycle_returns_every_byte_of_device_memory(pkg):
    """Library-owned device memory (SURVEY §8b "Ownership": device buffers are freed by *_destroy, nothing is handed to the caller): contexts,
    datasets and models created, used on both kernel families with growing batches — so that scratch is outgrown and retired mid-stream — through
    host-buffer, registered-array and device entry points, and destroyed again, 40 times over: the device's free memory ends where it started."""
    import torch
    capi = pkg.capi
    torch.cuda.synchronize()
    rng = np.random.default_rng(77)

    def one_cycle(k):
        cfg = synth.config_astrom(n_epochs=60 + 7 * (k % 5), n_walkers=700 + 300 * (k % 4), seed=100 + k)
        obs, planet = synth.to_mirror(pkg, cfg)
        fn = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
        el = np.ascontiguousarray(cfg["elems"])
        for W in (1, 40, el.shape[1]):                       # fused small-batch launch, mapped staging, throughput kernels; scratch grows
            ll, g, _ = fn.ln_like_arrays(el[:, :W], None, grad=True)
            assert np.isfinite(ll).any()
        ll_h = np.empty(el.shape[1]); g_h = np.empty_like(el)
        fn.host_register(el, ll_h, g_h)
        a_ = (fn._ctx, fn._ds, capi._dptr(el), None, el.shape[1], el.shape[1], capi._dptr(ll_h), capi._dptr(g_h), None)
        assert fn.lib.octo_eval(*a_) == 0 and np.array_equal(ll_h, ll, equal_nan=True)
        fn.host_unregister(el, ll_h, g_h)
        fn.close()
        if k % 4 == 0:                                        # the standard parameterisation on top: model buffers, both routes
            import test_model as tm
            model = pkg.LogDensityModel(tm._reference_test_model(pkg))
            th = model.link(model.sample_priors(rng, 600))
            lp, gr = model.logdensity_and_gradient(th)
            lp1, _ = model.logdensity_and_gradient(th[:, :3])
            assert np.isfinite(lp).any() and np.allclose(lp1[np.isfinite(lp1)], lp[:3][np.isfinite(lp1)], rtol=1e-12, atol=0)
            model.close()

    for k in range(4):                                        # warm-up: the runtime's own pools (code objects, streams, events) settle
        one_cycle(k)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(40):
        one_cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 <= 8 << 20, f"device memory not returned: {(free0 - free1) / 2**20:.1f} MiB after 40 create/use/destroy cycles"


def test_host_swap_step_is_the_device_swap_step(pkg):
    """octo_pt_step (HOST arrays: the communication step of julia/OctofitterHIP.jl: octofit_pigeons_hip, through its executable twin
    TemperedSwap.swap_step_host) against octo_pt_step_device on the same log-likelihoods, seeds and steps: identical label matrices and
    acceptance counts after every step, with and without the one-rank RCCL communicator."""
    import torch
    from octofitter_jl_amd.host.tempering import TemperedSwap
    cfg = synth.config_astrom(n_epochs=50, n_walkers=64, seed=35)
    obs_m, planet = synth.to_mirror(pkg, cfg)
    dev = torch.device("cuda", 0)
    n_temps, chains = 8, 37
    for force_rccl in (False, True):
        fn_d = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
        fn_h = pkg.make_ln_like(pkg.System(name="s", companions=[planet]), cfg["theta_example"])
        pt_d = TemperedSwap(fn_d, n_temps_total=n_temps, n_chains=chains, rank=0, world=1, device=dev, seed=11)
        pt_h = TemperedSwap(fn_h, n_temps_total=n_temps, n_chains=chains, rank=0, world=1, device=dev, seed=11)
        pt_d.create_comm(force_rccl=force_rccl); pt_h.create_comm(force_rccl=force_rccl)
        rng = np.random.default_rng(4)
        for step in range(1, 9):
            ll = rng.normal(0, 3, n_temps * chains)
            s_d = pt_d.swap_step(torch.tensor(ll, device=dev), step).cpu().numpy()
            s_h = pt_h.swap_step_host(ll, step)
            assert np.array_equal(s_d, s_h), (force_rccl, step)
        assert np.array_equal(pt_d.accepted.cpu().numpy(), pt_h.accepted_host) and pt_h.accepted_host.sum() > 0
        fn_d.lib.octo_comm_destroy(fn_d._ctx); fn_h.lib.octo_comm_destroy(fn_h._ctx)
        fn_d.close(); fn_h.close()
