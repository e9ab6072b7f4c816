"""
world_size-2 gloo tests (CPU, -m "not gpu") of the multi-GPU plumbing: walker sharding with an all-gather of the
log-likelihoods, and the parallel-tempering swap step (all_gather + deterministic label swap).

The per-rank evaluator and the swap kernel are injected stand-ins here — on a GPU box they are
BatchedLnLike.ln_like_device and octo_pt_swap_device; what is under test is the host logic around them:
partitioning, gather order, determinism across ranks. The HIP swap kernel itself is checked against the same
NumPy restatement in test_gpu_parity.py::test_pt_swap_kernel.
"""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _fake_eval(elems, nuis, grad):
    ll = elems[0] * 2.0 - elems[1]          # element-wise, so slicing the batch cannot change rounding
    if grad:
        return ll, -2 * elems, None
    return ll


def _numpy_swap(ll_cr, beta, slot2rep, parity, seed, step, accepted):
    from test_gpu_parity import _pt_swap_reference
    out, acc = _pt_swap_reference(ll_cr.numpy(), beta.numpy(), slot2rep.numpy(), parity, seed, step)
    slot2rep.copy_(torch.from_numpy(out))
    accepted.add_(torch.from_numpy(acc))


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()
    try:
        # ---- walker sharding: uneven split, gathered ll identical on every rank and equal to the unsharded result
        W = 37
        g = torch.Generator().manual_seed(3)
        elems = torch.randn(9, W, generator=g, dtype=torch.float64)
        sh = pkg.ShardedLnLike(_fake_eval, rank, world)
        ll = sh(elems, None, grad=False, gather=True)
        assert torch.equal(ll, _fake_eval(elems, None, False))
        res = sh(elems, None, grad=True, gather=True)
        sl = sh.local_slice(W)
        assert torch.equal(res[0], _fake_eval(elems, None, False)) and torch.equal(res[1], -2 * elems[:, sl])
        # ---- parallel tempering: 8 temperatures over 2 ranks, 5 chains
        n_temps, n_chains = 8, 5
        pt = pkg.TemperedSwap(None, n_temps_total=n_temps, n_chains=n_chains, rank=rank, world=world, device="cpu",
                              seed=99, swap_impl=_numpy_swap)
        assert (pt.lo, pt.hi) == ((0, 4) if rank == 0 else (4, 8))
        gen = torch.Generator().manual_seed(7)
        history = []
        for step in range(6):
            ll_all = torch.randn(n_temps, n_chains, generator=gen, dtype=torch.float64) * 3      # [replica][chain], same on both ranks
            ll_local = ll_all[pt.lo:pt.hi].reshape(-1).contiguous()
            s2r = pt.swap_step(ll_local, step).clone()
            history.append(s2r)
            assert torch.equal(torch.sort(s2r, dim=1).values, torch.arange(n_temps, dtype=torch.int32).repeat(n_chains, 1))
            b = pt.local_betas().view(pt.hi - pt.lo, n_chains)
            # β of local replica r in chain c = β[slot currently holding r]
            for c in range(n_chains):
                for r in range(pt.lo, pt.hi):
                    slot = int((s2r[c] == r).nonzero()[0])
                    assert b[r - pt.lo, c] == pt.beta[slot]
        # the same host function with an INJECTED gather (what octo_pt_step_device does inside the library on a GPU box)
        calls = []

        def gather(local, out):
            calls.append(local.numel())
            dist.all_gather_into_tensor(out, local)
            return out
        pt2 = pkg.TemperedSwap(None, n_temps_total=n_temps, n_chains=n_chains, rank=rank, world=world, device="cpu",
                               seed=99, swap_impl=_numpy_swap, gather=gather)
        gen2 = torch.Generator().manual_seed(7)
        for step in range(6):
            ll_all = torch.randn(n_temps, n_chains, generator=gen2, dtype=torch.float64) * 3
            s2r = pt2.swap_step(ll_all[pt2.lo:pt2.hi].reshape(-1).contiguous(), step)
            assert torch.equal(s2r, history[step]), "injected gather and torch.distributed gather disagree"
        assert calls == [(n_temps // world) * n_chains] * 6
        assert pt2.comm == "torch" and pkg.TemperedSwap(None, 8, 5, device="cpu", swap_impl=_numpy_swap).comm == "torch"
        # every rank computed the same permutation history
        flat = torch.stack(history).to(torch.int64)
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], x) for x in gathered)
        assert int(pt.accepted.sum()) > 0
        Path(tmp, f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
