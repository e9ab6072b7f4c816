"""
Multi-GPU layout of the path: one process per GPU, walkers (or temperatures) sharded, dataset replicated.

The likelihood of a walker depends only on that walker's elements and the read-only tables
(src/likelihoods/system.jl:206-241 holds no cross-θ state), so the data path needs NO collective.
The reference's only cross-process step is Pigeons' replica swap (ext/OctofitterPigeonsExt/
OctofitterPigeonsExt.jl:115-126; docs/src/parallel-sampling.md:64-80) — see tempering.py.
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced split of n units: rank r owns [lo, hi). Sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedLnLike:
    """Evaluate a global batch of W walkers with each rank computing its contiguous slice.

    `evaluate(elems_local, nuis_local, grad)` is the per-rank evaluator (BatchedLnLike.ln_like_device on a GPU box).
    gather=True all-gathers the log-likelihoods so every rank sees ll[W] (needed by ensemble moves / tempering);
    gradients stay local to the rank that owns the walker."""

    def __init__(self, evaluate, rank: int, world: int, group=None):
        self.evaluate, self.rank, self.world, self.group = evaluate, rank, world, group

    def local_slice(self, W):
        return slice(*shard_range(W, self.rank, self.world))

    def __call__(self, elems, nuis=None, grad=False, gather=True):
        import torch
        import torch.distributed as dist
        W = elems.shape[1]
        sl = self.local_slice(W)
        res = self.evaluate(elems[:, sl].contiguous(), None if nuis is None else nuis[:, sl].contiguous(), grad)
        ll_local = res[0] if grad else res
        if not gather or self.world == 1:
            return res
        sizes = [shard_range(W, r, self.world) for r in range(self.world)]
        pad = max(hi - lo for lo, hi in sizes)
        buf = torch.zeros(pad, dtype=ll_local.dtype, device=ll_local.device)
        buf[: ll_local.numel()] = ll_local
        out = [torch.empty_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf, group=self.group)
        ll = torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)])
        return (ll, *res[1:]) if grad else ll
