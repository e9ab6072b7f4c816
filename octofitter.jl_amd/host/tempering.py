"""
Parallel-tempering swap step over RCCL — the one collective of the path (BASELINE config 5).

Reference: Pigeons.jl's communication step, reached through ext/OctofitterPigeonsExt/OctofitterPigeonsExt.jl:76-128
(target = model, reference chain = prior-only model :61-67). Replicas keep their states; neighbouring
temperatures exchange their β LABELS with probability min(1, exp((β_i − β_{i+1})(ℓ_{i+1} − ℓ_i))) on the
log-LIKELIHOOD (the reference chain is the prior), so one all_gather of ℓ per step is the whole exchange.

Sharding: temperatures (replicas) are split contiguously over ranks; every rank holds all `n_chains`
independent PT chains for its replicas: local walker index = r_local * n_chains + c.
Every rank runs the same deterministic swap (counter-based RNG keyed by (seed, step, chain, slot)) on the
gathered ℓ, so the permutation never has to be communicated.
"""
from __future__ import annotations

from .sharding import shard_range


class TemperedSwap:
    """comm: how the per-replica log-likelihoods are gathered before the swap.
         "c_abi"  (default on device tensors) octo_pt_step_device: ncclAllGather + swap kernel inside the library, on torch's current
                  stream; `create_comm()` must have been called when world > 1 (rank 0's octo_comm_unique_id is broadcast with
                  torch.distributed, every rank then joins with octo_comm_create).
         "torch"  torch.distributed.all_gather_into_tensor (RCCL through torch, or gloo on CPU tensors), then the swap kernel —
                  the harness the gloo test drives; `gather` / `swap_impl` inject stand-ins for CPU-only tests."""

    def __init__(self, fn, n_temps_total, n_chains, rank=0, world=1, device=None, seed=0, betas=None, group=None, swap_impl=None,
                 comm=None, gather=None):
        import torch
        self.torch = torch
        self.fn, self.rank, self.world, self.group = fn, rank, world, group
        self.n_temps, self.n_chains, self.seed = int(n_temps_total), int(n_chains), int(seed)
        if self.n_temps % world:
            raise ValueError("temperatures must divide evenly over ranks")
        self.lo, self.hi = shard_range(self.n_temps, rank, world)
        self.device = device
        if betas is None:
            betas = torch.linspace(1.0, 0.0, self.n_temps, dtype=torch.float64) ** 3     # geometric-ish ladder, β_0 = 1 (target)
        self.beta = torch.as_tensor(betas, dtype=torch.float64).to(device)
        self.slot2rep = torch.arange(self.n_temps, dtype=torch.int32).repeat(self.n_chains, 1).contiguous().to(device)
        self.accepted = torch.zeros(self.n_temps, dtype=torch.int32, device=device)
        self._ll_all = torch.empty(self.n_temps * self.n_chains, dtype=torch.float64, device=device)
        self._swap_impl = swap_impl       # tests inject a CPU stand-in; the product path is the HIP kernel below
        self._gather = gather             # tests inject a gather; default: torch.distributed
        on_device = device is not None and torch.device(device).type == "cuda"
        self.comm = comm or ("c_abi" if (on_device and swap_impl is None and gather is None) else "torch")
        self._comm_ready = world == 1

    def create_comm(self, force_rccl=False):
        """Join the library's own RCCL communicator (octo_comm_create). Rank 0 draws the 128-byte id, torch.distributed carries it
        to the other ranks (any backend: it is a host-side broadcast of 128 bytes). force_rccl: also for world = 1 (tests)."""
        import ctypes as C
        import torch.distributed as dist
        lib = self.fn.lib
        if self.world == 1 and not force_rccl:
            self.fn._check(lib.octo_comm_create(self.fn._ctx, None, 0, 1), "octo_comm_create")
            self._comm_ready = True
            return
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            st = lib.octo_comm_unique_id(ident)
            if st != 0:
                raise RuntimeError(f"octo_comm_unique_id failed with status {st} (librccl.so.1 not loadable?)")
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=self.group)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self.fn._check(lib.octo_comm_create(self.fn._ctx, ident, self.rank, self.world), "octo_comm_create")
        self._comm_ready = True

    def local_betas(self):
        """β of every local walker under the current label assignment, [n_local_temps * n_chains]."""
        torch = self.torch
        rep2slot = torch.empty_like(self.slot2rep)
        ar = torch.arange(self.n_temps, dtype=torch.int32, device=self.slot2rep.device).repeat(self.n_chains, 1)
        rep2slot.scatter_(1, self.slot2rep.long(), ar)
        b = self.beta[rep2slot[:, self.lo:self.hi].long()]        # [n_chains, n_local]
        return b.t().contiguous().reshape(-1)

    def swap_step_host(self, ll_local, step):
        """The communication step on HOST arrays (octo_pt_step): the executable twin of julia/OctofitterHIP.jl: octofit_pigeons_hip's swap — a driver whose
        replicas live in host memory. ll_local: numpy [n_local_temps * n_chains]; returns slot2rep as numpy [n_chains, n_temps] (kept on the object
        as `slot2rep_host`; `accepted_host` counts acceptances per slot). Same seed, same step -> the same permutation as swap_step on every rank."""
        import ctypes as C
        import numpy as np
        if not self._comm_ready:
            raise RuntimeError("call create_comm() on every rank before the first swap_step_host")
        if not hasattr(self, "slot2rep_host"):
            self.slot2rep_host = np.ascontiguousarray(np.tile(np.arange(self.n_temps, dtype=np.int32), (self.n_chains, 1)))
            self.accepted_host = np.zeros(self.n_temps, dtype=np.int32)
            self.beta_host = np.ascontiguousarray(self.beta.detach().cpu().numpy(), dtype=np.float64)
        ll = np.ascontiguousarray(ll_local, dtype=np.float64)
        i32p = C.POINTER(C.c_int32)
        self.fn._check(self.fn.lib.octo_pt_step(self.fn._ctx, ll.ctypes.data_as(C.POINTER(C.c_double)), self.beta_host.ctypes.data_as(C.POINTER(C.c_double)),
                                                self.slot2rep_host.ctypes.data_as(i32p), self.n_temps, self.n_chains, int(step) % 2, C.c_uint64(self.seed),
                                                C.c_uint64(int(step)), self.accepted_host.ctypes.data_as(i32p)), "octo_pt_step")
        return self.slot2rep_host

    def swap_step(self, ll_local, step):
        """ll_local: [n_local_temps * n_chains] log-likelihoods of this rank's replicas. Returns slot2rep."""
        import ctypes as C
        torch = self.torch
        parity = int(step) % 2
        if self.comm == "c_abi":
            if not ll_local.is_cuda:
                raise RuntimeError("TemperedSwap(comm='c_abi') needs device tensors: gather and swap run inside the HIP library (no CPU fallback)")
            if not self._comm_ready:
                raise RuntimeError("call create_comm() on every rank before the first swap_step")
            stream = torch.cuda.current_stream(ll_local.device).cuda_stream
            ll_local = ll_local.contiguous()
            self.fn._check(self.fn.lib.octo_pt_step_device(
                self.fn._ctx, ll_local.data_ptr(), self._ll_all.data_ptr(), self.beta.data_ptr(), self.slot2rep.data_ptr(), self.n_temps,
                self.n_chains, parity, C.c_uint64(self.seed), C.c_uint64(int(step)), self.accepted.data_ptr(), C.c_void_p(stream)), "octo_pt_step_device")
            self._keep = ll_local      # keep the buffer alive until the stream has consumed it
            return self.slot2rep
        if self._gather is not None:
            ll_all = self._gather(ll_local.contiguous(), self._ll_all)
        elif self.world > 1:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self._ll_all, ll_local.contiguous(), group=self.group)
            ll_all = self._ll_all
        else:
            ll_all = ll_local
        ll_cr = ll_all.view(self.n_temps, self.n_chains)                       # [replica][chain], exactly as gathered
        if self._swap_impl is not None:
            self._swap_impl(ll_cr, self.beta, self.slot2rep, parity, self.seed, int(step), self.accepted)
            return self.slot2rep
        if not ll_cr.is_cuda:
            raise RuntimeError("TemperedSwap needs device tensors: the swap step runs as a HIP kernel (no CPU fallback)")
        stream = torch.cuda.current_stream(ll_cr.device).cuda_stream
        self.fn._check(self.fn.lib.octo_pt_swap_device(
            self.fn._ctx, ll_cr.data_ptr(), self.beta.data_ptr(), self.slot2rep.data_ptr(), self.n_temps, self.n_chains,
            parity, C.c_uint64(self.seed), C.c_uint64(int(step)), self.accepted.data_ptr(), C.c_void_p(stream)), "octo_pt_swap_device")
        self._keep = ll_cr     # keep the buffer alive until the stream has consumed it
        return self.slot2rep
