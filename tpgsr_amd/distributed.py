"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, over xGMI).

The hot path has exactly ONE exchange per training step: a sum all-reduce of the flat gradient
arena; the 1/world average, the global-norm clip and Adam then run identically on every rank.  Parameters and BN
buffers are broadcast from rank 0 once at start-up.  Device-agnostic on purpose (the CPU/gloo tests exercise it)."""
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def exchange_gradients(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """sum-all-reduce ONE flat bucket in place (no bucketing logic needed: the arena is already contiguous)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def broadcast_state(flat_params: torch.Tensor, buffers: Iterable[torch.Tensor], src: int = 0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src, group=group)
        for b in buffers:
            dist.broadcast(b, src, group=group)


def shard_seed(base_seed: int, rank: Optional[int] = None) -> int:
    """per-rank synthetic-data seed (1234 + rank)"""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return base_seed + rank


class GradientExchanger:
    """Bucketed, overlapped sum-all-reduce of ONE flat gradient buffer.

    The buffer is cut into a few contiguous buckets in the order their gradients become final during the backward pass
    (TPGSR step: the SR network(s) first, then the student recognisers).  `launch(b)` issues bucket b's all-reduce
    asynchronously as soon as the caller's stream has produced it -- RCCL runs it on its own stream while the rest of the
    backward pass keeps the compute stream busy -- and `finish()` orders the caller's stream after every outstanding
    bucket and applies the 1/world average (`scale_fn(flat)`; a HIP kernel on the GPU path, `mul_` in the gloo tests).
    xGMI sizing: MI355X links are point-to-point (7 x ~153 GB/s), a ring all-reduce is per-link bound, so buckets are
    as large as the dependency structure allows (two for C3/C4: 14 MB + 33 MB) rather than DDP's 25 MB default."""

    def __init__(self, flat_grad: torch.Tensor, bounds, group=None, scale_fn=None, force: bool = False):
        """force: issue the collectives at world size 1 too (one rank: the all-reduce is the identity, the average a multiplication
        by 1.0) -- how a one-GPU box drives RCCL's stream semantics through exactly the code path of an 8-GPU job"""
        self.flat, self.group = flat_grad, group
        self.bounds = [(int(a), int(b)) for a, b in bounds]
        self.scale_fn = scale_fn
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force) and dist.is_available() and dist.is_initialized())
        self._work = {}

    def begin(self):
        """start of a step: forget handles a failed step left behind (a step that raised after launch(0) never reached finish())"""
        self._work = {}

    def launch(self, b: int):
        if not self.active:
            return
        if b in self._work:
            raise RuntimeError(f"bucket {b} was already launched in this step")
        lo, hi = self.bounds[b]
        self._work[b] = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """launch whatever has not been launched, wait for everything, average"""
        if not self.active:
            return
        for b in range(len(self.bounds)):
            if b not in self._work:
                self.launch(b)
        for b in sorted(self._work):
            self._work[b].wait()       # stream-ordered on the GPU path (no host block)
        self._work = {}
        if self.scale_fn is not None:
            self.scale_fn(self.flat, 1.0 / self.world)
        else:
            self.flat.mul_(1.0 / self.world)


class DataParallel(torch.nn.Module):
    """Drop-in for the reference's ``torch.nn.DataParallel(model, device_ids=range(ngpu))`` wrapper
    (interfaces/base.py:394-400) in the one-process-per-GPU world: exposes ``.module`` (so ``save_checkpoint``'s
    ``netG.module.state_dict()``, interfaces/base.py:546-585, works unchanged), forwards calls to the wrapped drop-in
    network, adopts rank 0's parameters / BN buffers at construction and averages the module's flat gradient arena
    over the ranks at the end of every backward pass (ONE RCCL all-reduce; the fused modules write all their parameter
    gradients inside a single autograd node, so "bucket ready" == "backward of the module done").  Averaging after each
    backward stays correct when a shared network accumulates several backward passes: the already-averaged part is
    identical on every rank.  With no initialised process group it is a transparent wrapper."""

    def __init__(self, module: torch.nn.Module, process_group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self._inv = None
        # fused networks (TSRN / TSRN_TL / CRNN: one autograd node per network, which calls _grad_sync) vs operator-by-operator ones
        self._fused = callable(getattr(module, "_engine", None)) and getattr(module._engine(), "FUSED", True)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1
        if self._fused:
            module._grad_sync = self._sync          # called by the module's fused autograd node at the end of its backward pass
        else:
            # operator-by-operator networks (SRResNet_TL / RDN_TL / VDSR_TL / SRCNN_TL, ASTER, MORAN: no fused engine, ordinary
            # autograd): average every parameter gradient as soon as autograd has accumulated it, like DDP without buckets
            self._hooks = [p.register_post_accumulate_grad_hook(self._sync_param) for p in module.parameters() if p.requires_grad]
        if broadcast and multi:
            if self._fused:
                dev = next(module.parameters()).device
                eng = module._engine()
                eng.bind(dev)
                broadcast_state(eng.arena.flat, module.buffers(), 0, process_group)
                eng._kernel_writes = getattr(eng, "_kernel_writes", 0) + 1      # eval-mode plans must re-pack (engine.pack_if_stale)
            else:
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, 0, group=process_group)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _sync_param(self, p):
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.process_group)
        if world == 1 or p.grad is None:
            return
        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.process_group)
        p.grad.mul_(1.0 / world)

    def _sync(self, eng):
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(self.process_group)
        if world == 1:
            return
        g = eng.arena.grad
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.process_group)
        if g.is_cuda:
            from . import kernels as K
            if self._inv is None or self._inv.device != g.device:
                self._inv = torch.full((1,), 1.0 / world, device=g.device)
            K.scale_(g, g.numel(), self._inv)
        else:
            g.mul_(1.0 / world)
