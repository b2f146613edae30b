"""Fused optimiser over the flat arenas: clip_grad_norm_(0.25) per SR network + one Adam over SR nets and students
(reference: interfaces/super_resolution.py:419-424, interfaces/base.py:449-450).  Three launches per module
(sum-of-squares partials, clip coefficient, Adam) instead of ~4 per parameter tensor."""
from typing import Iterable, Sequence

import torch

from . import kernels as K

_NBLK = 256


class FusedAdam:
    def __init__(self, modules: Sequence[torch.nn.Module], lr=1e-3, betas=(0.5, 0.999), eps=1e-8,
                 clip_modules: Iterable[torch.nn.Module] = (), max_norm=0.25, pool=None):
        self.modules = []
        for m in modules:
            if all(m is not q for q in self.modules):
                self.modules.append(m)
        self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
        self.clip = set(id(m) for m in clip_modules)
        self.state = {}
        self.pool = pool          # engine.ArenaPool: all gradient arenas are slices of one buffer (one memset)
        for m in self.modules:
            frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
            if frozen:
                # the fused step walks the whole flat arena; torch.optim.Adam would skip these tensors
                raise NotImplementedError(
                    f"FusedAdam updates every parameter of a module; {type(m).__name__} has frozen parameters "
                    f"({frozen[0]}, ...): drive it with torch.optim.Adam through the nn.Module API instead")

    def _st(self, m):
        eng = m._engine()
        if eng.device is None:
            dev = next(m.parameters()).device
            eng.bind(dev)
        a = eng.arena
        st = self.state.get(id(m))
        if st is None or st["n"] != a.numel or st["m"].device != a.flat.device:
            dev = a.flat.device
            st = dict(n=a.numel, m=torch.zeros(a.numel, device=dev), v=torch.zeros(a.numel, device=dev),
                      step=torch.zeros(1, dtype=torch.int32, device=dev), part=torch.empty(_NBLK, device=dev),
                      coef=torch.ones(1, device=dev), norm=torch.zeros(1, device=dev))
            self.state[id(m)] = st
        return a, st

    def zero_grad(self):
        if self.pool is not None and self.pool.grad is not None:
            K.zero(self.pool.grad, self.pool.grad.numel())
            return
        for m in self.modules:
            a, _ = self._st(m)
            K.zero(a.grad, a.numel)

    def step(self):
        # These launches run on the step's exposed tail (behind the last gradient): as few as possible.  ALL modules' step counters go up
        # in ONE launch -- the first clipped module's clip-coefficient launch when it comes first, else a launch of their own -- :
        # C3: sumsq, clip + counters, Adam, Adam (round 5: sumsq, clip, counter, Adam, counter, Adam)
        sts = [self._st(m) for m in self.modules]
        counters = [st["step"] for _, st in sts]
        fuse = len(counters) <= 8
        bumped = False
        for m, (a, st) in zip(self.modules, sts):
            gscale = None
            if id(m) in self.clip:
                K.sumsq_partial(a.grad, a.numel, st["part"], _NBLK)
                if fuse and not bumped:
                    K.clip_coef_steps(st["part"], _NBLK, self.max_norm, st["coef"], st["norm"], counters)
                    bumped = True
                else:
                    K.clip_coef(st["part"], _NBLK, self.max_norm, st["coef"], st["norm"])
                gscale = st["coef"]
            elif fuse and not bumped:
                K.clip_coef_steps(None, 0, 0.0, None, None, counters)
                bumped = True
            if not fuse:
                K.step_inc(st["step"])
            K.adam_step(a.flat, a.grad, st["m"], st["v"], a.numel, gscale, self.lr, self.betas[0], self.betas[1], self.eps,
                        st["step"])
            m._engine()._kernel_writes += 1       # the arena was rewritten through a raw pointer: eval-mode plans must re-pack

    def grad_norm(self, m):
        return self.state[id(m)]["norm"]
