"""Engine protocol for a drop-in network written OPERATOR BY OPERATOR over tpgsr_amd.functional (every operator a HIP kernel with a
hand-written backward, torch autograd only chains them) -- and the recorded-plan execution of such a network.

`TPGSRTrainStep` / `FusedAdam` / `ArenaPool` talk to a text-prior generator through its engine (`bind`, `forward`, `backward`, `arena`):
this adapter gives the `--tpg OPT` recogniser (`tpgsr_amd.model.crnn.model.Model`, reference model/crnn/model.py:25-110; selected for the
same training loop by interfaces/super_resolution.py:77-80 / interfaces/base.py:681-756) that interface, so it is a first-class
student / teacher of the fused train step: its parameters live in the pooled flat arena (one gradient-exchange bucket, one fused
clip + Adam).

Two ways to run it:

* RECORDED (default, `TPGSR_OPT_RECORD=0` switches it off): the module's forward and the autograd backward are executed ONCE per
  (batch size, mode, slot) under `kernels.recording` -- every operator's launches land in a `Plan` instead of on the GPU, against
  buffers the plan keeps alive -- and from then on a forward or backward pass is one replay of that plan by the native executor
  (csrc/plan.cpp): no Python per operator, no autograd, weight gradients on the weight-gradient stream with ONE batched slab reduce,
  exactly like `CRNNEngine`'s hand-recorded plans.  What makes the trace complete: parameter gradients are accumulated into the arena
  by the operators' own kernels (`functional.GRAD_SINK`), tensors with two consumers go through `functional.fork` (their gradients are
  summed by tpgsr_add), so autograd never runs an ATen kernel of its own that a replay would miss -- tests/test_opt_recorded_gpu.py
  compares replays on fresh inputs with the operator-by-operator run.
* operator by operator through autograd (the reference implementation of the above; gradients accumulated by autograd into the arena's
  `.grad` views)."""
import os
from typing import Dict, Optional

import torch

from . import functional as Fh
from . import kernels as K
from .engine import ParamArena
from .kernels import Plan, recording


class FunctionalEngine:
    FUSED = False        # no fused autograd node: tpgsr_amd.distributed.DataParallel hooks the parameters instead
    T, IMG_HW = 26, (32, 100)

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.arena = ParamArena(module)
        self.device = None
        self.record = os.environ.get("TPGSR_OPT_RECORD", "1") != "0"
        self._plans: Dict[tuple, dict] = {}      # (N, training, slot, role) -> recorded plans + their static buffers
        self._saved: Dict[int, tuple] = {}
        self._pending_batches = 0
        self._kernel_writes = 0      # bumped by FusedAdam / broadcasts (engine protocol); every plan here re-packs its weights per replay

    def invalidate_packed(self):
        """engine protocol (engine._EngineBase.invalidate_packed): every plan here re-packs its weights per replay, so only the counter moves"""
        self._kernel_writes += 1

    def bind(self, device):
        rebuilt = self.arena.ensure(device)
        if rebuilt or self.device != device:
            for name, b in self.module.named_buffers():
                if b.device != device:
                    raise RuntimeError(f"buffer {name} is on {b.device}, parameters on {device}: call module.to(device) first")
            self._saved.clear()
            self._plans.clear()       # recorded against the old parameter / gradient addresses
        self.device = device

    def flush_counters(self):
        """BatchNorm's num_batches_tracked of the replays since the last flush (a recorded forward does not touch the counters)"""
        if self._pending_batches:
            for name, b in self.module.named_buffers():
                if name.endswith("num_batches_tracked"):
                    b += self._pending_batches
        self._pending_batches = 0

    # ---- recorded mode ------------------------------------------------------------------------------------------------------------
    def plans(self, N: int, training: bool, slot: int = 0) -> dict:
        role = getattr(self, "role", "tpg")
        key = (N, bool(training), slot, role, K.POLICY)
        pl = self._plans.get(key)
        if pl is None:
            pl = self._plans[key] = self._trace(N, training, role)
        return pl

    def _trace(self, N, training, role):
        dev = self.device
        gray = torch.empty(N, 1, *self.IMG_HW, device=dev)
        fwd = Plan("opt_fwd")
        out = dict(fwd=fwd, gray=gray)
        with recording(fwd), K.conv_terms(K.terms_for(role, "fwd")):
            if training:
                x = gray.requires_grad_(True)
                with torch.enable_grad():
                    y = self.module(x)
            else:
                with torch.no_grad():
                    y = self.module(gray)
        logits = y.permute(1, 0, 2)                      # [N][T][C]: the contiguous tensor the prediction layer wrote
        assert logits.is_contiguous() and logits.shape[0] == N
        out["logits"] = logits
        if not training:
            return out
        bwd = Plan("opt_bwd")
        bwd.overlap = os.environ.get("TPGSR_OVERLAP_WGRAD", "1") != "0"
        bwd.deferred = [] if os.environ.get("TPGSR_DEFER_REDUCE", "1") != "0" else None
        dlogits = torch.empty_like(logits)
        self.arena.attach_grads()
        grads = {p.data_ptr(): p.grad for p in self.module.parameters() if p.requires_grad}
        got = []
        h = x.register_hook(lambda g_: got.append(g_))
        prev, Fh.GRAD_SINK = Fh.GRAD_SINK, grads
        try:
            with recording(bwd), K.conv_terms(K.terms_for("tpg", "bwd")):
                torch.autograd.backward(logits, dlogits)
                if bwd.deferred is not None:
                    K.flush_wgrad_reduces()
                K._REC.join()
        finally:
            Fh.GRAD_SINK = prev
            h.remove()
        x.grad = None                                     # (assigned by autograd at trace time; the hook's tensor is the live one)
        out.update(bwd=bwd, dlogits=dlogits, dgray=got[0] if got else None, graph=y)     # `graph` keeps the saved activations alive
        return out

    # ---- the engine protocol --------------------------------------------------------------------------------------------------------
    def forward(self, gray: torch.Tensor, training: bool, slot: int = 0, late_stream=None, after_late=None) -> torch.Tensor:
        """gray (N, 1, 32, 100) -> logits [N][T][nclass] (batch-major, like CRNNEngine.forward; late_stream / after_late: CRNNEngine's
        protocol -- nothing is packed apart here, the callback runs first)"""
        if after_late is not None:
            after_late()
        if bool(self.module.training) != bool(training):
            raise RuntimeError(f"{type(self.module).__name__}: forward(training={training}) on a module in "
                               f"{'train' if self.module.training else 'eval'}() mode")
        self.bind(gray.device)
        if self.record:
            N = gray.shape[0]
            pl = self.plans(N, training, slot)
            K.copy(gray.contiguous(), pl["gray"], gray.numel())
            pl["fwd"].run()
            if training:
                self._pending_batches += 1
                self._saved[slot] = (N, pl)
            res = torch.empty_like(pl["logits"])
            K.copy(pl["logits"], res, res.numel())
            return res
        if not training:
            with torch.no_grad():
                y = self.module(gray)                     # (T, N, C), a permuted view of the contiguous [N][T][C] result
            return y.permute(1, 0, 2).contiguous()
        x = gray.detach().requires_grad_(True)            # later cascade stages ask for d gray
        with torch.enable_grad():
            y = self.module(x)
        logits = y.permute(1, 0, 2)
        self._saved[slot] = (x, logits)
        return logits.detach().contiguous()

    def backward(self, N: int, gray: torch.Tensor, dlogits: torch.Tensor, need_dgray: bool = False, slot: int = 0) -> Optional[torch.Tensor]:
        if slot not in self._saved:
            raise RuntimeError(f"{type(self.module).__name__}: backward without a training-mode forward in slot {slot}")
        if self.record:
            n, pl = self._saved.pop(slot)
            assert n == N
            self.arena.attach_grads()                       # (same addresses as at trace time: the arena does not move)
            K.copy(dlogits.contiguous(), pl["dlogits"], pl["dlogits"].numel())
            pl["bwd"].run()
            if not need_dgray:
                return None
            res = torch.empty_like(pl["dgray"])
            K.copy(pl["dgray"], res, res.numel())
            return res
        x, logits = self._saved.pop(slot)
        self.arena.attach_grads()
        torch.autograd.backward(logits, dlogits.reshape(logits.shape))
        return x.grad if need_dgray else None
