"""Engine protocol for a drop-in network that runs OPERATOR BY OPERATOR over tpgsr_amd.functional instead of a recorded whole-network
plan: every operator is a HIP kernel with a hand-written backward, torch autograd only chains them.

`TPGSRTrainStep` / `FusedAdam` / `ArenaPool` talk to a text-prior generator through its engine (`bind`, `forward`, `backward`, `arena`):
this adapter gives the `--tpg OPT` recogniser (`tpgsr_amd.model.crnn.model.Model`, reference model/crnn/model.py:25-110; selected for the
same training loop by interfaces/super_resolution.py:77-80 / interfaces/base.py:681-756) that interface, so it is a first-class
student / teacher of the fused train step: its parameters live in the pooled flat arena (one gradient-exchange bucket, one fused
clip + Adam), its gradients are accumulated by autograd straight into the arena's `.grad` views."""
from typing import Dict, Optional

import torch

from .engine import ParamArena


class FunctionalEngine:
    FUSED = False        # no fused autograd node: tpgsr_amd.distributed.DataParallel hooks the parameters instead

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.arena = ParamArena(module)
        self.device = None
        self._plans: Dict[tuple, dict] = {}      # (no recorded plans: bench.py's launch census finds nothing here)
        self._saved: Dict[int, tuple] = {}

    def bind(self, device):
        rebuilt = self.arena.ensure(device)
        if rebuilt or self.device != device:
            for name, b in self.module.named_buffers():
                if b.device != device:
                    raise RuntimeError(f"buffer {name} is on {b.device}, parameters on {device}: call module.to(device) first")
            self._saved.clear()
        self.device = device

    def flush_counters(self):
        pass

    def forward(self, gray: torch.Tensor, training: bool, slot: int = 0) -> torch.Tensor:
        """gray (N, 1, 32, 100) -> logits [N][T][nclass] (batch-major, like CRNNEngine.forward)"""
        if bool(self.module.training) != bool(training):
            raise RuntimeError(f"{type(self.module).__name__}: forward(training={training}) on a module in "
                               f"{'train' if self.module.training else 'eval'}() mode")
        self.bind(gray.device)
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
        x, logits = self._saved.pop(slot)
        self.arena.attach_grads()
        torch.autograd.backward(logits, dlogits.reshape(logits.shape))
        return x.grad if need_dgray else None
