"""Training-step drivers mirroring interfaces/super_resolution.py:295-424 of the reference (the hot loop body only:
no LMDB loaders, logging, evaluation or checkpoint rotation -- SURVEY.md section 8, rows T1-T3).

`TSRNTrainStep` is config C2 (`--arch tsrn`): sr = model(lr); loss = ImageLoss(sr, hr).mean()*100; zero_grad;
backward; clip_grad_norm_(model, 0.25); Adam(lr 1e-3, betas (0.5, 0.999)).  The whole step is a fixed sequence of
HIP kernel launches on one stream (forward plan, loss, backward plan, optional RCCL all-reduce of the flat gradient
arena, clip + Adam), so it can be captured once into a hipGraph and replayed (`capture()`)."""
from typing import Optional

import torch

from .. import kernels as K
from ..optim import FusedAdam

_NBLK = 128


class TSRNTrainStep:
    def __init__(self, model, gradient=True, loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25,
                 process_group=None, world_size: int = 1):
        self.model = model
        self.gradient, self.w0, self.w1 = bool(gradient), float(loss_weight[0]), float(loss_weight[1])
        self.opt = FusedAdam([model], lr=lr, betas=betas, clip_modules=[model], max_norm=max_norm)
        self.pg, self.world = process_group, world_size
        self._graph = None
        self._static = None

    # -- one step as plain launches ---------------------------------------------------------------------------
    def _buffers(self, lr_img):
        dev = lr_img.device
        if self._static is None or self._static["dev"] != dev or self._static["shape"] != tuple(lr_img.shape):
            N, C, H, W = lr_img.shape
            self._static = dict(dev=dev, shape=tuple(lr_img.shape),
                                part=torch.empty(_NBLK, 2, device=dev), loss=torch.zeros((), device=dev),
                                dloss=torch.full((1,), 100.0, device=dev), dsr=torch.empty(N, C, 2 * H, 2 * W, device=dev),
                                inv_world=torch.full((1,), 1.0 / self.world, device=dev))
        return self._static

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        """Returns the (device) loss scalar = ImageLoss(sr, hr).mean() * 100 of this step."""
        model = self.model
        if not model.training:
            raise RuntimeError("TSRNTrainStep.step needs model.train()")
        eng = model._engine()
        eng.bind(lr_img.device)
        st = self._buffers(lr_img)
        N, C, H, W = lr_img.shape
        self.opt.zero_grad()
        sr = eng.forward(lr_img, True)
        H2, W2 = 2 * H, 2 * W
        hr = hr_img.contiguous()
        K.image_loss_fwd(sr, hr, N, C, H2, W2, self.gradient, st["part"], _NBLK)
        n_gp = N * min(C, 3) * H2 * W2 if self.gradient else 0
        K.image_loss_finalize(st["part"], _NBLK, sr.numel(), n_gp, self.w0 * 100.0, self.w1 * 100.0, st["loss"])
        K.image_loss_bwd(sr, hr, st["dloss"], N, C, H2, W2, self.gradient, self.w0, self.w1, st["dsr"])
        eng.backward(tuple(lr_img.shape), sr, st["dsr"])
        if self.world > 1:
            # one flat bucket: RCCL all-reduce over xGMI, then the 1/world average
            torch.distributed.all_reduce(eng.arena.grad, group=self.pg)
            K.scale_(eng.arena.grad, eng.arena.numel, st["inv_world"])
        self.opt.step()
        self.last_sr = sr
        return st["loss"]

    # -- hipGraph replay ----------------------------------------------------------------------------------------
    def capture(self, lr_img: torch.Tensor, hr_img: torch.Tensor, warmup: int = 2):
        """Capture one full step on static input buffers; afterwards `replay(lr, hr)` copies new data in and launches
        the graph.  (Collectives are kept outside graphs: with world_size > 1 use step().)"""
        if self.world > 1:
            raise RuntimeError("graph capture is single-process; multi-GPU steps run eagerly around the RCCL call")
        self._lr = lr_img.clone()
        self._hr = hr_img.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step(self._lr, self._hr)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._graph_loss = self.step(self._lr, self._hr)
        self._graph = g
        return g

    def replay(self, lr_img: Optional[torch.Tensor] = None, hr_img: Optional[torch.Tensor] = None) -> torch.Tensor:
        if lr_img is not None:
            self._lr.copy_(lr_img)
        if hr_img is not None:
            self._hr.copy_(hr_img)
        self._graph.replay()
        return self._graph_loss
