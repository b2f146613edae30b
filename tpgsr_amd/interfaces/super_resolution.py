"""Training-step drivers mirroring interfaces/super_resolution.py:295-424 of the reference (the hot loop body only:
no LMDB loaders, logging, evaluation or checkpoint rotation -- SURVEY.md section 8, rows T1-T3).

`TSRNTrainStep` is config C2 (`--arch tsrn`): sr = model(lr); loss = ImageLoss(sr, hr).mean()*100; zero_grad;
backward; clip_grad_norm_(model, 0.25); Adam(lr 1e-3, betas (0.5, 0.999)).  The whole step is a fixed sequence of
HIP kernel launches on one stream (forward plan, loss, backward plan, optional RCCL all-reduce of the flat gradient
arena, clip + Adam), so it can be captured once into a hipGraph and replayed (`capture()`)."""
from typing import Optional

import torch

from .. import kernels as K
from ..distributed import broadcast_state, exchange_gradients
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

    def _phase_a(self, lr_img, hr_img):
        """zero_grad + forward + loss + backward (everything before the gradient exchange)"""
        model = self.model
        eng = model._engine()
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
        self.last_sr = sr
        return st["loss"]

    def _exchange(self):
        """ONE flat bucket: RCCL all-reduce (sum) of the gradient arena over xGMI; the 1/world average is in phase B"""
        if self.world > 1:
            exchange_gradients(self.model._engine().arena.grad, self.pg)

    def _phase_b(self):
        eng = self.model._engine()
        if self.world > 1:
            K.scale_(eng.arena.grad, eng.arena.numel, self._static["inv_world"])
        self.opt.step()

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        """Returns the (device) loss scalar = ImageLoss(sr, hr).mean() * 100 of this step (this rank's shard)."""
        if not self.model.training:
            raise RuntimeError("TSRNTrainStep.step needs model.train()")
        self.model._engine().bind(lr_img.device)
        loss = self._phase_a(lr_img, hr_img)
        self._exchange()
        self._phase_b()
        return loss

    def broadcast_parameters(self, src: int = 0):
        """DDP start-up: every rank adopts rank `src`'s parameters and BN buffers (one flat broadcast + buffers)."""
        if self.world > 1:
            eng = self.model._engine()
            eng.bind(next(self.model.parameters()).device)
            broadcast_state(eng.arena.flat, self.model.buffers(), src, self.pg)

    # -- hipGraph replay ----------------------------------------------------------------------------------------
    def capture(self, lr_img: torch.Tensor, hr_img: torch.Tensor, warmup: int = 2):
        """Capture the step on static input buffers: one graph for world_size 1; two graphs (before / after the RCCL
        all-reduce, which stays an ordinary stream-ordered call between them) for world_size > 1.
        `replay(lr, hr)` copies new data into the static buffers and launches the graph(s)."""
        self._lr = lr_img.clone()
        self._hr = hr_img.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step(self._lr, self._hr)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self._graph_loss = self._phase_a(self._lr, self._hr)
            if self.world == 1:
                self._phase_b()
        self._graph = ga
        self._graph_b = None
        if self.world > 1:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb):
                self._phase_b()
            self._graph_b = gb
        return ga

    def replay(self, lr_img: Optional[torch.Tensor] = None, hr_img: Optional[torch.Tensor] = None) -> torch.Tensor:
        if lr_img is not None:
            self._lr.copy_(lr_img)
        if hr_img is not None:
            self._hr.copy_(hr_img)
        self._graph.replay()
        if self._graph_b is not None:
            self._exchange()
            self._graph_b.replay()
        return self._graph_loss
