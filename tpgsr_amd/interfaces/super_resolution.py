"""Training-step drivers mirroring interfaces/super_resolution.py:295-424 of the reference (the hot loop body only:
no LMDB loaders, logging, evaluation or checkpoint rotation -- SURVEY.md section 8, rows T1-T3).

`TSRNTrainStep` is config C2 (`--arch tsrn`): sr = model(lr); loss = ImageLoss(sr, hr).mean()*100; zero_grad;
backward; clip_grad_norm_(model, 0.25); Adam(lr 1e-3, betas (0.5, 0.999)).  The whole step is a fixed sequence of
HIP kernel launches on one stream (forward plan, loss, backward plan, optional RCCL all-reduce of the flat gradient
arena, clip + Adam), so it can be captured once into a hipGraph and replayed (`capture()`)."""
import os
from typing import Optional

import torch

from .. import kernels as K
from ..distributed import GradientExchanger, broadcast_state
from ..engine import ArenaPool
from ..optim import FusedAdam

_NBLK = 128
_NBLK_IMG = 1024     # partial rows of the image loss: 786 K elements with two 4-neighbour gradient magnitudes each -- at 128 workgroups the
                     # launch was a 24-iteration latency chain per thread (29 us on the critical path between the SR forward and backward)


def _warn_hw_queues(collective: bool):
    """The step's three streams plus RCCL's are more than the four hardware queues HIP multiplexes streams onto by default: two of them
    share a queue and an event wait of one blocks the other (+0.9 ms per C3 step measured).  The variable is read
    before the process creates its streams: `import tpgsr_amd` sets it to 8 unless the caller chose a value (tpgsr_amd/__init__.py;
    measured effective even after torch.cuda.init(), profiles/r05a_hw_queues_probe.md).  A caller's own smaller value is warned about."""
    try:
        nq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    except ValueError:          # an unparsable value is as good as unset: this is an advisory warning, never a start-up error
        nq = 4
    if collective and not K.DRYRUN and nq < 8:
        import warnings
        warnings.warn("tpgsr_amd: a gradient exchange is on but GPU_MAX_HW_QUEUES is %s (< 8): the train step's streams and RCCL's will "
                      "share hardware queues (~0.9 ms per step on MI355X).  Export GPU_MAX_HW_QUEUES=8 before the process starts "
                      "(INTEGRATION.md section 4)." % os.environ.get("GPU_MAX_HW_QUEUES", "unset (HIP default 4)"), RuntimeWarning, stacklevel=3)


class TSRNTrainStep:
    def __init__(self, model, gradient=True, loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25,
                 process_group=None, world_size: int = 1, force_collectives: bool = False, precision: Optional[str] = None):
        self.model = model
        # arithmetic policy of the step's GEMMs (kernels.py): `precision`, else an explicit TPGSR_CONV_PREC / set_conv_prec, else "x2"
        self.precision = K.train_step_policy(precision)
        self.collective = world_size > 1 or bool(force_collectives)   # force: drive RCCL at world size 1 too (tests)
        _warn_hw_queues(self.collective)
        self.gradient, self.w0, self.w1 = bool(gradient), float(loss_weight[0]), float(loss_weight[1])
        self.pool = ArenaPool([model])
        self.opt = FusedAdam([model], lr=lr, betas=betas, clip_modules=[model], max_norm=max_norm, pool=self.pool)
        self.pg, self.world = process_group, world_size
        self._graph = None
        self._static = None
        self._exch = None

    # -- one step as plain launches ---------------------------------------------------------------------------
    def _buffers(self, lr_img):
        dev = lr_img.device
        if self._static is None or self._static["dev"] != dev or self._static["shape"] != tuple(lr_img.shape):
            N, C, H, W = lr_img.shape
            self._static = dict(dev=dev, shape=tuple(lr_img.shape),
                                part=torch.empty(_NBLK_IMG, 2, device=dev), loss=torch.zeros((), device=dev),
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
        # the loss VALUE is for the caller's log (the gradient below does not read it): its pass over sr / hr and the single-wave finalize run
        # on the auxiliary stream next to the backward pass, not in front of it
        main, aux = K.current_stream(), K.aux_stream(lr_img.device)
        K.order(aux, main)
        with K.stream_ctx(aux):
            K.image_loss_fwd(sr, hr, N, C, H2, W2, self.gradient, st["part"], _NBLK_IMG)
            n_gp = N * min(C, 3) * H2 * W2 if self.gradient else 0
            K.image_loss_finalize(st["part"], _NBLK_IMG, sr.numel(), n_gp, self.w0 * 100.0, self.w1 * 100.0, st["loss"])
        K.image_loss_bwd(sr, hr, st["dloss"], N, C, H2, W2, self.gradient, self.w0, self.w1, st["dsr"])
        eng.backward(tuple(lr_img.shape), sr, st["dsr"])
        K.order(main, aux)
        self.last_sr = sr
        return st["loss"]

    def _exchanger(self):
        if self._exch is None or self._exch.flat.data_ptr() != self.pool.grad.data_ptr():
            inv = self._static["inv_world"]
            self._exch = GradientExchanger(self.pool.grad, [(0, self.pool.grad.numel())], self.pg,
                                           scale_fn=lambda flat, _s: K.scale_(flat, flat.numel(), inv), force=self.collective)
        return self._exch

    def _exchange(self):
        """ONE flat bucket: RCCL all-reduce (sum) of the gradient arena over xGMI, then the 1/world average"""
        if self.collective:
            self._exchanger().finish()

    def _phase_b(self):
        self.opt.step()

    def step(self, lr_img: torch.Tensor, hr_img: torch.Tensor) -> torch.Tensor:
        """Returns the (device) loss scalar = ImageLoss(sr, hr).mean() * 100 of this step (this rank's shard)."""
        if not self.model.training:
            raise RuntimeError("TSRNTrainStep.step needs model.train()")
        with K.policy(self.precision):
            self.pool.bind(lr_img.device)
            loss = self._phase_a(lr_img, hr_img)
            self._exchange()
            self._phase_b()
        return loss

    def broadcast_parameters(self, src: int = 0):
        """DDP start-up: every rank adopts rank `src`'s parameters and BN buffers (one flat broadcast + buffers)."""
        if self.world > 1:
            self.pool.bind(next(self.model.parameters()).device)
            broadcast_state(self.pool.flat, self.model.buffers(), src, self.pg)
            self.model._engine()._kernel_writes += 1      # the arena changed behind the parameters' version counters

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
        with torch.cuda.graph(ga), K.policy(self.precision):
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
        # the captured Adam rewrote the arena behind every version counter: an eval-mode forward after this must re-pack
        self.model._engine().invalidate_packed()
        return self._graph_loss


def parse_crnn_data(imgs_input: torch.Tensor) -> torch.Tensor:
    """TextBase.parse_crnn_data (reference interfaces/base.py:806-829): bicubic resize of the RGB channels to (32, 100)
    and luminance -> (N, 1, 32, 100).  Differentiable (later cascade stages back-propagate into the previous SR)."""
    return _ParseCrnnFn.apply(imgs_input)


class _ParseCrnnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        if not x.is_cuda:
            raise RuntimeError("tpgsr_amd runs on the GPU only (no CPU fallback)")
        x = x.contiguous().float()
        N, C, H, W = x.shape
        out = torch.empty(N, 1, 32, 100, device=x.device)
        K.bicubic_gray_fwd(x, N, C, H, W, 32, 100, out)
        ctx.shape = (N, C, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, C, H, W = ctx.shape
        din = torch.empty(N, C, H, W, device=dout.device)
        K.bicubic_gray_bwd(dout.contiguous().float(), N, C, H, W, 32, 100, din)
        return din


class TPGSRTrainStep:
    """Configs C3-C5 (`--arch tsrn_tl_cascade --use_distill`), interfaces/super_resolution.py:295-406 + :419-424:
    teacher CRNN(HR) -> q (no grad); per stage: student CRNN(prev image) -> softmax p -> distill loss*100 -> prior
    (N,37,1,26) with samples [0, N//4) zeroed -> TSRN_TL(LR, prior) -> image loss*100; sum; backward through every
    stage (the student receives gradient from the distill loss AND through the prior; later stages back-propagate
    through parse_crnn_data into the previous SR image); clip each SR net at 0.25 (students are not clipped); ONE Adam
    over SR nets + students."""

    def __init__(self, sr_models, students, teacher, stu_iter=1, sr_share=True, tpg_share=False, gradient=True,
                 loss_weight=(1.0, 1e-4), lr=1e-3, betas=(0.5, 0.999), max_norm=0.25, process_group=None, world_size=1,
                 force_collectives=False, precision: Optional[str] = None, ssim_loss: bool = False, use_label: bool = False,
                 use_distill: bool = True):
        # `--ssim_loss` (interfaces/super_resolution.py:388-391): every stage adds (1 - ssim(cascade_images, images_hr).mean()) * 10 to its image loss
        self.ssim_loss = bool(ssim_loss)
        # `--use_label` (:347-366): every stage adds mean(CTC(log_softmax(student logits), labels) * weighted_tics) -- step(..., labels=) then
        # takes what the reference's collate hands the loop: (label_vecs (N, 37, 1, L) one-hot, weighted_mask (sum L_n) concatenated indices,
        # weighted_tics (N)).  `--use_distill` (:370-373, the launch scripts' default): the SemanticLoss term against the teacher's prior.
        self.use_label, self.use_distill = bool(use_label), bool(use_distill)
        if not (self.use_label or self.use_distill):
            raise ValueError("TPGSRTrainStep: the cascade branch trains the text-prior generator through --use_distill and / or --use_label")
        # arithmetic policy of the step's GEMMs (kernels.py): `precision`, else an explicit TPGSR_CONV_PREC / set_conv_prec, else "x2" --
        # the benchmarked policy, gated at full size against the oracle on both north_star gates (tests/test_policy_x2*_gpu.py)
        self.precision = K.train_step_policy(precision)
        self.collective = world_size > 1 or bool(force_collectives)   # force: drive RCCL at world size 1 too (tests)
        _warn_hw_queues(self.collective)
        self.sr = list(sr_models) if isinstance(sr_models, (list, tuple)) else [sr_models]
        self.stu = list(students) if isinstance(students, (list, tuple)) else [students]
        self.teacher = teacher
        if hasattr(teacher, "_engine") and all(teacher is not m for m in self.stu):
            teacher._engine().role = "teacher"      # arithmetic policy: its output is a soft target only (kernels.terms_for)
        self.stu_iter, self.sr_share, self.tpg_share = stu_iter, sr_share, tpg_share
        self.gradient, self.w0, self.w1 = bool(gradient), float(loss_weight[0]), float(loss_weight[1])
        mods = self.sr + self.stu
        # one flat parameter / gradient buffer: SR net(s) first, then the students = the order their gradients become final
        self.pool = ArenaPool(mods)
        self.opt = FusedAdam(mods, lr=lr, betas=betas, clip_modules=self.sr, max_norm=max_norm, pool=self.pool)
        self.pg, self.world = process_group, world_size
        self._static = None
        self._graph = None
        self._dbg = {}
        self._exch = None
        self._early_bucket = None          # index of the text-prior generator's early gradient bucket (see _exchanger)
        self._overlap_exchange = True      # False while capturing hipGraphs (the all-reduce stays between the graphs)
        # The SR network's backward plan does not join the weight-gradient stream at its end (the student's backward pass starts right
        # away); _join_side() orders the main stream after it before clip + Adam.  With a gradient exchange the SR bucket is launched
        # FROM the weight-gradient stream (its tail is ordered after every gradient of the SR network: the weight gradients and their
        # reduces run there, the leaf stream is ordered into it, and it is made to wait for the main stream's BatchNorm / PReLU
        # gradients), so the collective starts when the gradients are final and the main stream still does not wait.
        self._defer_join = os.environ.get("TPGSR_DEFER_JOIN", "1") != "0"
        self._sr_pre_side = os.environ.get("TPGSR_SR_PRE_SIDE", "1") != "0"

    def _mark(self, name):
        """diagnostics (tools/lab/step_phases.py): with `self._marks = []` set, an event on the caller's stream at every phase boundary"""
        m = getattr(self, "_marks", None)
        if m is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            m.append((name, ev))

    def _buffers(self, lr_img):
        dev, N = lr_img.device, lr_img.shape[0]
        if self._static is None or self._static["key"] != (dev, tuple(lr_img.shape)):
            _, C, H, W = lr_img.shape
            S = self.stu_iter
            st = dict(key=(dev, tuple(lr_img.shape)), q=torch.empty(N, 26, 37, device=dev),
                      gray_hr=torch.empty(N, 1, 32, 100, device=dev), loss=torch.zeros((), device=dev),
                      dloss=torch.full((1,), 100.0, device=dev), inv_world=torch.full((1,), 1.0 / self.world, device=dev),
                      part_img=[torch.empty(_NBLK_IMG, 2, device=dev) for _ in range(S)],
                      part_sem=[torch.empty(_NBLK, 2, device=dev) for _ in range(S)],
                      l_img=[torch.zeros((), device=dev) for _ in range(S)], l_sem=[torch.zeros((), device=dev) for _ in range(S)],
                      gray=[torch.empty(N, 1, 32, 100, device=dev) for _ in range(S)],
                      p=[torch.empty(N, 26, 37, device=dev) for _ in range(S)],
                      prior=[torch.empty(N, 37, 1, 26, device=dev) for _ in range(S)],
                      dsr=[torch.empty(N, C, 2 * H, 2 * W, device=dev) for _ in range(S)],
                      dcas=torch.empty(N, C, 2 * H, 2 * W, device=dev), dlogits=torch.empty(N, 26, 37, device=dev))
            if self.use_label:
                st.update(ctc_nll=[torch.empty(N, device=dev) for _ in range(S)])
            if self.ssim_loss:
                from ..utils.ssim_psnr import create_window
                cc = min(C, 3)
                st.update(ssim_win=create_window(11, 1)[0, 0].contiguous().to(dev), ssim_part=torch.empty(256, dtype=torch.float64, device=dev),
                          ssim_out=torch.empty(1, device=dev), ssim_gm=torch.empty(3 * N * cc * 4 * H * W, device=dev))
            self._static = st
        return self._static

    def _labels(self, labels, N, dev):
        """(label_vecs, weighted_mask, weighted_tics) of the reference's collate (dataset/dataset.py:1239-1323) -> the CTC kernel's operands:
        text_len as interfaces/super_resolution.py:349-352 derives it from the one-hot tensor, target offsets = its prefix sums"""
        label_vecs, weighted_mask, weighted_tics = labels
        text_sum = label_vecs.sum(1).squeeze(1)                                  # [N, L]
        text_len = (text_sum > 0).float().sum(1).reshape(-1).to(torch.int32)
        if text_len.numel() != N or weighted_tics.numel() != N:
            raise ValueError("labels: label_vecs / weighted_tics must have one entry per image")
        lens = text_len.cpu()
        if int(lens.sum()) != weighted_mask.numel():
            raise ValueError(f"labels: weighted_mask has {weighted_mask.numel()} indices, the label lengths add up to {int(lens.sum())}")
        off = torch.zeros(N, dtype=torch.int32)
        off[1:] = torch.cumsum(lens, 0)[:-1]
        return (weighted_mask.to(torch.int32).to(dev).contiguous(), off.to(dev), text_len.to(dev).contiguous(),
                weighted_tics.float().to(dev).contiguous(), int(lens.max()))

    def _phase_a(self, lr_img, hr_img, labels=None):
        st = self._buffers(lr_img)
        N, C, H, W = lr_img.shape
        ctc = self._labels(labels, N, lr_img.device) if self.use_label else None
        wsem = 100.0 if self.use_distill else 0.0
        H2, W2 = 2 * H, 2 * W
        hr = hr_img.contiguous()
        lr_img = lr_img.contiguous().float()
        if self.collective:
            self._exchanger().begin()
        self._mark("start")
        # teacher on HR (eval mode, no gradient): independent of the student / SR forward until the semantic loss, so it runs
        # on its own stream next to them (interfaces/super_resolution.py:372-382 computes it inline)
        main, aux = K.current_stream(), K.aux_stream(lr_img.device)
        K.order(aux, main)
        with K.stream_ctx(aux):
            # (zero_grad: one memset of the pooled gradient arena -- nothing writes a gradient before the caller's stream has waited for this
            #  stream, below, for the teacher's distribution)
            self.opt.zero_grad()
            K.bicubic_gray_fwd(hr, N, C, H2, W2, 32, 100, st["gray_hr"])
            t_logits = self.teacher._engine().forward(st["gray_hr"], False)
            K.softmax_prior_fwd(t_logits, None, N, 26, 37, 0, st["q"], None, None, _NBLK)
        cascade, ch, cw = lr_img, H, W
        srs, logits_keep = [], []
        # the SR network's prior-independent prologue (operand packing, STN head, rectification, block1: ~0.3 ms of small launches) runs
        # on the weight-gradient stream -- idle during the forward pass -- next to the text-prior generator's forward pass
        pre_side = self._sr_pre_side and not K.DRYRUN
        side = K.side_stream(lr_img.device) if pre_side else None
        for i in range(self.stu_iter):
            stu = self.stu[0 if self.tpg_share else i]
            srm = self.sr[0 if self.sr_share else i]
            K.bicubic_gray_fwd(cascade, N, C, ch, cw, 32, 100, st["gray"][i])

            def sr_prologue(srm=srm, i=i):
                K.order(side, main)
                with K.stream_ctx(side):
                    srm._engine().forward_pre(lr_img, True, slot=i, defer_join=self._defer_join)
            # the generator packs the operands behind its third convolution on the weight-gradient stream too (next to conv0..conv2 on this
            # one) and queues the SR prologue behind that packing
            logits = stu._engine().forward(st["gray"][i], True, slot=i, late_stream=side, after_late=sr_prologue if pre_side else None)
            self._mark(f"student{i} fwd")
            logits_keep.append(logits)
            if i == 0:
                K.order(main, aux)              # the teacher's distribution q is needed from here on
                self._mark("wait teacher")
            K.softmax_prior_fwd(logits, st["q"], N, 26, 37, N // 4, st["p"][i], st["prior"][i], st["part_sem"][i], _NBLK)
            if pre_side:
                K.order(main, side)
                self._mark(f"wait SR prologue{i}")
            sr = srm._engine().forward(lr_img, True, st["prior"][i], slot=i, defer_join=self._defer_join, pre_done=pre_side)
            # The loss VALUE is for the caller's log: no gradient depends on it (image_loss_bwd / softmax_prior_bwd recompute what they
            # need from sr, hr, p, q).  Its launches -- a full pass over sr and hr + two single-wave finalizes per stage -- run on the teacher's
            # stream, idle by now, instead of between the SR network's forward and backward passes (-30 us on the caller's stream per stage);
            # _join_side() orders the caller's stream after it before the step returns
            K.order(aux, main)
            with K.stream_ctx(aux):
                K.semantic_loss_finalize(st["part_sem"][i], _NBLK, N * 26 * 37, wsem, st["l_sem"][i])
                if ctc is not None:      # + mean(ctc * weighted_tics): the per-sample values by the kernel (no gradient here), the mean by ATen
                    K.ctc_loss(logits, 26 * 37, 37, ctc[0], ctc[1], ctc[2], None, N, 26, 37, 0, 0.0, st["ctc_nll"][i], None, False, ctc[4])
                    if not K.DRYRUN:
                        st["l_sem"][i].add_((st["ctc_nll"][i] * ctc[3]).mean())
                K.image_loss_fwd(sr, hr, N, C, H2, W2, self.gradient, st["part_img"][i], _NBLK_IMG)
                n_gp = N * min(C, 3) * H2 * W2 if self.gradient else 0
                K.image_loss_finalize(st["part_img"][i], _NBLK_IMG, sr.numel(), n_gp, self.w0 * 100.0, self.w1 * 100.0, st["l_img"][i])
                if self.ssim_loss:      # loss_img += (1 - ssim.mean()) * 10: the mean by tpgsr_ssim, the scalar arithmetic by three ATen launches
                    K.ssim(sr, hr, st["ssim_win"], 11, N, C, H2, W2, st["ssim_part"], 256, st["ssim_out"])
                    if not K.DRYRUN:
                        st["l_img"][i].add_(st["ssim_out"][0].neg().add_(1.0).mul_(10.0))
            srs.append(sr)
            self._mark(f"SR{i} fwd + loss")
            cascade, ch, cw = sr, H2, W2
        # total loss (device scalar) = sum of the 2*stu_iter scalars (same stream as its addends, same order as before)
        with K.stream_ctx(aux):
            K.copy(st["l_img"][0], st["loss"], 1)
            for i in range(self.stu_iter):
                if i > 0:
                    K.add(st["loss"], st["l_img"][i], 1, st["loss"])
                K.add(st["loss"], st["l_sem"][i], 1, st["loss"])
        # backward, last stage first
        for i in range(self.stu_iter - 1, -1, -1):
            stu = self.stu[0 if self.tpg_share else i]
            srm = self.sr[0 if self.sr_share else i]
            K.image_loss_bwd(srs[i], hr, st["dloss"], N, C, H2, W2, self.gradient, self.w0, self.w1, st["dsr"][i])
            if self.ssim_loss:             # d/d sr of (1 - mean ssim) * 10, added to the first three channels of the image-loss gradient
                K.ssim_bwd(srs[i], hr, st["ssim_win"], 11, N, C, H2, W2, st["ssim_gm"], None, -10.0 / (N * min(C, 3) * H2 * W2), st["dsr"][i], True)
            if i < self.stu_iter - 1:      # gradient arriving through the next stage's parse_crnn_data
                K.add(st["dsr"][i], st["dcas"], st["dsr"][i].numel(), st["dsr"][i])
            dprior = srm._engine().backward(tuple(lr_img.shape), srs[i], st["dsr"][i], slot=i, defer_join=self._defer_join)
            overlap = self.collective and self._overlap_exchange
            if overlap and self._final_stage(srm) == i:
                # every gradient of this SR net is final here (a shared one: after the LAST of its backward passes, stage 0): its bucket
                # travels over xGMI while the text-prior generators' backward passes below run
                self._launch_bucket_from_side(self._bucket("sr", srm), lr_img.device, sr_net=True)
            self._mark(f"SR{i} bwd")
            K.softmax_prior_bwd(st["p"][i], st["q"], dprior, None, N, 26, 37, N // 4, wsem, st["dlogits"], _NBLK)
            if ctc is not None:          # d/d logits of mean(ctc * weighted_tics), added to the softmax's gradient
                K.ctc_loss(logits_keep[i], 26 * 37, 37, ctc[0], ctc[1], ctc[2], ctc[3], N, 26, 37, 0, 1.0 / N, st["ctc_nll"][i], st["dlogits"], True, ctc[4])
            if getattr(self, "_debug", False):
                self._dbg.setdefault("dprior", {})[i] = dprior.clone()
                self._dbg.setdefault("dlogits", {})[i] = st["dlogits"].clone()
            kw = {}
            stu_final = overlap and self._final_stage(stu) == i
            if stu_final and self._bucket("early", stu) is not None:
                # between the two plans of this generator's backward pass everything from conv3 on (95.6 % of it) is final: with several
                # generators (C5) the cascade runs them last stage first, so all but stage 0's bucket hide under the stages that follow
                kw["after_early"] = lambda b=self._bucket("early", stu): self._launch_bucket_from_side(b, lr_img.device)
            dgray = stu._engine().backward(N, st["gray"][i], st["dlogits"], need_dgray=i > 0, slot=i, **kw)
            if stu_final and i > 0:
                # what its second plan produced (conv0..conv2: 1.5 MB) -- final now; stage 0's goes out with finish()
                self._launch_bucket_from_side(self._bucket("rest", stu), lr_img.device)
            if i > 0:
                self._dbg_dgray = dgray
                K.bicubic_gray_bwd(dgray, N, C, H2, W2, 32, 100, st["dcas"])
        self._mark("student bwd")
        self._join_side(lr_img.device)
        self._mark("join side")
        self.last_sr, self.last_p = srs[-1], st["p"][self.stu_iter - 1]
        return st["loss"]

    def _final_stage(self, module) -> int:
        """the cascade stage whose backward pass is the LAST to add to `module`'s gradients (the backward loop runs the stages last to
        first, so that is the lowest stage using it): a shared network is final at stage 0, a per-stage one at its own stage"""
        for i in range(self.stu_iter):
            if module is self.sr[0 if self.sr_share else i] or module is self.stu[0 if self.tpg_share else i]:
                return i
        return 0

    def _bucket(self, kind, module):
        self._exchanger()
        return self._buckets.get((kind, id(module)))

    def _exchanger(self):
        """Buckets of the ONE flat gradient buffer, each launched when its gradients are final (reference: nn.DataParallel's reduction
        of the replicas' gradients, interfaces/base.py:394-400):
          * one per SR network (a shared one after the last of its backward passes);
          * two per text-prior generator whose backward pass is recorded as two plans (CRNNEngine): everything from conv3 to the end of
            its arena (both BiLSTMs, conv6..conv3: 95.6 %) between the plans, its first layers (1.5 MB) after the second."""
        if self._exch is None or self._exch.flat.data_ptr() != self.pool.grad.data_ptr():
            inv = self._static["inv_world"]
            bounds, self._buckets = [], {}
            prev_end = 0
            for m in self.pool.modules:
                a, b = self.pool.ranges[id(m)]
                eng = m._engine()
                off = eng.early_final_offset() if (any(m is q for q in self.stu) and hasattr(eng, "early_final_offset")) else None
                if off:
                    self._buckets[("early", id(m))] = len(bounds)
                    bounds.append((a + off, b))
                    self._buckets[("rest", id(m))] = len(bounds)
                    bounds.append((prev_end, a + off))         # (with the alignment gap in front of the slice: one contiguous cover)
                else:
                    self._buckets[("sr" if any(m is q for q in self.sr) else "rest", id(m))] = len(bounds)
                    bounds.append((prev_end, b))
                prev_end = b
            self._early_bucket = next((v for (k, _), v in self._buckets.items() if k == "early"), None)
            self._exch = GradientExchanger(self.pool.grad, bounds, self.pg,
                                           scale_fn=lambda flat, _s: K.scale_(flat, flat.numel(), inv), force=self.collective)
        return self._exch

    def _launch_bucket_from_side(self, b, device, sr_net=False):
        """launch bucket b's all-reduce FROM the weight-gradient stream: its tail is ordered after the weight gradients and slab reduces
        recorded so far; it is made to wait for this stream's BatchNorm / PReLU-slope gradients; this stream does not wait for anything.
        sr_net: an SR network's bucket -- its STN head's gradients come off the LEAF stream (their slab reduce included), the last of
        the three to finish, so the launch is made from THAT stream, ordered after the other two: the weight-gradient stream, with the
        text-prior generator's weight gradients queued on it, does not wait for the leaf chain"""
        if K.DRYRUN:
            self._exchanger().launch(b)
            return
        # ALWAYS from a stream ordered after the weight-gradient stream, also with TPGSR_DEFER_JOIN=0: the text-prior generator's first
        # backward plan never joins it (K.continue_in), so a launch from this stream could read gradients its weight-gradient /
        # slab-reduce launches are still writing
        side = K.side_stream(device)
        if sr_net:
            aux = K.aux_stream(device)
            K.order(aux, side)
            K.order(aux, K.current_stream())
            with K.stream_ctx(aux):
                self._exchanger().launch(b)
            return
        K.order(side, K.current_stream())
        with K.stream_ctx(side):
            self._exchanger().launch(b)

    def _exchange(self):
        """buckets of ONE flat buffer (SR nets | students, the single student cut in two); all but the last were launched inside the backward pass"""
        if self.collective:
            self._exchanger().finish()

    def _join_side(self, device):
        if K.DRYRUN:
            return
        if self._defer_join:
            K.order(K.current_stream(), K.side_stream(device))
        # the SR network's backward plan leaves its leaf stream (STN head backward + the slab reduce of its weight gradients) unjoined too,
        # and the loss value is computed there (always: with and without the deferred join)
        K.order(K.current_stream(), K.aux_stream(device))

    def _phase_b(self):
        self.opt.step()

    def broadcast_parameters(self, src: int = 0):
        """DDP start-up: every rank adopts rank `src`'s parameters (one flat broadcast) and BN buffers."""
        if self.world > 1:
            dev = next(self.sr[0].parameters()).device
            self.pool.bind(dev)
            self.teacher._engine().bind(dev)
            bufs = [b for m in self.pool.modules + [self.teacher] for b in m.buffers()]
            broadcast_state(self.pool.flat, bufs, src, self.pg)
            broadcast_state(self.teacher._engine().arena.flat, [], src, self.pg)
            for m in self.pool.modules + [self.teacher]:      # the arenas changed behind the parameters' version counters
                m._engine()._kernel_writes += 1

    def step(self, lr_img, hr_img, labels=None):
        for m in self.sr + self.stu:
            if not m.training:
                raise RuntimeError("TPGSRTrainStep.step needs the SR nets and students in train() mode")
        if self.use_label and labels is None:
            raise ValueError("TPGSRTrainStep(use_label=True).step needs labels=(label_vecs, weighted_mask, weighted_tics)")
        with K.policy(self.precision):
            self.pool.bind(lr_img.device)
            self.teacher._engine().bind(lr_img.device)
            loss = self._phase_a(lr_img, hr_img, labels)
            self._exchange()
            self._phase_b()
        self._mark("optimiser")
        return loss

    def capture(self, lr_img, hr_img, warmup=2):
        self._lr, self._hr = lr_img.clone(), hr_img.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.step(self._lr, self._hr)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ga = torch.cuda.CUDAGraph()
        self._overlap_exchange = False       # the all-reduce stays between the two graphs ...
        try:
            with torch.cuda.graph(ga), K.policy(self.precision):
                self._graph_loss = self._phase_a(self._lr, self._hr)
                if self.world == 1:
                    self._phase_b()
        finally:
            self._overlap_exchange = True    # ... and eager step() calls after a capture overlap it with the backward pass again
        self._graph, self._graph_b = ga, None
        if self.world > 1:
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb):
                self._phase_b()
            self._graph_b = gb
        return ga

    def replay(self, lr_img=None, hr_img=None):
        if lr_img is not None:
            self._lr.copy_(lr_img)
        if hr_img is not None:
            self._hr.copy_(hr_img)
        self._graph.replay()
        if self._graph_b is not None:
            self._exchange()
            self._graph_b.replay()
        # the captured Adam rewrote the pooled arena behind every version counter: eval-mode forwards after this must re-pack
        for m in self.pool.modules:
            m._engine().invalidate_packed()
        return self._graph_loss


def parse_aster_data(imgs_input: torch.Tensor, max_len: int = 100):
    """interfaces/base.py:844-864 (fixed-resolution branch): the ASTER recognizer's input dict from (N, C >= 3, H, W) images in [0, 1]:
    RGB planes, bicubic resize to 32 x 128, [0, 1] -> [-1, 1] -- one HIP kernel; 'rec_targets' / 'rec_lengths' are the reference's dummies"""
    if not imgs_input.is_cuda:
        raise RuntimeError("parse_aster_data runs on the GPU only")
    x = imgs_input.contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty(N, 32, 128, 3, device=x.device)
    K.bicubic_resize(x, N, C, 3, H, W, 32, 128, 2.0, -1.0, out)
    from .. import functional as Fh
    return {"images": Fh.to_nchw(out), "rec_targets": torch.ones(N, max_len, dtype=torch.int32), "rec_lengths": [max_len] * N}


def parse_moran_data(imgs_input: torch.Tensor, max_iter: int = 20):
    """interfaces/base.py:608-632 (fixed-resolution branch): the MORAN recognizer's inputs from (N, C >= 3, H, W) images in [0, 1]:
    luminance of the bicubic 32 x 100 resize (one HIP kernel, the same as parse_crnn_data's) and the reference's dummy targets --
    every sample is decoded for max_iter = 20 steps (text = 20 x '0', index 0 of the alphabet).  -> (tensor, length, text, text)"""
    if not imgs_input.is_cuda:
        raise RuntimeError("parse_moran_data runs on the GPU only")
    x = imgs_input.contiguous().float()
    N, C, H, W = x.shape
    gray = torch.empty(N, 1, 32, 100, device=x.device)
    K.bicubic_gray_fwd(x, N, C, H, W, 32, 100, gray)
    text = torch.zeros(N * max_iter, dtype=torch.long)
    length = torch.full((N,), max_iter, dtype=torch.int32)
    return gray, length, text, text


class TextSREvaluator:
    """The evaluation pass of interfaces/super_resolution.py:540-900 for the `tsrn_tl` / cascade architectures, on the HIP
    kernels end to end: eval-mode networks (BatchNorm from running statistics, folded into the consumer convs' loaders; STN
    skipped, model/tsrn.py:183), per stage: parse_crnn_data -> text-prior generator -> softmax -> (N, 37, 1, 26) prior ->
    SR network; then PSNR / SSIM of the last SR image against HR (tpgsr_psnr / tpgsr_ssim) and recognition of LR / SR / HR by
    an evaluation recogniser -- CRNN with on-device CTC greedy decoding, or the ASTER recognizer (model/recognizer) with its greedy
    attention decode -- + string comparison (utils/metrics.py, utils/util.py).
    No dropout of the prior, no gradient state, no host round trip before the strings are built."""

    def __init__(self, sr_models, tpg_models, recognizer=None, stu_iter=1, sr_share=True, tpg_share=False, voc_type="lower"):
        self.sr = list(sr_models) if isinstance(sr_models, (list, tuple)) else [sr_models]
        self.tpg = list(tpg_models) if isinstance(tpg_models, (list, tuple)) else [tpg_models]
        self.recognizer = recognizer if recognizer is not None else self.tpg[0]
        self.stu_iter, self.sr_share, self.tpg_share, self.voc_type = stu_iter, sr_share, tpg_share, voc_type
        from ..utils.ssim_psnr import SSIM
        self._ssim = SSIM()
        self._buf = None

    @torch.no_grad()
    def super_resolve(self, images_lr):
        """-> (list of the stu_iter SR images, list of their (N, 26, 37) text priors)"""
        for m in self.sr + self.tpg:
            if m.training:
                raise RuntimeError("TextSREvaluator needs the networks in eval() mode")
        lr = images_lr.contiguous().float()
        N, C, H, W = lr.shape
        dev = lr.device
        cascade, ch, cw = lr, H, W
        srs, priors = [], []
        for i in range(self.stu_iter):
            tpg = self.tpg[0 if self.tpg_share else i]
            srm = self.sr[0 if self.sr_share else i]
            gray = torch.empty(N, 1, 32, 100, device=dev)
            K.bicubic_gray_fwd(cascade, N, cascade.shape[1], ch, cw, 32, 100, gray)
            logits = tpg._engine().forward(gray, False)                       # [N][T][C]
            p = torch.empty(N, 26, 37, device=dev)
            prior = torch.empty(N, 37, 1, 26, device=dev)
            K.softmax_prior_fwd(logits, None, N, 26, 37, 0, p, prior, None, _NBLK)
            sr = srm._engine().forward(lr, False, prior)
            srs.append(sr)
            priors.append(p)
            cascade, ch, cw = sr, 2 * H, 2 * W
        return srs, priors

    @torch.no_grad()
    def recognize(self, images):
        """evaluation recogniser -> list of strings.  CRNN (`--test_model CRNN`): CTC greedy decoding; an ASTER `RecognizerBuilder`
        (`--test_model ASTER`, interfaces/super_resolution.py:107-135): parse_aster_data + its greedy attention decode; a `MORAN`
        (`--test_model MORAN`): parse_moran_data + rectifier + its left-to-right attention decoder"""
        from ..utils.metrics import get_string_crnn
        from ..model.recognizer import RecognizerBuilder
        if isinstance(self.recognizer, RecognizerBuilder):
            from ..utils.metrics import get_string_aster, get_vocabulary
            out = self.recognizer(parse_aster_data(images, self.recognizer.max_len_labels))["output"]
            return get_string_aster(out["pred_rec"], get_vocabulary("all"))
        from ..model.moran import MORAN
        if isinstance(self.recognizer, MORAN):      # `--test_model MORAN`, interfaces/super_resolution.py:1389-1396
            from ..utils.metrics import get_string_moran
            x, length, text, text_rev = parse_moran_data(images)
            preds = self.recognizer(x, length, text, text_rev, test=True)
            return get_string_moran(preds[0] if isinstance(preds, tuple) else preds, length)
        x = images.contiguous().float()
        N, C, H, W = x.shape
        gray = torch.empty(N, 1, 32, 100, device=x.device)
        K.bicubic_gray_fwd(x, N, C, H, W, 32, 100, gray)
        logits = self.recognizer._engine().forward(gray, False)
        return get_string_crnn(logits.permute(1, 0, 2))

    @torch.no_grad()
    def eval_batch(self, images_lr, images_hr, label_strs=None):
        from ..utils.metrics import str_filt
        from ..utils.ssim_psnr import calculate_psnr
        srs, priors = self.super_resolve(images_lr)
        sr = srs[-1]
        out = dict(images_sr=srs, priors=priors, psnr=calculate_psnr(sr, images_hr), ssim=self._ssim(sr, images_hr),
                   pred_sr=self.recognize(sr), pred_lr=self.recognize(images_lr), pred_hr=self.recognize(images_hr))
        if label_strs is not None:
            tgt = [str_filt(s, self.voc_type) for s in label_strs]
            for k in ("sr", "lr", "hr"):
                out["n_correct_" + k] = sum(str_filt(a, self.voc_type) == b for a, b in zip(out["pred_" + k], tgt))
        return out
