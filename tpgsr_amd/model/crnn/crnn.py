"""CRNN text recogniser = the text-prior generator (teacher and students) with the reference's constructor and
state_dict layout (reference: model/crnn/crnn.py:5-26 BidirectionalLSTM, :29-90 CRNN), executed by the fused MI355X
plan in tpgsr_amd/engine_crnn.py.

    crnn = CRNN(32, 1, 37, 256).cuda()
    logits = crnn(gray)            # gray (N, 1, 32, 100) -> (T=26, N, 37), seq-first like the reference
"""
import weakref

import torch
from torch import nn

from ..nn_params import BatchNormParams, Conv2dParams, EngineHolder, LinearParams, LSTMParams, _NoForward


class BidirectionalLSTM(_NoForward):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = LSTMParams(nIn, nHidden, bidirectional=True)
        self.embedding = LinearParams(nHidden * 2, nOut)


class _CRNNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gray, anchor, net):
        eng = net._engine()
        eng.bind(gray.device)
        ctx.slot, ctx.gen = eng.acquire_slot() if net.training else (0, 0)
        if net.training:
            # a graph that is dropped without a backward pass (a logged loss, an exception) must not keep its workspace slot:
            # release it when autograd destroys this node (release_slot is a no-op once backward has released it)
            weakref.finalize(ctx, eng.release_slot, ctx.slot, ctx.gen)
        logits = eng.forward(gray, net.training, slot=ctx.slot)
        ctx.net, ctx.mode, ctx.N = net, net.training, gray.shape[0]
        ctx.save_for_backward(gray.contiguous().float())
        ctx.need_dgray = gray.requires_grad
        return logits.permute(1, 0, 2)

    @staticmethod
    def backward(ctx, dlogits_tnc):
        if not ctx.mode:
            raise RuntimeError("backward through an eval-mode CRNN forward is not supported")
        (gray,) = ctx.saved_tensors
        eng = ctx.net._engine()
        eng.check_slot(ctx.slot, ctx.gen)
        dgray = eng.backward(ctx.N, gray, dlogits_tnc.permute(1, 0, 2).contiguous(), ctx.need_dgray, slot=ctx.slot)
        eng.release_slot(ctx.slot, ctx.gen)
        sync = getattr(ctx.net, "_grad_sync", None)   # set by tpgsr_amd.distributed.DataParallel
        if sync is not None:
            sync(eng)
        return dgray, None, None


class CRNN(EngineHolder, nn.Module):
    def __init__(self, imgH, nc, nclass, nh, n_rnn=2, leakyRelu=False):
        super().__init__()
        assert imgH % 16 == 0, "imgH has to be a multiple of 16"
        if leakyRelu or nc != 1 or imgH != 32:
            raise NotImplementedError("the TPGSR path uses CRNN(32, 1, 37, 256) with ReLU (interfaces/base.py:635)")
        ks = [3, 3, 3, 3, 3, 3, 2]
        ps = [1, 1, 1, 1, 1, 1, 0]
        nm = [64, 128, 256, 256, 512, 512, 512]
        cnn = nn.Sequential()
        for i in range(7):
            n_in = nc if i == 0 else nm[i - 1]
            cnn.add_module(f"conv{i}", Conv2dParams(n_in, nm[i], ks[i], padding=ps[i]))
            if i in (2, 4, 6):
                cnn.add_module(f"batchnorm{i}", BatchNormParams(nm[i]))
            cnn.add_module(f"relu{i}", _NoForward())
            if i in (0, 1, 3, 5):
                cnn.add_module(f"pooling{[0, 1, 3, 5].index(i)}", _NoForward())
        self.cnn = cnn
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nclass))

    def _engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            from ...engine_crnn import CRNNEngine
            eng = CRNNEngine(self)
            self.__dict__["_eng"] = eng
        return eng

    def forward(self, input):
        from ... import kernels as _K
        if not input.is_cuda and not _K.DRYRUN:
            raise RuntimeError("tpgsr_amd.model.crnn runs on an MI355X only (no CPU / stock-PyTorch fallback)")
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or input.requires_grad)
        if not needs_grad:
            return self._engine().forward(input, self.training).permute(1, 0, 2)
        anchor = next((p for p in self.parameters() if p.requires_grad), None)
        return _CRNNFunction.apply(input, anchor, self)

    def state_dict(self, *args, **kwargs):
        eng = self.__dict__.get("_eng")
        if eng is not None:
            eng.flush_counters()
        return super().state_dict(*args, **kwargs)
