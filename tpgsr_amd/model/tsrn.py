"""TSRN / TSRN_TL super-resolution networks with the reference's constructor signatures and state_dict layout
(reference: model/tsrn.py:18-78 TSRN, :111-215 TSRN_TL, :373-426 recurrent residual blocks, :464-508 upsampler,
mish, GruBlock), executed by the fused MI355X plan in tpgsr_amd/engine.py.

    net = TSRN(scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=True, hidden_units=32).cuda()
    sr = net(lr)                    # (N, 4, 16, 64) -> (N, 4, 32, 128), autograd-visible
    loss.backward()                 # parameter gradients land in net's flat gradient arena (p.grad are views)

Semantics kept from the reference: STN rectification only in training mode (tsrn.py:64), train-mode BatchNorm with
running-stat updates, tanh output, F.grid_sample default align_corners=False (torch >= 1.3; `grid_align_corners=True`
reproduces the authors' torch 1.2 behaviour).  Not differentiated: the input image (it is data on every call path
of interfaces/super_resolution.py).
"""
import math

import weakref

import torch
from torch import nn

from .. import functional as Fh
from .nn_params import BatchNormParams, Conv2dParams, ConvTranspose2dParams, EngineHolder, GRUParams, PReLUParams, _NoForward
from .stn_head import STNHead
from .tps_spatial_transformer import TPSSpatialTransformer


# The block classes below are executed by the fused plan when they sit inside a TSRN / TSRN_TL; called on their own they run the
# same HIP kernels operator by operator (tpgsr_amd/functional.py), NCHW in / NCHW out like the reference's modules.
class mish(nn.Module):
    """x * tanh(softplus(x)) (reference :480-488)"""

    def __init__(self):
        super().__init__()
        self.activated = True

    def forward(self, x):
        return Fh.mish(x)


class GruBlock(nn.Module):
    """1x1 conv + bidirectional GRU over the last spatial axis (reference :491-508)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        assert out_channels % 2 == 0
        self.conv1 = Conv2dParams(in_channels, out_channels, 1, padding=0)
        self.gru = GRUParams(out_channels, out_channels // 2, bidirectional=True)

    def nhwc(self, x, axis=0):
        """NHWC in / out; axis 1 = the reference's `.transpose(-1, -2)` around the block, without the copies"""
        return self.gru(self.conv1(x), axis)

    def forward(self, x):
        return Fh.to_nchw(self.nhwc(Fh.to_nhwc(x), 0))


class RecurrentResidualBlock(nn.Module):
    """conv-bn-mish-conv-bn, vertical BiGRU, then horizontal BiGRU of (x + residual) (reference :373-394)."""

    def __init__(self, channels):
        super().__init__()
        self.conv1 = Conv2dParams(channels, channels, 3, padding=1)
        self.bn1 = BatchNormParams(channels)
        self.gru1 = GruBlock(channels, channels)
        self.prelu = mish()
        self.conv2 = Conv2dParams(channels, channels, 3, padding=1)
        self.bn2 = BatchNormParams(channels)
        self.gru2 = GruBlock(channels, channels)

    def forward(self, x):
        xh = Fh.to_nhwc(x)
        r = self.bn1(self.conv1(xh), act="mish")
        r = self.bn2(self.conv2(r))
        r = self.gru1.nhwc(r, 1)
        return Fh.to_nchw(self.gru2.nhwc(Fh.add(xh, r), 0))


class RecurrentResidualBlockTL(nn.Module):
    """as above with the text-prior strip concatenated in front of gru1 (reference :397-426)."""

    def __init__(self, channels, text_channels):
        super().__init__()
        self.conv1 = Conv2dParams(channels, channels, 3, padding=1)
        self.bn1 = BatchNormParams(channels)
        self.gru1 = GruBlock(channels + text_channels, channels)
        self.prelu = mish()
        self.conv2 = Conv2dParams(channels, channels, 3, padding=1)
        self.bn2 = BatchNormParams(channels)
        self.gru2 = GruBlock(channels, channels)

    def forward(self, x, text_emb):
        xh = Fh.to_nhwc(x)
        r = self.bn1(self.conv1(xh), act="mish")
        r = self.bn2(self.conv2(r))
        r = self.gru1.nhwc(Fh.cat([r, Fh.to_nhwc(text_emb)]), 1)
        return Fh.to_nchw(self.gru2.nhwc(Fh.add(xh, r), 0))


class UpsampleBLock(nn.Module):
    """conv3x3 C->4C, PixelShuffle(2), mish (reference :464-477)."""

    def __init__(self, in_channels, up_scale):
        super().__init__()
        if up_scale != 2:
            raise NotImplementedError("the pixel-shuffle store of the conv kernel is specialised for up_scale 2 (the reference's)")
        self.conv = Conv2dParams(in_channels, in_channels * up_scale ** 2, 3, padding=1)
        self.pixel_shuffle = _NoForward()          # fused into the conv's store
        self.prelu = mish()

    def forward(self, x):
        return Fh.to_nchw(Fh.mish(self.conv(Fh.to_nhwc(x), out_ps=True)))


class InfoGen(nn.Module):
    """text prior (N,37,1,26) -> (N,32,1,203) by four ConvTranspose2d+BN+ReLU (reference :81-108)."""

    def __init__(self, t_emb, output_size):
        super().__init__()
        self.tconv1 = ConvTranspose2dParams(t_emb, 512, 3, 2, padding=1)
        self.bn1 = BatchNormParams(512)
        self.tconv2 = ConvTranspose2dParams(512, 128, 3, 2, padding=1)
        self.bn2 = BatchNormParams(128)
        self.tconv3 = ConvTranspose2dParams(128, 64, 3, 2, padding=1)
        self.bn3 = BatchNormParams(64)
        self.tconv4 = ConvTranspose2dParams(64, output_size, 3, (2, 1), padding=(1, 0))
        self.bn4 = BatchNormParams(output_size)

    def forward(self, t_embedding):
        x = Fh.to_nhwc(t_embedding)
        x = self.bn1(self.tconv1(x), act="relu")
        x = self.bn2(self.tconv2(x), act="relu")
        x = self.bn3(self.tconv3(x), act="relu")
        x = self.bn4(self.tconv4(x), act="relu")
        return Fh.to_nchw(x)


class _TSRNFunction(torch.autograd.Function):
    """autograd bridge: forward/backward are the recorded HIP plans; parameter gradients are accumulated straight
    into the module's gradient arena (p.grad views), so no per-parameter gradient tensors flow through autograd."""

    @staticmethod
    def forward(ctx, x, anchor, net, prior):
        eng = net._engine()
        eng.bind(x.device)
        ctx.slot, ctx.gen = eng.acquire_slot() if net.training else (0, 0)
        if net.training:
            # a graph that is dropped without a backward pass (a logged loss, an exception) must not keep its workspace slot:
            # release it when autograd destroys this node (release_slot is a no-op once backward has released it)
            weakref.finalize(ctx, eng.release_slot, ctx.slot, ctx.gen)
        sr = eng.forward(x, net.training, prior, slot=ctx.slot)
        ctx.net, ctx.x_shape, ctx.mode = net, tuple(x.shape), net.training
        ctx.save_for_backward(sr)
        ctx.has_prior = prior is not None
        return sr

    @staticmethod
    def backward(ctx, dsr):
        (sr,) = ctx.saved_tensors
        if not ctx.mode:
            raise RuntimeError("backward through an eval-mode TSRN forward is not supported (the reference only "
                               "back-propagates in training mode)")
        eng = ctx.net._engine()
        eng.check_slot(ctx.slot, ctx.gen)
        dprior = eng.backward(ctx.x_shape, sr, dsr, slot=ctx.slot)
        eng.release_slot(ctx.slot, ctx.gen)
        sync = getattr(ctx.net, "_grad_sync", None)   # set by tpgsr_amd.distributed.DataParallel
        if sync is not None:
            sync(eng)
        return None, None, None, (dprior if ctx.has_prior else None)


class _TSRNBase(EngineHolder, nn.Module):
    def _engine(self):
        eng = self.__dict__.get("_eng")
        if eng is None:
            from ..engine import TSRNEngine
            eng = TSRNEngine(self, grid_align_corners=self.grid_align_corners)
            self.__dict__["_eng"] = eng
        return eng

    def _run(self, x, prior=None):
        from .. import kernels as _K
        if not x.is_cuda and not _K.DRYRUN:
            raise RuntimeError("tpgsr_amd.model.tsrn runs on an MI355X only (no CPU / stock-PyTorch fallback); "
                               "call .cuda() on the module and its inputs")
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                                  or (prior is not None and prior.requires_grad))
        if not needs_grad:
            return self._engine().forward(x, self.training, prior)
        anchor = next(p for p in self.parameters() if p.requires_grad) if any(
            p.requires_grad for p in self.parameters()) else None
        return _TSRNFunction.apply(x, anchor, self, prior)

    def state_dict(self, *args, **kwargs):
        eng = self.__dict__.get("_eng")
        if eng is not None:
            eng.flush_counters()
        return super().state_dict(*args, **kwargs)


class TSRN(_TSRNBase):
    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32,
                 grid_align_corners=False):
        super().__init__()
        in_planes = 4 if mask else 3
        assert math.log(scale_factor, 2) % 1 == 0
        upsample_block_num = int(math.log(scale_factor, 2))
        C = 2 * hidden_units
        self.in_planes = in_planes
        self.grid_align_corners = grid_align_corners
        self.block1 = nn.Sequential(Conv2dParams(in_planes, C, 9, padding=4), PReLUParams())
        self.srb_nums = srb_nums
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), RecurrentResidualBlock(C))
        setattr(self, "block%d" % (srb_nums + 2), nn.Sequential(Conv2dParams(C, C, 3, padding=1), BatchNormParams(C)))
        block_ = [UpsampleBLock(C, 2) for _ in range(upsample_block_num)]
        block_.append(Conv2dParams(C, in_planes, 9, padding=4))
        setattr(self, "block%d" % (srb_nums + 3), nn.Sequential(*block_))
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.stn = STN
        if self.stn:
            self.tps = TPSSpatialTransformer(output_image_size=tuple(self.tps_inputsize), num_control_points=20,
                                             margins=(0.05, 0.05))
            self.stn_head = STNHead(in_planes=in_planes, num_ctrlpoints=20, activation="none")

    def forward(self, x):
        return self._run(x)


class TSRN_TL(_TSRNBase):
    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=True, hidden_units=32,
                 word_vec_d=300, text_emb=37, out_text_channels=32, grid_align_corners=False):
        super().__init__()
        in_planes = 4 if mask else 3
        assert math.log(scale_factor, 2) % 1 == 0
        upsample_block_num = int(math.log(scale_factor, 2))
        C = 2 * hidden_units
        self.in_planes = in_planes
        self.grid_align_corners = grid_align_corners
        self.block1 = nn.Sequential(Conv2dParams(in_planes, C, 9, padding=4), PReLUParams())
        self.srb_nums = srb_nums
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), RecurrentResidualBlockTL(C, out_text_channels))
        self.feature_enhancer = None
        self.infoGen = InfoGen(text_emb, out_text_channels)
        self.emb_cls = text_emb
        setattr(self, "block%d" % (srb_nums + 2), nn.Sequential(Conv2dParams(C, C, 3, padding=1), BatchNormParams(C)))
        block_ = [UpsampleBLock(C, 2) for _ in range(upsample_block_num)]
        block_.append(Conv2dParams(C, in_planes, 9, padding=4))
        setattr(self, "block%d" % (srb_nums + 3), nn.Sequential(*block_))
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.stn = STN
        if self.stn:
            self.tps = TPSSpatialTransformer(output_image_size=tuple(self.tps_inputsize), num_control_points=20,
                                             margins=(0.05, 0.05))
            self.stn_head = STNHead(in_planes=in_planes, num_ctrlpoints=20, activation="none",
                                    input_size=self.tps_inputsize)

    def forward(self, x, text_emb=None):
        if text_emb is None:  # reference :191-193 (profiling path): an all-zero prior
            text_emb = torch.zeros(x.shape[0], self.emb_cls, 1, 26, device=x.device)
        return self._run(x, text_emb)
