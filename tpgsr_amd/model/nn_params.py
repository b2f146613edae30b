"""Parameter-holder modules: the reference's layer objects WITHOUT an ATen forward.

The reference builds its networks from nn.Conv2d / nn.BatchNorm2d / nn.GRU / nn.Linear / nn.PReLU instances; their
only durable contract is the state_dict (names, shapes) and the default initialisation.  These holders register the
same tensors under the same names with the same init.  Inside the fused networks (TSRN / TSRN_TL / CRNN) they are executed by
the recorded plans of tpgsr_amd/engine*.py; called directly they run the same HIP kernels one operator at a time through
tpgsr_amd/functional.py -- on NHWC activations (N, H, W, C): block / network modules convert at their NCHW boundary.
Nothing can silently fall back to a stock PyTorch kernel."""
import math

import torch
from torch import nn


def _uniform(t, bound):
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class EngineHolder:
    """Mixin of the modules that own a lazily built HIP engine (`_eng`: recorded plans, ctypes argument blocks, device workspaces, the flat
    parameter arena).  The engine is process-local state: `copy.deepcopy(module)`, `pickle` and `torch.save(module)` carry the parameters
    and buffers only, and the copy builds its own engine (and arena) on first use.  (Before round 6 all three raised "ctypes objects
    containing pointers cannot be pickled" once the module had run.)"""

    def __getstate__(self):
        eng = self.__dict__.get("_eng")
        if eng is not None and hasattr(eng, "flush_counters"):
            eng.flush_counters()                     # num_batches_tracked is kept lazily by the engine
        d = self.__dict__.copy()
        d.pop("_eng", None)
        d.pop("_grad_sync", None)                    # (tpgsr_amd.distributed.DataParallel's hook: a closure over a process group)
        return d

    def _replicate_for_data_parallel(self):
        # torch.nn.DataParallel over several devices (the reference's interfaces/base.py:394-400 with --ngpu > 1) copies a module's
        # __dict__ per device and threads: the replicas would share ONE engine bound to one device.  Say so instead of computing garbage.
        raise RuntimeError(f"{type(self).__name__}: torch.nn.DataParallel over several GPUs is not supported by tpgsr_amd (one engine = one "
                           "device).  Run one process per GPU (python -m torch.distributed.run ...) and wrap the module in "
                           "tpgsr_amd.distributed.DataParallel, or use TPGSRTrainStep(..., world_size=W): INTEGRATION.md section 4.  "
                           "(torch.nn.DataParallel with ONE device id calls the module directly and works.)")


class _NoForward(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} only holds parameters; it is executed by the fused HIP plan of its "
                           "parent network (tpgsr_amd.engine), not layer by layer")


class Conv2dParams(_NoForward):
    """nn.Conv2d(in, out, k, padding=p) parameters, default PyTorch init (kaiming_uniform(a=sqrt(5)))."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, bias=True):
        super().__init__()
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, (kh, kw)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kh, kw))
        bound = 1.0 / math.sqrt(in_channels * kh * kw)
        _uniform(self.weight, bound)
        if bias:
            self.bias = nn.Parameter(_uniform(torch.empty(out_channels), bound))
        else:
            self.register_parameter("bias", None)

    def forward(self, x, out_ps=False, wscale=1.0):
        """x NHWC -> NHWC (out_ps: store nn.PixelShuffle(2)'s layout directly)"""
        from .. import functional as Fh
        return Fh.conv2d(x, self.weight, self.bias, self.padding, out_ps=out_ps, wscale=wscale)


class ConvTranspose2dParams(_NoForward):
    """nn.ConvTranspose2d(in, out, k, stride, padding, bias=False) parameters ([in][out][kh][kw])."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, bias=False):
        super().__init__()
        assert not bias
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, (k, k)
        self.stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, k, k))
        _uniform(self.weight, 1.0 / math.sqrt(out_channels * k * k))

    def forward(self, x):
        from .. import functional as Fh
        return Fh.conv_transpose2d(x, self.weight, self.stride, self.padding)


class LinearParams(_NoForward):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(_uniform(torch.empty(out_features, in_features), bound))
        self.bias = nn.Parameter(_uniform(torch.empty(out_features), bound))

    def forward(self, x, wscale=1.0):
        from .. import functional as Fh
        return Fh.linear(x, self.weight, self.bias, wscale=wscale)


class BatchNormParams(_NoForward):
    """nn.BatchNorm{1,2}d(C): weight 1, bias 0, running_mean 0, running_var 1, num_batches_tracked 0."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x, act=None):
        """x (..., C) channels-last; act ('relu' | 'mish' | None) is fused into the normalisation pass"""
        from .. import functional as Fh
        return Fh.batch_norm(x, self, self.training, act)


class PReLUParams(_NoForward):
    def __init__(self, init=0.25):
        super().__init__()
        self.weight = nn.Parameter(torch.full((1,), float(init)))

    def forward(self, x):
        from .. import functional as Fh
        return Fh.prelu(x, self.weight)


class _RNNParams(_NoForward):
    GATES = 1

    def __init__(self, input_size, hidden_size, bidirectional=True):
        super().__init__()
        assert bidirectional
        self.input_size, self.hidden_size = input_size, hidden_size
        bound = 1.0 / math.sqrt(hidden_size)
        g = self.GATES * hidden_size
        for suf in ("", "_reverse"):
            self.register_parameter("weight_ih_l0" + suf, nn.Parameter(_uniform(torch.empty(g, input_size), bound)))
            self.register_parameter("weight_hh_l0" + suf, nn.Parameter(_uniform(torch.empty(g, hidden_size), bound)))
            self.register_parameter("bias_ih_l0" + suf, nn.Parameter(_uniform(torch.empty(g), bound)))
            self.register_parameter("bias_hh_l0" + suf, nn.Parameter(_uniform(torch.empty(g), bound)))

    def flatten_parameters(self):  # API parity with nn.GRU / nn.LSTM (model/tsrn.py:503); nothing to do
        pass


class GRUParams(_RNNParams):
    GATES = 3

    def forward(self, x, axis=0):
        """bidirectional GRU along W (axis 0) or H (axis 1) of an NHWC map"""
        from .. import functional as Fh
        return Fh.bigru(x, self, axis)


class LSTMParams(_RNNParams):
    GATES = 4
