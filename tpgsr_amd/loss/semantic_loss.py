"""SemanticLoss with the reference's signature (reference: loss/semantic_loss.py:10-39):
mean|gt - pred| + KLDivLoss(reduction='mean')(log(pred + 1e-20), gt + 1e-20) on probability tensors of shape
(T, N, C) (what interfaces/super_resolution.py:372 passes).  The fused training step computes the same quantity inside
tpgsr_softmax_prior_fwd; this module serves callers that hold probabilities."""
import torch
from torch import nn

from .. import kernels as K

_NBLK = 64


class _SemLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        if not pred.is_cuda:
            raise RuntimeError("tpgsr_amd losses run on the GPU only (no CPU fallback)")
        # the kernels work on logits: log(pred) reproduces pred through the softmax when rows sum to 1; rows that do
        # not sum to 1 (not produced on the TPGSR path) are not supported by the fused kernel
        C = pred.shape[-1]
        rows = pred.numel() // C
        logits = torch.log(pred.detach().reshape(rows, C).float().clamp_min(1e-30)).contiguous()
        gtc = gt.detach().reshape(rows, C).float().contiguous()
        p = torch.empty(rows, C, device=pred.device)
        part = torch.empty(_NBLK, 2, device=pred.device)
        loss = torch.empty((), device=pred.device)
        K.softmax_prior_fwd(logits, gtc, rows, 1, C, 0, p, None, part, _NBLK)
        K.semantic_loss_finalize(part, _NBLK, rows * C, 1.0, loss)
        ctx.save_for_backward(p, gtc)
        ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, dloss):
        p, q = ctx.saved_tensors
        count = p.numel()
        diff = q - p
        dp = (-torch.sign(diff) - (q + 1e-20) / (p + 1e-20)) / count      # tiny (T*N*37 elements): host-side torch ops
        return (dp * dloss).reshape(ctx.shape), None


class SemanticLoss(nn.Module):
    def __init__(self, margin=0.1):
        super().__init__()
        self.margin = margin
        self.lambda1 = 1.0
        self.lambda2 = 1.0

    def forward(self, pred_vec, gt_vec):
        return _SemLossFn.apply(pred_vec, gt_vec)
