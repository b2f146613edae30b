"""SemanticLoss with the reference's signature (reference: loss/semantic_loss.py:10-39):
mean|gt - pred| + KLDivLoss(reduction='mean')(log(pred + 1e-20), gt + 1e-20) on probability tensors of shape
(T, N, C) (what interfaces/super_resolution.py:372 passes).  The fused training step computes the same quantity inside
tpgsr_softmax_prior_fwd; this module serves callers that hold probabilities (forward and backward are the
tpgsr_semantic_loss_fwd / _bwd kernels: no renormalisation, rows need not sum to 1)."""
import torch
from torch import nn

from .. import kernels as K

_NBLK = 64


class _SemLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        if not (pred.is_cuda and gt.is_cuda):
            raise RuntimeError("tpgsr_amd losses run on the GPU only (no CPU fallback)")
        if pred.shape != gt.shape:
            raise ValueError(f"SemanticLoss: pred {tuple(pred.shape)} vs gt {tuple(gt.shape)}")
        p = pred.detach().contiguous().float()
        q = gt.detach().contiguous().float()
        part = torch.empty(_NBLK, 2, device=pred.device)
        loss = torch.empty((), device=pred.device)
        K.semantic_loss_fwd(p, q, p.numel(), part, _NBLK)
        K.semantic_loss_finalize(part, _NBLK, p.numel(), 1.0, loss)
        ctx.save_for_backward(p, q)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        p, q = ctx.saved_tensors
        dp = torch.empty_like(p)
        K.semantic_loss_bwd(p, q, dloss.contiguous().float().reshape(1), p.numel(), dp)
        return dp, None


class SemanticLoss(nn.Module):
    def __init__(self, margin=0.1):
        super().__init__()
        self.margin = margin
        self.lambda1 = 1.0
        self.lambda2 = 1.0

    def forward(self, pred_vec, gt_vec):
        return _SemLossFn.apply(pred_vec, gt_vec)
