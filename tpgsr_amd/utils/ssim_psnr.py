"""PSNR and SSIM of the evaluation path with the reference's signatures (reference: utils/ssim_psnr.py:9-15 calculate_psnr,
:18-78 gaussian / create_window / SSIM): whole batch, first 3 channels, PSNR on x255 values; SSIM with the 11x11 Gaussian
(sigma 1.5) window, zero padding, size_average.  CUDA tensors go through the tpgsr_psnr / tpgsr_ssim reduction kernels (one
pass each, fp64 combine).  Since round 6 the SSIM module is differentiable with respect to its FIRST argument (the reference's
`--ssim_loss` branch, interfaces/super_resolution.py:388-391: `(1 - ssim(cascade_images, images_hr).mean()) * 10`): its backward is
tpgsr_ssim_bwd (csrc/metrics.hip); the second image (HR) gets no gradient, as in the reference's use.  PSNR stays a pure metric."""
from math import exp

import torch

_NBLK = 256


def _scratch(dev):
    return torch.empty(_NBLK, dtype=torch.float64, device=dev), torch.empty(1, dtype=torch.float32, device=dev)


def calculate_psnr(img1, img2):
    """img1, img2 (N, >=3, H, W) in [0, 1] -> 0-dim tensor (float('inf') for identical batches, like the reference)"""
    from .. import kernels as K
    if not (img1.is_cuda and img2.is_cuda) and not K.DRYRUN:
        raise RuntimeError("tpgsr_amd.utils.ssim_psnr runs on the GPU only (no CPU fallback)")
    if img1.shape != img2.shape:
        raise ValueError(f"calculate_psnr: {tuple(img1.shape)} vs {tuple(img2.shape)}")
    a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
    N, C, H, W = a.shape
    part, out = _scratch(a.device)
    K.psnr(a, b, N, C, H, W, part, _NBLK, out)
    return out[0]


def gaussian(window_size, sigma):
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel=1):
    """utils/ssim_psnr.py:23-27: (channel, 1, window_size, window_size), the same Gaussian taps for every channel"""
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


class SSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        if not size_average:
            raise NotImplementedError("the TPGSR evaluation path uses size_average=True (interfaces/super_resolution.py)")
        self.window_size = window_size
        self.size_average = size_average
        self.register_buffer("window", create_window(window_size, 1)[0, 0].contiguous(), persistent=False)   # the kernel takes the 2-D taps

    def forward(self, img1, img2):
        from .. import kernels as K
        if not (img1.is_cuda and img2.is_cuda) and not K.DRYRUN:
            raise RuntimeError("tpgsr_amd.utils.ssim_psnr runs on the GPU only (no CPU fallback)")
        win = self.window.to(img1.device)
        if torch.is_grad_enabled() and img1.requires_grad:
            return _SSIMFn.apply(img1, img2.detach(), win, self.window_size)
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        N, C, H, W = a.shape
        part, out = _scratch(a.device)
        K.ssim(a, b, win, self.window_size, N, C, H, W, part, _NBLK, out)
        return out[0]


class _SSIMFn(torch.autograd.Function):
    """mean SSIM of the first min(C, 3) channels, differentiable in the first image (tpgsr_ssim / tpgsr_ssim_bwd)"""

    @staticmethod
    def forward(ctx, img1, img2, win, ks):
        from .. import kernels as K
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        N, C, H, W = a.shape
        part, out = _scratch(a.device)
        K.ssim(a, b, win, ks, N, C, H, W, part, _NBLK, out)
        ctx.save_for_backward(a, b, win)
        ctx.ks = ks
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        from .. import kernels as K
        a, b, win = ctx.saved_tensors
        N, C, H, W = a.shape
        cc = min(C, 3)
        gm = torch.empty(3 * N * cc * H * W, dtype=torch.float32, device=a.device)
        da = torch.zeros_like(a)
        coef = g.detach().reshape(1).float().contiguous()
        K.ssim_bwd(a, b, win, ctx.ks, N, C, H, W, gm, coef, 1.0 / (N * cc * H * W), da, False)
        return da, None, None, None
