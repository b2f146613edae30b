"""PSNR and SSIM of the evaluation path with the reference's signatures (reference: utils/ssim_psnr.py:9-15 calculate_psnr,
:18-78 gaussian / create_window / SSIM): whole batch, first 3 channels, PSNR on x255 values; SSIM with the 11x11 Gaussian
(sigma 1.5) window, zero padding, size_average.  CUDA tensors go through the tpgsr_psnr / tpgsr_ssim reduction kernels (one
pass each, fp64 combine); the metric is a pure function of two image batches, no gradient is defined (the reference never
back-propagates through it)."""
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
        a, b = img1.detach().contiguous().float(), img2.detach().contiguous().float()
        N, C, H, W = a.shape
        win = self.window.to(a.device)
        part, out = _scratch(a.device)
        K.ssim(a, b, win, self.window_size, N, C, H, W, part, _NBLK, out)
        return out[0]
