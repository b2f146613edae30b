"""PSNR, the parity metric (reference: utils/ssim_psnr.py:9-15): whole batch, RGB only, on x255 values."""
import torch


def calculate_psnr(img1, img2):
    mse = ((img1[:, :3, :, :] * 255 - img2[:, :3, :, :] * 255) ** 2).mean()
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(255.0 / torch.sqrt(mse))
