"""tpgsr_amd -- MI355X-native (gfx950 / CDNA4) TPGSR-TSRN training / inference hot path.

Drop-in module tree (same names as the reference): tpgsr_amd.model.tsrn.{TSRN, TSRN_TL}, tpgsr_amd.model.stn_head,
tpgsr_amd.model.tps_spatial_transformer, tpgsr_amd.loss.{image_loss, gradient_loss}, tpgsr_amd.utils.ssim_psnr,
tpgsr_amd.interfaces.super_resolution.  All compute is hand-written HIP behind the C ABI of include/tpgsr_hip.h
(libtpgsr_hip.so); there is no CPU or stock-PyTorch fallback."""
__version__ = "0.1.0"
