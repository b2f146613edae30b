"""tpgsr_amd -- MI355X-native (gfx950 / CDNA4) TPGSR-TSRN training / inference hot path.

Drop-in module tree (same names as the reference): tpgsr_amd.model.tsrn.{TSRN, TSRN_TL}, tpgsr_amd.model.stn_head,
tpgsr_amd.model.tps_spatial_transformer, tpgsr_amd.loss.{image_loss, gradient_loss}, tpgsr_amd.utils.ssim_psnr,
tpgsr_amd.interfaces.super_resolution.  All compute is hand-written HIP behind the C ABI of include/tpgsr_hip.h
(libtpgsr_hip.so); there is no CPU or stock-PyTorch fallback."""
import os as _os

# A train step uses three HIP streams, a gradient exchange adds RCCL's: more than the four hardware queues HIP multiplexes streams onto by
# default.  Two streams sharing a queue serialise, and an event wait of one blocks the other (+0.9 ms per step).
# The runtime reads the variable when it initialises (the first HIP call), so this works as long as the package is imported before the
# process touches the GPU; a job that does so earlier exports GPU_MAX_HW_QUEUES=8 itself.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
