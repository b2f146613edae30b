"""tpgsr_amd -- MI355X-native (gfx950 / CDNA4) TPGSR-TSRN training / inference hot path.

Drop-in module tree (same names as the reference): tpgsr_amd.model.tsrn.{TSRN, TSRN_TL}, tpgsr_amd.model.stn_head,
tpgsr_amd.model.tps_spatial_transformer, tpgsr_amd.loss.{image_loss, gradient_loss}, tpgsr_amd.utils.ssim_psnr,
tpgsr_amd.interfaces.super_resolution.  All compute is hand-written HIP behind the C ABI of include/tpgsr_hip.h
(libtpgsr_hip.so); there is no CPU or stock-PyTorch fallback."""
__version__ = "0.1.0"

import os as _os

# The train step runs on three HIP streams and a gradient exchange adds RCCL's: more than the FOUR hardware queues HIP multiplexes streams
# onto by default -- two streams then share a queue and an event wait of one blocks the other (C3 with the collectives forced at world
# size 1: 7.39 ms per step on four queues, 6.21 on eight; profiles/r05a_hw_queues_probe.md).  The package asks for eight unless the caller
# chose a value.  Measured on this ROCm (same file): the setting takes effect when made before the process creates its streams -- also
# after `import torch`, and even after torch.cuda.init() -- so importing the package is enough; a trainer need not export anything.
# TPGSR_NO_ENV_DEFAULTS=1: the package leaves the process environment alone (the variable is process-wide: it also applies to torch's and any
# other HIP user's streams; and "effective when set before the streams exist" is a measured property of this ROCm build, not a contract).
if _os.environ.get("TPGSR_NO_ENV_DEFAULTS", "0") != "1":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
