"""tpgsr_amd -- MI355X-native (gfx950 / CDNA4) TPGSR-TSRN training / inference hot path.

Drop-in module tree (same names as the reference): tpgsr_amd.model.tsrn.{TSRN, TSRN_TL}, tpgsr_amd.model.stn_head,
tpgsr_amd.model.tps_spatial_transformer, tpgsr_amd.loss.{image_loss, gradient_loss}, tpgsr_amd.utils.ssim_psnr,
tpgsr_amd.interfaces.super_resolution.  All compute is hand-written HIP behind the C ABI of include/tpgsr_hip.h
(libtpgsr_hip.so); there is no CPU or stock-PyTorch fallback."""
__version__ = "0.1.0"

import os as _os

# The train step runs on three HIP streams and a gradient exchange adds RCCL's: more than the FOUR hardware queues HIP multiplexes streams
# onto by default -- two streams then share a queue and an event wait of one blocks the other (+0.9 ms per C3 step with collectives,
# DESIGN.md section 6).  The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises (the first HIP call of the process, which
# `import torch` alone does not make), so the package asks for 8 here unless the caller chose a value; a process that has already
# initialised HIP keeps what it has and the train steps warn (interfaces.super_resolution._warn_hw_queues).
import sys as _sys

_t = _sys.modules.get("torch")
# True: HIP was already up when this package was imported and nobody had chosen a value -- the default of 4 queues is in force
HW_QUEUES_LATE = bool(_t is not None and _t.cuda.is_initialized() and "GPU_MAX_HW_QUEUES" not in _os.environ)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
del _t
