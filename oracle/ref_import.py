"""Import the genuine reference (/root/reference) in the BUILD CONTAINER only.

Used solely by ``tests/golden/make_golden.py`` to pin the oracle and to generate the committed
fixtures.  /root/reference does not exist on the GPU box; nothing in tests/-m gpu, smoke() or
bench.py imports this module.  Recipe: SURVEY.md appendix A (stub IPython / torchvision, which
the reference imports but the image lacks)."""
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


def load():
    if not available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    tv = types.ModuleType("torchvision")
    for sub in ("models", "transforms", "datasets"):
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    sys.modules.setdefault("IPython", ip)
    sys.modules.setdefault("torchvision", tv)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    from model import tsrn, srcnn, stn_head, tps_spatial_transformer
    from model.crnn import crnn
    from loss import image_loss, semantic_loss
    from utils import ssim_psnr
    return types.SimpleNamespace(tsrn=tsrn, srcnn=srcnn, stn_head=stn_head, tps=tps_spatial_transformer,
                                 crnn=crnn, image_loss=image_loss, semantic_loss=semantic_loss,
                                 ssim_psnr=ssim_psnr)
