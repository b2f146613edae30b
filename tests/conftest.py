import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The parity suite checks the HIP path against the fp32 oracle at fp32 tolerances: it pins the fp32-equivalent policy "x3" (the fused
# train steps would otherwise default to the benchmarked two-term policy "x2", tpgsr_amd/kernels.py); the tests of "x2" / "bf16" / "f32"
# select those explicitly (kernels.set_conv_prec, `precision=`).
os.environ.setdefault("TPGSR_CONV_PREC", "x3")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class _GoldenPolicy:
    """what a reference-fixture test needs to know about the arithmetic policy it runs under"""

    def __init__(self, name):
        self.name = name
        # two-term split operands carry 16 significand bits: 3 * 2^-18 ~ 1.1e-5 per product at worst, ~4e-6 typically, against
        # ~1e-7 of the fp32-equivalent policy.  Forward-value tolerances written for x3 are widened by this factor under x2; the gates that
        # the north star sets (PSNR within 1e-3 dB, identical arg-max priors) and the gradient-NORM tolerances are NOT widened.
        self.fwd = 1.0 if name == "x3" else 8.0

    def tol(self, x3_tolerance):
        return x3_tolerance * self.fwd


@pytest.fixture(params=["x3", "x2"])
def golden_policy(request):
    """The tests against data the REFERENCE produced (tests/golden/model_*.npz, op_*.npz, train_c3.npz) run under the fp32-equivalent
    policy AND under the benchmarked two-term policy (until round 6 `x2` was checked through the oracle only)."""
    from tpgsr_amd import kernels as K
    prev = K.POLICY
    K.set_conv_prec(request.param)
    try:
        yield _GoldenPolicy(request.param)
    finally:
        K.set_conv_prec(prev)
