"""CPU: the C-ABI shared library loads and exports every symbol include/tpgsr_hip.h declares (no kernel is launched)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    from tpgsr_amd import build
    path = build.build()
    import torch  # noqa: F401  (HIP runtime load order, see tpgsr_amd/_lib.py)
    return ctypes.CDLL(path)


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tpgsr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tpgsr_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    from tpgsr_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_struct_layouts(lib):
    from tpgsr_amd import _lib
    lib.tpgsr_sizeof.restype = ctypes.c_int
    for which, st in enumerate((_lib.ConvArgs, _lib.WgradArgs, _lib.PackDesc)):
        assert lib.tpgsr_sizeof(which) == ctypes.sizeof(st), st.__name__


def test_error_reporting_without_gpu(lib):
    lib.tpgsr_last_error.restype = ctypes.c_char_p
    lib.tpgsr_conv_fwd.restype = ctypes.c_int
    rc = lib.tpgsr_conv_fwd(None, None)          # rejected before any launch
    assert rc == -1 and b"null" in lib.tpgsr_last_error()
    assert lib.tpgsr_version() >= 1


def test_product_path_has_no_cpu_fallback():
    import torch
    from tpgsr_amd.model import tsrn
    net = tsrn.TSRN(STN=False, mask=True)
    with pytest.raises(RuntimeError, match="no CPU"):
        net(torch.zeros(1, 4, 16, 64))
    # the product package never imports the oracle
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import sys, tpgsr_amd.engine, tpgsr_amd.interfaces.super_resolution, "
                          "tpgsr_amd.loss.image_loss; print(any(m.startswith('oracle') for m in sys.modules))"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip() == "False", out.stdout + out.stderr
