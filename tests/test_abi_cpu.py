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


def test_release_build_has_no_lab_switches(lib):
    """the kernels' debug switches (parts of a kernel switched off for timing: results are garbage with any bit set) exist in LAB builds only
    (-DTPGSR_LAB, tools/lab/*_probe.py): a release library neither exports them nor declares them"""
    if os.environ.get("TPGSR_LAB"):
        pytest.skip("this IS a lab build")
    for name in ("tpgsr_wgh_debug", "tpgsr_gp_debug", "tpgsr_gru_debug"):
        assert not hasattr(lib, name), f"{name} is exported by a release build"
        assert name not in declared_symbols()


def test_binding_table_matches_header():
    from tpgsr_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_struct_layouts(lib):
    from tpgsr_amd import _lib
    lib.tpgsr_sizeof.restype = ctypes.c_int
    for which, st in enumerate((_lib.ConvArgs, _lib.WgradArgs, _lib.PackDesc, _lib.WgradReduceDesc, _lib.ComposeBwdDesc)):
        assert lib.tpgsr_sizeof(which) == ctypes.sizeof(st), st.__name__


def test_error_reporting_without_gpu(lib):
    lib.tpgsr_last_error.restype = ctypes.c_char_p
    lib.tpgsr_conv_fwd.restype = ctypes.c_int
    rc = lib.tpgsr_conv_fwd(None, None)          # rejected before any launch
    assert rc == -1 and b"null" in lib.tpgsr_last_error()
    assert lib.tpgsr_version() >= 1


def test_conv_rejects_operands_past_the_buffer_window(lib):
    """the conv loaders address operands with 32-bit byte offsets in a 2 GiB window: a bigger operand is rejected before any
    launch instead of being read as zeros"""
    from tpgsr_amd import _lib
    lib.tpgsr_last_error.restype = ctypes.c_char_p
    a = _lib.ConvArgs()
    a.in_ = 0x1000
    a.wt = 0x2000
    a.out = 0x3000
    a.N, a.H, a.W, a.Cin, a.in_ld = 4096, 32, 128, 64, 64            # 4 GiB of fp32 activations
    a.Cout, a.KH, a.KW, a.pad_h, a.pad_w, a.OH, a.OW, a.out_ld = 64, 3, 3, 1, 1, 32, 128, 64
    rc = lib.tpgsr_conv_fwd(ctypes.byref(a), None)
    assert rc == -1 and b"2 GiB" in lib.tpgsr_last_error()
    a.N = 1024                                                        # 1 GiB input, but a dy of 1024 x 4096 x 640 floats
    w = _lib.WgradArgs()
    w.c = a
    w.dy, w.part, w.dy_ld = 0x4000, 0x5000, 640
    rc = lib.tpgsr_conv_wgrad(ctypes.byref(w), None)
    assert rc == -1 and b"2 GiB" in lib.tpgsr_last_error()


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


def test_native_plan_records_without_gpu():
    """The plan executor (csrc/plan.cpp) can be built, inspected and rejected arguments reported with no device."""
    from tpgsr_amd import _lib
    lib = _lib.load()
    h = lib.tpgsr_plan_create()
    try:
        args = (_lib.PlanArg * 4)()
        args[0].p, args[1].p, args[2].i, args[3].p = 0x1000, 0x2000, 16, 0x3000
        assert lib.tpgsr_plan_add_launch(h, b"tpgsr_add", args, 4, 0) == 0
        assert lib.tpgsr_plan_add_fork(h) == 1
        ca = _lib.ConvArgs()
        one = (_lib.PlanArg * 1)()
        one[0].p = ctypes.addressof(ca)
        assert lib.tpgsr_plan_add_launch(h, b"tpgsr_conv_fwd", one, 1, 1) == 2
        assert lib.tpgsr_plan_add_join(h) == 3
        assert lib.tpgsr_plan_size(h) == 4
        v = _lib.PlanArg()
        v.p = 0x4000
        assert lib.tpgsr_plan_set_arg(h, 0, 3, ctypes.byref(v)) == 0
        assert lib.tpgsr_plan_set_arg(h, 2, 0, ctypes.byref(v)) < 0          # the copied argument struct is not patchable
        assert lib.tpgsr_plan_add_launch(h, b"tpgsr_add", args, 3, 0) < 0     # wrong arity
        assert b"arguments" in lib.tpgsr_last_error()
        assert lib.tpgsr_plan_add_launch(h, b"tpgsr_version", args, 0, 0) < 0  # not a launch entry point
        assert lib.tpgsr_plan_size(h) == 4
    finally:
        lib.tpgsr_plan_destroy(h)


def test_cu_mask_spec_parsing():
    from tpgsr_amd import kernels as K
    assert K.parse_cu_mask("0xff") == [0xFF]
    assert K.parse_cu_mask("40") == [0xFFFFFFFF, 0xFF]
    assert K.parse_cu_mask("4/16") == [0x00010001, 0x00010001]
    assert K.parse_cu_mask("0x1" + "0" * 16) == [0, 0, 1]


def test_plan_recording_sections_on_cpu():
    """Plan bookkeeping without a device: side sections tag launches, fork/join edges are emitted once per section / plan,
    DynPtr slots are tracked, and the native handle is built lazily (not here)."""
    from tpgsr_amd import kernels as K
    plan = K.Plan("t")
    plan.overlap = True
    with K.recording(plan):
        K._launch("tpgsr_zero", 0x1000, 16)
        with K.side():
            K._launch("tpgsr_zero", K.DynPtr("buf"), 16)
            K._launch("tpgsr_zero", 0x3000, 16)
        K._launch("tpgsr_zero", 0x4000, 16)
        plan.join()
        plan.join()                                   # idempotent: nothing to join any more
    kinds = [(op[0], op[3]) for op in plan.ops]
    assert kinds == [("tpgsr_zero", 0), ("fork", 0), ("tpgsr_zero", 1), ("tpgsr_zero", 1), ("tpgsr_zero", 0), ("join", 0)]
    assert plan.dyn == {"buf": [(2, 0)]} and plan._native is None
    plan.set_ptr("buf", 0x2000)
    assert plan.ops[2][2][0] == 0x2000
    plain = K.Plan("p")                               # overlap off: side() is a no-op
    with K.recording(plain):
        with K.side():
            K._launch("tpgsr_zero", 0x1000, 16)
        plain.join()
    assert [(op[0], op[3]) for op in plain.ops] == [("tpgsr_zero", 0)]


def test_plan_side_batch_on_cpu():
    """side_batch_begin / _end: the side sections in between are held and go out behind ONE fork at the end of the batch, in order,
    DynPtr slots pointing at their final op index; the caller's launches keep their order"""
    from tpgsr_amd import kernels as K
    plan = K.Plan("b")
    plan.overlap = True
    with K.recording(plan):
        K._launch("tpgsr_zero", 0x1000, 16)
        sb = K.side_batch_begin()
        assert sb
        with K.side():
            K._launch("tpgsr_zero", 0x2000, 16)
        K._launch("tpgsr_zero", K.DynPtr("m"), 16)
        with K.side():
            K._launch("tpgsr_zero", K.DynPtr("s"), 16)
        K._launch("tpgsr_zero", 0x5000, 16)
        K.side_batch_end(sb)
        with K.side():                                  # outside a batch: its own fork, as before
            K._launch("tpgsr_zero", 0x6000, 16)
        plan.join()
    kinds = [(op[0], op[3], op[2][0] if op[1] else None) for op in plan.ops]
    assert kinds == [("tpgsr_zero", 0, 0x1000), ("tpgsr_zero", 0, None), ("tpgsr_zero", 0, 0x5000), ("fork", 0, None),
                     ("tpgsr_zero", 1, 0x2000), ("tpgsr_zero", 1, None), ("fork", 0, None), ("tpgsr_zero", 1, 0x6000), ("join", 0, None)]
    assert plan.dyn == {"m": [(1, 0)], "s": [(5, 0)]}
    empty = K.Plan("e")                                 # a batch without side sections leaves nothing behind
    empty.overlap = True
    with K.recording(empty):
        sb = K.side_batch_begin()
        K._launch("tpgsr_zero", 0x1000, 16)
        K.side_batch_end(sb)
        empty.join()
    assert [(op[0], op[3]) for op in empty.ops] == [("tpgsr_zero", 0)]


def test_plan_continue_in_on_cpu():
    """K.continue_in: a pass recorded as two plans (the caller regains control in between).  The first plan leaves its side-stream work
    pending -- no join -- and the second plan's join is emitted even when it has no side section of its own"""
    from tpgsr_amd import kernels as K
    a, b = K.Plan("a"), K.Plan("b")
    a.overlap = b.overlap = True
    with K.recording(a):
        K._launch("tpgsr_zero", 0x1000, 16)
        with K.side():
            K._launch("tpgsr_zero", 0x2000, 16)
        K.continue_in(b)
        K._launch("tpgsr_zero", 0x3000, 16)
        K._REC.join()
    assert [(op[0], op[3]) for op in a.ops] == [("tpgsr_zero", 0), ("fork", 0), ("tpgsr_zero", 1)]
    assert [(op[0], op[3]) for op in b.ops] == [("tpgsr_zero", 0), ("join", 0)]
    c, d = K.Plan("c"), K.Plan("d")                    # nothing pending: no join needed in the second plan
    with K.recording(c):
        K._launch("tpgsr_zero", 0x1000, 16)
        K.continue_in(d)
        K._launch("tpgsr_zero", 0x3000, 16)
        K._REC.join()
    assert [(op[0], op[3]) for op in d.ops] == [("tpgsr_zero", 0)]
    e, f = K.Plan("e"), K.Plan("f")                    # not from inside a side section / a side batch
    e.overlap = True
    with K.recording(e):
        sb = K.side_batch_begin()
        with pytest.raises(AssertionError):
            K.continue_in(f)
        K.side_batch_end(sb)
