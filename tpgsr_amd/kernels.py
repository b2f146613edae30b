"""Tensor-level wrappers over the C ABI (one Python function per kernel entry point).

All tensors are fp32 CUDA tensors; activations are NHWC-contiguous (shape is informational only: the kernels get
explicit geometry).  Launches go to the current torch stream, so they compose with torch ops and can be captured
into a hipGraph (torch.cuda.graph).  No fallbacks: a failed/rejected launch raises TpgsrKernelError."""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import ConvArgs, GruWgradArgs, WgradArgs, act_code, check


# TPGSR_PLAN_DRYRUN=1: record and validate launch plans WITHOUT a GPU -- every wrapper checks its argument list against the
# C-ABI signature, plans are handed to the native executor (which checks entry point and argument count), workspaces are
# host tensors, and NOTHING is computed (outputs are uninitialised).  It exists so that the plan-recording host logic is
# covered by the CPU test suite (tests/test_plan_dryrun_cpu.py); it is not a fallback and no product entry point enables it.
DRYRUN = os.environ.get("TPGSR_PLAN_DRYRUN") == "1"


# Arithmetic of the MFMA GEMMs (tpgsr_conv_args.terms): 0 = fp32 matrix cores; 3 = fp32-equivalent on the bf16 matrix cores
# (every operand split exactly into three bf16 terms, csrc/conv_xbf.hip); 1 = plain bf16 operands, fp32 accumulate.
# POLICY (TPGSR_CONV_PREC) names what the engines record:
#   "x3"   : (default) split operands everywhere -- fp32-equivalent results (kernel-level error vs fp64 3.5e-7 rms against 4.2e-7
#            for the fp32 matrix cores, tools/lab/xbf_numerics.py), every parity test at its fp32 tolerance, 1.1-1.4x faster
#   "f32"  : fp32 matrix cores everywhere (v_mfma_f32_32x32x2_f32; bit-for-bit an fmaf chain)
#   "bf16" : BASELINE.json's bf16 configurations: bf16 operands in the SR network and in every backward pass, while the
#            FORWARD pass of the text-prior generator (CRNN) stays fp32-equivalent, so the arg-max text priors are identical
#            to the fp32 oracle's by construction; fp32 accumulation, activations, statistics, losses and optimiser throughout
# CONV_TERMS is the value make_conv_args stamps into launches while a plan is recorded / a kernel is called directly.
#   "x2"   : two-term split (3 MFMAs per product block instead of 6, ~16 significand bits per operand: 256x tighter than bf16) in the SR
#            network and in every backward pass; the FORWARD pass of the text-prior generator stays fp32-equivalent like under
#            "bf16" (TPGSR_X2_TPG_FWD=2 lowers it as well)
#   "x3b2" : forward passes fp32-equivalent (x3: SR images, text priors and losses are those of "x3" bit for bit), every BACKWARD GEMM
#            (data and weight gradients) on the two-term split: per product 3 * 2^-18 relative, i.e. the size of the rounding noise an
#            fp32 GEMM's own accumulation order leaves in a gradient (tests/test_policy_x2_gpu.py measures both against fp64)
_TERMS = {"f32": 0, "x3": 3, "bf16": 1, "x2": 2, "x3b2": 3}
_X2_TPG_FWD = int(os.environ.get("TPGSR_X2_TPG_FWD", "3"))
POLICY = os.environ.get("TPGSR_CONV_PREC", "x3")
if POLICY not in _TERMS:
    raise ValueError(f"TPGSR_CONV_PREC={POLICY!r}: expected one of {sorted(_TERMS)}")
CONV_TERMS = _TERMS[POLICY]
# a packed fp32 operand tensor carries its bf16 planes as the attribute `_tpgsr_twin` = (planes tensor, kp): the twin lives and
# dies with the operand (a registry keyed by address would hand a stale twin to the next tensor allocated there)


# Who chose the policy.  The fused TRAIN STEPS (interfaces.super_resolution.TSRNTrainStep / TPGSRTrainStep) default to "x2" -- the policy
# that is benchmarked and gated at full size against the oracle on both north_star gates (|dPSNR| < 1e-3 dB, identical arg-max priors:
# tests/test_policy_x2_gpu.py, test_policy_x2_gates_gpu.py) -- unless somebody chose one: the TPGSR_CONV_PREC variable, set_conv_prec(), or
# the step's own `precision=` argument.  Everything else (module forwards, the functional layer, the evaluators) records "x3" by default.
_POLICY_EXPLICIT = "TPGSR_CONV_PREC" in os.environ
TRAIN_STEP_DEFAULT = "x2"


def set_conv_prec(name: str):
    """'f32' | 'x3' | 'x3b2' | 'x2' | 'bf16' for plans recorded from now on (plans are cached per policy)"""
    global CONV_TERMS, POLICY, _POLICY_EXPLICIT
    if name not in _TERMS:
        raise ValueError(f"set_conv_prec({name!r}): expected one of {sorted(_TERMS)}")
    POLICY, CONV_TERMS, _POLICY_EXPLICIT = name, _TERMS[name], True


def train_step_policy(precision=None) -> str:
    """the arithmetic policy a fused train step records under: its `precision=` argument, else whatever was chosen explicitly
    (TPGSR_CONV_PREC / set_conv_prec), else TRAIN_STEP_DEFAULT"""
    if precision is not None:
        if precision not in _TERMS:
            raise ValueError(f"precision={precision!r}: expected one of {sorted(_TERMS)}")
        return precision
    return POLICY if _POLICY_EXPLICIT else TRAIN_STEP_DEFAULT


class policy:
    """``with policy(name):`` -- plans recorded / launches stamped inside run under `name`; restores the previous policy on exit and
    does not count as an explicit choice"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global CONV_TERMS, POLICY
        self.prev = (POLICY, CONV_TERMS)
        POLICY, CONV_TERMS = self.name, _TERMS[self.name]
        return self

    def __exit__(self, *exc):
        global CONV_TERMS, POLICY
        POLICY, CONV_TERMS = self.prev
        return False


def terms_for(net_kind: str, phase: str) -> int:
    """what an engine records under the current policy; net_kind 'sr' | 'tpg' (text-prior generator) | 'teacher', phase 'fwd' | 'bwd'"""
    if net_kind == "teacher" and POLICY != "x2":
        net_kind = "tpg"
    if POLICY == "bf16":
        return 3 if (net_kind == "tpg" and phase == "fwd") else 1
    if POLICY == "x2":
        # "teacher": the frozen recogniser whose softmax output is only the distillation TARGET q (interfaces/super_resolution.py:372-382):
        # nothing thresholds it, so it runs two-term like the rest; "tpg" = a student, whose arg-max prior must match the fp32 oracle's
        return _X2_TPG_FWD if (net_kind == "tpg" and phase == "fwd") else 2
    if POLICY == "x3b2":
        return 3 if phase == "fwd" else 2
    return _TERMS[POLICY]


class conv_terms:
    """``with conv_terms(t):`` stamps `t` into the launches recorded inside"""

    def __init__(self, terms):
        self.terms = terms

    def __enter__(self):
        global CONV_TERMS
        self.prev, CONV_TERMS = CONV_TERMS, self.terms

    def __exit__(self, *exc):
        global CONV_TERMS
        CONV_TERMS = self.prev
        return False


def register_bf_twin(wt: torch.Tensor, twin: torch.Tensor, kp: int, cin: int = 0):
    wt._tpgsr_twin = (twin, kp, cin)


def block_order_cin(Kd: int, cin: int) -> int:
    """the `cin` a packed [K = taps * cin][N] operand is split with: cin (rows in channel-block order, the halo kernel's) when
    it is a multi-tap operand over a multiple of 32 channels, else 0 (natural order)"""
    return cin if (cin and cin % 32 == 0 and Kd % cin == 0 and Kd // cin > 1) else 0


class _DummyStream:
    cuda_stream = None
    device = torch.device("cpu")

    def wait_stream(self, other):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def current_stream():
    return _DummyStream() if DRYRUN else torch.cuda.current_stream()


def stream_ctx(st):
    return st if DRYRUN else torch.cuda.stream(st)


def _stream():
    return torch.cuda.current_stream().cuda_stream


_SIDE = {}

# Schedule modes (tests / diagnostics; tpgsr_plan_set_mode in include/tpgsr_hip.h): SERIAL folds the side / leaf / teacher streams onto
# the caller's stream -- the whole train step then runs in recording order on ONE stream, the reference the three-stream schedule must
# equal bit for bit; FUZZ delays random streams around every stream edge.
SERIAL = False
FUZZ = False


def set_schedule(serial: bool = False, fuzz_us: int = 0, seed: int = 0, noise_blocks: int = 0):
    """serial: one stream, recording order; fuzz_us > 0: spin kernels of 0..fuzz_us microseconds around every stream edge (seeded);
    noise_blocks > 0: plus a co-running busy kernel of that many workgroups.  set_schedule() restores the default schedule."""
    global SERIAL, FUZZ
    SERIAL, FUZZ = bool(serial), bool(fuzz_us > 0 or noise_blocks > 0) and not serial
    _lib.load().tpgsr_plan_set_mode(int(SERIAL), int(fuzz_us), int(seed), int(noise_blocks))


def order(dst, src):
    """stream `dst` waits for everything enqueued on `src` so far -- every stream edge a train step makes outside a recorded plan goes
    through here, so the schedule modes see it"""
    if SERIAL or DRYRUN or dst is src or getattr(dst, "cuda_stream", 0) == getattr(src, "cuda_stream", 1):
        return
    if FUZZ:
        check(_lib.load().tpgsr_plan_fuzz_point(src.cuda_stream), "tpgsr_plan_fuzz_point")
    dst.wait_stream(src)
    if FUZZ:
        check(_lib.load().tpgsr_plan_fuzz_point(dst.cuda_stream), "tpgsr_plan_fuzz_point")


def event_record(stream):
    """an event at the current tail of `stream` (None under the serial schedule / dry run: program order is the order)"""
    if SERIAL or DRYRUN:
        return None
    if FUZZ:
        check(_lib.load().tpgsr_plan_fuzz_point(stream.cuda_stream), "tpgsr_plan_fuzz_point")
    ev = torch.cuda.Event()
    ev.record(stream)
    return ev


def event_wait(stream, ev):
    """`stream` waits for an event_record() event (and for nothing enqueued behind it on the recording stream)"""
    if ev is None:
        return
    stream.wait_event(ev)
    if FUZZ:
        check(_lib.load().tpgsr_plan_fuzz_point(stream.cuda_stream), "tpgsr_plan_fuzz_point")


def parse_cu_mask(spec: str):
    """'0xffff...': hex bit mask (bit i = CU i); 'N' or 'N/S': N CUs, every S-th (default: the first N)"""
    if spec.lower().startswith("0x"):
        v = int(spec, 16)
    else:
        n, _, stride = spec.partition("/")
        n, stride = int(n), int(stride or 1)
        v = 0
        for i in range(n):
            v |= 1 << (i * stride)
    words = []
    while v:
        words.append(v & 0xFFFFFFFF)
        v >>= 32
    return words or [0]


def side_stream(device=None) -> "torch.cuda.Stream":
    """The per-device second HIP stream weight-gradient launches are recorded on (see Plan.side).
    TPGSR_SIDE_CUMASK confines it to a subset of the compute units (parse_cu_mask)."""
    if SERIAL:
        return torch.cuda.current_stream(device)
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _SIDE.get(idx)
    if st is None:
        spec = os.environ.get("TPGSR_SIDE_CUMASK", "")
        if spec:
            words = parse_cu_mask(spec)
            arr = (C.c_uint * len(words))(*words)
            with torch.cuda.device(idx):
                raw = _lib.load().tpgsr_stream_create(arr, len(words))
            if not raw:
                check(-2, "tpgsr_stream_create")
            st = torch.cuda.ExternalStream(raw, device=idx)
        else:
            # TPGSR_SIDE_PRIORITY (experiment switch): HIP priority of the weight-gradient stream
            st = torch.cuda.Stream(device=idx, priority=int(os.environ.get("TPGSR_SIDE_PRIORITY", "0")))
        _SIDE[idx] = st
    return st


_AUX = {}


def aux_stream(device=None) -> "torch.cuda.Stream":
    """A further per-device stream for host-level overlap of independent sub-networks (the frozen teacher recogniser of the
    text-prior path runs on it next to the student's forward pass)."""
    if DRYRUN:
        return _DummyStream()
    if SERIAL:
        return torch.cuda.current_stream(device)
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _AUX.get(idx)
    if st is None:
        st = _AUX[idx] = torch.cuda.Stream(device=idx)
    return st


class Plan:
    """A recorded, replayable list of kernel launches (static pointers + geometry).  Built once per shape by running
    the wrappers below under ``with recording(plan)``; ``run()`` replays it on the current stream with no Python
    argument marshalling beyond the ctypes call itself -- the host-side analogue of a hipGraph, and capturable into
    one (torch.cuda.graph)."""

    def __init__(self, name=""):
        self.name = name
        self.ops = []      # [name, cfunc, [args...], stream id]; cfunc None: "fork" / "join" / "edge" (args = (src, dst)) stream dependencies
        self.keep = []     # tensors / arg structs referenced by raw pointer
        self.meta = {}     # op index -> what a table-driven launch consists of (bench.py's launch census reads it)
        self.dyn = {}      # key -> [(op index, arg index)]
        self.sid = 0       # stream the next recorded launch goes to: 0 = the caller's stream, 1 = side stream, 2 = leaf stream
        self.forks = 0
        self.leaf_pending = False   # launches on the leaf stream nothing has been ordered after yet
        self.hold = None            # side_batch(): side-stream launches held back until the batch closes [(op, [(key, arg index)])]
        self._native = None  # tpgsr_plan handle, built on first run()
        self._has_side = False
        self._has_leaf = False

    def mark_dynamic(self, key):
        """The NEXT recorded pointer argument equal to the DynPtr placeholder `key` becomes patchable."""
        return DynPtr(key)

    def set_ptr(self, key, ptr):
        for oi, ai in self.dyn[key]:
            self.ops[oi][2][ai] = ptr
            if self._native:
                v = _lib.PlanArg()
                v.p = ptr
                check(_lib.load().tpgsr_plan_set_arg(self._native, oi, ai, C.byref(v)), "tpgsr_plan_set_arg")

    def side(self):
        """``with plan.side():`` -- the launches recorded inside go to the side stream, ordered after everything
        recorded so far (fork) and in order among themselves; ``join()`` orders the caller's stream after them.
        The backward pass puts every weight-gradient GEMM + its slab reduce there: they are leaves of the dependency
        graph (nothing but the optimiser reads them), so they fill the machine while the latency-bound data-gradient
        chain (dgrad -> BN backward -> BiGRU BPTT -> ...) runs on the main stream.  Under torch.cuda.graph capture
        the same calls become fork/join edges of the hipGraph.  Results do not depend on the interleaving: every
        kernel reduces in a fixed order and side launches only read buffers the main stream never rewrites."""
        return _SideCtx(self)

    def leaf(self):
        """``with plan.leaf():`` -- the launches recorded inside go to the LEAF stream (a third stream), ordered after everything
        recorded so far on the caller's stream; side() sections inside it stay on the leaf stream (in order).  For a chain of
        launches that only produces parameter gradients (the STN head's backward): it runs next to whatever the caller's stream
        does afterwards.  leaf_to_side() orders the side stream after it (before the batched slab reduce, which reads its slabs)."""
        return _LeafCtx(self)

    def leaf_to_side(self):
        if self.leaf_pending:
            self.ops.append(["edge", None, (2, 1), 0])
            self.leaf_pending = False
            self.forks += 1          # the side stream now carries the leaf stream's work: a join must follow

    def side_batch_begin(self):
        """Side-stream sections opened from here on are HELD and go out behind ONE fork when side_batch_end() is called, instead of one
        fork each: an event record on the caller's stream costs it ~5 us of queue time (the 64->64 trunk's backward pass has four
        side sections per block).  The held launches start later, never earlier: they only read buffers the caller's stream does
        not rewrite before the join (the rule side() already lives by), so the result is the same."""
        assert self.sid == 0 and self.hold is None
        self.hold = []

    def side_batch_end(self):
        held, self.hold = self.hold, None
        if held:
            self.ops.append(["fork", None, None, 0])
            self.forks += 1
            for op, dyns in held:
                for key, ai in dyns:
                    self.dyn.setdefault(key, []).append((len(self.ops), ai))
                self.ops.append(op)

    def join(self):
        assert self.hold is None, "join inside a side batch"
        self.leaf_to_side()
        if self.forks:
            self.ops.append(["join", None, None, 0])
            self.forks = 0

    def _build_native(self):
        """Hand the recorded launches to the C-ABI plan executor (csrc/plan.cpp): replay = ONE foreign call."""
        lib = _lib.load()
        h = lib.tpgsr_plan_create()
        try:
            for name, fn, args, sid in self.ops:
                if fn is None:
                    if name == "edge":
                        rc = lib.tpgsr_plan_add_edge(h, args[0], args[1])
                    else:
                        rc = lib.tpgsr_plan_add_fork(h) if name == "fork" else lib.tpgsr_plan_add_join(h)
                else:
                    types = fn.argtypes[:-1]          # the trailing stream is supplied at run time
                    arr = (_lib.PlanArg * max(1, len(types)))()
                    for i, (t, a) in enumerate(zip(types, args)):
                        if t is _lib.cf:
                            arr[i].f = float(a)
                        elif t is _lib.vp:
                            arr[i].p = a
                        elif t in (_lib.ci, _lib.ll):
                            arr[i].i = int(a)
                        else:                          # POINTER(struct): recorded as ctypes.byref(struct)
                            arr[i].p = C.addressof(a._obj)
                    rc = lib.tpgsr_plan_add_launch(h, name.encode(), arr, len(types), sid)
                if rc < 0:
                    check(rc, f"{self.name}: native plan, op {name}")
        except Exception:
            lib.tpgsr_plan_destroy(h)
            raise
        assert lib.tpgsr_plan_size(h) == len(self.ops)
        self._native = h
        self._has_side = any(fn is None for _, fn, _, _ in self.ops)
        self._has_leaf = any(sid == 2 for _, _, _, sid in self.ops)

    def run(self):
        if not self._native:
            self._build_native()
        if DRYRUN:
            return
        main = torch.cuda.current_stream()
        side = side_stream(main.device).cuda_stream if self._has_side else None
        leaf = aux_stream(main.device).cuda_stream if self._has_leaf else None
        rc = _lib.load().tpgsr_plan_run3(self._native, main.cuda_stream, side, leaf)
        if rc:
            check(rc, self.name)

    def run_interpreted(self):
        """The same replay op by op through ctypes (reference implementation of run(); tests compare the two)."""
        main = torch.cuda.current_stream()
        side, leaf = side_stream(main.device), aux_stream(main.device)
        objs = (main, side, leaf)
        streams = (main.cuda_stream, side.cuda_stream, leaf.cuda_stream)
        for name, fn, args, sid in self.ops:
            if fn is None:
                if SERIAL:
                    continue
                if name == "fork":
                    side.wait_stream(main)
                elif name == "join":
                    main.wait_stream(side)
                else:
                    objs[args[1]].wait_stream(objs[args[0]])
                continue
            rc = fn(*args, streams[sid])
            if rc:
                check(rc, f"{self.name}:{name}")

    def __del__(self):
        h, self._native = getattr(self, "_native", None), None
        if h:
            try:
                _lib.load().tpgsr_plan_destroy(h)
            except Exception:
                pass

    def __len__(self):
        return len(self.ops)


class _SideCtx:
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        p = self.plan
        assert p.sid == 0, "nested side-stream sections"
        if p.hold is None:
            p.ops.append(["fork", None, None, 0])
            p.forks += 1
        p.sid = 1

    def __exit__(self, *exc):
        self.plan.sid = 0
        return False


class _LeafCtx:
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        p = self.plan
        assert p.sid == 0, "leaf section inside another stream section"
        p.ops.append(["edge", None, (0, 2), 0])
        p.sid = 2
        p.leaf_pending = True

    def __exit__(self, *exc):
        self.plan.sid = 0
        return False


class _NoSide:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def side():
    """Side-stream section of the plan being recorded (no-op when launching eagerly outside a plan, and inside a leaf section:
    the leaf stream runs its own weight gradients in order)."""
    return _REC.side() if _REC is not None and getattr(_REC, "overlap", False) and _REC.sid != 2 else _NoSide()


SIDE_BATCH = os.environ.get("TPGSR_SIDE_BATCH", "1") != "0"


def side_batch_begin():
    """see Plan.side_batch_begin (no-op outside a recording with side streams, and with TPGSR_SIDE_BATCH=0)"""
    if SIDE_BATCH and _REC is not None and getattr(_REC, "overlap", False) and _REC.sid == 0 and _REC.hold is None:
        _REC.side_batch_begin()
        return True
    return False


def side_batch_end(opened: bool):
    if opened:
        _REC.side_batch_end()


def leaf():
    """Leaf-stream section of the plan being recorded (Plan.leaf); no-op unless the plan opted in (`use_leaf`)."""
    return _REC.leaf() if _REC is not None and getattr(_REC, "overlap", False) and getattr(_REC, "use_leaf", False) else _NoSide()


def leaf_join():
    """the caller's stream waits for everything recorded on the leaf stream so far (no-op when nothing is pending there / outside a
    recording): for leaf sections whose result the caller's stream needs later in the same plan"""
    if _REC is not None and _REC.sid == 0 and _REC.leaf_pending:
        _REC.ops.append(["edge", None, (2, 0), 0])


def stream_tag() -> str:
    """suffix for stream-ordered scratch buffers: launches recorded on the leaf / side stream must not share them with the main stream's"""
    return "_leaf" if (_REC is not None and _REC.sid == 2) else "_side" if (_REC is not None and _REC.sid == 1) else ""


class DynPtr:
    def __init__(self, key):
        self.key = key


_REC: Optional[Plan] = None


class recording:
    def __init__(self, plan: Plan):
        self.plan = plan

    def __enter__(self):
        global _REC
        assert _REC is None, "nested recording"
        _REC = self.plan
        return self.plan

    def __exit__(self, *exc):
        global _REC
        _REC = None
        return False


def continue_in(plan: Plan):
    """The recording goes on in `plan` (same streams, run right after the current one): the caller of the two plans regains control in
    between -- a train step launches a gradient bucket there.  Side-stream work of the first plan stays pending (no join): the second
    plan's launches follow it in stream order, and its join covers both."""
    global _REC
    assert _REC is not None and _REC.sid == 0 and _REC.hold is None, "continue_in inside a stream section / side batch"
    assert not getattr(_REC, "deferred", None), "deferred slab reduces must be flushed before the recording moves on"
    if _REC.forks or _REC.leaf_pending:
        plan.forks += 1            # something of the first plan may still run on the side stream: the second plan's join must be emitted
    _REC = plan


def _launch(name, *args):
    fn = getattr(_lib.load(), name)
    if _REC is not None:
        args = list(args)
        if _REC.hold is not None and _REC.sid == 1:      # side-stream launch inside a side batch: recorded when the batch closes
            dyns = [(a.key, ai) for ai, a in enumerate(args) if isinstance(a, DynPtr)]
            _REC.hold.append(([name, fn, [None if isinstance(a, DynPtr) else a for a in args], 1], dyns))
            return
        oi = len(_REC.ops)
        for ai, a in enumerate(args):
            if isinstance(a, DynPtr):
                _REC.dyn.setdefault(a.key, []).append((oi, ai))
                args[ai] = None
        _REC.ops.append([name, fn, args, _REC.sid])
        return
    if DRYRUN:
        if len(args) + 1 != len(fn.argtypes):
            raise TypeError(f"{name}: {len(args)} arguments + stream, the C ABI takes {len(fn.argtypes)}")
        for t, a in zip(fn.argtypes, args):
            t.from_param(a)
        return
    check(fn(*args, _stream()), name)


def _p(t):
    if t is None or isinstance(t, (int, DynPtr)):
        return t
    assert (t.is_cuda or DRYRUN) and t.dtype in (torch.float32, torch.int32, torch.uint8, torch.float64) and t.is_contiguous(), \
        f"tpgsr kernels need contiguous fp32 CUDA tensors (got {t.dtype}, cuda={t.is_cuda}, contiguous={t.is_contiguous()})"
    if _REC is not None:
        _REC.keep.append(t)
    return t.data_ptr()


@dataclass
class ConvGeom:
    """Geometry of one stride-1 convolution call (see tpgsr_conv_args in include/tpgsr_hip.h)."""
    N: int
    H: int
    W: int
    Cin: int
    Cout: int
    KH: int = 1
    KW: int = 1
    pad_h: int = 0
    pad_w: int = 0
    OH: Optional[int] = None
    OW: Optional[int] = None

    def __post_init__(self):
        if self.OH is None:
            self.OH = self.H + 2 * self.pad_h - self.KH + 1
        if self.OW is None:
            self.OW = self.W + 2 * self.pad_w - self.KW + 1

    @property
    def M(self):
        return self.N * self.OH * self.OW

    @property
    def K(self):
        return self.KH * self.KW * self.Cin

    def dgrad(self) -> "ConvGeom":
        """Geometry of the data-gradient conv (input = dy, output = dx)."""
        return ConvGeom(self.N, self.OH, self.OW, self.Cout, self.Cin, self.KH, self.KW,
                        self.KH - 1 - self.pad_h, self.KW - 1 - self.pad_w, self.H, self.W)


def make_conv_args(g: ConvGeom, inp, wt=None, out=None, *, bias=None, in2=None, in_scale=None, in_shift=None, in_act=None,
                   in_ps=False, in_ld=None, in_coff=0, in2_ld=None, out_act=None, out_ps=False, out_ld=None, out_coff=0,
                   bn_partial=None, in_b=None, cin_a=0, in_b_ld=None, in_dil_w=1, wt_ld=0, wt_coff=0, stride_w=1, bnb=None, bn_fin=None,
                   bn_coarse=False, in2_scale=None) -> ConvArgs:
    """`bnb` (a dict from engine.BNLayer.fuse_stats): this convolution produces the gradient that enters a BatchNorm's backward pass --
    its epilogue also writes that BatchNorm's two reduction sums per 64-pixel row block (tpgsr_conv_args.bnb_y).
    `bn_fin` (a dict from engine.BNLayer.fin / fuse_stats(...)["fin"]): the launch also FINALIZES the BatchNorm whose statistics it
    leaves in bn_partial -- by its last workgroup where the kernel can, by an appended launch otherwise (tpgsr_conv_args.fin_mode)"""
    a = ConvArgs()
    a.in_, a.in2, a.in_scale, a.in_shift = _p(inp), _p(in2), _p(in_scale), _p(in_shift)
    a.in2_scale = _p(in2_scale)        # a = in * in_scale + in_shift + in2 * in2_scale (whole-CU halo kernel only: conv_in2_scale_ok)
    a.wt, a.bias, a.out, a.bn_partial = _p(wt), _p(bias), _p(out), _p(bn_partial)
    a.N, a.H, a.W, a.Cin = g.N, g.H, g.W, g.Cin
    a.in_ld = (cin_a if in_b is not None else g.Cin) if in_ld is None else in_ld
    a.in_coff = in_coff
    a.in2_ld = g.Cin if in2_ld is None else in2_ld
    a.in_act = act_code(in_act)
    a.in_ps = int(bool(in_ps))
    a.Cout, a.KH, a.KW, a.pad_h, a.pad_w, a.OH, a.OW = g.Cout, g.KH, g.KW, g.pad_h, g.pad_w, g.OH, g.OW
    a.out_ld = g.Cout if out_ld is None else out_ld
    a.out_coff = out_coff
    a.out_act = act_code(out_act)
    a.out_ps = int(bool(out_ps))
    a.in_b = _p(in_b)
    a.cin_a = cin_a if in_b is not None else 0
    a.in_b_ld = (g.Cin - cin_a) if in_b_ld is None else in_b_ld
    a.in_dil_w = in_dil_w
    a.wt_ld, a.wt_coff = wt_ld, wt_coff
    a.stride_w = stride_w
    a.terms = 0
    if CONV_TERMS and g.Cin % 4 == 0:
        if wt is None or isinstance(wt, (int, DynPtr)):
            a.terms = CONV_TERMS                       # weight-gradient use: no weight operand
        else:
            tw = getattr(wt, "_tpgsr_twin", None)
            if tw is not None:
                a.terms, a.kp, a.wt_bf, a.wt_bf_cin = CONV_TERMS, tw[1], tw[0].data_ptr(), tw[2]
                a._dev = tw[0].device               # (where a split-K scratch buffer of this launch has to live: conv_fwd)
                assert tw[2] in (0, g.Cin), f"operand split for Cin {tw[2]}, used by a convolution over {g.Cin} channels"
                if _REC is not None:
                    _REC.keep.append(tw[0])
    if bnb is not None:
        if not (a.terms and a.wt_bf and wt_coff % 32 == 0):
            raise RuntimeError("BatchNorm-backward statistics ride on the split-bf16 convolution kernels only (check bnb_fusable first)")
        a.bn_partial = _p(bnb.get("partial"))
        a.bnb_y, a.bnb_mean, a.bnb_rstd = _p(bnb["y"]), _p(bnb.get("mean")), _p(bnb.get("rstd"))
        a.bnb_scale, a.bnb_shift, a.bnb_act = _p(bnb.get("scale")), _p(bnb.get("shift")), act_code(bnb["act"])
        a.bnb_store_dz = int(bool(bnb.get("store_dz", False)))   # without "partial": a plain activation backward on the way out
        if bn_fin is None:
            bn_fin = bnb.get("fin")
    if bn_fin is not None and BN_FIN_FUSE:
        f = bn_fin
        a.fin_mode, a.fin_count, a.fin_counter, a.fin_gamma = f["mode"], f["count"], _p(f["counter"]), _p(f["gamma"])
        if f["mode"] == 1:
            a.fin_beta, a.fin_bias = _p(f["beta"]), _p(f.get("bias"))
            a.fin_scale, a.fin_shift, a.fin_mean, a.fin_rstd = _p(f["scale"]), _p(f["shift"]), _p(f.get("save_mean")), _p(f.get("save_rstd"))
            a.fin_rm, a.fin_rv = _p(f.get("running_mean")), _p(f.get("running_var"))
            a.fin_momentum, a.fin_eps = f.get("momentum", 0.1), f.get("eps", 1e-5)
        else:
            a.fin_scale, a.fin_shift, a.fin_mean = _p(f["coef"]), _p(f.get("dgamma")), _p(f.get("dbeta"))
            a.fin_accumulate = int(bool(f.get("accumulate", True)))
    # bn_coarse (or bnb["coarse"]): whoever reduces the partial rows copes with ONE ROW PER 192 PIXELS when the launch lands on the
    # whole-CU halo kernel (tpgsr_conv_args.bn_row_tiles; a third of the rows for every workgroup of a consumer that finalizes the
    # BatchNorm itself, csrc/bn_derive.h); the caller reads the granularity back from `a.bn_row_tiles` / bnb["row_tiles"]
    if (bn_coarse or (bnb is not None and bnb.get("coarse"))) and a.bn_partial and not a.fin_mode and BN_COARSE_ROWS and CONV_TERMS:
        if _lib.load().tpgsr_conv_bn_row_tiles(C.byref(a)) == 3:
            a.bn_row_tiles = 3
    if bnb is not None:
        bnb["row_tiles"] = max(1, a.bn_row_tiles)
    return a


def DRYRUN_NO_LIB() -> bool:
    """a dry run (TPGSR_PLAN_DRYRUN=1) still loads the library for host-side queries; kept as a function so engines can ask in one place"""
    return False


def conv_in2_scale_ok(a: ConvArgs) -> bool:
    """will tpgsr_conv_fwd take this launch with its scaled residual operand (in2_scale)?  Only the whole-CU halo kernel's loader has it"""
    return bool(_lib.load().tpgsr_conv_in2_scale_ok(C.byref(a)))


# TPGSR_BNB_APPLY_FOLD=1 (round 6; OFF by default): the apply pass of a BatchNorm's backward runs on the
# WEIGHT-GRADIENT stream only (it still produces dy for the weight gradient) and the caller's stream takes dy through the loader of the
# consuming data-gradient convolution (tpgsr_conv_args.in2_scale, whole-CU halo kernel).  Built, tested (tests/test_conv_halo3_gpu.py,
# the parity suite under the switch), measured on one box, interleaved: 5.456 / 5.492 ms per C3 step folded against 5.439 / 5.441 with
# the eleven apply launches in place (profiles/r06i_bnb_apply_fold_ab.md) -- the two-operand loader has ONE register set (no load of the
# next channel block in flight while this one is split), which costs the trunk's data gradient more than the 6-us launch it replaces.
BNB_APPLY_FOLD = os.environ.get("TPGSR_BNB_APPLY_FOLD", "0") == "1"
BN_COARSE_ROWS = os.environ.get("TPGSR_BN_COARSE_ROWS", "1") != "0"


def bn_rows(M: int, row_tiles: int = 1) -> int:
    """rows of a bn_partial buffer over M pixels at `row_tiles` 64-pixel blocks per row"""
    nblk = (M + 63) // 64
    return (nblk + max(1, row_tiles) - 1) // max(1, row_tiles)


# BatchNorm finalize (forward statistics -> scale / shift; backward sums -> dgamma / dbeta / coefficients) as part of the convolution
# launch that produces the partial sums: TPGSR_BN_FIN_FUSE=1.  OFF by default -- measured slower (C3 6.19 vs 6.08 ms per step,
# profiles/r04aa_bn_fin_fuse_ab.md): the last workgroup of the whole-CU kernel reduces 393 KB of partial rows alone, through
# L1-bypassing loads of write-through data, in ~14 us; the separate launch spreads the same reduction over 64 workgroups and costs
# ~8 us including its launch boundary.  Correct and tested either way (tests/test_bn_fin_fuse_gpu.py).
BN_FIN_FUSE = os.environ.get("TPGSR_BN_FIN_FUSE", "0") == "1"


def bn_fin_fused() -> bool:
    return bool(BN_FIN_FUSE and CONV_TERMS)


# BatchNorm-backward reduction fused into the producing data-gradient convolution (TPGSR_BNB_FUSE=0: its own launch, as before)
BNB_FUSE = os.environ.get("TPGSR_BNB_FUSE", "1") != "0"


def bnb_fusable(cin: int) -> bool:
    """can a convolution over `cin` input channels carry a BatchNorm's backward statistics under the current arithmetic policy?"""
    return bool(BNB_FUSE and CONV_TERMS and cin % 4 == 0)


def conv_fwd(args: ConvArgs):
    """tpgsr_conv_fwd.  The launch is first offered to the split-K planner (tpgsr_conv_splitk_plan: the few
    launches with fewer output tiles than CUs and a long contraction -- BiLSTM projections' data gradients, InfoGen's transposed
    convolutions, the STN head's 3 x 3 convolutions on 96 pixels); a taker gets a scratch buffer of its OWN (plans of different engines
    run on different physical streams at the same time: nothing may be shared), kept alive by the argument block the plan holds."""
    if not DRYRUN and args.terms and not args.sk_splits:
        # (eager launches too -- the operator-by-operator path gives the recorded plan's bits; its buffer goes back to torch's
        #  stream-ordered allocator right after the call, which is safe on the stream the launch was queued on)
        nb = C.c_longlong(0)
        S = _lib.load().tpgsr_conv_splitk_plan(C.byref(args), C.byref(nb))
        if S > 1:
            buf = torch.empty(nb.value // 4, dtype=torch.float32, device=getattr(args, "_dev", None) or torch.device("cuda", torch.cuda.current_device()))
            args._sk_buf = buf
            args.sk_part, args.sk_splits = buf.data_ptr(), S
    _launch("tpgsr_conv_fwd", C.byref(args))


def wgrad_splits(M, K, Cout, geom: "ConvGeom" = None) -> int:
    """number of pixel splits Z of a weight gradient (the caller sizes part [Z][K][Cout] / dbpart [Z][Cout] with it).
    With `geom` and the bf16 matrix-core policy on, convolutions the halo weight-gradient kernel takes get ITS split count
    (one workgroup per CU); pass the same Z to make_wgrad_args(zsplits=Z)."""
    if geom is not None and CONV_TERMS and not DRYRUN:
        plan = wgrad_halo_plan(geom)
        if plan is not None:
            return plan[0]
    return _lib.load().tpgsr_wgrad_splits(M, K, Cout)


def wgrad_halo_plan(g: "ConvGeom"):
    """(Z, scratch bytes) when the halo weight-gradient kernel takes this geometry under the current policy, else None"""
    a = ConvArgs()
    a.N, a.H, a.W, a.Cin, a.Cout, a.KH, a.KW = g.N, g.H, g.W, g.Cin, g.Cout, g.KH, g.KW
    a.pad_h, a.pad_w, a.OH, a.OW = g.pad_h, g.pad_w, g.OH, g.OW
    a.terms = CONV_TERMS
    z, nbytes = C.c_int(0), C.c_longlong(0)
    if not _lib.load().tpgsr_wgrad_halo_plan(C.byref(a), C.byref(z), C.byref(nbytes)):
        return None
    return z.value, nbytes.value


# scratch for the pre-split dy of the halo weight-gradient kernel: one buffer per (device, stream), grown by allocating a larger one
# (earlier buffers stay alive: recorded plans keep pointing at them; everything on one stream is ordered, so sharing is safe)
_DY_SCRATCH = {}


def _dy_scratch(nbytes: int, device) -> torch.Tensor:
    key = (str(device), current_stream().cuda_stream, stream_tag())     # (the leaf stream's launches get their own)
    bufs = _DY_SCRATCH.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(nbytes, dtype=torch.uint8, device=device))
    return bufs[-1]


def make_wgrad_args(cargs: ConvArgs, dy, part, dbpart=None, *, dy_ld=None, dy_coff=0, dy_ps=False, zsplits=0) -> WgradArgs:
    """zsplits: the Z the caller sized `part` with when it came from wgrad_splits(..., geom=...) (0: the legacy count)"""
    w = WgradArgs()
    w.c = cargs
    w.dy = _p(dy)
    w.dy_ld = cargs.Cout if dy_ld is None else dy_ld
    w.dy_coff = dy_coff
    w.dy_ps = int(bool(dy_ps))
    w.part = _p(part)
    w.dbpart = _p(dbpart)
    w.zsplits = int(zsplits)
    if zsplits and cargs.terms and not DRYRUN and isinstance(dy, torch.Tensor):
        z, nbytes = C.c_int(0), C.c_longlong(0)
        if _lib.load().tpgsr_wgrad_halo_plan(C.byref(cargs), C.byref(z), C.byref(nbytes)):
            buf = _dy_scratch(nbytes.value, dy.device)
            w.dy_bf = buf.data_ptr()
            if _REC is not None:
                _REC.keep.append(buf)
    return w


def conv_wgrad(w: WgradArgs):
    _launch("tpgsr_conv_wgrad", C.byref(w))


def conv_wgrad_batch(wargs):
    """Several independent weight-gradient GEMMs (WgradArgs from make_wgrad_args) as ONE launch -- tpgsr_conv_wgrad_batch; launches the
    batch does not take (not 1x1 / not on the split-bf16 tile-loop kernel, or a different loader variant than the first) go out singly."""
    lib = _lib.load()
    items, rest, key = [], [], None
    for w in wargs:
        it = _lib.WgradBatchItem()
        ld = lib.tpgsr_conv_wgrad_batch_prepare(C.byref(w), C.byref(it))
        k = (ld, w.c.terms)
        if ld >= 0 and (key is None or k == key):
            key = k
            items.append(it)
        else:
            rest.append(w)
    if len(items) < 2:
        for w in wargs:
            conv_wgrad(w)
        return
    arr = (_lib.WgradBatchItem * len(items))(*items)
    blk = 0
    for it in arr:
        it.blk0 = blk
        blk += it.nblk
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(torch.device("cpu") if DRYRUN else torch.device("cuda", torch.cuda.current_device()))
    if _REC is not None:
        _REC.keep.append(list(wargs))
        if not (_REC.hold is not None and _REC.sid == 1):
            _REC.meta[len(_REC.ops)] = [it.w for it in arr]
    _launch("tpgsr_conv_wgrad_batch", _p(table), len(items), blk, key[0], key[1])
    if rest:
        conv_wgrad_batch(rest)      # (another loader variant: a batch of its own, or a single launch)


def deferring() -> bool:
    """True while recording a plan that batches its weight-gradient slab reduces into one launch (Plan.deferred)."""
    return _REC is not None and getattr(_REC, "deferred", None) is not None


def wgrad_reduce(part, dbpart, Z, g: ConvGeom, dw, db=None, *, layout=0, accumulate=True, gscale=1.0, real=None):
    """real = (Cin, KH, KW, cin_ld) of the parameter when the GEMM ran on a zero-padded operand (slab rows k = tap*cin_ld + ci
    with ci >= Cin or tap >= KH*KW are skipped); default: the geometry's own"""
    Cin, KH, KW, cin_ld = real if real is not None else (g.Cin, g.KH, g.KW, 0)
    item = (part, dbpart, Z, g.K, Cin, g.Cout, KH, KW, layout, dw, db, int(accumulate), float(gscale), cin_ld)
    if deferring():   # the caller gave this layer its own slab buffers; reduced by flush_wgrad_reduces()
        _REC.deferred.append(item + (_REC.sid == 2,))       # (tagged: produced on the leaf stream)
    elif real is not None:
        _reduce_program([item])
    else:
        _launch("tpgsr_wgrad_reduce", _p(part), _p(dbpart), Z, g.K, Cin, g.Cout, KH, KW, layout, _p(dw), _p(db),
                int(accumulate), gscale)


def _reduce_program(items):
    lib = _lib.load()
    arr = (_lib.WgradReduceDesc * len(items))()
    blk = 0
    seen = set()
    for d, (part, dbpart, Z, Kd, Cin, Cout, KH, KW, layout, dw, db, acc, gscale, cin_ld, *_leaf) in zip(arr, items):
        has_b = db is not None and dbpart is not None
        d.part, d.dbpart = _p(part), (_p(dbpart) if has_b else None)
        d.dw, d.db = _p(dw), (_p(db) if has_b else None)
        d.Z, d.K, d.Cin, d.Cout, d.KH, d.KW, d.layout, d.accumulate, d.gscale, d.blk0 = Z, Kd, Cin, Cout, KH, KW, layout, acc, gscale, blk
        d.cin_ld = cin_ld
        blk += lib.tpgsr_wgrad_reduce_blocks2(Kd, Cin, Cout, KH, KW, layout, cin_ld, int(has_b))
        assert part.data_ptr() % 16 == 0, "weight-gradient slabs must be 16-byte aligned"
        assert dw.data_ptr() not in seen, "two deferred reduces of one program target the same gradient"
        seen.add(dw.data_ptr())
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(items[0][0].device)
    with (side() if (_REC is None or _REC.sid == 0) else _NoSide()):   # already inside a side / leaf section: stay there
        _launch("tpgsr_wgrad_reduce_program", _p(table), len(items), blk)


def flush_wgrad_reduces(split_leaf=False):
    """Emit ONE tpgsr_wgrad_reduce_program launch (side stream) for every reduce deferred so far in this plan.
    split_leaf: the slabs written on the LEAF stream get a program of their own, at the end of that stream -- the weight-gradient stream
    then never waits for the leaf chain (the STN head's backward: ~45 small launches in a row, the last to finish in a TPGSR step; the
    one shared program made the stream -- and with it the text-prior generator's weight gradients queued behind -- wait ~0.6 ms for it).
    Whoever reads the gradients next must be ordered after BOTH streams (Plan.join() does; TPGSRTrainStep._join_side)."""
    rec = _REC
    items, rec.deferred = rec.deferred, []
    if not items:
        return
    leaf_items = [it for it in items if split_leaf and len(it) > 14 and it[14]]
    side_items = [it for it in items if not (split_leaf and len(it) > 14 and it[14])]
    if not leaf_items:
        rec.leaf_to_side()          # one program: it reads the leaf stream's slabs too
    if side_items:
        _reduce_program(side_items)
    if leaf_items:
        with rec.leaf():
            _reduce_program(leaf_items)


def pack_conv_weight(w, Cout, Cin, KH, KW, wt_f=None, wt_d=None, *, transposed=False, wscale=1.0):
    _launch("tpgsr_pack_conv_weight", _p(w), Cout, Cin, KH, KW, int(transposed), wscale, _p(wt_f), _p(wt_d))


def pack_tail_weight(w, Co, Cc, KS, wt_f=None, wt_d=None):
    _launch("tpgsr_pack_tail_weight", _p(w), Co, Cc, KS, _p(wt_f), _p(wt_d))


def compose_bwd_program(descs_dev, ndesc, total_blocks):
    _launch("tpgsr_compose_bwd_program", _p(descs_dev), ndesc, total_blocks)


def pack_program(descs_dev, ndesc, total_blocks):
    _launch("tpgsr_pack_program", _p(descs_dev), ndesc, total_blocks)


def mfma_probe(out, blocks, iters):
    _launch("tpgsr_mfma_probe", _p(out), blocks, iters)


def copy(src, dst, n):
    _launch("tpgsr_copy", _p(src), _p(dst), n)


def zero(dst, n):
    _launch("tpgsr_zero", _p(dst), n)


# ---- BatchNorm ------------------------------------------------------------------------------------------------
def bn_finalize(partial, nblk, C_, count, conv_bias, gamma, beta, running_mean, running_var, scale, shift, save_mean=None,
                save_rstd=None, *, momentum=0.1, eps=1e-5, eval_mode=False):
    _launch("tpgsr_bn_finalize", _p(partial), nblk, C_, count, _p(conv_bias), _p(gamma), _p(beta), _p(running_mean),
                                        _p(running_var), momentum, eps, int(eval_mode), _p(scale), _p(shift), _p(save_mean),
                                        _p(save_rstd))


def bn_stats(x, M, C_, partial, nblk, ld=None):
    _launch("tpgsr_bn_stats", _p(x), M, C_, C_ if ld is None else ld, _p(partial), nblk)


def bn_bwd_reduce(da, da2, y, M, C_, scale, shift, save_mean, save_rstd, act, partial, nblk):
    _launch("tpgsr_bn_bwd_reduce", _p(da), _p(da2), _p(y), M, C_, _p(scale), _p(shift), _p(save_mean), _p(save_rstd),
                                          act_code(act), _p(partial), nblk)


def bn_bwd_finalize(partial, nblk, C_, count, gamma, save_mean, save_rstd, dgamma, dbeta, coef, accumulate=True):
    _launch("tpgsr_bn_bwd_finalize", _p(partial), nblk, C_, count, _p(gamma), _p(save_mean), _p(save_rstd), _p(dgamma),
                                            _p(dbeta), int(accumulate), _p(coef))


def bn_bwd_apply(da, da2, y, M, C_, scale, shift, act, coef, dy):
    _launch("tpgsr_bn_bwd_apply", _p(da), _p(da2), _p(y), M, C_, _p(scale), _p(shift), act_code(act), _p(coef), _p(dy))


# BatchNorm finalized inside its first consumer's launch (csrc/bn_derive.h, round 5) instead of by tpgsr_bn_finalize / tpgsr_bn_bwd_finalize
# launches of their own: the first ceil(C / 16) workgroups of the consumer's grid sum the partial rows and publish, everybody waits on a
# flag.  OFF by default: three forms of "no finalize launch" were built and measured against the separate launches inside the C3 step
# (profiles/r05e_bn_derive_ab.md) -- every workgroup summing the rows itself (256-thread workgroups: 6.34 vs 6.06 ms; one 1024-thread
# workgroup per CU: 6.09 vs 5.93), and this deriver + flag hand-off (6.10 vs 5.99) -- and round 4's last-workgroup finalize before them
# (6.19 vs 6.08): the separate 64-workgroup launch (3 us + a ~2 us boundary) is the cheapest way this chip has of putting a grid-wide
# reduction between two launches.  TPGSR_BN_DERIVE=1 records the in-launch form (results agree to the last bit or two of scale / shift).
BN_DERIVE = os.environ.get("TPGSR_BN_DERIVE", "0") == "1"


def bn_derive_ok(C_: int, M: int = 1 << 30) -> bool:
    """can a launch over an [M][C] map finalize its BatchNorm itself?  (channel counts the derivers split evenly; enough workgroups)"""
    D = (C_ + 15) // 16
    return BN_DERIVE and (C_ == 8 or (C_ % 16 == 0 and 16 <= C_ <= 512)) and M * C_ >= 1024 * D


def plan_flag(device):
    """one int32 word, zero when the launch that uses it starts: a fresh tensor for a direct call; inside a recorded plan a slot of the
    plan's flag block, which the plan zeroes with ONE launch at its very start (inserted here when the first flag is asked for)"""
    rec = _REC
    if rec is None:
        return torch.zeros(1, dtype=torch.int32, device=device)
    if getattr(rec, "flags", None) is None:
        rec.flags, rec.nflags = torch.zeros(128, dtype=torch.int32, device=device), 0
        rec.keep.append(rec.flags)
        rec.ops.insert(0, ["tpgsr_zero", getattr(_lib.load(), "tpgsr_zero"), [rec.flags.data_ptr(), 128], 0])
        rec.dyn = {k: [(oi + 1, ai) for oi, ai in v] for k, v in rec.dyn.items()}      # every recorded op moved down by one
        rec.meta = {oi + 1: v for oi, v in rec.meta.items()}
    if rec.nflags >= rec.flags.numel():
        raise RuntimeError("a recorded plan finalizes more than 128 BatchNorms inside their consumers")
    rec.nflags += 1
    return rec.flags[rec.nflags - 1:rec.nflags]


def make_bn_derive(rows, nrows, C_, count, gamma, *, flag=None, bias=None, beta=None, running_mean=None, running_var=None, momentum=0.1,
                   eps=1e-5, scale=None, shift=None, save_mean=None, save_rstd=None, dgamma=None, dbeta=None, coef=None, accumulate=False):
    """flag: one zeroed int32 word (default: plan_flag -- a slot of the recorded plan's block, or a fresh tensor for a direct call, which
    then serves ONE launch)"""
    d = _lib.BnDerive()
    if flag is None:
        flag = plan_flag(rows.device if isinstance(rows, torch.Tensor) else None)
    d._flag_keep = flag
    d.rows, d.nrows, d.C, d.count = _p(rows), int(nrows), int(C_), int(count)
    d.bias, d.gamma, d.beta = _p(bias), _p(gamma), _p(beta)
    d.running_mean, d.running_var, d.momentum, d.eps = _p(running_mean), _p(running_var), momentum, eps
    d.scale, d.shift, d.save_mean, d.save_rstd = _p(scale), _p(shift), _p(save_mean), _p(save_rstd)
    d.dgamma, d.dbeta, d.coef, d.accumulate = _p(dgamma), _p(dbeta), _p(coef), int(bool(accumulate))
    d.flag = _p(flag)
    return d


def affine_act_bnd(d, x, M, act, out):
    _launch("tpgsr_affine_act_bnd", C.byref(d), _p(x), M, act_code(act), _p(out))


def affine_act_pool_bnd(d, x, N, H, W, act, ph, pw, out):
    _launch("tpgsr_affine_act_pool_bnd", C.byref(d), _p(x), N, H, W, act_code(act), ph, pw, _p(out))


def bn_bwd_apply_bnd(d, da, da2, y, M, scale, shift, act, dy):
    _launch("tpgsr_bn_bwd_apply_bnd", C.byref(d), _p(da), _p(da2), _p(y), M, _p(scale), _p(shift), act_code(act), _p(dy))


def affine_act(x, M, C_, scale, shift, act, out):
    _launch("tpgsr_affine_act", _p(x), M, C_, _p(scale), _p(shift), act_code(act), _p(out))


def affine_act_pool(x, N, H, W, C_, scale, shift, act, ph, pw, out):
    _launch("tpgsr_affine_act_pool", _p(x), N, H, W, C_, _p(scale), _p(shift), act_code(act), ph, pw, _p(out))


def affine_act_pool_bwd(x, dout, N, H, W, C_, scale, shift, act, ph, pw, dz):
    _launch("tpgsr_affine_act_pool_bwd", _p(x), _p(dout), N, H, W, C_, _p(scale), _p(shift), act_code(act), ph, pw, _p(dz))


# ---- elementwise ----------------------------------------------------------------------------------------------
def prelu_fwd(x, alpha, n, y):
    _launch("tpgsr_prelu_fwd", _p(x), _p(alpha), n, _p(y))


def prelu_bwd(x, alpha, dy, dy2, n, dx, dalpha_partial, nblk):
    _launch("tpgsr_prelu_bwd", _p(x), _p(alpha), _p(dy), _p(dy2), n, _p(dx), _p(dalpha_partial), nblk)


def add(a, b, n, out):
    _launch("tpgsr_add", _p(a), _p(b), n, _p(out))


def act_bwd(x, dy, n, act, dx):
    _launch("tpgsr_act_bwd", _p(x), _p(dy), n, act_code(act), _p(dx))


def nchw_to_nhwc(x, N, C_, H, W, out):
    _launch("tpgsr_nchw_to_nhwc", _p(x), N, C_, H, W, _p(out))


def nhwc_to_nchw(x, N, C_, H, W, out):
    _launch("tpgsr_nhwc_to_nchw", _p(x), N, C_, H, W, _p(out))


def reduce_partials(part, Z, n, out, accumulate=True):
    _launch("tpgsr_reduce_partials", _p(part), Z, n, _p(out), int(accumulate))


# ---- GRU ------------------------------------------------------------------------------------------------------
def bigru_fwd(gi, w_hh, b_hh, N, H, W, axis, h_out, gates=None):
    """gates [P][256]: (r, z, n, W_hn h + b_hn) per direction, stored for bigru_bwd (None at inference)"""
    _launch("tpgsr_bigru_fwd", _p(gi), _p(w_hh), _p(b_hh), N, H, W, axis, _p(h_out), _p(gates))


def make_bigru_proj_args(cargs: ConvArgs, w_hh, b_hh, axis, h_out, gates=None) -> "_lib.BigruProjArgs":
    """cargs: make_conv_args(ConvGeom(N, H, W, Cin, 192), x, Wc, None, bias=bc, **loader) -- the GruBlock's composed input projection"""
    a = _lib.BigruProjArgs()
    a.c = cargs
    a.w_hh, a.b_hh, a.h_out, a.gates, a.axis = _p(w_hh), _p(b_hh), _p(h_out), _p(gates), int(axis)
    return a


def bigru_proj_supported(pargs) -> bool:
    """does the one-launch GruBlock forward (csrc/gru_proj.hip) take this block under the current arithmetic policy?"""
    return bool(_lib.load().tpgsr_bigru_proj_supported(C.byref(pargs)))


def bigru_proj_fwd(pargs):
    _launch("tpgsr_bigru_proj_fwd", C.byref(pargs))


def bigru_bwd(gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dgh):
    _launch("tpgsr_bigru_bwd", _p(gates), _p(h_out), _p(dh_out), _p(dh_out2), _p(w_hh), N, H, W, axis, _p(dgi), _p(dgh))


def bigru_bwd2(gates, h_out, dh_out, dh_out2, w_hh, N, H, W, axis, dgi, dghn):
    """bigru_bwd with the hidden-side gradient written compactly: dghn [P][64] = dn_pre * r (its r / z planes are dgi's)"""
    _launch("tpgsr_bigru_bwd2", _p(gates), _p(h_out), _p(dh_out), _p(dh_out2), _p(w_hh), N, H, W, axis, _p(dgi), _p(dghn))


# every weight gradient of a GruBlock in ONE launch (csrc/gru_wgrad.hip) instead of three tile-loop weight-gradient launches
# (input side + 2 x hidden side); TPGSR_GRU_WGRAD=0 records the three launches as before.  Split-bf16 policies only.
GRU_WGRAD = os.environ.get("TPGSR_GRU_WGRAD", "1") != "0"


def gru_wgrad_fused() -> bool:
    return bool(GRU_WGRAD and CONV_TERMS)


def gru_wgrad_splits(P: int) -> int:
    return _lib.load().tpgsr_gru_wgrad_splits(P)


def gru_wgrad(cargs: ConvArgs, dgi, dghn, h, axis, Z, partC, dbC, partH, dbH):
    """cargs: make_conv_args(ConvGeom(N, H, W, Cin, 192), x, **loader) -- the A side of the composed projection"""
    w = GruWgradArgs()
    w.c = cargs
    w.dgi, w.dghn, w.h = _p(dgi), _p(dghn), _p(h)
    w.partC, w.dbC, w.partH, w.dbH = _p(partC), _p(dbC), _p(partH), _p(dbH)
    w.axis, w.zsplits = int(axis), int(Z)
    _launch("tpgsr_gru_wgrad", C.byref(w))


# ---- STN ------------------------------------------------------------------------------------------------------
def tps_grid_fwd(ctrl, inv_kernel, coord_repr, N, HW, NC, grid, src=None):
    _launch("tpgsr_tps_grid_fwd", _p(ctrl), _p(inv_kernel), _p(coord_repr), N, HW, NC, _p(grid), _p(src))


def tps_grid_bwd(dgrid, src, inv_kernel, coord_repr, N, HW, NC, dctrl):
    _launch("tpgsr_tps_grid_bwd", _p(dgrid), _p(src), _p(inv_kernel), _p(coord_repr), N, HW, NC, _p(dctrl))


def grid_sample_fwd(inp, grid, N, H, W, C_, OH, OW, align_corners, out):
    _launch("tpgsr_grid_sample_fwd", _p(inp), _p(grid), N, H, W, C_, OH, OW, int(align_corners), _p(out))


def grid_sample_bwd(inp, grid, dout, N, H, W, C_, OH, OW, align_corners, din, dgrid):
    _launch("tpgsr_grid_sample_bwd", _p(inp), _p(grid), _p(dout), N, H, W, C_, OH, OW, int(align_corners), _p(din),
                                            _p(dgrid))


def strip_resample_fwd(inp, scale, shift, act, N, Win, Wout, C_, out):
    _launch("tpgsr_strip_resample_fwd", _p(inp), _p(scale), _p(shift), act_code(act), N, Win, Wout, C_, _p(out))


def strip_resample_bwd(inp, scale, shift, act, dout, N, Win, Wout, C_, dz):
    _launch("tpgsr_strip_resample_bwd", _p(inp), _p(scale), _p(shift), act_code(act), _p(dout), N, Win, Wout, C_, _p(dz))


def hsum(d, N, H, W, C_, dstrip, accumulate=True):
    _launch("tpgsr_hsum", _p(d), N, H, W, C_, _p(dstrip), int(accumulate))


# ---- CRNN pieces ----------------------------------------------------------------------------------------------
def bicubic_gray_fwd(x_nchw, N, Ctot, H, W, OH, OW, out):
    _launch("tpgsr_bicubic_gray_fwd", _p(x_nchw), N, Ctot, H, W, OH, OW, _p(out))


def bicubic_gray_bwd(dout, N, Ctot, H, W, OH, OW, din_nchw):
    _launch("tpgsr_bicubic_gray_bwd", _p(dout), N, Ctot, H, W, OH, OW, _p(din_nchw))


def pool2d_fwd(x, N, H, W, C_, scale, shift, act, k, s, pd, out):
    _launch("tpgsr_pool2d_fwd", _p(x), N, H, W, C_, _p(scale), _p(shift), act_code(act), k[0], k[1], s[0], s[1], pd[0], pd[1], _p(out))


def pool2d_bwd(x, dout, N, H, W, C_, scale, shift, act, k, s, pd, dz):
    _launch("tpgsr_pool2d_bwd", _p(x), _p(dout), N, H, W, C_, _p(scale), _p(shift), act_code(act), k[0], k[1], s[0], s[1], pd[0], pd[1],
            _p(dz))


# 32-deep K chunks per workgroup of the BiLSTM backward's recurrent GEMM (K = 1024): more chunks = fewer slabs for the gate kernel to add
LSTM_BWD_KCHUNKS = int(os.environ.get("TPGSR_LSTM_BWD_KCHUNKS", "4"))
LSTM_FWD_KCHUNKS = int(os.environ.get("TPGSR_LSTM_FWD_KCHUNKS", "1"))      # forward: K = 256


def lstm_rec_gemm(a0, a1, a_stride, b0, b1, Nrows, Kd, Nc, S, out):
    """a0 / a1: raw device addresses (ints) of row 0 of the two directions' A operands inside a kept-alive tensor"""
    _launch("tpgsr_lstm_rec_gemm", a0, a1, a_stride, _p(b0), _p(b1), Nrows, Kd, Nc, S, _p(out))


# persistent BiLSTM recurrences (Hh == 256, N <= 64): ONE launch per BiLSTM and pass instead of two launches per time step
# (csrc/lstm_seq.hip: per-step exchange between the 2 x 32 workgroups by write-through stores + one relaxed arrival counter).
# TPGSR_LSTM_SEQ=0 records the per-step launches (tpgsr_lstm_rec_gemm + tpgsr_lstm_step_{fwd,bwd}) instead.
LSTM_SEQ = os.environ.get("TPGSR_LSTM_SEQ", "1") == "1"
LSTM_SEQ_BWD = os.environ.get("TPGSR_LSTM_SEQ_BWD", "1" if LSTM_SEQ else "0") == "1"
_LSTM_CORESIDENT = {}


def lstm_seq_coresident(device) -> bool:
    """Can the persistent BiLSTM launches run on `device`?  Asked once per device when an engine records its plans
    (tpgsr_lstm_seq_probe: 64 workgroups with the backward kernel's footprint must become resident together within ~50 ms).  False --
    a shared / partitioned / smaller GPU -- makes the engines record the per-step launches instead of a step that ends in NaN.
    TPGSR_LSTM_SEQ_PROBE=0 skips the question (the persistent form is recorded unconditionally, as before round 5)."""
    if DRYRUN or os.environ.get("TPGSR_LSTM_SEQ_PROBE", "1") == "0":
        return True
    if torch.cuda.is_current_stream_capturing():
        return True                       # (plans are recorded by the eager warm-up steps before a capture; a capture cannot synchronise)
    key = str(device)
    if key not in _LSTM_CORESIDENT:
        rc = 0
        for attempt in range(2):          # a negative answer is asked again once, on a drained device: the first plan is recorded while the
            words = torch.zeros(2, dtype=torch.int32, device=device)      # step's other streams may still be busy
            rc = _lib.load().tpgsr_lstm_seq_probe(words.data_ptr(), _stream())
            if rc < 0:
                check(rc, "tpgsr_lstm_seq_probe")
            if rc:
                break
            torch.cuda.synchronize(device)
        _LSTM_CORESIDENT[key] = bool(rc)
        if not rc:
            import warnings
            warnings.warn("tpgsr_amd: the persistent BiLSTM kernels' 64 workgroups do not become co-resident on %s (a shared or partitioned GPU?): "
                          "recording the per-step recurrence instead (slower, never wrong)" % key, RuntimeWarning, stacklevel=2)
    return _LSTM_CORESIDENT[key]


def lstm_seq_probe_reset(device=None):
    """forget the cached co-residency answer (of one device, or of all): the next recorded plan asks again -- for a process whose GPU
    becomes shared or un-shared during its lifetime.  Plans already recorded keep the form they were recorded with (clear the engines'
    plan caches, `engine._plans.clear()`, to re-record).  Under a stream capture the question cannot be asked (a capture cannot
    synchronise) and the answer is taken as yes: capture relies on the eager warm-up steps having probed."""
    if device is None:
        _LSTM_CORESIDENT.clear()
    else:
        _LSTM_CORESIDENT.pop(str(device), None)


# the five weight-gradient GEMMs of a BidirectionalLSTM layer as one launch (tpgsr_conv_wgrad_batch); 0: five launches
LSTM_WGRAD_BATCH = os.environ.get("TPGSR_LSTM_WGRAD_BATCH", "1") == "1"


# fused recurrent projection + gate step (Hh == 256, N <= 64): one 32-workgroup launch per time step instead of a 128-workgroup
# split-K GEMM + a 96-workgroup gate kernel.  Correct, measured slower (C3 10.18 vs 10.02 ms/step): with one wave per SIMD on 32 CUs
# every L2 round trip of the step is exposed, the two wide launches hide them -- opt-in
LSTM_STEPX = os.environ.get("TPGSR_LSTM_STEPX", "0") == "1"


def lstm_wfrag(whhT, wfr, Hh):
    _launch("tpgsr_lstm_wfrag", _p(whhT), _p(wfr), Hh)


def lstm_stepx_fwd(G, wfr, bhh, Cst, out, hx, N, T, Hh, step):
    _launch("tpgsr_lstm_stepx_fwd", _p(G), _p(wfr), _p(bhh), _p(Cst), _p(out), _p(hx), N, T, Hh, step)


def lstm_stepx_buffers(device):
    """(wfr, hx): fragment planes of W_hh^T (rebuilt per pass) and the zero-initialised h exchange buffer"""
    lib = _lib.load()
    return (torch.empty(lib.tpgsr_lstm_wfrag_bytes(), dtype=torch.uint8, device=device),
            torch.zeros(lib.tpgsr_lstm_seq_hx_bytes(), dtype=torch.uint8, device=device))


def lstm_seq_fwd(G, whhT, bhh, Cst, out, hx, sync, N, T, Hh):
    _launch("tpgsr_lstm_seq_fwd", _p(G), _p(whhT), _p(bhh), _p(Cst), _p(out), _p(hx), _p(sync), N, T, Hh)


def lstm_seq_buffers(device):
    """(hx, sync) for lstm_seq_fwd: the exchange buffer must start zeroed (rows of sequences >= N are never written)"""
    hx = torch.zeros(_lib.load().tpgsr_lstm_seq_hx_bytes(), dtype=torch.uint8, device=device)
    return hx, torch.zeros(4, dtype=torch.int32, device=device)


# data-tagged hand-off for the persistent forward kernel (8-byte {h terms, tag} granules, no arrival counter): TPGSR_LSTM_GRANULE=0 -> counter form
LSTM_GRANULE = os.environ.get("TPGSR_LSTM_GRANULE", "1") == "1"
LSTM_GRANULE_BWD = os.environ.get("TPGSR_LSTM_GRANULE_BWD", "0") == "1"


def lstm_seq_fwdg(G, whhT, bhh, Cst, out, hg, sync, N, T, Hh):
    _launch("tpgsr_lstm_seq_fwdg", _p(G), _p(whhT), _p(bhh), _p(Cst), _p(out), _p(hg), _p(sync), N, T, Hh)


def lstm_seq_granule_buffers(device):
    """(hg, sync) for lstm_seq_fwdg: both zeroed ONCE here, then owned by the launches (tags / launch epoch)"""
    return (torch.zeros(_lib.load().tpgsr_lstm_seq_hg_bytes(), dtype=torch.uint8, device=device),
            torch.zeros(8, dtype=torch.int32, device=device))


def lstm_seq_bwd(G, Cst, dout, w0, w1, px, sync, N, T, Hh):
    _launch("tpgsr_lstm_seq_bwd", _p(G), _p(Cst), _p(dout), _p(w0), _p(w1), _p(px), _p(sync), N, T, Hh)


def lstm_seq_bwdg(G, Cst, dout, w0, w1, pg, sync, N, T, Hh):
    _launch("tpgsr_lstm_seq_bwdg", _p(G), _p(Cst), _p(dout), _p(w0), _p(w1), _p(pg), _p(sync), N, T, Hh)


def lstm_seq_bwd_granule_buffers(device):
    """(pg, sync) for lstm_seq_bwdg: zeroed ONCE here, then owned by the launches"""
    return (torch.zeros(_lib.load().tpgsr_lstm_seq_pg_bytes(), dtype=torch.uint8, device=device),
            torch.zeros(8, dtype=torch.int32, device=device))


def lstm_seq_bwd_buffers(device):
    """(px, sync) for lstm_seq_bwd (no initialisation needed: every word read in a step was written in the step before)"""
    px = torch.empty(_lib.load().tpgsr_lstm_seq_px_bytes(), dtype=torch.uint8, device=device)
    return px, torch.zeros(4, dtype=torch.int32, device=device)


def lstm_step_fwd(G, gh, nsplit, bhh, Cst, out, N, T, Hh, step):
    _launch("tpgsr_lstm_step_fwd", _p(G), _p(gh), nsplit, _p(bhh), _p(Cst), _p(out), N, T, Hh, step)


def lstm_step_bwd(G, Cst, dout, dhc, nsplit, dcc, N, T, Hh, step):
    _launch("tpgsr_lstm_step_bwd", _p(G), _p(Cst), _p(dout), _p(dhc), nsplit, _p(dcc), N, T, Hh, step)


def softmax_prior_fwd(logits, q, N, T, C_, drop_n, p, prior, partial, nblk):
    _launch("tpgsr_softmax_prior_fwd", _p(logits), _p(q), N, T, C_, drop_n, _p(p), _p(prior), _p(partial), nblk)


def semantic_loss_finalize(partial, nblk, count, w, loss):
    _launch("tpgsr_semantic_loss_finalize", _p(partial), nblk, count, w, _p(loss))


def softmax_prior_bwd(p, q, dprior, dp_in, N, T, C_, drop_n, wsem, dlogits, nblk):
    _launch("tpgsr_softmax_prior_bwd", _p(p), _p(q), _p(dprior), _p(dp_in), N, T, C_, drop_n, wsem, _p(dlogits), nblk)


# ---- tail / loss / optimiser ----------------------------------------------------------------------------------
def tail_shiftsum_tanh(P, bias, N, H, W, Co, KS, out_nchw):
    _launch("tpgsr_tail_shiftsum_tanh", _p(P), _p(bias), N, H, W, Co, KS, _p(out_nchw))


def shiftsum_nhwc(P, N, H, W, Co, KS, out):
    _launch("tpgsr_shiftsum_nhwc", _p(P), N, H, W, Co, KS, _p(out))


def tail_bwd_blocks(N, H, W, Co, KS) -> int:
    return _lib.load().tpgsr_tail_bwd_blocks(N, H, W, Co, KS)


def tail_bwd(out_nchw, dout_nchw, N, H, W, Co, KS, dP, dbias_partial, nblk):
    _launch("tpgsr_tail_bwd", _p(out_nchw), _p(dout_nchw), N, H, W, Co, KS, _p(dP), _p(dbias_partial), nblk)


def image_loss_fwd(out, tgt, N, C_, H, W, gradient, partial, nblk):
    _launch("tpgsr_image_loss_fwd", _p(out), _p(tgt), N, C_, H, W, int(gradient), _p(partial), nblk)


def image_loss_finalize(partial, nblk, n_mse, n_gp, w0, w1, loss):
    _launch("tpgsr_image_loss_finalize", _p(partial), nblk, n_mse, n_gp, w0, w1, _p(loss))


def image_loss_bwd(out, tgt, dloss, N, C_, H, W, gradient, w0, w1, dout):
    _launch("tpgsr_image_loss_bwd", _p(out), _p(tgt), _p(dloss), N, C_, H, W, int(gradient), w0, w1, _p(dout))


def sumsq_partial(x, n, partial, nblk):
    _launch("tpgsr_sumsq_partial", _p(x), n, _p(partial), nblk)


def clip_coef(partial, nblk, max_norm, coef, norm_out=None):
    _launch("tpgsr_clip_coef", _p(partial), nblk, max_norm, _p(coef), _p(norm_out))


def adam_step(p, g, m, v, n, gscale, lr, beta1, beta2, eps, step_dev):
    _launch("tpgsr_adam_step", _p(p), _p(g), _p(m), _p(v), n, _p(gscale), lr, beta1, beta2, eps, _p(step_dev))


def step_inc(step_dev):
    _launch("tpgsr_step_inc", _p(step_dev))


def clip_coef_steps(partial, nblk, max_norm, coef, norm_out, step_devs):
    """clip_coef (partial None: skipped) + step_inc of every counter in `step_devs` (<= 8 device tensors) in one launch"""
    arr = (C.c_void_p * max(1, len(step_devs)))(*[t.data_ptr() for t in step_devs])
    _launch("tpgsr_clip_coef_steps", _p(partial), nblk, max_norm, _p(coef), _p(norm_out), C.cast(arr, C.c_void_p), len(step_devs))


def scale_(x, n, coef):
    _launch("tpgsr_scale_", _p(x), n, _p(coef))


def im2col3x3_c1(inp, N, H, W, col):
    _launch("tpgsr_im2col3x3_c1", _p(inp), N, H, W, _p(col))


def col2im3x3_c1(dcol, N, H, W, din):
    _launch("tpgsr_col2im3x3_c1", _p(dcol), N, H, W, _p(din))


def pad_channels(src, M, Cs, Cd, dst):
    _launch("tpgsr_pad_channels", _p(src), M, Cs, Cd, _p(dst))


def semantic_loss_fwd(p, q, n, partial, nblk):
    _launch("tpgsr_semantic_loss_fwd", _p(p), _p(q), n, _p(partial), nblk)


def semantic_loss_bwd(p, q, dloss, n, dp):
    _launch("tpgsr_semantic_loss_bwd", _p(p), _p(q), _p(dloss), n, _p(dp))


def make_bf_twin(wt: torch.Tensor, cin: int = 0):
    """split ONE packed fp32 operand [K][ld] into its bf16 planes now (single-op callers / tests; the engines batch all their
    operands into one tpgsr_split_bf_program launch per step).  cin: input channels of the convolution that will consume it
    (rows go in channel-block order when block_order_cin says so)"""
    lib = _lib.load()
    Kd, N = wt.shape
    cin = block_order_cin(Kd, cin)
    kp = (Kd + 31) // 32 * 32
    twin = torch.zeros(3 * ((N + 31) // 32 * 32) * kp, dtype=torch.bfloat16, device=wt.device)
    d = (_lib.SplitDesc * 1)()
    d[0].src, d[0].dst, d[0].K, d[0].N, d[0].ld, d[0].kp, d[0].blk0, d[0].cin = wt.data_ptr(), twin.data_ptr(), Kd, N, N, kp, 0, cin
    table = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(wt.device)
    split_bf_program(table, 1, lib.tpgsr_split_bf_blocks(Kd, N))
    register_bf_twin(wt, twin, kp, cin)
    return twin, kp


def split_bf_program(descs_dev, ndesc, total_blocks):
    _launch("tpgsr_split_bf_program", _p(descs_dev), ndesc, total_blocks)


# ---- layout / resampling glue (csrc/glue.hip) ---------------------------------------------------------------------
def copy_strided(src, src_ld, src_coff, dst, dst_ld, dst_coff, M, C_, accumulate=False):
    _launch("tpgsr_copy_strided", _p(src), src_ld, src_coff, _p(dst), dst_ld, dst_coff, M, C_, int(accumulate))


def resize_nearest_fwd(inp, N, H, W, C_, s, out):
    _launch("tpgsr_resize_nearest_fwd", _p(inp), N, H, W, C_, s, _p(out))


def resize_nearest_bwd(dout, N, H, W, C_, s, din):
    _launch("tpgsr_resize_nearest_bwd", _p(dout), N, H, W, C_, s, _p(din))


def resize_bilinear_fwd(inp, N, H, W, C_, OH, OW, out):
    _launch("tpgsr_resize_bilinear_fwd", _p(inp), N, H, W, C_, OH, OW, _p(out))


def resize_bilinear_bwd(dout, N, H, W, C_, OH, OW, din):
    _launch("tpgsr_resize_bilinear_bwd", _p(dout), N, H, W, C_, OH, OW, _p(din))


def dilate2d(inp, N, H, W, C_, sh, sw, out):
    _launch("tpgsr_dilate2d", _p(inp), N, H, W, C_, sh, sw, _p(out))


def subsample2d(inp, N, H, W, C_, sh, sw, out):
    _launch("tpgsr_subsample2d", _p(inp), N, H, W, C_, sh, sw, _p(out))


def hreduce(inp, N, H, W, C_, scale, out):
    _launch("tpgsr_hreduce", _p(inp), N, H, W, C_, scale, _p(out))


def hbroadcast(dout, N, H, W, C_, scale, din):
    _launch("tpgsr_hbroadcast", _p(dout), N, H, W, C_, scale, _p(din))


# ---- evaluation path (csrc/metrics.hip) ---------------------------------------------------------------------------
def ctc_greedy_decode(logits, N, T, C_, labels, lengths):
    _launch("tpgsr_ctc_greedy_decode", _p(logits), N, T, C_, _p(labels), _p(lengths))


def psnr(a, b, N, Ctot, H, W, partial, nblk, out):
    _launch("tpgsr_psnr", _p(a), _p(b), N, Ctot, H, W, _p(partial), nblk, _p(out))


def ssim(a, b, window, KS, N, Ctot, H, W, partial, nblk, out):
    _launch("tpgsr_ssim", _p(a), _p(b), _p(window), KS, N, Ctot, H, W, _p(partial), nblk, _p(out))


def ctc_loss(logits, sn, st, targets, tgt_off, tgt_len, weight, N, T, C_, blank, scale, nll, dlogits, accumulate, max_len):
    """per-sample CTC negative log-likelihood (+ its gradient, scaled by scale * weight[n], into dlogits): csrc/crnn.hip, `--use_label`"""
    _launch("tpgsr_ctc_loss", _p(logits), int(sn), int(st), _p(targets), _p(tgt_off), _p(tgt_len), _p(weight), N, T, C_, int(blank), float(scale),
            _p(nll), _p(dlogits), int(bool(accumulate)), int(max_len))


def ssim_bwd(a, b, window, KS, N, Ctot, H, W, gm, coef, mult, da, accumulate):
    """da[:, :min(Ctot,3)] (+)= mult * coef[0] * d(sum of the SSIM map)/da  (`--ssim_loss`; gm: scratch 3 N min(Ctot,3) H W floats)"""
    _launch("tpgsr_ssim_bwd", _p(a), _p(b), _p(window), KS, N, Ctot, H, W, _p(gm), _p(coef), float(mult), _p(da), int(bool(accumulate)))


# ---- ASTER evaluation recognizer, greedy decode (csrc/aster.hip) ----------------------------------------------------
def bicubic_resize(x_nchw, N, Ctot, C_, H, W, OH, OW, scale, shift, out_nhwc):
    _launch("tpgsr_bicubic_resize", _p(x_nchw), N, Ctot, C_, H, W, OH, OW, float(scale), float(shift), _p(out_nhwc))


def aster_attention(xproj, sproj, wv, bv, x, N, T, A, D, alpha, context):
    _launch("tpgsr_aster_attention", _p(xproj), _p(sproj), _p(wv), _p(bv), _p(x), N, T, A, D, _p(alpha), _p(context))


def embed_concat(ids, emb, V, E, ctx, D, N, out):
    _launch("tpgsr_embed_concat", _p(ids), _p(emb), V, E, _p(ctx), D, N, _p(out))


def gru_cell(gi, gh, h, N, Hd, hnew):
    _launch("tpgsr_gru_cell", _p(gi), _p(gh), _p(h), N, Hd, _p(hnew))


def softmax_max(logits, N, C_, ids, score, ld, col, ids_next=None):
    _launch("tpgsr_softmax_max", _p(logits), N, C_, _p(ids), _p(score), ld, col, _p(ids_next))
