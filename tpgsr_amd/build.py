"""Build libtpgsr_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtpgsr_hip.so")
SOURCES = ["conv_mfma.hip", "conv_xbf.hip", "conv_panel.hip", "conv_halo3.hip", "elementwise.hip", "gru.hip", "gru_proj.hip", "gru_wgrad.hip", "stn.hip", "loss_optim.hip", "crnn.hip", "lstm_seq.hip", "glue.hip", "metrics.hip", "preprocess.hip", "aster.hip", "error.cpp", "plan.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("TPGSR_FAST_MATH"):   # A/B switch only: v_exp_f32 / v_rcp_f32 activations (costs gradient parity, see common.h)
    FLAGS.append("-DTPGSR_FAST_MATH")
if os.environ.get("TPGSR_LAB"):         # lab build: compiles the kernels' debug switches in (tpgsr_wgh_debug, tools/lab/wgh_probe.py)
    FLAGS.append("-DTPGSR_LAB")

if os.environ.get("TPGSR_FRAG_PRELOAD") is not None:   # A/B switch: conv fragment preload (conv_mfma.hip)
    FLAGS.append("-DTPGSR_FRAG_PRELOAD=" + os.environ["TPGSR_FRAG_PRELOAD"])


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _headers_digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + ["../../include/tpgsr_hip.h"]:
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and p.endswith(".h"):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _src_digest(src, hdr):
    return hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + hdr.encode()).hexdigest()


def _digest():
    hdr = _headers_digest()
    return hashlib.sha256("".join(_src_digest(s, hdr) for s in SOURCES).encode()).hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Incremental: an object is recompiled when its source, any header under csrc/ (or include/tpgsr_hip.h) or the flags changed --
    per-object stamps under build/, the library's stamp beside it."""
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr = _headers_digest()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        sd = _src_digest(src, hdr)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".stamp") and open(obj + ".stamp").read() == sd:
            continue
        cmd = [_hipcc()] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, sd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, sd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        with open(obj + ".stamp", "w") as f:
            f.write(sd)
        if verbose and out:
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
