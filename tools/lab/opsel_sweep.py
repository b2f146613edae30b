#!/usr/bin/env python
"""Round 6: sweep EVERY kernel source for the operand form behind the unrepeatable GruBlock forward (profiles/r06_gru_proj_root_cause.md):
a packed VALU instruction whose op_sel takes a HIGH register of a source pair for its low half.  Compiles csrc/*.hip to gfx950 assembly
(six at a time, a few minutes on 8 cores; no GPU), lists the hits per source and instruction, and names the instruction that last wrote the
selected odd register: a hit fed by an LDS / memory return (ds_read_*, buffer_load_*, global_load_*) is the dangerous kind; one fed by a VALU
result is interlocked like any other VALU dependency.   usage: python tools/lab/opsel_sweep.py [outdir]"""
import collections
import glob
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tpgsr_amd import build as B  # noqa: E402


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/opsel_sweep"
    os.makedirs(out, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(ROOT, "tpgsr_amd", "csrc", "*.hip")))

    def compile_one(src):
        dst = os.path.join(out, os.path.basename(src) + ".s")
        r = subprocess.run([B._hipcc()] + B.FLAGS + ["-x", "hip", "-S", "--cuda-device-only", src, "-o", dst], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return dst
    with ThreadPoolExecutor(6) as ex:
        asms = list(ex.map(compile_one, srcs))
    total = collections.Counter()
    for f in asms:
        k = open(f).read().splitlines()
        packed = 0
        for i, ln in enumerate(k):
            m = re.match(r"\s*(v_pk_\w+) (.*)", ln)
            if not m:
                continue
            packed += 1
            if not re.search(r"op_sel:\[(0,1|1)", ln):
                continue
            ops = [t.strip() for t in m.group(2).split(" op_sel")[0].split(",")]
            sel = re.search(r"op_sel:\[([01,]+)\]", ln).group(1).split(",")
            for si, s in enumerate(ops[1:]):
                if si < len(sel) and sel[si] == "1":
                    rr = regs(s)
                    if len(rr) < 2:
                        continue
                    writer = "?"
                    for j in range(i - 1, max(0, i - 400), -1):
                        mm = re.match(r"\s*(\w+) ([^,]+)", k[j])
                        if mm and rr[1] in regs(mm.group(2).strip()):
                            writer = mm.group(1)
                            break
                    total[(os.path.basename(f)[:-2], m.group(1), re.sub(r"_e(32|64)$", "", writer))] += 1
        print(f"{os.path.basename(f)[:-2]:22s} packed instructions {packed:6d}")
    print("\nhits (source, instruction, last writer of the selected high register):")
    for key, v in sorted(total.items(), key=lambda kv: -kv[1]):
        danger = re.match(r"(ds_read|buffer_load|global_load|flat_load|scratch_load)", key[2]) is not None
        print(f"  {key[0]:20s} {key[1]:16s} <- {key[2]:20s} x{v}{'   <-- fed by a memory return' if danger else ''}")
    if not total:
        print("  none")


if __name__ == "__main__":
    main()
