#!/usr/bin/env python
"""LAB (round 6): the staged one-launch GruBlock forward (tools/lab/gp_staged_lab.hip) at full batch, the failing case
of round 5 (affine + text-strip loader, H-axis scan, two-term arithmetic): repeated launches, then WHERE and WHAT differs --
  * which sequences' h / gates differ from the majority result,
  * whether the panel fragments (xf checksum per wave), the gi block right after the barrier, or the gi block after the scan differ,
  * the differing cells' rows / columns / values next to the majority values,
  * which CU / LDS range the workgroup and its co-resident neighbours had (HW_ID, XCC_ID, LDS_ALLOC, wall clock).
Build first (no GPU needed): hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -x hip -c tools/lab/gp_staged_lab.hip -o /tmp/gp_lab.o; ... -shared
usage: python tools/lab/gp_staged_probe.py [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tpgsr_amd import _lib, kernels as K  # noqa: E402
from test_gru_proj_gpu import _case  # noqa: E402

DEV = "cuda"
N, H, W = 48, 16, 64
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
vp, ci = C.c_void_p, C.c_int


class LabArgs(C.Structure):
    _fields_ = [("p", _lib.BigruProjArgs), ("dump_x", vp), ("dump_g1", vp), ("dump_g2", vp), ("dump_s", vp), ("dump_i", vp), ("mode", ci), ("extra", ci)]


lab = C.CDLL(os.path.join(ROOT, "tools", "lab", "libgp_lab.so"))
lab.gp_lab_launch.argtypes = [C.POINTER(LabArgs), ci, vp]
lab.gp_lab_launch.restype = ci
lab.tpgsr_last_error.restype = C.c_char_p


def launch(pargs, variant, mode=0, extra=0, dumps=None):
    la = LabArgs()
    la.p = pargs
    la.mode, la.extra = mode, extra
    if dumps:
        la.dump_x, la.dump_g1, la.dump_g2, la.dump_i = (d.data_ptr() for d in dumps)
    rc = lab.gp_lab_launch(C.byref(la), variant, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lab.tpgsr_last_error()


def majority(ts):
    """element-wise: the value most launches agree on (median of the bit patterns works for 'one launch differs')"""
    st = torch.stack([t.view(torch.int32) for t in ts])
    return st.median(0).values.view(torch.float32)


def run_case(loader, terms, variant, extra=0, mode=0, reps=REPS, verbose=True):
    Cin = 96 if loader == "affine+strip" else 64
    axis = 1
    t, kw = _case(N, H, W, Cin, loader, seed=5)
    P = N * H * W
    geom = K.ConvGeom(N, H, W, Cin, 192)
    nwg = N * W // 4
    with K.conv_terms(terms):
        K.make_bf_twin(t["wc"], 0)
        outs, dumps_all = [], []
        for rep in range(reps):
            h, gt = torch.full((P, 64), float("nan"), device=DEV), torch.full((P, 512 if variant == 4 else 256), float("nan"), device=DEV)
            dumps = None
            if variant == 11:
                dumps = (torch.zeros(4, dtype=torch.int32, device=DEV), torch.zeros(4, device=DEV),
                         torch.full((nwg * 4 * 96 * 64,), float("nan"), device=DEV), torch.zeros(8, dtype=torch.int32, device=DEV))
            elif mode:
                dumps = (torch.zeros(nwg * 4 * 64, dtype=torch.int32, device=DEV), torch.zeros(nwg * 64 * 192, device=DEV),
                         torch.zeros(nwg * 64 * 192, device=DEV), torch.zeros(nwg * 4 * 8, dtype=torch.int32, device=DEV))
            pa = K.make_bigru_proj_args(K.make_conv_args(geom, t["x"], t["wc"], None, bias=t["bc"], **kw), t["whh"], t["bhh"], axis, h, gt)
            launch(pa, variant, mode, extra, dumps)
            outs.append((h, gt))
            dumps_all.append(dumps)
    torch.cuda.synchronize()
    hm = majority([o[0] for o in outs])
    nbad_total, events = 0, []
    for i, (h, gt) in enumerate(outs):
        diff = (h.view(torch.int32) != hm.view(torch.int32))
        if diff.any():
            pix = diff.nonzero()[:, 0].unique()
            n, r = pix // (H * W), pix % (H * W)
            seqs = (n * W + r % W).unique()
            nbad_total += len(seqs)
            events.append((i, seqs.tolist()))
    print(f"== {loader} x{terms} variant {variant} extra LDS {extra} mode {mode}: {reps} launches, {len(events)} with a sequence off the majority, "
          f"{nbad_total} (launch, sequence) events; NaN left in h: {int(torch.isnan(outs[0][0]).sum())}", flush=True)
    if not verbose:
        return nbad_total
    shown = 0
    gm = None
    for i, seqs in events:
        for s in seqs:
            if shown >= 8:
                break
            shown += 1
            n, col = s // W, s % W
            wg, wave = s // 4, s % 4
            hh = outs[i][0].view(N, H, W, 64)[n, :, col]          # [T][64]
            hr = hm.view(N, H, W, 64)[n, :, col]
            d = (hh.view(torch.int32) != hr.view(torch.int32))
            steps_f = d[:, :32].any(1).nonzero().flatten().tolist()
            steps_r = d[:, 32:].any(1).nonzero().flatten().tolist()
            print(f"-- launch {i} sequence {s} (wg {wg} wave {wave}): h differs at rows (forward half) {steps_f} (reverse half) {steps_r}; "
                  f"max |dh| {float((hh - hr).abs().max()):.3e}")
            NG = 8 if variant == 4 else 4
            if gm is None:
                gm = majority([o[1] for o in outs])
            gg = outs[i][1].view(N, H, W, 2, NG, 32)[n, :, col]      # [T][dir][r z n an (ir iz in rz.x)][32]
            gr = gm.view(N, H, W, 2, NG, 32)[n, :, col]
            dg = (gg.view(torch.int32) != gr.view(torch.int32))
            first = dg.nonzero()
            if len(first):
                # the first damaged scan step per direction
                for dd in (0, 1):
                    rows = dg[:, dd].any(-1).any(-1).nonzero().flatten().tolist()
                    if rows:
                        row0 = rows[0] if dd == 0 else rows[-1]
                        which = dg[row0, dd].any(-1).tolist()
                        units = dg[row0, dd].any(0).nonzero().flatten().tolist()
                        print(f"   dir {dd}: first damaged row {row0}: gates (r z n an [ir iz in rz.x]) {which}, units {units[:32]}")
                        if variant == 4:
                            u = units[0]
                            names = ["r", "z", "n", "an", "ir", "iz", "in", "rz.x"]
                            print("      unit", u, " ".join(f"{nm}: {float(gg[row0, dd, k, u]):+.7e} / {float(gr[row0, dd, k, u]):+.7e}" for k, nm in enumerate(names)))
                            # which terms of the hidden-side product explain the damage?  delta_j = sum_k W_hr[j][k] dh[k] over the damaged lanes j:
                            # per group of four k (one ds_read_b128 of the exchange), least squares with 16 equations and 4 unknowns
                            dl = (gg[row0, dd, 7, 16:32] - gr[row0, dd, 7, 16:32]).double().cpu()
                            Wr = t["whh"][dd, 16:32, :].double().cpu()                   # W_hr rows of units 16..31: [16][32]
                            prev_row = row0 + 1 if dd == 1 else row0 - 1                  # the step before (its h is what this step multiplies)
                            hprev = hr[prev_row, 32 * dd:32 * dd + 32].double().cpu() if 0 <= prev_row < H else torch.zeros(32, dtype=torch.float64)
                            best = []
                            for g4 in range(8):
                                A = Wr[:, 4 * g4:4 * g4 + 4]
                                sol = torch.linalg.lstsq(A, dl.unsqueeze(1)).solution.squeeze(1)
                                res = float((A @ sol - dl).norm() / dl.norm())
                                best.append((res, g4, sol))
                            best.sort(key=lambda x: x[0])
                            res, g4, sol = best[0]
                            print(f"      damage of rz.x over units 16..31 explained by ONE group of four h values: best group k = {4 * g4}..{4 * g4 + 3} relative residual {res:.2e} "
                                  f"(next best {best[1][0]:.2e}); implied h used = {[f'{float(hprev[4 * g4 + q] + sol[q]):+.6f}' for q in range(4)]} instead of {[f'{float(hprev[4 * g4 + q]):+.6f}' for q in range(4)]}")
                            # do the implied values occur anywhere among this sequence's h values (any row, this direction)?
                            allh = hr[:, 32 * dd:32 * dd + 32].double().cpu()
                            for q in range(4):
                                v = float(hprev[4 * g4 + q] + sol[q])
                                idx = (allh - v).abs().argmin()
                                print(f"         k {4 * g4 + q}: nearest h of this sequence / direction: row {int(idx) // 32} unit {int(idx) % 32} ({float(allh.flatten()[idx]):+.6f}, off by {float((allh.flatten()[idx] - v).abs()):.1e})")
                            # is the wrong ir some OTHER cell's value?  search the majority ir values of this sequence (all rows, both directions)
                            bad_ir = gg[row0, dd, 4, u]
                            hits = (gr[:, :, 4:7].view(torch.int32) == bad_ir.view(torch.int32)).nonzero().tolist()
                            print(f"      the consumed ir (0x{int(bad_ir.view(torch.int32)) & 0xffffffff:08x}) equals the expected input of (row, dir, [ir iz in], unit): {hits[:6]}")
            if variant == 11:
                wd = dumps_all[i][2].view(nwg, 4, 96, 64)[wg, wave]                 # [96][lane]: wrz[k].x, wrz[k].y interleaved (64 rows), then wn2 pairs (32 rows)
                whh = t["whh"]                                                         # [2][96][32]
                lanes = torch.arange(64, device=DEV)
                dl, jl = lanes // 32, lanes % 32
                exp = torch.empty(96, 64, device=DEV)
                for k in range(32):
                    exp[2 * k] = whh[dl, jl, k]              # W_hr[j][k]
                    exp[2 * k + 1] = whh[dl, 32 + jl, k]     # W_hz[j][k]
                    exp[64 + k] = whh[dl, 64 + jl, k]        # W_hn[j][k] (pairs (2k', 2k'+1) in order)
                bad = (wd.view(torch.int32) != exp.view(torch.int32))
                print(f"   recurrent weights in registers after the scan: {int(bad.sum())} of {bad.numel()} differ from W_hh; rows (0-63: r/z interleaved by k, 64-95: n) "
                      f"{bad.any(1).nonzero().flatten().tolist()[:24]} lanes {bad.any(0).nonzero().flatten().tolist()}")
                for r_, l_ in bad.nonzero()[:6].tolist():
                    v = wd[r_, l_]
                    where = (whh.view(torch.int32) == v.view(torch.int32)).nonzero().tolist()[:3]
                    print(f"      row {r_} lane {l_}: holds {float(v):+.7e} (0x{int(v.view(torch.int32)) & 0xffffffff:08x}), expected {float(exp[r_, l_]):+.7e}; that value occurs in W_hh at [dir][row][k] {where}")
            elif mode:
                dx, g1, g2, di = dumps_all[i]
                # majority dumps over the OTHER launches
                g1m = majority([dumps_all[k][1] for k in range(reps)]).view(nwg, 64, 192)
                xm = torch.stack([dumps_all[k][0] for k in range(reps)]).median(0).values.view(nwg, 4, 64)
                g1w, g2w = g1.view(nwg, 64, 192)[wg], g2.view(nwg, 64, 192)[wg]
                dd1 = (g1w.view(torch.int32) != g1m[wg].view(torch.int32))
                dd12 = (g1w.view(torch.int32) != g2w.view(torch.int32))
                dxw = (dx.view(nwg, 4, 64)[wg] != xm[wg])
                print(f"   xf checksum differs from majority in waves {dxw.any(1).nonzero().flatten().tolist()} lanes {dxw.any(0).nonzero().flatten().tolist()[:16]}; "
                      f"xf checksums equal across the 4 waves: {bool((dx.view(nwg, 4, 64)[wg] == dx.view(nwg, 4, 64)[wg][0]).all())}")
                print(f"   gi right after the barrier differs from majority in {int(dd1.sum())} cells; gi after the scan differs from gi at the barrier in {int(dd12.sum())} cells")
                for name, dm, ref in (("at barrier vs majority", dd1, g1m[wg]), ("after scan vs at barrier", dd12, g1w)):
                    if dm.any():
                        rows = dm.any(1).nonzero().flatten().tolist()
                        cols = dm.any(0).nonzero().flatten().tolist()
                        print(f"      {name}: rows {rows} cols {cols[:48]}{'...' if len(cols) > 48 else ''}")
                        rc = dm.nonzero()[:6]
                        for r_, c_ in rc.tolist():
                            cur = (g1w if name.startswith("at") else g2w)[r_, c_]
                            print(f"         [{r_}][{c_}] = {float(cur):+.6e} (0x{int(cur.view(torch.int32)) & 0xffffffff:08x}) expected {float(ref[r_, c_]):+.6e} (0x{int(ref[r_, c_].view(torch.int32)) & 0xffffffff:08x})")
                info = di.view(nwg, 4, 8).cpu()

                def dec(rec):
                    hw, xcc, la = int(rec[0]) & 0xffffffff, int(rec[1]) & 15, int(rec[2]) & 0xffffffff
                    t0 = (int(rec[4]) & 0xffffffff) | ((int(rec[5]) & 0xffffffff) << 32)
                    t1 = (int(rec[6]) & 0xffffffff) | ((int(rec[7]) & 0xffffffff) << 32)
                    return dict(wave_slot=hw & 15, simd=(hw >> 4) & 3, cu=(hw >> 8) & 15, sh=(hw >> 12) & 1, se=(hw >> 13) & 7, xcc=xcc,
                                lds_base=la & 0xff, lds_size=(la >> 12) & 0x1ff, raw_lds=la, t0=t0, t1=t1)
                me = [dec(info[wg, w]) for w in range(4)]
                print(f"   placement: xcc {me[0]['xcc']} se {me[0]['se']} sh {me[0]['sh']} cu {me[0]['cu']} simds {[m['simd'] for m in me]} slots {[m['wave_slot'] for m in me]} "
                      f"LDS_ALLOC base {me[0]['lds_base']} size {me[0]['lds_size']} (raw 0x{me[0]['raw_lds']:08x}); wave lifetimes (ticks) {[m['t1'] - m['t0'] for m in me]}")
                key = (me[0]["xcc"], me[0]["se"], me[0]["sh"], me[0]["cu"])
                t0 = min(m["t0"] for m in me)
                t1 = max(m["t1"] for m in me)
                nb = []
                for o in range(nwg):
                    if o == wg:
                        continue
                    r0 = dec(info[o, 0])
                    if (r0["xcc"], r0["se"], r0["sh"], r0["cu"]) == key:
                        ot0 = min(dec(info[o, w])["t0"] for w in range(4))
                        ot1 = max(dec(info[o, w])["t1"] for w in range(4))
                        if ot0 < t1 and ot1 > t0:
                            nb.append((o, r0["lds_base"], r0["lds_size"], ot0 - t0, ot1 - t0, [dec(info[o, w])["simd"] for w in range(4)]))
                print(f"   co-resident workgroups on that CU (wg, LDS base, size, start - my start, end - my start, simds): {nb}")
    return nbad_total


def time_case(loader, terms, variant, reps=30):
    Cin = 96 if loader == "affine+strip" else 64
    t, kw = _case(N, H, W, Cin, loader, seed=5)
    P = N * H * W
    geom = K.ConvGeom(N, H, W, Cin, 192)
    with K.conv_terms(terms):
        K.make_bf_twin(t["wc"], 0)
        h, gt = torch.empty(P, 64, device=DEV), torch.empty(P, 512 if variant == 4 else 256, device=DEV)
        pa = K.make_bigru_proj_args(K.make_conv_args(geom, t["x"], t["wc"], None, bias=t["bc"], **kw), t["whh"], t["bhh"], 1, h, gt)
        for _ in range(5):
            launch(pa, variant)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch(pa, variant)
        e1.record()
        torch.cuda.synchronize()
    print(f"time: {loader} x{terms} variant {variant}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch", flush=True)


def main():
    torch.manual_seed(0)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    what = sys.argv[2] if len(sys.argv) > 2 else "round2"
    if what == "round1":
        # 1. does it reproduce in the lab copy, and which structural variants cure it?
        for loader, terms in (("affine+strip", 2), ("affine+strip", 3), ("affine", 2)):
            for variant in (0, 1, 2, 3):
                run_case(loader, terms, variant, verbose=False)
        run_case("affine+strip", 2, 0, extra=30 * 1024, verbose=False)       # one workgroup per CU (round 5's last data point)
        run_case("affine+strip", 2, 0, extra=2 * 1024, verbose=False)        # a little more LDS (still several per CU)
        # 2. the failing case with the dumps on
        run_case("affine+strip", 2, 0, mode=1 | 2 | 4 | 16, verbose=True)
        run_case("affine+strip", 3, 0, mode=1 | 2 | 4 | 16, verbose=True)
        # 3. without dumps but verbose (timing undisturbed): where in h / gates
        run_case("affine+strip", 2, 0, mode=0, verbose=True)
        return
    # round 2: gi in LDS is right before and after the scan, yet the r gate of units 16-31 of the reverse direction comes out wrong at
    # steps 2-5, only in the first-dispatched workgroup of a CU.  What went into the gate, and which change to the scan cures it?
    if what == "round2":
        for variant in (0, 5, 6, 7, 8, 9, 10, 4):
            run_case("affine+strip", 2, variant, verbose=False)
        run_case("affine+strip", 2, 4, verbose=True)
        run_case("affine+strip", 3, 4, verbose=True)
        for variant in (0, 5, 6, 7, 8, 9, 10):
            run_case("affine+strip", 3, variant, verbose=False)
        return
    # round 3: the inputs of the damaged gate are right, its hidden-side product W_hr h is wrong (lanes 48-63), W_hz h of the SAME packed
    # instructions is right: are the W_hr registers themselves damaged when the scan ends, and with what?
    if what == "round3":
        run_case("affine+strip", 2, 11, verbose=True)
        run_case("affine+strip", 3, 11, verbose=True)
        return
    if what == "round5":
        # the trigger: is it the packed instruction whose LOW half reads the ODD register of a just-returned pair (op_sel), or any consumer behind a
        # partial wait?  14 = no op_sel cross-read, partial waits kept; 16 = hand-written partial waits with all eight reads in flight; + time per launch
        for rep in range(2):
            for variant in (0, 14, 16, 13):
                run_case("affine+strip", 2, variant, verbose=False)
        for variant in (0, 14, 16, 13):
            run_case("affine+strip", 3, variant, verbose=False)
        for variant in (0, 13, 14, 16, 12):
            time_case("affine+strip", 2, variant)
        return
    # round 4: which terms of W_hr h are off (VAR 4's dump, least squares per group of four h values), and do landed-before-use exchange reads cure it?
    run_case("affine+strip", 2, 4, verbose=True)
    for variant in (0, 12, 13, 0, 12, 13):
        run_case("affine+strip", 2, variant, verbose=False)
    for variant in (0, 12, 13):
        run_case("affine+strip", 3, variant, verbose=False)


if __name__ == "__main__":
    main()
