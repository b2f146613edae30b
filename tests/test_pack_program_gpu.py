"""GPU: tpgsr_pack_program (csrc/conv_mfma.hip) -- every per-step operand packing of a network in one launch; convolution / linear weights
go through LDS in 32 x 32-channel tiles -- against tpgsr_pack_conv_weight, the element-by-element kernel, bit for bit: ragged channel
counts, 1x1 / 3x3 / 2x2 / 1x3 taps, padded leading dimensions (f_ld, f_coff, d_ld, cin_ld), a missing data-gradient operand, a weight
scale, a 9x9 kernel (element-by-element path) and plain copies mixed into the same program."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [  # Cout, Cin, KH, KW, f_ld, f_coff, d_ld, cin_ld, with_d, wscale
    (64, 64, 3, 3, 0, 0, 0, 0, True, 1.0),             # small layer: element by element
    (256, 256, 3, 3, 0, 0, 0, 0, True, 1.0),
    (300, 250, 1, 3, 304, 0, 252, 252, True, 1.0),     # ragged tiles on both sides, padded leading dimensions
    (512, 256, 3, 3, 0, 0, 0, 0, True, 1.0),
    (37, 512, 1, 1, 40, 0, 0, 0, True, 1.0),          # the recogniser's class layer, padded to 40 columns
    (1024, 256, 1, 1, 2048, 1024, 0, 0, True, 1.0),     # one direction of a BiLSTM input projection inside the shared operand
    (1024, 256, 1, 1, 1024, 0, 0, 0, False, 1.0),      # recurrent weights: forward operand only
    (64, 9, 1, 1, 64, 0, 12, 12, True, 1.0),           # conv0 as a 1x1 convolution over the 12-channel im2col map
    (512, 512, 2, 2, 0, 0, 0, 0, True, 0.5),
    (70, 45, 1, 3, 72, 0, 48, 48, True, 1.0),
    (64, 4, 9, 9, 0, 0, 0, 0, True, 1.0),              # 81 taps: element by element
]


def test_pack_program_equals_elementwise_pack():
    from tpgsr_amd import _lib, kernels as K
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    arr = (_lib.PackDesc * (len(CASES) + 1))()
    keep, want, blk = [], [], 0
    for d, (Co, Ci, KH, KW, f_ld, f_coff, d_ld, cin_ld, with_d, ws) in zip(arr, CASES):
        w = torch.randn(Co, Ci, KH, KW, generator=g).to(DEV)
        fl, dl, cl = f_ld or Co, d_ld or Ci, cin_ld or Ci
        wt_f = torch.full((KH * KW * cl, fl), -7.0, device=DEV)
        wt_d = torch.full((KH * KW * Co, dl), -7.0, device=DEV) if with_d else None
        d.src, d.dst_f, d.dst_d = w.data_ptr(), wt_f.data_ptr(), (wt_d.data_ptr() if with_d else None)
        d.Cout, d.Cin, d.KH, d.KW, d.kind, d.f_ld, d.f_coff, d.wscale = Co, Ci, KH, KW, 0, fl, f_coff, ws
        d.d_ld, d.cin_ld, d.numel, d.blk0 = d_ld, cin_ld, w.numel(), blk
        blk += lib.tpgsr_pack_blocks(0, Co, Ci, KH, KW, w.numel())
        # reference: the element-by-element kernel into dense operands, then placed into the padded layout on the host
        rf, rd = torch.empty(KH * KW * Ci, Co, device=DEV), torch.empty(KH * KW * Co, Ci, device=DEV)
        K.pack_conv_weight(w, Co, Ci, KH, KW, rf, rd, wscale=ws)
        ef, ed = torch.full_like(wt_f, -7.0), (torch.full_like(wt_d, -7.0) if with_d else None)
        ef.view(KH * KW, cl, fl)[:, :Ci, f_coff:f_coff + Co] = rf.view(KH * KW, Ci, Co)
        if with_d:
            ed[:, :Ci] = rd
        keep += [w, wt_f, wt_d]
        want.append((wt_f, ef, wt_d, ed))
    src = torch.randn(1000, generator=g).to(DEV)            # a plain copy (kind 2) in the same program
    dst = torch.zeros(1000, device=DEV)
    d = arr[len(CASES)]
    d.src, d.dst_f, d.kind, d.wscale, d.numel, d.blk0 = src.data_ptr(), dst.data_ptr(), 2, 1.0, 1000, blk
    blk += lib.tpgsr_pack_blocks(2, 0, 0, 1, 1, 1000)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    K.pack_program(table, len(CASES) + 1, blk)
    torch.cuda.synchronize()
    for i, (wt_f, ef, wt_d, ed) in enumerate(want):
        assert torch.equal(wt_f, ef), (i, "forward operand")
        if wt_d is not None:
            assert torch.equal(wt_d, ed), (i, "data-gradient operand")
    assert torch.equal(dst, src)
