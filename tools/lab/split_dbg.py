import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tpgsr_amd import kernels as K
for (Kd, N) in [(576, 64), (64, 64), (32, 32), (100, 192)]:
    k = torch.arange(Kd).view(-1, 1); n = torch.arange(N).view(1, -1)
    w = ((k % 16) * 16 + (n % 16)).float().cuda().contiguous()
    twin, kp = K.make_bf_twin(w)
    torch.cuda.synchronize()
    NB, KB = (N + 31) // 32, kp // 16
    planes = twin.view(3, NB, KB, 2, 32, 8).permute(0, 1, 4, 2, 3, 5).reshape(3, NB * 32, kp).float().cpu()
    rec = planes[0][:N, :Kd].t().contiguous()
    bad = (rec != w.cpu()).nonzero()
    print((Kd, N), "kp", kp, "mismatches", len(bad), "of", Kd * N, "first", bad[:6].tolist(), "planes1/2 abs max", float(planes[1].abs().max()), float(planes[2].abs().max()))
    if len(bad):
        kk, nn = bad[0].tolist()
        print("   at", kk, nn, "got", float(rec[kk, nn]), "want", float(w[kk, nn]))
        # where did the wanted value of (k=0, n=1) go?
        flat = twin[: NB * KB * 512].float().cpu()
        print("   flat[0:40]", flat[:40].tolist())
