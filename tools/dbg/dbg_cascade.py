import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
from oracle import tpgsr_oracle as O
import test_crnn_gpu as T
from tpgsr_amd.interfaces.super_resolution import TPGSRTrainStep
DEV='cuda'
NSR=int(os.environ.get('NSR','1'))
srs, stus, teacher, sds, sd_s, sd_t = T._c3_models(stn=False, n_sr=NSR, n_stu=2)
lr, hr = O.synthetic_batch(4, 77)
if os.environ.get('SAME'):
    srs[1].load_state_dict(sds[0]); sds[1] = sds[0]

psl = [O.as_params(x) for x in sds]; ps = psl[0]; pt = O.as_params(sd_t, False); pu = [O.as_params(x) for x in sd_s]
# oracle forward with retained intermediates
with torch.no_grad():
    q = F.softmax(O.crnn_forward(pt, O.parse_crnn_data(hr[:, :3]), training=False), -1)
cascade = lr; inter = {}
loss = 0
for i in range(2):
    gray = O.parse_crnn_data(cascade[:, :3]); gray.retain_grad() if gray.requires_grad else None
    logits = O.crnn_forward(pu[i], gray, training=True)
    pv = F.softmax(logits, -1)
    prior = pv.permute(1,0,2).unsqueeze(1).permute(0,3,1,2)
    l_d = O.semantic_loss(pv, q) * 100
    drop = torch.ones(4); drop[:1] = 0
    prior = prior * drop.view(-1,1,1,1); prior.retain_grad()
    sr = O.tsrn_forward(psl[i if NSR>1 else 0], lr, prior, training=True, stn=False, text_prior=True); sr.retain_grad()
    l_i = O.image_loss(sr, hr).mean() * 100
    loss = loss + l_i + l_d
    inter[i] = dict(gray=gray, prior=prior, sr=sr, logits=logits, l_i=l_i.item(), l_d=l_d.item()); logits.retain_grad()
    cascade = sr
loss.backward()
ts = TPGSRTrainStep(srs, stus, teacher, stu_iter=2, sr_share=(NSR==1), tpg_share=False); ts._debug = True
for m in srs + stus: m._engine().bind(torch.device(DEV, 0))
teacher._engine().bind(torch.device(DEV, 0))
l = ts._phase_a(lr.to(DEV), hr.to(DEV)); torch.cuda.synchronize()
st = ts._static
print('loss', l.item(), loss.item())
for i in range(2):
    print(i, 'l_img', st['l_img'][i].item(), inter[i]['l_i'], 'l_sem', st['l_sem'][i].item(), inter[i]['l_d'])
    print(i, 'dsr err', ((st['dsr'][i].cpu() - inter[i]['sr'].grad).norm() / inter[i]['sr'].grad.norm()).item())
    print(i, 'gray err', ((st['gray'][i].cpu() - inter[i]['gray'].detach()).abs().max()).item())
    print(i, 'prior err', (st['prior'][i].cpu() - inter[i]['prior'].detach()).abs().max().item())
g1 = inter[1]['gray'].grad
print('dgray1 err', ((ts._dbg_dgray.cpu() - g1).norm()/g1.norm()).item())
from tpgsr_amd import kernels as K
dc = torch.empty(4,4,32,128, device=DEV); g1d = g1.to(DEV).contiguous(); K.bicubic_gray_bwd(g1d, 4, 4, 32, 128, 32, 100, dc); torch.cuda.synchronize()
il = O.image_loss(inter[0]['sr'].detach().requires_grad_(True), hr)
x0 = inter[0]['sr'].detach().clone().requires_grad_(True); (O.image_loss(x0, hr).mean()*100).backward()
ref_dcas = inter[0]['sr'].grad - x0.grad
print('bicubic adjoint of oracle dgray vs oracle dcas', ((dc.cpu()-ref_dcas).norm()/ref_dcas.norm()).item(), 'mine', ((st['dcas'].cpu()-ref_dcas).norm()/ref_dcas.norm()).item())
print('logits1 grad err', 0)
print('dcas err', ((st['dcas'].cpu()[:, :3] - 0).norm()).item())
# isolate: student 1 alone, oracle's dlogits
stu = stus[1]
for p_ in stu.parameters():
    if p_.grad is not None: p_.grad.zero_()
g_in = inter[1]['gray'].detach().to(DEV).requires_grad_(True)
y = stu(g_in)
dl = inter[1]['logits'].grad.to(DEV)
(y * dl).sum().backward()
print('isolated dgray err', ((g_in.grad.cpu() - g1).norm() / g1.norm()).item(), 'fwd err', (y.detach().cpu() - inter[1]['logits'].detach()).abs().max().item())
keys = O.trainable_keys(pu[1]); Ps = dict(stu.named_parameters())
num = sum((Ps[k].grad.cpu() - pu[1][k].grad).double().pow(2).sum().item() for k in keys); den = sum(pu[1][k].grad.double().pow(2).sum().item() for k in keys)
print('isolated student-1 param grads global rel err', (num/den)**0.5)
xs = inter[1]['gray'].detach()
print('gray range', xs.min().item(), xs.max().item())

for i in range(2):
    dp_ref = inter[i]['prior'].grad
    print(i, 'dprior err', ((ts._dbg['dprior'][i].cpu() - dp_ref).norm() / dp_ref.norm()).item(), 'norm', dp_ref.norm().item())
    dl_ref = inter[i]['logits'].grad.permute(1, 0, 2)
    print(i, 'dlogits err', ((ts._dbg['dlogits'][i].cpu() - dl_ref).norm() / dl_ref.norm()).item())
# in-cascade student grads (before the isolated run zeroed them? no: recompute cascade)
for m in stus + srs:
    for p_ in m.parameters():
        if p_.grad is not None: p_.grad.zero_()
l = ts._phase_a(lr.to(DEV), hr.to(DEV)); torch.cuda.synchronize()
for j in range(2):
    keys = O.trainable_keys(pu[j]); Ps = dict(stus[j].named_parameters())
    num = sum((Ps[k].grad.cpu() - pu[j][k].grad).double().pow(2).sum().item() for k in keys); den = sum(pu[j][k].grad.double().pow(2).sum().item() for k in keys)
    worst = max(((Ps[k].grad.cpu() - pu[j][k].grad).norm().item() / max(pu[j][k].grad.norm().item(), 1e-9), k) for k in keys)
    print('in-cascade student', j, 'global rel err', (num/den)**0.5, 'worst', worst)
