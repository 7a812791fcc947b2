import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, numpy as np
from buddy_amd.config import compose
from buddy_amd.instantiate import instantiate
from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
from buddy_amd.testing.operators.reverb import RIROperator
from buddy_amd.utils.losses import get_loss
from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S
def rel(a,b):
    a=a.detach().cpu().double(); b=b.detach().cpu().double()
    return float((a-b).abs().max()/(b.abs().max()+1e-30))
args = compose(tester="informed_dereverberation_DPS", overrides=["tester.sampling_params.T=4","network.nf=32"])
sd = synth_state_dict(7, 32)
net = instantiate(args.network); net.load_state_dict({k: torch.from_numpy(v) for k,v in sd.items()}); net=net.cuda().eval()
P = ncsnpp_ref.to_torch(sd)
onet = lambda x, cn: ncsnpp_ref.ncsnpp_time(P, x, cn, 510, 128)
edm = instantiate(args.diff_params); oedm = S.EDMRef(args.diff_params.sde_hp)
L=8192
clean, rir = torch.from_numpy(synth_clean(0, L)), torch.from_numpy(synth_rir(0, 2000))
op_hp = args.tester.informed_dereverberation.op_hp
oop = O.RIROperatorRef(op_hp); oop.update_params(rir); y = oop.degradation(clean[None])
gop = RIROperator(op_hp, device="cuda"); gop.update_params(rir)
yg = gop.degradation(clean[None].cuda())
print("degradation", rel(yg, y))
print("apply_stft", rel(torch.view_as_real(gop.apply_stft(yg)), torch.view_as_real(oop.apply_stft(y))))
t = torch.tensor(0.3)
x = torch.randn(1, L)*0.3
xo = x.clone().requires_grad_(True); xg = x.clone().cuda().requires_grad_(True)
do = oedm.denoiser(xo.unsqueeze(1), onet, t).squeeze(1)
dg = edm.denoiser(xg.unsqueeze(1), net, t.cuda()).squeeze(1)
print("x_den", rel(dg, do))
lo = O.get_loss_ref(args.tester.posterior_sampling.rec_loss, oop); lg = get_loss(args.tester.posterior_sampling.rec_loss, gop)
yho = oop.degradation(do); yhg = gop.degradation(dg)
print("y_hat", rel(yhg, yho))
ro = lo(y, yho); rg = lg(yg, yhg)
print("rec", float(ro), float(rg))
# gradient wrt x_den
gdo, = torch.autograd.grad(ro, do, retain_graph=True); gdg, = torch.autograd.grad(rg, dg, retain_graph=True)
print("d rec / d x_den", rel(gdg, gdo), float(gdo.abs().max()))
go, = torch.autograd.grad(ro, xo); gg, = torch.autograd.grad(rg, xg)
print("d rec / d x", rel(gg, go), float(go.abs().max()))
# vjp of denoiser alone with the oracle's cotangent
xg2 = x.clone().cuda().requires_grad_(True)
dg2 = edm.denoiser(xg2.unsqueeze(1), net, t.cuda()).squeeze(1)
gg2, = torch.autograd.grad(dg2, xg2, gdo.cuda())
print("denoiser vjp with oracle cotangent", rel(gg2, go))
