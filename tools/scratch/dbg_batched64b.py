import sys, torch
sys.path.insert(0, '.')
from oracle import precision, ncsnpp_ref
from oracle.arbiter_runs import _on_device, run_blind_batched
from buddy_amd.synth import synth_state_dict
L = 64000
for fp64 in (True, False):
  for B in (1, 2, 4):
    with (precision.fp64("cuda") if fp64 else _on_device("cuda")), _on_device("cuda"):
        P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
        x = (0.3 * torch.randn(B, 1, L)).requires_grad_(True)
        cn = torch.full((B,), -0.6)
        y = ncsnpp_ref.ncsnpp_time(P, x, cn, 510, 128)
        g, = torch.autograd.grad(y, x, torch.randn_like(y))
        print("fp64" if fp64 else "fp32", B, "net finite", torch.isfinite(y).all().item(), torch.isfinite(g).all().item(), float(y.abs().max()))
tr, clean, k = run_blind_batched([0, 1], L, 1, 128, 1, 8000, fp64=True)
print("one step, B=2, fp64:", torch.isfinite(tr).all().item())
