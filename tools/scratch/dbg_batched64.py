import sys, torch
sys.path.insert(0, '.')
from buddy_amd.config import compose
from oracle import precision, sampler_ref as S, ncsnpp_ref, operators_ref as O
from oracle.arbiter_runs import overrides, _on_device
from oracle.batched.operators import BlindSubbandFiltering
from oracle.batched.losses import get_loss
from buddy_amd.synth import synth_state_dict
args = compose(overrides=overrides(50, 10, 128))
op_hp = args.tester.informed_dereverberation.op_hp
L = 64000
for U in (1, 2, 4):
    with precision.fp64("cuda"), _on_device("cuda"):
        ns = [S.NoiseStream(9000 + s) for s in range(U)]
        bo = BlindSubbandFiltering(op_hp, 16000, num_utts=U, noise=ns, device="cuda")
        bo.update_H(use_noise=True)
        print(U, "H finite", torch.isfinite(torch.view_as_real(bo.H)).all().item(), bo.H.dtype)
        x = 0.05 * torch.randn(U, L)
        X = bo.apply_stft(x)
        print("  stft finite", torch.isfinite(torch.view_as_real(X)).all().item())
        Y = bo.subband_filtering(X, bo.H)
        print("  fir finite", torch.isfinite(torch.view_as_real(Y)).all().item(), Y.shape)
        # reference computation of the FIR via explicit loop over a few bands
        Hh = bo.H
        u, f = U - 1, 37
        pre = 1
        xp = torch.nn.functional.pad(X[u, f], (Hh.shape[-1] - 1 - pre, pre))
        ref = torch.stack([(xp[t:t + Hh.shape[-1]] * torch.flip(Hh[u, f], dims=[0])).sum() for t in range(Y.shape[-1])])
        print("  fir vs direct", float((Y[u, f] - ref).abs().max() / ref.abs().max()))
        y = bo.degradation(x)
        print("  degrade finite", torch.isfinite(y).all().item())
        lp = get_loss(args.tester.posterior_sampling.rec_loss_params, bo)
        print("  loss", float(lp(x, y)))
