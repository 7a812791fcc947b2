"""One-off validation of the north-star tolerance: a FULL blind DPS run (T steps, order 1, 10 operator updates/step) of one 4 s utterance
on the MI355X path vs the CPU oracle with identical weights, inputs and noise draws.  Prints SI-SDR(build; oracle) and the SI-SDR of
both outputs w.r.t. the clean signal (their difference is the number the north star bounds by 0.1 dB).
usage: python tools/validate_full_run.py [T] [L] [blind|informed]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from buddy_amd.config import compose
from buddy_amd.instantiate import instantiate
from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
from buddy_amd.testing.tester import Tester
from buddy_amd.utils.metrics import si_sdr
from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
BLIND = (sys.argv[3] if len(sys.argv) > 3 else "blind") == "blind"
ov = [f"tester.sampling_params.T={T}", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled"]
args = compose(overrides=ov)
sd = synth_state_dict(0, 128)
net = instantiate(args.network); net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net = net.cuda().eval()
edm = instantiate(args.diff_params)
item = (synth_clean(0, L), synth_rir(0, 8000), "u0.wav")
t = Tester(args, net, edm, test_set=None, device="cuda", in_training=True)
ns = [S.NoiseStream(9000)]
t.sampler.noise = ns
seg, y, op, _ = t.prepare_batch([item], blind=BLIND, noise=ns)
t0 = time.time()
smp = t.sampler
smp.bind(y, op, BLIND)
sched = smp.create_schedule().cuda(); gam = smp.get_gamma(sched).cuda()
xg = smp.initialize_x(tuple(y.shape), "cuda", sched)
trace_g = []
for i in range(T):
    xg, xdg = smp.step(xg, sched[i], sched[i + 1], gam[i], blind=BLIND)
    trace_g.append(xdg.cpu())
pred = xdg
torch.cuda.synchronize(); tg = time.time() - t0
print(f"MI355X run: {tg:.1f} s for {T} steps", flush=True)

torch.set_num_threads(32)
P = ncsnpp_ref.to_torch(sd)
onet = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
nr = S.NoiseStream(9000)
ref = S.EulerHeunDPSRef(onet, S.EDMRef(args.diff_params.sde_hp), args, nr)
op_hp = args.tester.informed_dereverberation.op_hp
oo = O.RIROperatorRef(op_hp); oo.update_params(torch.from_numpy(item[1]))
c0 = torch.from_numpy(item[0]); c0 = 0.05 * c0 / c0.std()
y0 = oo.degradation(c0[None])
bo = oo
if BLIND:
    bo = O.BlindSubbandFilteringRef(op_hp, 16000, nr); bo.update_H(use_noise=True, noise=nr)
t0 = time.time()
trace_o = []
pr = ref.predict_conditional(y0, bo, shape=(1, L), blind=BLIND, trace=trace_o)
tc = time.time() - t0
print(f"oracle run: {tc:.1f} s", flush=True)
assert nr.k == ns[0].k
p = pred.cpu()
res = {"T": T, "L": L, "mode": "blind" if BLIND else "informed", "si_sdr_build_vs_oracle_dB": float(si_sdr(p, pr)), "si_sdr_build_vs_clean_dB": float(si_sdr(p, c0[None])),
       "si_sdr_oracle_vs_clean_dB": float(si_sdr(pr, c0[None])), "rel_max_err": float((p - pr).abs().max() / pr.abs().max()),
       "gpu_seconds": tg, "oracle_seconds_32_threads": tc,
       "note": "random-init network (no checkpoint offline): absolute SI-SDR to clean is meaningless, the DIFFERENCE between build and oracle is the parity measure"}
res["per_step_x_den_si_sdr_build_vs_oracle_dB"] = [round(float(si_sdr(a, b[1])), 1) for a, b in zip(trace_g, trace_o)]
res["delta_si_sdr_to_clean_dB"] = res["si_sdr_build_vs_clean_dB"] - res["si_sdr_oracle_vs_clean_dB"]
print(json.dumps(res))
