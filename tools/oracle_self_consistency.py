"""How reproducible is a full blind DPS run of the REFERENCE ALGORITHM itself?  Runs the CPU oracle twice with different intra-op thread
counts (different fp32 summation orders inside the same torch kernels) and reports SI-SDR between the two outputs.  This bounds what any
re-implementation can be asked to match over T chained, Adam-coupled, norm-guided steps on random-init weights.
usage: python tools/oracle_self_consistency.py [T] [L] [threadsA] [threadsB]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd.config import compose
from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
from buddy_amd.utils.metrics import si_sdr
from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
ta = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tb = int(sys.argv[4]) if len(sys.argv) > 4 else 3
args = compose(overrides=[f"tester.sampling_params.T={T}", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled"])
sd = synth_state_dict(0, 128)
P = ncsnpp_ref.to_torch(sd)
onet = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
c0 = torch.from_numpy(synth_clean(0, L)); c0 = 0.05 * c0 / c0.std()
rir = torch.from_numpy(synth_rir(0, 8000))

def run(threads, trace):
    torch.set_num_threads(threads)
    nr = S.NoiseStream(9000)
    ref = S.EulerHeunDPSRef(onet, S.EDMRef(args.diff_params.sde_hp), args, nr)
    op_hp = args.tester.informed_dereverberation.op_hp
    oo = O.RIROperatorRef(op_hp); oo.update_params(rir)
    y0 = oo.degradation(c0[None])
    bo = O.BlindSubbandFilteringRef(op_hp, 16000, nr); bo.update_H(use_noise=True, noise=nr)
    t0 = time.time()
    out = ref.predict_conditional(y0, bo, shape=(1, L), blind=True, trace=trace)
    print(f"threads={threads}: {time.time()-t0:.0f} s", flush=True)
    return out

tra, trb = [], []
a = run(ta, tra); b = run(tb, trb)
per_step = [float(si_sdr(x1[1], x2[1])) for x1, x2 in zip(tra, trb)]
print(json.dumps({"T": T, "L": L, "threads": [ta, tb], "si_sdr_run_a_vs_run_b_dB": float(si_sdr(a, b)),
                  "si_sdr_a_vs_clean": float(si_sdr(a, c0[None])), "si_sdr_b_vs_clean": float(si_sdr(b, c0[None])),
                  "per_step_x_den_si_sdr_dB": [round(v, 1) for v in per_step]}))
