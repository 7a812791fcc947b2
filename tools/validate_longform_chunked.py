"""Long-form policy check: one long clip dereverberated un-chunked and as overlapping chunks (testing/longform.py), full informed
DPS run (T steps, order 2, full-width network on seeded weights): SI-SDR of both estimates to the clean signal and to each other.
usage: python tools/validate_longform_chunked.py [seconds] [chunk_seconds] [overlap_seconds] [T] > profiles/archive/r02_longform_chunked.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from buddy_amd.config import compose
from buddy_amd.instantiate import instantiate
from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
from buddy_amd.testing.tester import Tester
from buddy_amd.utils.metrics import si_sdr
from oracle.sampler_ref import NoiseStream          # deterministic noise stream only

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
chunk = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ov = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
T = int(sys.argv[4]) if len(sys.argv) > 4 else 50
L = int(secs * 16000)
args = compose(tester="informed_dereverberation_DPS", overrides=[f"tester.sampling_params.T={T}"])
net = instantiate(args.network)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, 128).items()})
net = net.cuda().eval()
t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
clean, rir = synth_clean(0, L), synth_rir(0, 8000)
mk = lambda n: [NoiseStream(9100 + u) for u in range(n)]
res = {}
for name, cs in (("unchunked", secs + 1.0), ("chunked", chunk)):
    torch.cuda.synchronize(); t0 = time.time()
    seg, y, pred = t.dereverberate_long(clean, rir, blind=False, chunk_seconds=cs, overlap_seconds=ov, noise=mk)
    torch.cuda.synchronize()
    res[name] = (pred.cpu().double(), time.time() - t0)
    c = seg.cpu().double()
sd = lambda a, b: float(si_sdr(a[None], b[None]))
yv = y.cpu().double()
print(json.dumps({"clip_seconds": secs, "chunk_seconds": chunk, "overlap_seconds": ov, "T": T, "run": "informed DPS, order 2, NCSN++ nf=128 seeded weights, synthetic clip + 8000-tap RIR",
                  "si_sdr_reverberant_input_to_clean_dB": sd(yv, c),
                  "unchunked": {"si_sdr_to_clean_dB": sd(res["unchunked"][0], c), "seconds": res["unchunked"][1]},
                  "chunked": {"si_sdr_to_clean_dB": sd(res["chunked"][0], c), "seconds": res["chunked"][1]},
                  "si_sdr_chunked_vs_unchunked_dB": sd(res["chunked"][0], res["unchunked"][0]),
                  "delta_si_sdr_to_clean_dB": sd(res["chunked"][0], c) - sd(res["unchunked"][0], c)}))
