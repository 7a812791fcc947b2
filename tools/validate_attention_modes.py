"""Gate for the attention variants: one full informed DPS run (T steps, order 2, full-width network, one 4 s utterance, injected
noise) per attention mode -- matrix (materialised T x T, the r01 path), flash (fp32 online softmax, default), bf16 / f16 (16-bit MFMA operands) --
each in its own process (BUDDY_ATTN is read once), then SI-SDR of every mode against the matrix run and the difference of SI-SDR-to-clean.
A reduced-precision mode may only be offered as a fast option if |delta SI-SDR to clean| <= 0.1 dB.
usage: python tools/validate_attention_modes.py [T] [L] > profiles/archive/r02_attention_modes.json"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def one(mode, T, L, out):
    import torch
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    args = compose(tester="informed_dereverberation_DPS", overrides=[f"tester.sampling_params.T={T}"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, 128).items()})
    net = net.cuda().eval()
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    t.sampler.noise = [NoiseStream(9000)]
    seg, y, op, _ = t.prepare_batch([(synth_clean(0, L), synth_rir(0, 8000), "u0.wav")], blind=False)
    torch.cuda.synchronize(); t0 = time.time()
    pred = t.sampler.predict_conditional(y, op, shape=(1, L), blind=False)
    torch.cuda.synchronize()
    np.savez(out, pred=pred.cpu().numpy(), clean=seg.cpu().numpy(), seconds=time.time() - t0)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 64000
    import torch
    from buddy_amd.utils.metrics import si_sdr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    res = {}
    for mode in ("matrix", "flash", "bf16", "f16"):
        out = os.path.join(ROOT, "gpurun_out", f"attn_{mode}.npz")
        env = dict(os.environ, BUDDY_ATTN=mode)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", mode, str(T), str(L), out], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res[mode] = np.load(out)
    sd = lambda a, b: float(si_sdr(torch.from_numpy(np.asarray(a, dtype=np.float64)), torch.from_numpy(np.asarray(b, dtype=np.float64))))
    ref = res["matrix"]
    rep = {"T": T, "L": L, "run": "informed DPS, order 2, NCSN++ nf=128 on seeded weights, one utterance, NoiseStream(9000)", "modes": {}}
    for mode, r in res.items():
        rep["modes"][mode] = {"si_sdr_vs_matrix_dB": None if mode == "matrix" else sd(r["pred"], ref["pred"]),
                              "si_sdr_to_clean_dB": sd(r["pred"], r["clean"]),
                              "delta_si_sdr_to_clean_dB": sd(r["pred"], r["clean"]) - sd(ref["pred"], ref["clean"]),
                              "seconds": float(r["seconds"])}
        os.remove(os.path.join(ROOT, "gpurun_out", f"attn_{mode}.npz"))
    rep["gate_0p1_dB"] = {m: abs(v["delta_si_sdr_to_clean_dB"]) <= 0.1 for m, v in rep["modes"].items()}
    print(json.dumps(rep))
