"""GPU: the reference command line end to end (test.py drop-in): paired clean/RIR wav tree -> harness -> sampler -> wav tree
(reference testing/tester.py:155-203), blind mode with the shipped `wpe_scaled` warm start and informed mode, 2 steps, small net."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu


def _dataset(root, n=2, L=16000):
    from buddy_amd.synth import synth_clean, synth_rir
    for u in range(n):
        for sub, data in (("clean", synth_clean(u, L)), ("rir", np.concatenate([np.zeros(37, np.float32), synth_rir(u, 3000)]))):
            d = os.path.join(root, sub, "p001")
            os.makedirs(d, exist_ok=True)
            wavfile.write(os.path.join(d, f"p001_{u:03d}.wav"), 16000, data.astype(np.float32))


@pytest.mark.parametrize("tester,mode", [("blind_dereverberation_BUDDy", "blind_dereverberation"), ("informed_dereverberation_DPS", "informed_dereverberation")])
def test_cli_end_to_end(tmp_path, tester, mode):
    import test as cli
    data = str(tmp_path / "data")
    _dataset(data)
    out = str(tmp_path / "exp")
    cli.main(["--config-name=conf_VCTK.yaml", f"tester={tester}", "tester.sampling_params.T=2", f"model_dir={out}", "+gpu=0",
              f"dset.test.path={data}", "dset.test.num_examples=2", "network.nf=32", "+batch_size=2", "tester.overriden_name=run"])
    base = os.path.join(out, "run", mode, "VCTK_16k_4s_time")
    subs = ["original", "degraded", "reconstructed", "true_rir"] + (["estimated_rir"] if "blind" in mode else [])
    for s in subs:
        files = sorted(os.listdir(os.path.join(base, s)))
        assert len(files) == 2, (s, files)
        sr, a = wavfile.read(os.path.join(base, s, files[0]))
        assert sr == 16000 and np.isfinite(a).all() and np.abs(a).max() > 0
    assert os.path.exists(os.path.join(base, ".argv"))
    # RIR preprocessing of the paired loader: direct path first, peak-normalised (reference datasets/vctk.py:211-214)
    _, r = wavfile.read(os.path.join(base, "true_rir", sorted(os.listdir(os.path.join(base, "true_rir")))[0]))
    assert abs(np.abs(r).max() - 1.0) < 1e-6 and abs(r[0]) == np.abs(r).max()
