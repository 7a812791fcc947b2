#!/usr/bin/env python3
"""Drop-in for the reference CLI (reference test.py:9-106): same Hydra-style command line

    python test.py --config-name=conf_VCTK.yaml tester=blind_dereverberation_BUDDy tester.checkpoint=<ckpt.pt> \\
        tester.sampling_params.T=201 model_dir=experiments/run +gpu=0 dset.test.path=audio_examples dset.test.num_examples=2

without Hydra (not installable offline): `group=name` picks conf/<group>/<name>.yaml, `a.b.c=value` overrides a key.
Extra keys: +batch_size=N (utterances per sampler call), +allow_random_init=true (no checkpoint: synthetic runs), and the torchrun
environment variables for utterance sharding (rank 0 ends up with every prediction: one RCCL gather at the end)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from buddy_amd import dist as bdist  # noqa: E402
from buddy_amd.config import compose, AttrDict, to_attrdict  # noqa: E402
from buddy_amd.instantiate import instantiate  # noqa: E402
from buddy_amd.testing.tester import Tester  # noqa: E402

GROUPS = ("tester", "network", "diff_params", "exp")


def parse(argv):
    groups, overrides = {}, []
    for a in argv:
        if a.startswith("--config-name") or a.startswith("--config-path"):
            continue
        if "=" not in a:
            raise SystemExit(f"unrecognised argument {a!r}")
        k, v = a.split("=", 1)
        if k in GROUPS:
            groups[k] = v
        elif k in ("dset",):
            continue                      # dataset group: only dset.test.* keys are used here
        else:
            overrides.append(a)
    return groups, overrides


def _main(args):
    rank, local_rank, world = bdist.init(device=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))) if torch.cuda.is_available() else None)
    gpu = args.get("gpu", local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("test.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(gpu if world == 1 else local_rank)
    device = torch.device("cuda", torch.cuda.current_device())

    diff_params = instantiate(args.diff_params)                          # reference test.py:26
    network = instantiate(args.network).to(device)                       # :32
    dcfg = args.get("dset", AttrDict()).get("test", AttrDict())
    test_set = None
    if dcfg.get("path", None):
        from buddy_amd.datasets.vctk import VCTKTestPaired
        test_set = VCTKTestPaired(fs=args.exp.sample_rate, path=dcfg.path, num_examples=int(dcfg.get("num_examples", 8)),
                                  speakers_test=dcfg.get("speakers_test", ()), speakers_discard=dcfg.get("speakers_discard", ()))
    tester = Tester(args, network, diff_params, test_set=test_set, device=device, batch_size=int(args.get("batch_size", 1)),
                    rank=rank, world_size=world)                          # :59
    ckpt = args.tester.get("checkpoint", None)
    if ckpt is not None:
        ok = tester.load_checkpoint(ckpt if os.path.isabs(ckpt) or os.path.exists(ckpt) else os.path.join(args.model_dir, ckpt))   # :73-93
        if not ok:                                                        # load_state_dict raises on an unusable file; belt and braces
            raise ValueError(f"checkpoint {ckpt} could not be loaded")
    elif bool(args.get("allow_random_init", False)):
        print("+allow_random_init: sampling with the randomly initialised network (synthetic runs only)")
    else:
        print("trying to load latest checkpoint")
        tester.load_latest_checkpoint()                                   # :95-96 -- raises "No checkpoint found"
    tester.do_test()                                                      # :98


def main(argv=None):
    groups, overrides = parse(sys.argv[1:] if argv is None else argv)
    args = compose(tester=groups.get("tester", "only_unconditional"), network=groups.get("network", "ncsnpp"),
                   diff_params=groups.get("diff_params", "edm_VCTK"), exp=groups.get("exp", "VCTK_16k_4s_time"), overrides=overrides)
    _main(args)


if __name__ == "__main__":
    main()
