"""Checkpoint loading with the reference's fallback order (reference ``utils/training_utils.py:6-111`` ``load_state_dict``):

1. strict load of ``state_dict['network']`` / ``['optimizer']`` / ``['ema']`` into whichever targets were passed;
2. the same with ``strict=False`` (no optimizer);
3. shape-matched assignment: every checkpoint tensor whose name AND shape match the target is taken, the rest keep their
   current values; fails when nothing matched;
4. a checkpoint that stores the weights under ``'state_dict'``.

Returns True as soon as one strategy succeeds, False otherwise.  The sampler loads the EMA weights (``ema=network``), never
silently the raw training weights -- exactly the reference (testing/tester.py:60-67)."""
from __future__ import annotations


def _shape_matched(target, source, log):
    cur = target.state_dict()
    n = 0
    for name, param in source.items():
        if name in cur and cur[name].shape == param.shape:
            cur[name] = param
            n += 1
            if log:
                print("assigning", name)
    target.load_state_dict(cur)
    return n


def load_state_dict(state_dict, network=None, ema=None, optimizer=None, log=True):
    if log:
        print("Loading state dict")
        print(state_dict.keys())
    try:
        if log:
            print("Attempt 1: trying with strict=True")
        if network is not None:
            network.load_state_dict(state_dict["network"])
        if optimizer is not None:
            optimizer.load_state_dict(state_dict["optimizer"])
        if ema is not None:
            ema.load_state_dict(state_dict["ema"])
        return True
    except Exception as e:
        if log:
            print("Could not load state dict")
            print(e)
    try:
        if log:
            print("Attempt 2: trying with strict=False")
        if network is not None:
            network.load_state_dict(state_dict["network"], strict=False)
        if ema is not None:
            ema.load_state_dict(state_dict["ema"], strict=False)
        return True
    except Exception as e:
        if log:
            print("Could not load state dict")
            print(e)
    try:
        if log:
            print("Attempt 3: trying with strict=False, but making sure that the shapes are fine")
        n = 0
        if network is not None:
            n += _shape_matched(network, state_dict["network"], log)
        if ema is not None:
            n += _shape_matched(ema, state_dict["ema"], log)
        if n == 0:
            raise Exception("No parameters were loaded")
        if log:
            print("loaded", n, "parameters")
        return True
    except Exception as e:
        print(e)
        print("the second strict=False failed")
    try:
        if log:
            print("Attempt 4: Assuming the naming is different, with the network and ema called 'state_dict'")
        if network is not None:
            network.load_state_dict(state_dict["state_dict"])
        if ema is not None:
            ema.load_state_dict(state_dict["state_dict"])
        return True
    except Exception as e:
        if log:
            print("Could not load state dict")
            print(e)
    return False
