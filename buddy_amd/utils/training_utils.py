"""Checkpoint loading with the reference's fallback order (reference ``utils/training_utils.py:6-111`` ``load_state_dict``):

1. strict load of ``state_dict['network']`` / ``['optimizer']`` / ``['ema']`` into whichever targets were passed;
2. the same with ``strict=False`` (no optimizer);
3. shape-matched assignment: every checkpoint tensor whose name AND shape match the target is taken, the rest keep their
   current values; fails when nothing matched;
4. a checkpoint that stores the weights under ``'state_dict'``;
5. legacy layout: names from ``state_dict['model']`` zipped with the tensor LIST ``state_dict['ema_weights']`` (:103-113);
6. the same, but only the entries of ``'model'`` with ``requires_grad`` consume an ``'ema_weights'`` tensor, the others (buffers)
   are taken from ``'model'`` itself (:115-129);
7. ``state_dict['state_dict']`` with ``diffusion.`` / ``diffusion_ema.`` name prefixes, shape-matched, ``strict=False`` (:132-173);
8. finally the file itself as a bare state dict, ``strict=True`` -- which either loads or RAISES (:174-178), so a checkpoint that fits
   no layout never leaves the sampler on randomly initialised weights.

Returns True as soon as one strategy succeeds.  The sampler loads the EMA weights (``ema=network``), never silently the raw training
weights (testing/tester.py:60-67).  Two deliberate differences, both stricter or equal in effect: attempt 3 also works for ema-only calls
(the reference's dereferences ``network`` and so always falls through when ``network=None``), and attempt 4 returns on success (the
reference forgets the ``return`` there, falls through to 8 and raises on a checkpoint it has just loaded)."""
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
    try:
        if log:
            print("Attempt 5: model='model' and ema='ema_weights' (a list in the order of the model's entries)")
        if ema is not None:
            names, tensors = list(state_dict["model"].keys()), list(state_dict["ema_weights"])
            ema.load_state_dict(dict(zip(names, tensors)))
            return True
    except Exception as e:
        if log:
            print(e)
    try:
        if log:
            print("Attempt 6: 'ema_weights' holds only the trainable entries of 'model'; buffers come from 'model'")
        if ema is not None:
            rest = iter(state_dict["ema_weights"])
            ema.load_state_dict({k: (next(rest) if v.requires_grad else v) for k, v in state_dict["model"].items()})
            return True
    except Exception as e:
        if log:
            print(e)
    try:
        if log:
            print("Attempt 7: parameters named 'diffusion.*' / 'diffusion_ema.*' under 'state_dict'")
        n = 0
        for target, prefix in ((network, "diffusion."), (ema, "diffusion_ema.")):
            if target is None:
                continue
            cur = target.state_dict()
            for name, param in state_dict["state_dict"].items():
                if name.startswith(prefix):
                    n += 1
                    short = name.replace(prefix, "")
                    if cur[short].shape == param.shape:
                        cur[short] = param
            target.load_state_dict(cur, strict=False)
        if n == 0:
            raise Exception("No parameters were loaded")
        if log:
            print("loaded", n, "parameters")
        return True
    except Exception as e:
        if log:
            print(e)
    if network is not None:
        network.load_state_dict(state_dict, strict=True)
    if ema is not None:
        ema.load_state_dict(state_dict, strict=True)
    return True
