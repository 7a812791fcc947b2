"""Checkpoint loading: which layouts a ``.pt`` file may have and in which order they are tried -- the behaviour of the reference's
``utils/training_utils.py:6-178`` ``load_state_dict(state_dict, network, ema, optimizer, log)``, expressed as a table of strategies.

A strategy is ``(label, loader)``; ``loader(ckpt, targets)`` either loads and returns the number of tensors it placed (> 0) or raises.
They run in order; the first one that succeeds ends the search (``True``).  The LAST strategy is not guarded: a file that fits no layout
raises out of ``load_state_dict``, so the sampler never runs on randomly initialised weights by accident.

| # | layout of the file | reference lines |
|---|---|---|
| 1 | ``{'network', 'optimizer', 'ema'}`` state dicts, strict | :13-27 |
| 2 | the same, ``strict=False`` (optimizer skipped) | :29-41 |
| 3 | the same, tensor by tensor where name AND shape match; fails if nothing matched | :43-79 |
| 4 | weights under ``'state_dict'`` | :81-95 |
| 5 | legacy: names from ``'model'``, tensors from the list ``'ema_weights'``, one per entry | :103-113 |
| 6 | legacy: ``'ema_weights'`` lists only the entries of ``'model'`` that require grad; the others come from ``'model'`` | :115-129 |
| 7 | ``'state_dict'`` with ``diffusion.`` / ``diffusion_ema.`` name prefixes, shape-matched | :132-173 |
| 8 | the file itself is a state dict, strict -- loads or raises | :174-178 |

The sampler passes ``ema=network`` (testing/tester.py:60-67): it loads the EMA weights, never the raw training weights.

Deliberate differences to the reference, each stricter or equal in effect:
* strategy 3 also serves ema-only calls (the reference dereferences ``network`` there and always falls through when it is None);
* strategy 4 stops on success (the reference forgets the ``return``, falls through to 8 and raises on a checkpoint it has just loaded);
* strategy 5 loads the complete zipped dict in one strict call and only when ``'ema_weights'`` has exactly one tensor per entry of ``'model'``.
  The reference calls ``ema.load_state_dict`` INSIDE its zip loop on a one-key dict, which always raises for a real model, so a legacy
  ``{'model', 'ema_weights'}`` file ends up in its strategy 6 -- and there, because tensors saved through ``state_dict()`` carry
  ``requires_grad=False``, it loads the raw ``'model'`` weights.  Loading the EMA list is the evident intent of both strategies and is what
  happens here; a truncated or surplus list can never be accepted by 5 (length check) and goes to 6.
* strategy 7 copies the prefixed tensors whose stripped name and shape match a parameter and SKIPS the others (a surplus
  ``diffusion_ema.foo`` entry, a tensor of another shape).  The reference indexes the target by the stripped name, raises ``KeyError`` on
  the first unknown one and falls through to its strict bare load, which then raises on the prefixed file: a checkpoint with one extra
  prefixed entry is unusable there and loads here (pinned by tests/test_host_logic.py case 7b).
A failed strict attempt may have copied some tensors before raising (torch copies matching tensors first); every later success overwrites
all of them."""
from __future__ import annotations


def _targets(network, ema, optimizer):
    return {"network": network, "ema": ema, "optimizer": optimizer}


def _strict(ckpt, t):
    n = 0
    for key in ("network", "optimizer", "ema"):
        if t[key] is not None:
            t[key].load_state_dict(ckpt[key])
            n += 1
    return max(n, 1)


def _non_strict(ckpt, t):
    for key in ("network", "ema"):
        if t[key] is not None:
            t[key].load_state_dict(ckpt[key], strict=False)
    return 1


def _assign_matching(target, source, prefix=""):
    """copy every tensor of ``source`` whose (prefix-stripped) name and shape match an entry of ``target``; returns how many"""
    cur = target.state_dict()
    n = 0
    for name, param in source.items():
        if not name.startswith(prefix):
            continue
        short = name[len(prefix):]
        if short in cur and cur[short].shape == param.shape:
            cur[short] = param
            n += 1
    target.load_state_dict(cur, strict=not prefix)
    return n


def _shape_matched(ckpt, t):
    n = sum(_assign_matching(t[key], ckpt[key]) for key in ("network", "ema") if t[key] is not None)
    if n == 0:
        raise KeyError("no tensor of the checkpoint matches a parameter by name and shape")
    return n


def _under_state_dict(ckpt, t):
    for key in ("network", "ema"):
        if t[key] is not None:
            t[key].load_state_dict(ckpt["state_dict"])
    return 1


def _legacy_ema_list(ckpt, t):
    if t["ema"] is None:
        raise KeyError("legacy layout only carries EMA weights")
    names, tensors = list(ckpt["model"].keys()), list(ckpt["ema_weights"])
    if len(names) != len(tensors):
        raise ValueError(f"'ema_weights' has {len(tensors)} tensors for {len(names)} entries of 'model'")
    t["ema"].load_state_dict(dict(zip(names, tensors)))
    return len(names)


def _legacy_ema_trainable_only(ckpt, t):
    if t["ema"] is None:
        raise KeyError("legacy layout only carries EMA weights")
    rest = iter(ckpt["ema_weights"])
    t["ema"].load_state_dict({k: (next(rest) if v.requires_grad else v) for k, v in ckpt["model"].items()})
    return len(ckpt["model"])


def _prefixed(ckpt, t):
    n = 0
    for key, prefix in (("network", "diffusion."), ("ema", "diffusion_ema.")):
        if t[key] is not None:
            n += sum(1 for name in ckpt["state_dict"] if name.startswith(prefix))
            _assign_matching(t[key], ckpt["state_dict"], prefix)
    if n == 0:
        raise KeyError("no 'diffusion.' / 'diffusion_ema.' entries under 'state_dict'")
    return n


def _bare(ckpt, t):
    for key in ("network", "ema"):
        if t[key] is not None:
            t[key].load_state_dict(ckpt, strict=True)
    return 1


STRATEGIES = (
    ("'network' / 'optimizer' / 'ema', strict", _strict),
    ("'network' / 'ema', strict=False", _non_strict),
    ("'network' / 'ema', tensors matching by name and shape", _shape_matched),
    ("weights under 'state_dict'", _under_state_dict),
    ("legacy: names of 'model' + list 'ema_weights'", _legacy_ema_list),
    ("legacy: 'ema_weights' for the trainable entries of 'model'", _legacy_ema_trainable_only),
    ("'state_dict' with diffusion. / diffusion_ema. prefixes", _prefixed),
    ("the file is a bare state dict, strict", _bare),
)


def load_state_dict(state_dict, network=None, ema=None, optimizer=None, log=True):
    say = print if log else (lambda *a, **k: None)
    t = _targets(network, ema, optimizer)
    say("checkpoint keys:", list(state_dict.keys())[:12])
    for i, (label, loader) in enumerate(STRATEGIES, 1):
        last = i == len(STRATEGIES)
        try:
            n = loader(state_dict, t)
            say(f"checkpoint layout {i} ({label}): loaded" + (f" ({n} entries)" if n > 1 else ""))
            return True
        except Exception as e:
            if last:
                raise
            say(f"checkpoint layout {i} ({label}) does not fit: {type(e).__name__}: {str(e)[:200]}")
    return False
