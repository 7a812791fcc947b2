"""fp64 arbiter mode of the oracle.  TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

The reference computes in fp32 and its blind sampler is chaotic on the time scale of a 50-step run (scale-free Adam updates of the
operator coupled to a guidance term normalised by its own norm), so two fp32 executions of the SAME algorithm with different summation
orders drift apart.  To judge an implementation against that, the same restated algorithm is run in float64 -- whose round-off is nine
orders of magnitude below fp32's, i.e. the "exact" trajectory of the algorithm for these inputs -- and every fp32 execution (the oracle at
several thread counts, the MI355X build) is measured against it.

    with oracle.precision.fp64():
        P = ncsnpp_ref.to_torch(sd)          # parameters, windows, noise draws, schedules: all float64 inside the context
        ...

Only the default dtype (and optionally the default device) changes; the code path is the one pinned in fp32 against the reference fixtures."""
from __future__ import annotations

import contextlib

import torch


@contextlib.contextmanager
def fp64(device=None):
    """default dtype float64 (and, optionally, default device ``device``: the float64 run of a full-size 50-step chain takes an hour per
    utterance on 8 CPU cores -- torch has no fast fp64 CPU convolution -- and under a minute on the MI355X through the same torch ops)"""
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    prev_dev = None
    if device is not None:
        prev_dev = torch.get_default_device()
        torch.set_default_device(device)
    try:
        yield
    finally:
        torch.set_default_dtype(prev)
        if prev_dev is not None:
            torch.set_default_device(prev_dev)
