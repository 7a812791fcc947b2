"""Utterance-batch data parallelism: one process per GPU, utterance u -> rank u mod world, no communication inside the
sampling loop, ONE all_gather of the outputs at the end (RCCL over xGMI on the GPU box; gloo in the CPU tests).
New capability -- the reference is single-process."""
from __future__ import annotations

import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def _group_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init(backend=None, device=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  World size 1 is a no-op unless ``force`` (then a one-rank group is
    brought up -- MASTER_PORT must be set -- so that the communicator and the collectives below run even on a single GPU)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard_indices(n_items, rank, world):
    return [i for i in range(n_items) if i % world == rank]


def gather_rows(local, n_items, rank, world):
    """local: (n_local, L) rows of the utterances in ``shard_indices(n_items, rank, world)`` order.
    Returns the (n_items, L) tensor in utterance order on every rank (one all_gather; ragged shards are padded).  With one rank and no
    process group this is the identity; with a process group the collective runs whatever the world size."""
    if world == 1 and not _group_up():
        return local
    import torch.distributed as dist
    per = (n_items + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        out[idx] = bufs[r][: len(idx)]
    return out


def gather_ragged(rows, n_items, rank, world, device=None):
    """End-of-run gather for the harness: ``rows`` = this rank's 1-D results (utterances ``shard_indices(n_items, rank, world)``, in that
    order, any lengths) -> list of all ``n_items`` rows in utterance order on every rank.  Two collectives in total (lengths, then the
    rows zero-padded to the longest): RCCL over xGMI on the GPU box, gloo in the CPU tests."""
    if world == 1 and not _group_up():
        return list(rows)
    import torch.distributed as dist
    dev = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    lens = torch.tensor([r.shape[-1] for r in rows], dtype=torch.int64, device=dev).reshape(-1, 1)
    all_lens = gather_rows(lens, n_items, rank, world).reshape(-1)
    lmax = int(all_lens.max())
    local = torch.zeros((len(rows), lmax), dtype=torch.float32, device=dev)
    for i, r in enumerate(rows):
        local[i, : r.shape[-1]] = r.to(dev)
    full = gather_rows(local, n_items, rank, world)
    return [full[i, : int(all_lens[i])] for i in range(n_items)]
