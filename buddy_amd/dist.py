"""Utterance-batch data parallelism: one process per GPU, utterance u -> rank u mod world, no communication inside the
sampling loop, ONE all_gather of the outputs at the end (RCCL over xGMI on the GPU box; gloo in the CPU tests).
New capability -- the reference is single-process."""
from __future__ import annotations

import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


# ---- placement of a rank on its node (round 6): device selection under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES, CPU affinity next to the GPU --------
def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def visible_device_ids():
    """Physical indices of the devices this process can see, in the order HIP enumerates them: HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES) re-maps
    on top of ROCR_VISIBLE_DEVICES; unset = identity.  Entries that are not plain indices (UUIDs) give None for that slot."""
    def parse(name):
        v = os.environ.get(name)
        if v is None or v.strip() == "":
            return None
        ids = []
        for tok in v.split(","):
            tok = tok.strip()
            ids.append(int(tok) if tok.lstrip("-").isdigit() else None)
        return ids
    rocr = parse("ROCR_VISIBLE_DEVICES")
    hipv = parse("HIP_VISIBLE_DEVICES") or parse("CUDA_VISIBLE_DEVICES")
    if hipv is None:
        return rocr
    if rocr is None:
        return hipv
    return [rocr[i] if (i is not None and 0 <= i < len(rocr)) else None for i in hipv]


def device_index(local_rank, device_count=None):
    """The torch device index of a rank: local rank r takes visible device r (the launcher exports one visible list for the node and every rank picks
    its slot); with fewer visible devices than ranks -- gloo smoke tests where ranks share a GPU -- the ranks wrap around."""
    n = device_count if device_count is not None else torch.cuda.device_count()
    if n <= 0:
        raise RuntimeError("no HIP device visible to this rank")
    return local_rank % n


def _gpu_numa_cpus(physical_index, sysfs="/sys"):
    """CPUs of the NUMA node the GPU hangs off (sysfs: the drm card's PCI device -> numa_node / local_cpulist); None when the kernel does not say."""
    import glob
    cards = sorted((c for c in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*")) if os.path.basename(c)[4:].isdigit()
                    and os.path.exists(os.path.join(c, "device/vendor"))), key=lambda c: int(os.path.basename(c)[4:]))
    amd = []
    for c in cards:
        try:
            if open(os.path.join(c, "device/vendor")).read().strip().lower() == "0x1002":
                amd.append(c)
        except OSError:
            pass
    if physical_index is None or not (0 <= physical_index < len(amd)):
        return None
    dev = os.path.join(amd[physical_index], "device")
    try:
        cpus = _parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        if cpus:
            return cpus
    except (OSError, ValueError):
        pass
    try:
        node = int(open(os.path.join(dev, "numa_node")).read())
        if node >= 0:
            return _parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read())
    except (OSError, ValueError):
        pass
    return None


def plan_affinity(local_rank, local_world, allowed, numa_cpus=None):
    """The CPU set of one rank: the cores this process may use (``allowed``: cgroup / taskset) that are local to its GPU's NUMA node, split evenly and
    DISJOINTLY among the ranks sharing them -- each rank issues ~2 000 kernel launches per sampler step from one Python thread and must not share
    that core.  Ranks whose GPU reports no NUMA node split the whole allowed set.  Pure function (tests/test_distributed_cpu.py)."""
    allowed = sorted(allowed)
    pool = [c for c in allowed if numa_cpus is None or c in set(numa_cpus)] or allowed
    per = max(1, len(pool) // max(1, local_world))
    lo = (local_rank * per) % len(pool)
    mine = pool[lo:lo + per]
    return mine or pool


def pin_rank(local_rank, local_world, device_count=None, sysfs="/sys"):
    """sched_setaffinity of this process to ``plan_affinity`` of its GPU (physical index through the visible-device lists).  Returns the record bench.py
    prints: {'cpus': [...], 'numa_cpus': n or None, 'physical_device': i}.  No-op (cpus = current set) where the platform has no affinity call."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"cpus": None, "numa_cpus": None, "physical_device": None}
    vis = visible_device_ids()
    n_dev = device_count if device_count else local_world          # ranks wrap around the visible devices when there are fewer (device_index)

    def physical(r):
        idx = r % n_dev
        return (vis[idx] if idx < len(vis) else None) if vis is not None else idx
    phys = physical(local_rank)
    numa = _gpu_numa_cpus(phys, sysfs)
    # ranks that share a NUMA pool must split it: the ranks whose GPU has the same pool as ours (all of them on a one-socket node, four per socket on
    # a two-socket 8-GPU node, every rank when they share one GPU)
    peers = [r for r in range(local_world) if _gpu_numa_cpus(physical(r), sysfs) == numa]
    slot = peers.index(local_rank) if local_rank in peers else local_rank
    mine = plan_affinity(slot, max(1, len(peers)), allowed, numa)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        mine = allowed
    return {"cpus": mine, "numa_cpus": (len(numa) if numa else None), "physical_device": phys}


def rccl_env_defaults(log=None):
    """Environment a multi-process RCCL job on this image needs (set only where unset) + what is in force, for the launch log: dmabuf IPC
    (HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver has no legacy IPC; without it hipIpcGetMemHandle fails), the RCCL version banner once
    (NCCL_DEBUG=VERSION) so a node's first run records what it ran on."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_DEBUG", "VERSION")
    rec = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE", "RCCL_MSCCL_ENABLE",
                                          "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT")}
    if log:
        log("rccl / device environment: " + ", ".join(f"{k}={v}" for k, v in rec.items() if v is not None))
    return rec


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
