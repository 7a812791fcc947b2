"""World-size-2 gloo test (CPU) of the utterance-sharded path: shard u -> rank u mod 2, sample independently, one
all_gather at the end; the gathered result must equal the single-process run (results independent of world size)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _sample(utts, L):
    """blind DPS (toy score function, batched operator) for the given utterance ids -> (len(utts), L)"""
    sys.path.insert(0, ROOT)
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from oracle.batched.operators import BlindSubbandFiltering
    from oracle.batched.sampler import EulerHeunSamplerDPSTorch
    from oracle.sampler_ref import NoiseStream
    from tests.test_host_logic import _ToyNet
    torch.set_num_threads(2)
    args = compose(overrides=["tester.sampling_params.T=2", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                              "tester.posterior_sampling.blind_hp.op_updates_per_step=1"])
    smp = EulerHeunSamplerDPSTorch(_ToyNet(), instantiate(args.diff_params), args)
    ns = [NoiseStream(900 + u) for u in utts]
    smp.noise = ns
    y = torch.stack([torch.from_numpy((0.05 * np.random.RandomState(u).standard_normal(L)).astype(np.float32)) for u in utts])
    op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, 16000, num_utts=len(utts), noise=ns, device="cpu")
    op.update_H(use_noise=True)
    return smp.predict_conditional(y, op, shape=(len(utts), L), blind=True)


def _worker(rank, world, port, n_utts, L, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from buddy_amd import dist as bd
    r, _, w = bd.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = bd.shard_indices(n_utts, rank, world)
    local = _sample(mine, L)
    full = bd.gather_rows(local, n_utts, rank, world)
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    n_utts, L = 3, 4096        # ragged shards: rank 0 gets utterances {0, 2}, rank 1 gets {1}
    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_utts, L, out_path), nprocs=2, join=True)
    gathered = torch.load(out_path)
    single = torch.cat([_sample([u], L) for u in range(n_utts)])
    assert gathered.shape == (n_utts, L)
    err = float((gathered - single).abs().max() / single.abs().max())
    assert err < 1e-4, err


def test_eight_ranks_64_utterance_partition(tmp_path):
    """BASELINE configs[2]'s partition at its real rank count on the CPU: 64 utterances over eight gloo ranks (u -> rank u mod 8, eight per rank,
    sampled as ONE batch of eight per rank like a GPU rank does), one all_gather; every row equals the utterance sampled alone (per-utterance
    semantics: results do not depend on the world size or on which utterances share a batch)."""
    n_utts, L = 64, 2048
    out_path = str(tmp_path / "gathered8.pt")
    mp.spawn(_worker, args=(8, _free_port(), n_utts, L, out_path), nprocs=8, join=True)
    gathered = torch.load(out_path)
    assert gathered.shape == (n_utts, L) and torch.isfinite(gathered).all()
    for u in (0, 7, 8, 13, 42, 63):        # spot rows of different ranks / batch positions against the utterance on its own
        single = _sample([u], L)[0]
        err = float((gathered[u] - single).abs().max() / single.abs().max())
        assert err < 1e-4, (u, err)


def _ragged_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from buddy_amd import dist as bd
    bd.init(backend="gloo")
    lengths = [5000, 4096, 7001, 300, 4096]
    mine = bd.shard_indices(len(lengths), rank, world)
    rows = [torch.full((lengths[i],), float(i + 1)) + torch.arange(lengths[i]) * 1e-4 for i in mine]
    full = bd.gather_ragged(rows, len(lengths), rank, world)
    ok = all(full[i].shape[0] == lengths[i] and torch.equal(full[i], torch.full((lengths[i],), float(i + 1)) + torch.arange(lengths[i]) * 1e-4)
             for i in range(len(lengths)))
    torch.save(ok, out_path + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_ragged_rows_two_ranks(tmp_path):
    """the harness's end-of-run gather: utterances of different lengths, uneven shards, every rank ends up with all rows in order"""
    out_path = str(tmp_path / "ok")
    mp.spawn(_ragged_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    assert torch.load(out_path + ".0") is True and torch.load(out_path + ".1") is True
