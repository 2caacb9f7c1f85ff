"""N>1 path on CPU: world_size 2, gloo.  Covers the deterministic per-rank sharding of the poisoned minibatch and the
bucketed all-reduce of the flat gradient (the only collective of the path)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from villandiffusion_amd.trainer import allreduce_flat_grad, shard_indices


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 100_003                                       # not a multiple of the bucket size
    g = torch.full((n,), float(rank + 1))
    g[rank::7] += 0.5
    ref = torch.full((n,), 3.0)
    ref[0::7] += 0.5
    ref[1::7] += 0.5
    allreduce_flat_grad(g, n_buckets=4)
    ok_sum = torch.equal(g, ref)
    ids0 = shard_indices(1001, epoch=3, rank=rank, world=world, seed=5)
    gathered = [torch.zeros_like(ids0) for _ in range(world)]
    dist.all_gather(gathered, ids0)
    allids = torch.cat(gathered)
    ok_cover = set(allids.tolist()) == set(range(1001)) and len(allids) == 1002      # padded to a multiple of world
    ok_epoch = not torch.equal(ids0, shard_indices(1001, epoch=4, rank=rank, world=world, seed=5))
    ok_det = torch.equal(ids0, shard_indices(1001, epoch=3, rank=rank, world=world, seed=5))
    out[rank] = bool(ok_sum and ok_cover and ok_epoch and ok_det)
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_and_flat_allreduce_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    import socket
    with socket.socket() as sk:                 # a port the OS says is free right now (a fixed one can be left in TIME_WAIT by an earlier test)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_single_process_is_a_noop():
    g = torch.arange(10.0)
    allreduce_flat_grad(g)
    assert torch.equal(g, torch.arange(10.0))
    assert shard_indices(10, 0, 0, 1, shuffle=False).tolist() == list(range(10))
