"""N>1 path on CPU: world_size-2 gloo processes; the fused all-gather of per-shard fitness inputs reassembles exactly
what a single process computes ("gathered == locally recomputed")."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from coevonet_amd.dist import allgather_shards


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pop, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.arange(3 * pop * 4, dtype=torch.float64).reshape(3, pop, 4) * 1.5  # [role][individual][3 rewards + dist]
    lo, hi = rank * pop // world, (rank + 1) * pop // world
    got = allgather_shards(full[:, lo:hi].clone(), world)
    ret[rank] = bool(torch.equal(got, full))
    dist.destroy_process_group()


def test_allgather_shards_world2_gloo():
    world, pop = 2, 10
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), pop, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_allgather_single_process_is_identity():
    x = torch.rand(3, 5, 4, dtype=torch.float64)
    assert allgather_shards(x, 1) is x
