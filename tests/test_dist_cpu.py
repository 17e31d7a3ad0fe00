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


class _FakeESEngine:
    """the fields DistContext.gather_es touches (evolutionary_strategy.ESEngine), on the CPU"""

    def __init__(self, rank, world, pop, blk):
        self.rank, self.world = rank, world
        self.lo, self.hi = rank * pop // world, (rank + 1) * pop // world
        self.stats = torch.zeros(3, pop, 2, dtype=torch.float64)
        self.part_block = blk
        self.partials = torch.zeros(world * blk, dtype=torch.float32)


def _worker_es(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from coevonet_amd.dist import DistContext
    ctx = DistContext(backend="gloo")
    pop, blk = 12, 40
    full_stats = torch.arange(3 * pop * 2, dtype=torch.float64).reshape(3, pop, 2) / 7
    full_parts = torch.arange(world * blk, dtype=torch.float32) * 0.25
    eng = _FakeESEngine(rank, world, pop, blk)
    eng.stats[:, eng.lo:eng.hi] = full_stats[:, eng.lo:eng.hi]
    eng.partials[rank * blk:(rank + 1) * blk] = full_parts[rank * blk:(rank + 1) * blk]
    ctx.gather_es(eng, "stats")
    ctx.gather_es(eng, "partials")
    ret[rank] = bool(torch.equal(eng.stats, full_stats) and torch.equal(eng.partials, full_parts))
    ctx.shutdown()


def test_gather_es_world2_gloo():
    """the two Co-ES exchanges: (reward, distance) pairs by individual, chunk partial sums rank-major"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_es, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_es_engine_rejects_unaligned_shards():
    import pytest
    from coevonet_amd.evolutionary_strategy import ESEngine
    with pytest.raises(ValueError, match="divisible"):
        ESEngine(pop=9, shard=(0, 2), device="cpu")
    with pytest.raises(ValueError, match="divisible"):
        ESEngine(pop=12, shard=(0, 3), device="cpu")     # 8 chunks over 3 ranks
    with pytest.raises(ValueError, match="even"):
        ESEngine(pop=9, antithetic=True, device="cpu")


def test_backend_choice():
    """gloo only for a KNOWN local world larger than the device count (several ranks rehearsing on one GPU); a launcher
    that exports no LOCAL_WORLD_SIZE (srun, mpirun over several nodes) gets RCCL; explicit requests win"""
    from coevonet_amd.dist import choose_backend
    assert choose_backend(None, {"LOCAL_WORLD_SIZE": "2"}, True, 1) == ("gloo", True)
    assert choose_backend(None, {"LOCAL_WORLD_SIZE": "8"}, True, 8) == ("nccl", False)
    assert choose_backend(None, {}, True, 8) == ("nccl", False)              # 16 ranks over 2 x 8 GPUs under srun
    assert choose_backend(None, {"COEVO_DIST_BACKEND": "gloo"}, True, 8) == ("gloo", False)
    assert choose_backend("nccl", {"LOCAL_WORLD_SIZE": "2", "COEVO_DIST_BACKEND": "gloo"}, True, 1)[0] == "nccl"
    assert choose_backend(None, {}, False, 0) == ("gloo", False)             # a CPU-only host: RCCL cannot start there
    assert choose_backend(None, {"LOCAL_WORLD_SIZE": "2"}, False, 0) == ("gloo", False)


def test_es_cohort_bounds_cover_every_game():
    """DQNESEngine's cohort cut (individual-major games, two per individual) for any cohort count"""
    from coevonet_amd.dqn_population import es_cohort_bounds
    assert es_cohort_bounds(250, 1) is None and es_cohort_bounds(1, 2) is None
    assert es_cohort_bounds(250, 2) == [0, 250, 500]
    for n_local, K in [(250, 3), (5, 4), (7, 7), (3, 9)]:
        b = es_cohort_bounds(n_local, K)
        assert b[0] == 0 and b[-1] == 2 * n_local and len(b) == min(K, n_local) + 1
        assert all(b1 > b0 and b1 % 2 == 0 for b0, b1 in zip(b, b[1:]))


class _FakePackedEngine:
    """the fields DistContext.gather_ga_packed touches (genetic_algorithm.GAEngine), on the CPU"""

    def __init__(self, rank, world, n_local):
        self.pack_local = torch.zeros(3, n_local, 4, dtype=torch.float64)
        self.pack_all = torch.zeros(world, 3, n_local, 4, dtype=torch.float64)


def _worker_packed(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from coevonet_amd.dist import DistContext
    ctx = DistContext(backend="gloo")
    n_local = 5
    full = torch.arange(world * 3 * n_local * 4, dtype=torch.float64).reshape(world, 3, n_local, 4) / 3
    eng = _FakePackedEngine(rank, world, n_local)
    eng.pack_local.copy_(full[rank])
    ctx.start_gather_timing() if torch.cuda.is_available() else None
    ctx._gather_ga_packed(eng)
    ret[rank] = bool(torch.equal(eng.pack_all, full))
    ctx.shutdown()


def test_gather_ga_packed_world2_gloo():
    """the fused Co-GA exchange: each rank's [role][j][4] record lands rank-major in every rank's gathered buffer - the layout
    coevo_ga_select_gathered indexes (individual i = rank i // n_local, j = i % n_local; genetic_algorithm.py:223-225 on
    every rank)"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_packed, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
