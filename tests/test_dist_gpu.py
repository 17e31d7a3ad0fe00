"""RCCL: the `nccl` backend (= RCCL on ROCm) initialises and moves the engines' two payload types (fp64 (reward, distance)
records, fp32 partial sums) with the very collective the engines use.  World size 1 is all a single GPU allows (RCCL
refuses two ranks on one device); whenever two devices are visible `test_rccl_two_ranks_equal_one_rank` runs the four
sharded engines over RCCL and compares them with one rank bit for bit (skipped on a one-GPU box, where the N > 1
semantics are covered with gloo: tests/test_dist_cpu.py, tests/test_ga_multirank_gpu.py, tests/test_dqn_population_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from coevonet_amd.dist import allgather_shards
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    ok = dist.get_backend() == "nccl"
    for dtype in (torch.float64, torch.float32):
        x = (torch.arange(3 * 5 * 4, device="cuda", dtype=dtype) * 0.25).reshape(3, 5, 4)
        out = torch.empty(x.numel(), dtype=dtype, device="cuda")
        dist.all_gather_into_tensor(out, x.reshape(-1).contiguous())      # the call allgather_shards makes
        ok = ok and torch.equal(out.reshape(3, 5, 4), x) and allgather_shards(x, 1) is x
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                               # bench.py's max-over-ranks timing
    ok = ok and float(t.item()) == 1.25
    dist.barrier()
    dist.destroy_process_group()
    ret[0] = bool(ok)


def test_rccl_backend_world1():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(port, ret), nprocs=1, join=True)
    assert ret[0] is True


# --------------------------------------------------------------------------------- RCCL with two ranks, two devices
def _worker_rccl2(rank, world, port, ret):
    """a fresh process per rank, one device each: DistContext must pick `nccl` by itself (no COEVO_DIST_BACKEND)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("COEVO_DIST_BACKEND", None)
    import torch.distributed as dist
    from coevonet_amd.dist import DistContext
    from tests import test_dqn_population_gpu as tdq
    from tests import test_ga_multirank_gpu as tga
    ctx = DistContext()
    out = {"backend": dist.get_backend(), "device": torch.cuda.current_device()}
    ctx.start_gather_timing()
    out["ga"] = tga._train(ctx)
    out["es"] = tga._train_es(ctx, False)
    out["dqn_ga"] = tdq._summary_ga(tdq._run_ga(tdq._ga_cfg(), ctx)[2])
    args, env, base, res = tdq._run_es(tdq._es_cfg(population=8), ctx)
    out["dqn_es"] = {"base": [tdq.sha(b) for b in base], "eval": [res.rewards[r] for r in tdq.rp.DQN_ROLES]}
    out["gathers"] = len(ctx.gather_times_us())
    ctx.shutdown()
    ret[rank] = out


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL refuses two ranks on one device: needs >= 2 GPUs")
def test_rccl_two_ranks_equal_one_rank():
    """whenever two devices are visible: the four sharded engines over RCCL (two ranks, two GPUs) == one rank, bit for
    bit - the fused fitness all-gather (genetic_algorithm.py:223-225's inputs) and the Co-ES reward + partial-sum gathers
    (evolutionary_strategy.py:120-148) on the backend the product ships with"""
    from tests import test_dqn_population_gpu as tdq
    from tests import test_ga_multirank_gpu as tga
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rccl2, args=(2, port, ret), nprocs=2, join=True)
    one_ga, one_es = tga._train(None), tga._train_es(None, False)
    one_dga = tdq._summary_ga(tdq._run_ga(tdq._ga_cfg())[2])
    args, env, base, res = tdq._run_es(tdq._es_cfg(population=8))
    for rank in (0, 1):
        got = ret[rank]
        assert got["backend"] == "nccl" and got["device"] == rank and got["gathers"] > 0
        for k in ("elite_ids", "eval", "sigma", "hof"):
            assert got["ga"][k] == one_ga[k], (rank, k)
            assert got["dqn_ga"][k] == one_dga[k], (rank, k)
        assert got["ga"]["fitness"] == one_ga["fitness"] and got["dqn_ga"]["fitness"] == one_dga["fitness"]
        for k in ("base", "eval", "sigma", "div"):
            assert got["es"][k] == one_es[k], (rank, k)
        assert got["dqn_es"]["base"] == [tdq.sha(b) for b in base]
        assert got["dqn_es"]["eval"] == [res.rewards[r] for r in tdq.rp.DQN_ROLES]
