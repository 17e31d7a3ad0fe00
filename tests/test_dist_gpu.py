"""RCCL smoke on the one-GPU box: the `nccl` backend (= RCCL on ROCm) initialises and moves the engines' two payload
types (fp64 (reward, distance) records, fp32 partial sums) with the very collective the engines use.  World size 1 is
all a single GPU allows (RCCL refuses two ranks on one device); the N > 1 semantics are covered with gloo
(tests/test_dist_cpu.py, tests/test_ga_multirank_gpu.py, tests/test_dqn_population_gpu.py)."""
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
