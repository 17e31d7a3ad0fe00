"""The population-sharded Co-GA (N>1 path) on real kernels: two ranks on ONE GPU over gloo must reproduce the
single-rank run bit for bit (results are independent of the number of ranks: game ordinals address the seeded reset
stream directly, the fitness all-gather reassembles the same vectors, every rank breeds the same offspring)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.util import Bag, sha

pytestmark = pytest.mark.gpu
CFG = dict(generations=3, population=8, hof_size=3, elites_number=2, fitness_sharing=True, max_timesteps_per_episode=40,
           max_evaluation_steps=75)


def _train(dist_ctx):
    from coevonet_amd import genetic_algorithm as ga
    from coevonet_amd.game_logic import initialize_env
    torch.manual_seed(5)
    np.random.seed(5)
    args = Bag(algorithm="GA", **CFG)
    env = initialize_env(args)
    res = ga.genetic_algorithm_train(env, env.agents[0], args, None, rng="device_philox", env_mode="device",
                                     dist_ctx=dist_ctx)
    eng = res.engine
    return {"elite_ids": res.elite_ids, "fitness": res.fitness, "eval": [res.rewards[r] for r in ga.ROLES],
            "sigma": res.sigma_after,
            "hof": {r: [sha(w) for w in eng.download(r, "hof", 0, CFG["hof_size"])] for r in ga.ROLES},
            "games": [g.tolist() for g in res.game_rewards], "shard": (eng.lo, eng.hi)}


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", COEVO_DIST_BACKEND="gloo")
    from coevonet_amd.dist import DistContext
    ctx = DistContext(backend="gloo")
    ret[rank] = _train(ctx)
    ctx.shutdown()


def test_two_ranks_equal_one_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    single = _train(None)
    pop, hof = CFG["population"], CFG["hof_size"]
    for rank in (0, 1):
        got = ret[rank]
        for k in ("elite_ids", "eval", "sigma", "hof"):
            assert got[k] == single[k], (rank, k)
        assert np.array_equal(np.array(got["fitness"], dtype=np.float32), np.array(single["fitness"], dtype=np.float32))
        lo, hi = got["shard"]
        assert (lo, hi) == (rank * pop // 2, (rank + 1) * pop // 2)
        for g, games in enumerate(got["games"]):  # this rank's games are the matching slice of each phase
            full = np.array(single["games"][g]).reshape(3, pop * hof, 3)
            assert np.array_equal(np.array(games).reshape(3, (hi - lo) * hof, 3), full[:, lo * hof:hi * hof])
