"""DeepQN population engine (BASELINE configs 4 / 5) against the oracle port.  DeepQN.forward is pinned by the reference's
logits (tests/test_deepqn_gpu.py, tests/golden/deepqn_forward.json); the two-role loops over the synthetic env are the
build's own definition ("loop parity unpinned") and are checked HIP-vs-oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from coevonet_amd import lib as L
from coevonet_amd.game_logic import create_agent, initialize_env, play_game
from oracle import ref_port as rp
from tests.util import Bag, sha

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _slab(flats, C, n):
    stride = int(L.load().coevo_dqn_slab_stride(C & 0xff, n))
    flat = torch.from_numpy(np.ascontiguousarray(np.stack(flats), dtype=np.float32)).to(DEV)
    slab = torch.zeros(len(flats), stride, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_pack", L._p(flat), L._p(slab), len(flats), C, n)
    return slab, stride


def _unpack(slab_ptr, k, C, n):
    P = int(L.load().coevo_dqn_param_count(C & 0xff, n))
    out = torch.zeros(k, P, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_unpack", slab_ptr, L._p(out), k, C, n)
    return out.cpu().numpy()


@pytest.mark.parametrize("tiled", [0, L.DQN_FC1_TILED])
@pytest.mark.parametrize("Cp,n", [(4, 6), (6, 18)])
def test_dqn_perturb_bit_exact_vs_oracle(Cp, n, tiled):
    """offspring / elite rebuild / distances over both fc1 layouts of the slab (the noise is indexed by the CANONICAL parameter
    position, so a child is the same net whatever the layout; the tiled block draws one Philox block per lane and transposes)"""
    C = Cp | tiled   # the channel argument of every layout-dependent entry point (sizes ignore the flag)
    torch.manual_seed(3)
    parents = [rp.dqn_init(Cp, n)[0] for _ in range(3)]
    slab, stride = _slab(parents, C, n)
    assert np.array_equal(_unpack(L._p(slab), 3, C, n), np.stack(parents))        # pack / unpack round trip
    other = Cp | (0 if tiled else L.DQN_FC1_TILED)                                  # ... and through the other fc1 layout
    twin = torch.zeros_like(slab)
    L.call("coevo_dqn_relayout", L._p(slab), L._p(twin), 3, C, other, n)
    assert np.array_equal(_unpack(L._p(twin), 3, other, n), np.stack(parents)) and not torch.equal(twin, slab)
    assert L.load().coevo_dqn_relayout(L._p(slab), L._p(slab), 3, C, other, n, None) == -1   # in place: refused
    P = len(parents[0])
    sigma = torch.tensor([0.05], dtype=torch.float32, device=DEV)
    seed, shi = 0xABCDEF0123, 9
    blocks = int(L.load().coevo_dqn_perturb_blocks(C, n))
    bn = rp.dqn_bn_segments(Cp, n)
    # GA (every parameter), ES (BatchNorm untouched), ES antithetic
    for flags, first in [(0, 100), (1, 100), (3, 10)]:
        k = 4
        pidx = torch.tensor([2, 0, 1, 2], dtype=torch.int32, device=DEV)
        child = torch.zeros(1 + k, stride, dtype=torch.float32, device=DEV)
        part = torch.zeros(k * blocks, dtype=torch.float64, device=DEV)
        L.call("coevo_dqn_perturb", L._p(slab), L._p(pidx), L._p(child), 1, k, C, n, L._p(sigma), seed, first, shi, flags,
               1, None, 0, L._p(slab), L._p(part))                                   # distances to parent 0
        dist = torch.zeros(1 + k, dtype=torch.float32, device=DEV)
        L.call("coevo_fc_distance_finalize", L._p(part), blocks, k, L._p(dist), 1, None)
        got = _unpack(L._p(child), 1 + k, C, n)
        assert not got[0].any()
        for c in range(k):
            j = first + c
            anti = bool(flags & 2)
            want = rp.perturb_philox_flat(parents[[2, 0, 1, 2][c]], np.float32(0.05), seed, (j >> 1) if anti else j, shi,
                                          bn if flags & 1 else (), negate=anti and bool(j & 1))
            assert np.array_equal(got[1 + c].view(np.uint32), want.view(np.uint32)), (flags, c)
            d = np.linalg.norm((want - parents[0]).astype(np.float64))
            np.testing.assert_allclose(dist[1 + c].item(), d, rtol=2e-7)
        noise = got[1] - parents[2]
        assert abs(noise[:P - 320].std() - 0.05) < 2e-4 and (flags & 1) == (not noise[P - 320:].any())
    # elite rebuild: child e = individual order[e] of the population bred from the E = 2 elites in `slab`
    order = torch.tensor([5, 0, 2], dtype=torch.int32, device=DEV)
    out = torch.zeros(3, stride, dtype=torch.float32, device=DEV)
    gen_dev = torch.tensor([3], dtype=torch.int32, device=DEV)
    L.call("coevo_dqn_perturb", L._p(slab), L._p(order), L._p(out), 0, 3, C, n, L._p(sigma), seed, 0, 1, 4, 2, L._p(gen_dev),
           -1, None, None)
    got = _unpack(L._p(out), 3, C, n)
    for e, idv in enumerate([5, 0, 2]):
        want = parents[0] if idv == 0 else rp.perturb_philox_flat(parents[(idv - 1) % 2], np.float32(0.05), seed, idv - 1,
                                                                  1 + 4 * 2)
        assert np.array_equal(got[e].view(np.uint32), want.view(np.uint32)), e


@pytest.mark.parametrize("game,C", [("pong_v3", 4), ("boxing_v2", 6)])
def test_play_game_atari_device_host_oracle_agree(game, C):
    """play_game over the synthetic env: whole episode on the device (coevo_synth_step + DeepQN kernels) == the host AEC
    loop with one device forward per step == the oracle's literal play_atari restatement"""
    torch.manual_seed(4)
    args = Bag(game=game, max_timesteps_per_episode=7, max_evaluation_steps=4, coevo_channels=C)
    env = initialize_env(args)
    assert env.n_resets == 1 and env.observation_space("first_0").shape == (84, 84, C)
    a, b = create_agent(env, args), create_agent(env, args)
    n = env.n_actions
    seen = []
    for i, ev in enumerate([False, True, False]):
        dev = play_game(env=env, player1=a.model, player2=b.model, args=args, eval=ev)
        ordinal = env.ordinal
        want = rp.dqn_play_game(a.model.flat(), b.model.flat(), C, n, env.seed_value, ordinal, 4 if ev else 7)
        assert list(dev) == want["rewards"], (i, dev, want)
        seen += want["rewards"] + [float(len(set(want["actions"])) > 1)]
    assert any(seen[0::3] + seen[1::3]) and any(seen[2::3])   # some hit was credited; the frames do steer the actions
    args.coevo_host_aec = True
    env2 = initialize_env(args)
    host = play_game(env=env2, player1=a.model, player2=b.model, args=args, eval=False)
    want = rp.dqn_play_game(a.model.flat(), b.model.flat(), C, n, env2.seed_value, 1, 7)
    assert list(host) == want["rewards"]


def _ga_cfg(**kw):
    d = dict(game="pong_v3", generations=3, population=4, hof_size=2, elites_number=2, fitness_sharing=True,
             max_timesteps_per_episode=6, max_evaluation_steps=4, mutation_power_agent_0=0.05,
             mutation_power_agent_1=0.08, coevo_channels=4)
    d.update(kw)
    return d


def _run_ga(cfg, dist_ctx=None):
    from coevonet_amd import genetic_algorithm as ga
    torch.manual_seed(11)
    args = Bag(algorithm="GA", **cfg)
    env = initialize_env(args)
    res = ga.genetic_algorithm_train(env, env.agents[0], args, None, dist_ctx=dist_ctx)
    return args, env, res


@pytest.mark.parametrize("graph", [True, False])
def test_dqn_ga_matches_oracle_port(graph):
    cfg = _ga_cfg(coevo_graph=graph)
    args, env, res = _run_ga(cfg)
    torch.manual_seed(11)
    oargs = Bag(algorithm="GA", **cfg)
    want = rp.dqn_ga_train(oargs, 4, 6, env_seed=env.seed_value)
    pop, hof = args.population, args.hof_size
    for g, w in enumerate(want):
        assert res.elite_ids[g] == w["elite_ids"], g
        for i in range(2 * pop * hof):
            assert list(res.game_rewards[g][i]) == w["games"][i]["rewards"], (g, i)
        for ph in range(2):
            np.testing.assert_allclose(res.fitness[g][ph], w["fitness"][ph], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(res.diversity[g], w["diversity"], rtol=1e-5, atol=1e-6)
        assert [res.rewards[r][g] for r in rp.DQN_ROLES] == w["eval_rewards"], g
        assert res.sigma_after[g] == w["sigma_after"], g
    eng = res.engine
    for r in rp.DQN_ROLES:
        assert [sha(x) for x in eng.download(r, "hof", 0, hof)] == [sha(x) for x in want[-1]["hof"][r]]
        assert [sha(x) for x in eng.download(r, "pop", 0, pop)] == [sha(x) for x in want[-1]["pop"][r]]
    assert env.n_resets == 1 + 3 * (2 * pop * hof + 10)


def _es_cfg(**kw):
    d = dict(game="boxing_v2", generations=2, population=4, hof_size=1, learning_rate=0.1, fitness_sharing=True,
             max_timesteps_per_episode=6, max_evaluation_steps=4, mutation_power_agent_0=0.05,
             mutation_power_agent_1=0.08, coevo_channels=6)
    d.update(kw)
    return d


def _run_es(cfg, dist_ctx=None):
    from coevonet_amd import evolutionary_strategy as es
    torch.manual_seed(12)
    args = Bag(algorithm="ES", **cfg)
    env = initialize_env(args)
    base, res = es.evolution_strategy_train(env, args, None, dist_ctx=dist_ctx)
    return args, env, base, res


@pytest.mark.parametrize("extension", [False, True])
def test_dqn_es_matches_oracle_port(extension):
    cfg = _es_cfg(coevo_antithetic=extension, coevo_centered_rank=extension)
    args, env, base, res = _run_es(cfg)
    torch.manual_seed(12)
    oargs = Bag(algorithm="ES", **{k: v for k, v in cfg.items() if k not in ("coevo_antithetic", "coevo_centered_rank")})
    want = rp.dqn_es_train(oargs, 6, 18, env_seed=env.seed_value, antithetic=extension, centered_rank=extension)
    pop = args.population
    for g, w in enumerate(want):
        for i in range(2 * pop):
            assert list(res.game_rewards[g][i]) == w["games"][i]["rewards"], (g, i)
        assert [res.rewards[r][g] for r in rp.DQN_ROLES] == w["eval_rewards"], g
        np.testing.assert_allclose(res.diversity[g], w["diversity"], rtol=1e-5, atol=1e-6)
        assert res.sigma_after[g] == w["sigma_after"]
    for ri, r in enumerate(rp.DQN_ROLES):
        # fitness sharing enters through an fp32 division by (1 + score): scores agree to ~1e-6, weights accordingly;
        # centered ranks do not depend on the score at all -> bit-exact
        if extension:
            assert sha(base[ri]) == sha(want[-1]["base"][r]), r
        else:
            np.testing.assert_allclose(base[ri], want[-1]["base"][r], rtol=1e-4, atol=5e-6)


# ------------------------------------------------------------------------------------------------ two ranks == one
def _summary_ga(res):
    eng = res.engine
    return {"elite_ids": res.elite_ids, "fitness": res.fitness, "eval": [res.rewards[r] for r in rp.DQN_ROLES],
            "sigma": res.sigma_after, "hof": {r: [sha(w) for w in eng.download(r, "hof", 0, eng.hof)] for r in rp.DQN_ROLES},
            "games": [g.tolist() for g in res.game_rewards], "shard": (eng.lo, eng.hi)}


def _worker(rank, world, port, algo, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", COEVO_DIST_BACKEND="gloo")
    from coevonet_amd.dist import DistContext
    ctx = DistContext(backend="gloo")
    if algo == "ga":
        ret[rank] = _summary_ga(_run_ga(_ga_cfg(), ctx)[2])
    else:
        args, env, base, res = _run_es(_es_cfg(population=8), ctx)
        ret[rank] = {"base": [sha(b) for b in base], "eval": [res.rewards[r] for r in rp.DQN_ROLES],
                     "games": [g.tolist() for g in res.game_rewards], "shard": (res.engine.lo, res.engine.hi)}
    ctx.shutdown()


@pytest.mark.parametrize("algo", ["ga", "es"])
def test_dqn_two_ranks_equal_one_rank(algo):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, algo, ret), nprocs=2, join=True)
    if algo == "ga":
        single = _summary_ga(_run_ga(_ga_cfg())[2])
        pop, hof = 4, 2
        for rank in (0, 1):
            got = ret[rank]
            for k in ("elite_ids", "eval", "sigma", "hof"):
                assert got[k] == single[k], (rank, k)
            assert np.array_equal(np.array(got["fitness"], dtype=np.float32), np.array(single["fitness"], dtype=np.float32))
            lo, hi = got["shard"]
            for g, games in enumerate(got["games"]):
                full = np.array(single["games"][g]).reshape(2, pop * hof, 2)
                assert np.array_equal(np.array(games).reshape(2, (hi - lo) * hof, 2), full[:, lo * hof:hi * hof])
    else:
        args, env, base, res = _run_es(_es_cfg(population=8))
        for rank in (0, 1):
            got = ret[rank]
            assert got["base"] == [sha(b) for b in base] and got["eval"] == [res.rewards[r] for r in rp.DQN_ROLES]
            lo, hi = got["shard"]
            for g, games in enumerate(got["games"]):
                assert games == res.game_rewards[g].tolist()[2 * lo:2 * hi]


# ------------------------------------------------------------------------------------------------ env in host memory
@pytest.mark.parametrize("cohorts,threads", [(1, 1), (2, 3), (3, 16)])
def test_dqn_ga_host_frames_equal_device_frames(monkeypatch, cohorts, threads):
    """coevo_dqn_host_frames_rollout (the env in host memory: frames rendered by the host cores, copied up every agent-step,
    actions copied down - utils/game_logic_functions.py:47-53,84-120) against the device-resident rollout: every game's
    reward pair, fitness, elite ids, evaluation means and the final HoF weights of three generations are identical, for any
    number of alternating cohorts and host threads; and a rendered frame is the device twin's, byte for byte"""
    from coevonet_amd import lib as L
    from coevonet_amd.atari_synthetic import synth_frame
    buf = np.zeros(84 * 84 * 4, np.uint8)
    assert L.load().coevo_synth_frame_host(buf.ctypes.data, 4, 1870300, (1 << 33) + 5, 7, 3) == 0
    assert np.array_equal(buf, synth_frame(1870300, (1 << 33) + 5, 7, 3, 4).reshape(-1))
    want = _summary_ga(_run_ga(_ga_cfg())[2])
    monkeypatch.setenv("COEVO_FRAME_COHORTS", str(cohorts))
    monkeypatch.setenv("COEVO_FRAME_THREADS", str(threads))
    args, env, res = _run_ga(_ga_cfg(coevo_frames="host"))
    ro = res.engine.ro
    assert type(ro).__name__ == "HostFrameRollout" and len(ro.lanes) == cohorts and ro.threads == threads
    assert _summary_ga(res) == want


def test_dqn_es_host_frames_equal_device_frames(monkeypatch):
    """the same for Co-ES (main rollout in two cohorts, the ten evaluation games through the host env as well)"""
    _, _, base_d, res_d = _run_es(_es_cfg())
    monkeypatch.setenv("COEVO_FRAME_THREADS", "4")
    monkeypatch.setenv("COEVO_FRAME_COHORTS", "2")
    _, _, base_h, res_h = _run_es(_es_cfg(coevo_frames="host"))
    assert type(res_h.engine.ro).__name__ == "HostFrameRollout" and type(res_h.engine.eval_ro).__name__ == "HostFrameRollout"
    assert [g.tolist() for g in res_h.game_rewards] == [g.tolist() for g in res_d.game_rewards]
    assert [res_h.rewards[r] for r in rp.DQN_ROLES] == [res_d.rewards[r] for r in rp.DQN_ROLES]
    assert res_h.sigma_after == res_d.sigma_after
    for ri in range(2):
        assert sha(base_h[ri]) == sha(base_d[ri])
