"""Parity surface the small fixtures never reached: the adaptive-sigma increase branch on the device (ga_adapt_kernel:
numpy-pairwise means over h[-10:] / h[-20:-10], quirk Q5), long horizons, early stopping, odd populations with 2 and 3
cohorts, the other PettingZoo integration order on the device, and the exact launches bench.py times (cfg 2 at pop 200 /
HoF 5 with two cohorts, cfg 3 at pop 1000) checked against the oracle on a sample of games + everything derived from
all of them (fitness, elite ids, the ES update)."""
import copy

import os

import numpy as np
import pytest
import torch

from coevonet_amd import evolutionary_strategy as es
from coevonet_amd import genetic_algorithm as ga
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from oracle import ref_port as rp
from tests.util import SAFE_MARGIN, Bag, load_golden, sha

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROLES = ga.ROLES
SIG = ("mutation_power_agent_0", "mutation_power_agent_1", "mutation_power_adversary")


def _seed(s):
    torch.manual_seed(s)
    np.random.seed(s)


def _ga(cfg, rng, env_mode="device", **kw):
    _seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    env = initialize_env(args)
    env.max_cycles = cfg.get("max_cycles", 25)
    res = ga.genetic_algorithm_train(env, env.agents[0], args, None, rng=rng, env_mode=env_mode, **kw)
    return args, env, res


# ------------------------------------------------------------------------------------ adaptive sigma, long horizon
@pytest.mark.parametrize("upto", [12, 15, 19, 20, 25])
def test_adapt_sigma_kernel_vs_numpy_rule(upto):
    """coevo_ga_adapt_sigma alone on a synthetic history: bit for bit the reference's rule as numpy evaluates it
    (genetic_algorithm.py:323-345), generation by generation up to `upto` (h[-20:-10] partial below 20 entries)"""
    g = np.random.Generator(np.random.PCG64(upto))
    cap = 40
    ev = g.normal(0, 1, size=(upto + 1, 10, 3))
    # a downward drift for agent_0 and the adversary so that "worse" fires, an upward one for agent_1 (Q5 matters
    # exactly when agent_0 is worse and agent_1 is not)
    ev[:, :, 0] -= 0.3 * np.arange(upto + 1)[:, None]
    ev[:, :, 1] += 0.3 * np.arange(upto + 1)[:, None]
    ev[:, :, 2] -= 0.1 * np.arange(upto + 1)[:, None] * (np.arange(upto + 1)[:, None] % 3 == 0)
    args = Bag(mutation_power_agent_0=0.05, mutation_power_agent_1=0.08, mutation_power_adversary=0.03,
               max_mutation_power=0.1, min_mutation_power=0.02)
    gen_dev = torch.zeros(1, dtype=torch.int32, device=DEV)
    hist = torch.zeros(3, cap, dtype=torch.float64, device=DEV)
    sig_hist = torch.zeros(3, cap, dtype=torch.float64, device=DEV)
    s64 = torch.tensor([getattr(args, a) for a in SIG], dtype=torch.float64, device=DEV)
    s32 = torch.zeros(3, dtype=torch.float32, device=DEV)
    rewards = torch.zeros(10 + 7, 3, dtype=torch.float64, device=DEV)
    h = {r: [] for r in ROLES}
    ups = 0
    for e in range(upto + 1):
        # python side: evaluate_current_weights' means in its accumulation order, then the rule
        tot = [0.0, 0.0, 0.0]
        for i in range(10):
            for s in range(3):
                tot[s] += float(ev[e, i, s])
        for s, r in enumerate(ROLES):
            h[r].append(tot[s] / 10)
        before = [getattr(args, a) for a in SIG]
        ga.adapt_mutation_power(args, e, h)
        ups += sum(getattr(args, a) > b for a, b in zip(SIG, before))
        # device side: generation counter e+1 sees generation e's evaluation games
        rewards[7:] = torch.from_numpy(ev[e]).to(DEV)
        gen_dev.fill_(e + 1)
        L.call("coevo_ga_adapt_sigma", L._p(rewards), 7, L._p(gen_dev), L._p(hist), L._p(sig_hist), cap, L._p(s64),
               L._p(s32), args.min_mutation_power, args.max_mutation_power, 1)
        want = [getattr(args, a) for a in SIG]
        assert s64.cpu().tolist() == want, e
        assert s32.cpu().tolist() == [float(np.float32(w)) for w in want]
    assert hist[:, :upto + 1].cpu().numpy().T.tolist() == [[h[r][e] for r in ROLES] for e in range(upto + 1)]
    assert ups >= 3


def test_ga_long_fixture_host_reference_on_device():
    """26 generations against what the reference produced (tests/golden/ga_long.json): elite ids, margin-safe game
    rewards, evaluation rewards, the whole sigma trajectory (both branches), final HoF weights"""
    fx = load_golden("ga_long.json")
    args, env, res = _ga(fx["config"], "host_reference")
    for g, ref in enumerate(fx["generations"]):
        assert res.elite_ids[g] == ref["elite_ids"], g
        for i, rg in enumerate(ref["games"][:12]):
            if rg["min_margin"] > SAFE_MARGIN:
                assert list(res.game_rewards[g][i]) == rg["rewards"], (g, i)
        np.testing.assert_allclose([res.rewards[r][g] for r in ROLES], ref["eval_rewards"], rtol=1e-12)
        assert res.sigma_after[g] == ref["sigma_after"], g
    last = {s["file"]: s["agents"] for s in fx["generations"][-1]["saves"]}
    eng = res.engine
    assert [sha(w) for w in eng.download("agent_1", "hof", 0, 1)] == [a["sha256"] for a in last["hall_of_fame_agent_1.pth"]]
    assert [sha(w) for w in eng.download("adversary_0", "elite", 0, 2)] == \
        [a["sha256"] for a in last["elite_weights_adversary.pth"]]


@pytest.mark.parametrize("pop,cohorts,gens", [(4, 2, 26), (11, 2, 3), (10, 3, 3), (9, 2, 3)])
def test_ga_device_loop_matches_oracle_port(pop, cohorts, gens):
    """the host-free generation loop (device_philox offspring, sigma rule on the device) against the sequential oracle
    port with the same counter-based noise.  (4, 2, 26): long horizon - ga_adapt_kernel's increase branch, Q5.
    Odd per-rank populations and three cohorts: the plan's cohort of every game == the cohort that breeds / resets it."""
    long_run = gens > 10
    cfg = {"seed": 21 if long_run else 5,
           "args": dict(generations=gens, population=pop, hof_size=1 if long_run else 3, elites_number=2,
                        fitness_sharing=True, max_timesteps_per_episode=9 if long_run else 40,
                        max_evaluation_steps=9 if long_run else 75, mutation_power_agent_0=0.05,
                        mutation_power_agent_1=0.08, mutation_power_adversary=0.03, max_mutation_power=0.1,
                        min_mutation_power=0.02, coevo_cohorts=cohorts)}
    args, env, res = _ga(cfg, "device_philox")
    eng = res.engine
    assert eng.K == cohorts and eng.ro.n_cohorts == cohorts
    for k in range(cohorts):  # one partition everywhere
        lo_k, hi_k = eng._cohort_individuals(k)
        games = np.nonzero(eng.plan.game_cohort_np[:eng.n_main] == k)[0]
        inds = np.unique((games % (pop * eng.hof)) // eng.hof)
        assert np.array_equal(inds, np.arange(lo_k, hi_k))
    oargs = copy.deepcopy(cfg["args"])
    oargs.pop("coevo_cohorts")
    _seed(cfg["seed"])
    oargs = Bag(algorithm="GA", **oargs)
    want = rp.ga_train(oargs, noise="philox", philox_seed=0)
    hof = oargs.hof_size
    ups, prev = 0, None
    for g, w in enumerate(want):
        assert res.elite_ids[g] == w["elite_ids"], g
        for i in range(3 * pop * hof):
            assert list(res.game_rewards[g][i]) == w["games"][i]["rewards"], (g, i)
        assert [res.rewards[r][g] for r in ROLES] == w["eval_rewards"], g
        assert res.sigma_after[g] == w["sigma_after"], g
        if prev is not None:
            ups += sum(a > b for a, b in zip(w["sigma_after"], prev))
        prev = w["sigma_after"]
    assert not long_run or ups >= 8
    for role in ROLES:
        assert [sha(x) for x in eng.download(role, "hof", 0, hof)] == [sha(x) for x in want[-1]["hof"][role]]
        assert [sha(x) for x in eng.download(role, "pop", 0, pop)] == \
            [sha(x) for x in [want[-1]["elites"][role][0]] + _children(want[-1], role, oargs, gens - 1, pop)]


def _children(rec, role, args, gen, pop):
    ri = ROLES.index(role)
    sig_before = rec["sigma_before"][ri]
    return [rp.mutate_philox(rec["elites"][role][c % args.elites_number], rp.ROLE_D[role], np.float32(sig_before), 0, c,
                             gen * 4 + ri) for c in range(pop - 1)]


# ------------------------------------------------------------------------------------ Co-ES long horizon
@pytest.mark.parametrize("name", ["es_long.json", "es_stop.json"])
def test_es_long_fixture_and_early_stopping_on_device(name):
    """every generation's margin-safe games == the reference's; sigma trajectory; early stopping stops where the
    reference stopped (evolutionary_strategy.py:320-354)"""
    fx = load_golden(name)
    cfg = fx["config"]
    _seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    env = initialize_env(args)
    agents, res = es.evolution_strategy_train(env, args, None, rng="host_reference", return_result=True)
    assert len(res.rewards["agent_0"]) == len(fx["generations"])
    assert res.stopped_at == fx["stopped_at"]
    pop = args.population
    exact, n_safe, n_all = True, 0, 0
    for g, ref in enumerate(fx["generations"]):
        for i, rg in enumerate(ref["games"][:3 * pop]):
            n_all += 1
            if rg["min_margin"] > SAFE_MARGIN:
                assert list(res.game_rewards[g][i]) == rg["rewards"], (g, i)
                n_safe += 1
        exact = exact and all(x["min_margin"] > SAFE_MARGIN for x in ref["games"][-10:])
        if exact:
            np.testing.assert_allclose([res.rewards[r][g] for r in ROLES], ref["eval_rewards"], rtol=1e-12)
            assert res.sigma_after[g] == ref["sigma_after"], g
    assert n_safe >= 0.8 * n_all
    assert env.n_resets == fx["env_resets"]


# ------------------------------------------------------------------------------------ the other integration order
@pytest.mark.parametrize("merged", [True, False])
def test_rollout_velocity_first_integration_order(merged):
    """INTEGRATE_POS_FIRST = False (the other PettingZoo release order) through the fused device env step"""
    from coevonet_amd.rollout import DeviceRollout, RolloutPlan
    from tests.test_kernels_gpu import make_nets, to_slab
    npop, nh, limit = 12, 3, 40
    nets10 = make_nets(npop + nh, 10, seed=91, mutate=False)
    nets8 = make_nets(nh, 8, seed=92, mutate=False)
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(nets10, 10).reshape(-1), to_slab(nets8, 8).reshape(-1)]).contiguous()
    off = [i * s10 for i in range(npop + nh)] + [(npop + nh) * s10 + k * s8 for k in range(nh)]
    D = [10] * (npop + nh) + [8] * nh
    games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]
    plan = RolloutPlan(np.array(games), off, D, device=DEV, heavy_rows=16)
    ro = DeviceRollout(plan, slab, merged=merged)
    ro.pos_first = 0
    ro.desc.pos_first = 0
    ro.set_limits(np.full(plan.n_games, limit))
    ro.reset(0, plan.n_games, 3)
    ro.run((limit + 2) // 3)
    torch.cuda.synchronize()
    ro.check_status()
    r = ro.rewards.cpu().numpy()
    stream = rp.Stream()
    differs = 0
    try:
        for g, (adv, a0, a1) in enumerate(games):
            rp.lib().oracle_mpe_set_pos_first(0)
            want = rp.play_game(stream, nets10[a0], nets10[a1], nets8[adv - npop - nh], limit, 25, ordinal=3 + g)
            assert list(r[g]) == want["rewards"], g
            rp.lib().oracle_mpe_set_pos_first(1)
            other = rp.play_game(stream, nets10[a0], nets10[a1], nets8[adv - npop - nh], limit, 25, ordinal=3 + g)
            differs += other["rewards"] != want["rewards"]
    finally:
        rp.lib().oracle_mpe_set_pos_first(1)
    assert differs > 0  # the two orders are not the same thing


# ------------------------------------------------------------------------------------ the launches bench.py times
def test_cfg2_full_size_generation_vs_oracle():
    """BASELINE configs[1] at full size (pop 200, HoF 5, two cohorts, lean kernel, pipelined breeding), two
    generations: all 600 deciding games of generation 0 (fitness and elite ids follow from them), a strided sample of
    the others and of generation 1's games (bred children, pushed HoF), the evaluation rewards."""
    pop, hof, E = 200, 5, 2
    cfg = {"seed": 0, "args": dict(generations=2, population=pop, hof_size=hof, elites_number=E, fitness_sharing=True,
                                   max_timesteps_per_episode=200, max_evaluation_steps=200)}
    args, env, res = _ga(cfg, "device_philox")
    eng = res.engine
    assert eng.K == 2 and eng.plan.heavy_max <= 16 and bool(eng.ro.desc.merged)
    assert 900 <= len(eng.plan.heavy_np) + len(eng.plan.light_np) <= 4 * 256  # both cohorts resident, 4 per CU
    _seed(0)
    hofs, popu = rp.ga_initial(pop, hof)
    M = 3 * pop * hof
    st = rp.Stream()

    def game(gen, ph, i, k, nets):
        a0, a1, adv = rp.ga_game_nets(ROLES[ph], nets[ROLES[ph]][i], hofs, k, hof)
        return rp.play_game(st, a0, a1, adv, 200, 25, ordinal=1 + gen * (M + 10) + ph * pop * hof + i * hof + k)

    got0 = res.game_rewards[0].reshape(3, pop, hof, 3)
    elites = {}
    for ph, role in enumerate(ROLES):
        D = rp.ROLE_D[role]
        div = rp.diversity(rp.weights_es(popu[role][-1], D), [rp.weights_es(w, D) for w in popu[role]])
        fit = []
        for i in range(pop):
            w = game(0, ph, i, hof - 1, popu)                     # Q2: the last HoF game decides
            assert list(got0[ph, i, hof - 1]) == w["rewards"], (ph, i)
            fit.append(w["rewards"][ph] / hof / (1 + div))
        for i in range(ph, pop, 23):                              # a sample of the games that only burn resets
            for k in range(hof - 1):
                assert list(got0[ph, i, k]) == game(0, ph, i, k, popu)["rewards"], (ph, i, k)
        np.testing.assert_allclose(res.fitness[0][ph], fit, rtol=2e-6)
        order = np.argsort(np.array(res.fitness[0][ph], dtype=np.float32))[::-1]
        assert res.elite_ids[0][ph] == [int(x) for x in order[:E]]
        assert res.elite_ids[0][ph] == [int(x) for x in np.argsort(fit)[::-1][:E]]
        elites[role] = [popu[role][i] for i in res.elite_ids[0][ph]]
    # generation 1: HoF pushed, population = [best] + children(elite[c % E], sigma 0.05, stream (c, ri))
    for role in ROLES:
        hofs[role].append(elites[role][0])
        hofs[role].pop(0)
    ev = [0.0, 0.0, 0.0]
    for j in range(10):
        w = rp.play_game(st, elites["agent_0"][0], elites["agent_1"][0], elites["adversary_0"][0], 200, 25,
                         ordinal=1 + M + j)
        for s in range(3):
            ev[s] += w["rewards"][s]
    assert [res.rewards[r][0] for r in ROLES] == [e / 10 for e in ev]
    got1 = res.game_rewards[1].reshape(3, pop, hof, 3)
    for ph, role in enumerate(ROLES):
        for i in [0, 1, 2, 99, 100, 101, 198, 199] + list(range(5 + ph, pop, 29)):   # both cohorts, their seam
            net = elites[role][0] if i == 0 else rp.mutate_philox(elites[role][(i - 1) % E], rp.ROLE_D[role],
                                                                  np.float32(0.05), 0, i - 1, ph)
            for k in (0, hof - 1) if i % 2 else (2,):
                a0, a1, adv = rp.ga_game_nets(role, net, hofs, k, hof)
                w = rp.play_game(st, a0, a1, adv, 200, 25, ordinal=1 + (M + 10) + ph * pop * hof + i * hof + k)
                assert list(got1[ph, i, k]) == w["rewards"], (ph, i, k)


def test_cfg3_full_size_generation_vs_oracle():
    """BASELINE configs[2] (reference_exact variant) at pop 1000: a strided sample of generation 0's 3000 games against
    the oracle, and the update of the full n = 1000 bit for bit (oracle update fed with the device's rewards)."""
    pop = 1000
    cfg = {"seed": 3, "args": dict(generations=1, population=pop, hof_size=1, learning_rate=0.1,
                                   max_timesteps_per_episode=200, max_evaluation_steps=200)}
    _seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    env = initialize_env(args)
    agents, res = es.evolution_strategy_train(env, args, None, rng="device_philox", return_result=True)
    _seed(cfg["seed"])
    base = {r: rp.init_net(rp.ROLE_D[r]) for r in ROLES}
    st = rp.Stream()
    got = res.game_rewards[0]
    for j in list(range(0, pop, 41)) + [pop - 1]:
        for s, r in enumerate(ROLES):
            nets = dict(base)
            nets[r] = rp.mutate_philox(base[r], rp.ROLE_D[r], np.float32(0.05), 0, j, s, skip_layernorm=True)
            w = rp.play_game(st, nets["agent_0"], nets["agent_1"], nets["adversary_0"], 200, 25, ordinal=1 + 3 * j + s)
            assert list(got[3 * j + s]) == w["rewards"], (j, r)
    new = {}
    for s, r in enumerate(ROLES):
        D = rp.ROLE_D[r]
        pert = np.stack([rp.mutate_philox(base[r], D, np.float32(0.05), 0, j, s, skip_layernorm=True)
                         for j in range(pop)])
        f = np.array([got[3 * j + s][ga.RET_SLOT[r]] for j in range(pop)], dtype=np.float32)
        new[r] = rp.es_update_from_pert(base[r], D, pert, f, 0.05, 0.1)
        del pert
    for a, r in zip(agents, ROLES):
        assert sha(a.model.flat()) == sha(new[r]), r
    ev = [0.0, 0.0, 0.0]
    for j in range(10):
        w = rp.play_game(st, new["agent_0"], new["agent_1"], new["adversary_0"], 200, 25, ordinal=1 + 3 * pop + j)
        for s in range(3):
            ev[s] += w["rewards"][s]
    assert [res.rewards[r][0] for r in ROLES] == [e / 10 for e in ev]


def test_cfg2_full_size_host_cores_equal_device(monkeypatch):
    """BASELINE configs[1] at full size with the env on the host cores (north_star's first architecture: 2 cores x 4 alternating
    cohorts of ~752 games, mapped page-locked buffers, the lean merged kernel fed with observations) against the device-resident
    loop: every one of the 2 x 3000 play_game triples, the fitness bits, elite ids and evaluation means of two generations are
    identical - the full-size form of test_host_cores_rollout_equals_device_rollout (utils/game_logic_functions.py:123-212)"""
    pop, hof, E = 200, 5, 2
    cfg = {"seed": 0, "args": dict(generations=2, population=pop, hof_size=hof, elites_number=E, fitness_sharing=True,
                                   max_timesteps_per_episode=200, max_evaluation_steps=200)}
    _, _, want = _ga(cfg, "device_philox")
    for var in ("COEVO_HOST_COHORTS", "COEVO_HOST_THREADS", "COEVO_HOST_ZERO_COPY", "COEVO_HOST_SIGNAL"):
        monkeypatch.delenv(var, raising=False)
    _, _, res = _ga(cfg, "device_philox", env_mode="host")
    ro = res.engine.ro
    assert ro.impl == "native" and ro.plan.n_cohorts == 4 and ro.threads == 2 and ro.zero_copy
    assert [len(g) for g in ro._cohort_games] == [750, 750, 750, 760] and ro.plan.heavy_max <= 16
    for g in range(2):
        assert res.elite_ids[g] == want.elite_ids[g]
        assert np.array_equal(np.asarray(res.game_rewards[g]).view(np.uint64), np.asarray(want.game_rewards[g]).view(np.uint64))
        assert np.array_equal(np.asarray(res.fitness[g], np.float32).view(np.uint32),
                              np.asarray(want.fitness[g], np.float32).view(np.uint32))
        assert [res.rewards[r][g] for r in ROLES] == [want.rewards[r][g] for r in ROLES]


@pytest.mark.parametrize("world,rank", [(8, 0), (8, 5), (4, 3)])
def test_cfg2_one_rank_of_the_split_vs_oracle(world, rank):
    """BASELINE's own split (pop 200 over 4 / 8 GPUs: `--gpus N`, genetic_algorithm.py:125-217 by individual index): the
    launches ONE rank runs - 25 / 50 individuals per role, shared opponents in hof-row chunks, every task through the
    small-launch body (fc2 on the vector ALU) and the whole rollout as ONE persistent launch (a rank of 8: a workgroup per CU,
    of 4: two) - against the oracle: all of the rank's deciding games of generation 0, a
    sample of the others, and generation 1's games of bred children (the rank's own children; the elites are whatever
    dist.ShardRehearsal's made-up gather selected, read back from the engine)"""
    from coevonet_amd.dist import ShardRehearsal
    pop, hof, E = 200, 5, 2
    n_local = pop // world
    lo = rank * n_local
    cfg = {"seed": 0, "args": dict(generations=2, population=pop, hof_size=hof, elites_number=E, fitness_sharing=True,
                                   max_timesteps_per_episode=200, max_evaluation_steps=200)}
    args, env, res = _ga(cfg, "device_philox", dist_ctx=ShardRehearsal(rank, world))
    eng = res.engine
    assert (eng.lo, eng.hi, eng.K) == (lo, lo + n_local, 1) and eng.plan.light_max == hof
    shape = (len(eng.plan.heavy_np), len(eng.plan.light_np), eng.plan.heavy_max, eng.plan.light_max, 1)
    if os.environ.get("COEVO_PERSISTENT", "1") != "0":
        assert eng.plan.heavy_max == hof and eng.ro.sync_words is not None
        assert L.load().coevo_mpe_persistent_fits(*shape) == 1
    else:   # (A/B runs of the suite: launches per env-cycle - the small-launch kernel for a rank of 8, the lean one for 4)
        assert eng.plan.heavy_max == (hof if world == 8 else 16)
    assert L.load().coevo_mpe_cycle_kernel_form(*shape) == (3 if world == 8 else 2)   # what COEVO_PERSISTENT=0 launches
    _seed(0)
    hofs, popu = rp.ga_initial(pop, hof)
    M = 3 * pop * hof
    st = rp.Stream()
    got0 = res.game_rewards[0].reshape(3, n_local, hof, 3)
    for ph, role in enumerate(ROLES):
        for j in range(n_local):
            i = lo + j
            for k in ([hof - 1] if (j + ph) % 5 else range(hof)):   # Q2: the deciding game of everyone, all games of a fifth
                a0, a1, adv = rp.ga_game_nets(role, popu[role][i], hofs, k, hof)
                w = rp.play_game(st, a0, a1, adv, 200, 25, ordinal=1 + ph * pop * hof + i * hof + k)
                assert list(got0[ph, j, k]) == w["rewards"], (ph, i, k)
    # generation 1: the elites every rank rebuilt are generation 0's individuals order[:E] (ids read back); child c =
    # individual c + 1 = mutate(elite[c % E]) with stream (c, role); HoF pushed
    ids = res.elite_ids[0]
    got1 = res.game_rewards[1].reshape(3, n_local, hof, 3)
    elites = {role: [popu[role][i] for i in ids[ph]] for ph, role in enumerate(ROLES)}
    for role in ROLES:
        hofs[role].append(elites[role][0])
        hofs[role].pop(0)
    for ph, role in enumerate(ROLES):
        for j in range(0, n_local, 4):
            i = lo + j
            net = elites[role][0] if i == 0 else rp.mutate_philox(elites[role][(i - 1) % E], rp.ROLE_D[role], np.float32(0.05),
                                                                  0, i - 1, ph)
            for k in (0, hof - 1):
                a0, a1, adv = rp.ga_game_nets(role, net, hofs, k, hof)
                w = rp.play_game(st, a0, a1, adv, 200, 25, ordinal=1 + (M + 10) + ph * pop * hof + i * hof + k)
                assert list(got1[ph, j, k]) == w["rewards"], (ph, i, k)
