"""Whole Co-GA generations on the MI355X (drop-in genetic_algorithm_train) against the golden fixtures minted from the
reference, and against the oracle port for the device-Philox offspring mode."""
import copy

import numpy as np
import pytest
import torch

from coevonet_amd import genetic_algorithm as ga
from coevonet_amd.game_logic import create_agent, initialize_env, play_game
from oracle import ref_port as rp
from tests.util import SAFE_MARGIN, Bag, load_golden, sha

pytestmark = pytest.mark.gpu
ROLE_FILES = {"agent_0": ("hall_of_fame_agent_0.pth", "elite_weights_agent_0.pth"),
              "agent_1": ("hall_of_fame_agent_1.pth", "elite_weights_agent_1.pth"),
              "adversary_0": ("hall_of_fame_adversary.pth", "elite_weights_adversary.pth")}


def _run(cfg, rng, env_mode):
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    env = initialize_env(args)
    env.max_cycles = cfg.get("max_cycles", 25)
    res = ga.genetic_algorithm_train(env, env.agents[0], args, None, rng=rng, env_mode=env_mode)
    return args, env, res


@pytest.mark.parametrize("name", ["ga_cfg1.json", "ga_hof2.json"])
@pytest.mark.parametrize("env_mode", ["device", "host"])
def test_ga_matches_reference_fixture(name, env_mode):
    """host_reference RNG: elite ids exact, fitness rel-err <= 1e-5, per-game rewards bit-exact on margin-safe games,
    final HoF / elite weights identical (sha256) to what the reference saved."""
    fx = load_golden(name)
    cfg = fx["config"]
    args, env, res = _run(cfg, "host_reference", env_mode)
    pop, hof = args.population, args.hof_size
    for g, ref in enumerate(fx["generations"]):
        assert res.elite_ids[g] == ref["elite_ids"], f"generation {g}"
        for ph in range(3):
            np.testing.assert_allclose(res.fitness[g][ph], ref["fitness"][ph], rtol=1e-5, atol=1e-7)
        # the score is a sum of (1 - d/mean d) terms: absolute error ~ fp32 eps of 1.0, whatever the score's size
        np.testing.assert_allclose(res.diversity[g], ref["diversity"], rtol=1e-5, atol=2e-7)
        got = res.game_rewards[g]
        n_safe = 0
        for i, rg in enumerate(ref["games"][:3 * pop * hof]):
            if rg["min_margin"] > SAFE_MARGIN:
                assert list(got[i]) == rg["rewards"], (g, i)
                n_safe += 1
        assert n_safe >= 0.9 * 3 * pop * hof
        np.testing.assert_allclose([res.rewards[r][g] for r in ga.ROLES], ref["eval_rewards"], rtol=1e-12)
        assert res.sigma_after[g] == ref["sigma_after"]
    last = {s["file"]: s["agents"] for s in fx["generations"][-1]["saves"]}
    eng = res.engine
    for role, (hf, ef) in ROLE_FILES.items():
        assert [sha(w) for w in eng.download(role, "hof", 0, hof)] == [a["sha256"] for a in last[hf]]
        assert [sha(w) for w in eng.download(role, "elite", 0, args.elites_number)] == [a["sha256"] for a in last[ef]]
    assert env.n_resets == fx["env_resets"]


@pytest.mark.parametrize("cohorts", [2, 1])
def test_ga_device_philox_matches_oracle_port(cohorts):
    """performance mode (offspring built on the device): every number equals the sequential CPU port run with the
    same counter-based noise - elite ids, fp32 fitness bits, fp64 game rewards, final weights.  cohorts=2: the pipelined
    generation (each cohort stream breeds, resets and rolls out its half); cohorts=1: the whole-generation hipGraph."""
    cfg = {"seed": 5, "args": dict(generations=3, population=10, hof_size=3, elites_number=2, fitness_sharing=True,
                                   max_timesteps_per_episode=40, max_evaluation_steps=75, coevo_cohorts=cohorts)}
    args, env, res = _run(cfg, "device_philox", "device")
    assert res.engine.ro.n_cohorts == cohorts and res.engine.K == cohorts
    cfg["args"].pop("coevo_cohorts")
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    oargs = Bag(algorithm="GA", **cfg["args"])
    want = rp.ga_train(oargs, noise="philox", philox_seed=0)
    pop, hof = oargs.population, oargs.hof_size
    for g, w in enumerate(want):
        assert res.elite_ids[g] == w["elite_ids"]
        got = res.game_rewards[g]
        for i in range(3 * pop * hof):
            assert list(got[i]) == w["games"][i]["rewards"], (g, i)
        for ph in range(3):  # the sharing score's distances are summed in fp64 here, by an fp32 BLAS dot in numpy
            np.testing.assert_allclose(res.fitness[g][ph], w["fitness"][ph], rtol=2e-6)
        assert [res.rewards[r][g] for r in ga.ROLES] == w["eval_rewards"]
        assert res.sigma_after[g] == w["sigma_after"]
    eng = res.engine
    for role in ga.ROLES:
        assert [sha(x) for x in eng.download(role, "hof", 0, hof)] == [sha(x) for x in want[-1]["hof"][role]]


@pytest.mark.parametrize("hof", [7, 10, 17])
def test_ga_larger_hof_sizes_match_oracle_port(hof):
    """HoF sizes beyond the fixtures' 1 - 5: 7 (an individual's 7 games of a cycle = one streaming task of the 8-row
    instantiation), 10 (more than 8 rows: the individuals' games become matrix-core tasks of <= 16 rows, as the HoF members'
    are) and 17 (tasks of 16 + 1 rows; HoF promotion outside the fused tail graph).  genetic_algorithm.py:125-290."""
    cfg = {"seed": 11 + hof, "args": dict(generations=2, population=6, hof_size=hof, elites_number=2, fitness_sharing=True,
                                          max_timesteps_per_episode=30, max_evaluation_steps=40)}
    args, env, res = _run(cfg, "device_philox", "device")
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    oargs = Bag(algorithm="GA", **cfg["args"])
    want = rp.ga_train(oargs, noise="philox", philox_seed=0)
    pop = oargs.population
    for g, w in enumerate(want):
        assert res.elite_ids[g] == w["elite_ids"]
        got = res.game_rewards[g]
        for i in range(3 * pop * hof):
            assert list(got[i]) == w["games"][i]["rewards"], (g, i)
        for ph in range(3):
            np.testing.assert_allclose(res.fitness[g][ph], w["fitness"][ph], rtol=2e-6)
        assert [res.rewards[r][g] for r in ga.ROLES] == w["eval_rewards"]
    for role in ga.ROLES:
        assert [sha(x) for x in res.engine.download(role, "hof", 0, hof)] == [sha(x) for x in want[-1]["hof"][role]]


def test_play_game_facade_matches_fixture():
    """play_game(env, p1, p2, adversary, args) through this package's surface == the reference's returns"""
    for case in load_golden("play_game.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        np.random.seed(case["torch_seed"])
        args = Bag(max_timesteps_per_episode=case["limit"], max_evaluation_steps=case["limit"])
        env = initialize_env(args)
        env.max_cycles = case["max_cycles"]
        a0, a1, adv = (create_agent(env, args, r) for r in ("agent_0", "agent_1", "adversary_0"))
        assert [sha(a.model.flat()) for a in (a0, a1, adv)] == [w["sha256"] for w in case["weights"]]
        for g in case["games"]:
            got = play_game(env=env, player1=a0.model, player2=a1.model, adversary=adv.model, args=args, eval=False)
            if g["min_margin"] > SAFE_MARGIN:
                assert list(got) == g["rewards"]
    with pytest.raises(ValueError):
        play_game(env=env, player1=a0.model, player2=a1.model, adversary=None, args=args)


def test_aec_loop_with_device_forward():
    """foreign-env path: the AEC loop with one HIP forward per agent-step gives the same triple"""
    from coevonet_amd.game_logic import _play_mpe_aec
    case = load_golden("play_game.json")["cases"][3]  # limit 7: short
    torch.manual_seed(case["torch_seed"])
    args = Bag(max_timesteps_per_episode=case["limit"], max_evaluation_steps=case["limit"])
    env = initialize_env(args)
    a0, a1, adv = (create_agent(env, args, r) for r in ("agent_0", "agent_1", "adversary_0"))
    env.reset()
    got = _play_mpe_aec(env, a0.model, a1.model, adv.model, args, False)
    assert list(got) == case["games"][0]["rewards"]


def test_save_and_metrics_roundtrip(tmp_path):
    """--save writes the reference's file names with lists of Agent objects that load back (what
    load_agent_for_testing + main.py --test consume), and one metrics line per generation"""
    import json
    import os
    from coevonet_amd.io_utils import GA_FILES, load_agents
    cfg = {"seed": 3, "args": dict(generations=2, population=5, hof_size=2, elites_number=2, save=True,
                                   max_timesteps_per_episode=20, max_evaluation_steps=20)}
    torch.manual_seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    env = initialize_env(args)
    res = ga.genetic_algorithm_train(env, env.agents[0], args, str(tmp_path), rng="device_philox")
    for role, (hof_file, elite_file) in GA_FILES.items():
        hof = load_agents(os.path.join(tmp_path, hof_file))
        assert len(hof) == 2 and hasattr(hof[-1], "model") and hasattr(hof[-1].model, "determine_action")
        assert [sha(a.model.flat()) for a in hof] == [sha(w) for w in res.engine.download(role, "hof", 0, 2)]
        assert len(load_agents(os.path.join(tmp_path, elite_file))) == 2
    # the weights_only-safe twins: tensors only, loadable without unpickling code, same weights
    from coevonet_amd.io_utils import agents_from_state_dicts, state_dict_path
    for role, (hof_file, elite_file) in GA_FILES.items():
        path = state_dict_path(os.path.join(tmp_path, hof_file))
        payload = torch.load(path, weights_only=True)
        assert payload["format"] == "coevonet_amd.state_dict.v1" and len(payload["agents"]) == 2
        assert set(payload["agents"][0]) == {"fc1.weight", "fc1.bias", "ln1.weight", "ln1.bias", "fc2.weight", "fc2.bias",
                                             "ln2.weight", "ln2.bias", "output.weight", "output.bias"}
        back = agents_from_state_dicts(env, args, role, path)
        assert [sha(a.model.flat()) for a in back] == [sha(w) for w in res.engine.download(role, "hof", 0, 2)]
    lines = [json.loads(x) for x in open(os.path.join(tmp_path, "metrics.jsonl"))]
    assert len(lines) == 2 and lines[1]["generation"] == 1 and set(lines[0]["eval_rewards"]) == set(ga.ROLES)
    # a reloaded trio plays through the drop-in play_game like main.py --test does
    trio = [load_agents(os.path.join(tmp_path, GA_FILES[r][0]))[-1] for r in ("agent_0", "agent_1", "adversary_0")]
    out = play_game(env=env, player1=trio[0].model, player2=trio[1].model, adversary=trio[2].model, args=args, eval=True)
    assert len(out) == 3 and all(np.isfinite(out))


@pytest.mark.parametrize("cohorts,threads,zero_copy,signal", [(1, 1, 0, "flag"), (2, 4, 0, "event"), (3, 2, 0, "flag"),
                                                             (2, 3, 1, "event"), (4, 8, 1, "flag")])
def test_host_cores_rollout_equals_device_rollout(monkeypatch, cohorts, threads, zero_copy, signal):
    """coevo_mpe_host_rollout (env on the host cores: K alternating cohorts, T threads, staged copies or mapped page-locked
    buffers, completion words or events) against the device env: every play_game triple, fitness bits, elite ids and evaluation means of three
    generations are identical - the partition and the thread count change the schedule, never a number
    (utils/game_logic_functions.py:123-212)."""
    cfg = {"seed": 11, "args": dict(generations=3, population=13, hof_size=3, elites_number=2, fitness_sharing=True,
                                    max_timesteps_per_episode=41, max_evaluation_steps=75)}
    _, _, want = _run(cfg, "device_philox", "device")
    monkeypatch.setenv("COEVO_HOST_COHORTS", str(cohorts))
    monkeypatch.setenv("COEVO_HOST_THREADS", str(threads))
    monkeypatch.setenv("COEVO_HOST_ZERO_COPY", str(zero_copy))
    monkeypatch.setenv("COEVO_HOST_SIGNAL", signal)   # completion through a stream-written host word / through an event
    _, _, res = _run(cfg, "device_philox", "host")
    ro = res.engine.ro
    assert ro.impl == "native" and ro.threads == min(threads, cohorts) and ro.plan.n_cohorts == cohorts and ro.zero_copy == bool(zero_copy)
    for g in range(3):
        assert res.elite_ids[g] == want.elite_ids[g]
        assert np.array_equal(np.asarray(res.game_rewards[g]).view(np.uint64), np.asarray(want.game_rewards[g]).view(np.uint64))
        assert np.array_equal(np.asarray(res.fitness[g], np.float32).view(np.uint32),
                              np.asarray(want.fitness[g], np.float32).view(np.uint32))
        assert [res.rewards[r][g] for r in ga.ROLES] == [want.rewards[r][g] for r in ga.ROLES]


def test_host_cores_rollout_tiny_and_ragged(monkeypatch):
    """edge shapes of the host-cores rollout: more cohorts asked for than individuals (clamped), a one-agent-step limit (only the
    adversary acts: no world step at all), an evaluation horizon longer than the training one, an odd population - all equal
    to the device env, and the env's reset counter ends where the reference's would (utils/game_logic_functions.py:197,217)"""
    cfg = {"seed": 3, "args": dict(generations=2, population=3, hof_size=1, elites_number=1, fitness_sharing=True,
                                   max_timesteps_per_episode=1, max_evaluation_steps=7)}
    _, env_d, want = _run(cfg, "device_philox", "device")
    monkeypatch.setenv("COEVO_HOST_COHORTS", "8")
    monkeypatch.setenv("COEVO_HOST_THREADS", "5")
    _, env_h, res = _run(cfg, "device_philox", "host")
    assert res.engine.ro.plan.n_cohorts <= 3 and res.engine.ro.threads <= res.engine.ro.plan.n_cohorts
    assert env_h.n_resets == env_d.n_resets
    for g in range(2):
        assert res.elite_ids[g] == want.elite_ids[g]
        assert np.array_equal(np.asarray(res.game_rewards[g]).view(np.uint64), np.asarray(want.game_rewards[g]).view(np.uint64))
        assert [res.rewards[r][g] for r in ga.ROLES] == [want.rewards[r][g] for r in ga.ROLES]


def _host_trainer_run(cfg, before_step=None):
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    env = initialize_env(args)
    tr = ga.GATrainer(env, args, rng="device_philox", env_mode="host")
    for _ in range(args.generations):
        if before_step:
            before_step(tr)
        tr.step()
    return tr, tr.finish()


@pytest.mark.parametrize("pin", ["1", "0", "far"])
def test_host_cores_placement_modes_change_no_number(monkeypatch, pin):
    """where the host cores run (csrc/host_placement.hip: the GPU's NUMA node by default, unpinned with COEVO_HOST_PIN=0, a
    node that is NOT the GPU's with =far) changes the schedule, never a number; the caller's affinity mask is the one it came
    with after every rollout, the chosen CPUs lie inside it, and the page-locked staging buffers come from the context
    (utils/game_logic_functions.py:138,179-190 on "the host cores")"""
    import os
    cfg = {"seed": 5, "args": dict(generations=2, population=9, hof_size=2, elites_number=2, fitness_sharing=True,
                                   max_timesteps_per_episode=30, max_evaluation_steps=75)}
    _, _, want = _run(cfg, "device_philox", "device")
    monkeypatch.setenv("COEVO_HOST_PIN", pin)
    monkeypatch.setenv("COEVO_HOST_COHORTS", "2")
    before = os.sched_getaffinity(0)
    tr, res = _host_trainer_run(cfg)
    assert os.sched_getaffinity(0) == before
    pl = tr.eng.ro.placement
    print("placement", pin, pl)
    assert pl["pinned"] == (pin != "0")
    if pin != "0":
        assert set(pl["cpus"]) <= before and len(set(pl["cpus"])) == len(pl["cpus"]) == tr.eng.ro.threads
        if pin == "1" and pl["gpu_numa_node"] >= 0 and pl["on_gpu_node"]:
            assert pl["cpu_numa_node"] == pl["gpu_numa_node"]
    for g in range(2):
        assert res.elite_ids[g] == want.elite_ids[g]
        assert np.array_equal(np.asarray(res.game_rewards[g]).view(np.uint64), np.asarray(want.game_rewards[g]).view(np.uint64))
    tr.eng.ro.close()


@pytest.mark.parametrize("seeded", [0x7FFFFFF0, 0xFFFFFFF0])
def test_host_cores_rollout_restarts_its_sequence_numbers(monkeypatch, seeded):
    """the completion / gate words are 32-bit sequence numbers waited for with '>='; left to grow across rollouts
    they would cross 2^31 / 2^32 after days and a pre-queued launch would then start on the previous cycle's observations.
    A rollout restarts them while its lanes are idle: seeded just below either wrap before EVERY rollout, results stay those
    of the device env"""
    from coevonet_amd import lib as L
    cfg = {"seed": 7, "args": dict(generations=2, population=8, hof_size=2, elites_number=2, fitness_sharing=True,
                                   max_timesteps_per_episode=75, max_evaluation_steps=75)}
    _, _, want = _run(cfg, "device_philox", "device")
    monkeypatch.setenv("COEVO_HOST_COHORTS", "2")

    def seed(tr):
        rc = L.load().coevo_host_rollout_debug_seed_counters(tr.eng.ro.ctx, seeded)
        assert rc in (0, -3)   # (-3: the runtime refused stream-written words - events are used, nothing to seed)

    tr, res = _host_trainer_run(cfg, before_step=seed)
    for g in range(2):
        assert res.elite_ids[g] == want.elite_ids[g]
        assert np.array_equal(np.asarray(res.game_rewards[g]).view(np.uint64), np.asarray(want.game_rewards[g]).view(np.uint64))
    tr.eng.ro.close()


@pytest.mark.parametrize("zero_copy", ["0", "1"])
def test_host_cores_rollout_refuses_pageable_staging_buffers(monkeypatch, zero_copy):
    """obs_host / actions_host are documented as page-locked; a pageable buffer would make the staged copy block
    the calling core behind a stream wait only that core can release.  Both copy modes validate them: COEVO_ERR_ARG"""
    from coevonet_amd import lib as L
    monkeypatch.setenv("COEVO_HOST_ZERO_COPY", zero_copy)
    cfg = {"seed": 7, "args": dict(generations=1, population=4, hof_size=1, elites_number=1, fitness_sharing=True,
                                   max_timesteps_per_episode=6, max_evaluation_steps=6)}
    tr, _ = _host_trainer_run(cfg)
    ro = tr.eng.ro
    good = ro.obs_host
    ro.obs_host = torch.zeros_like(good)           # pageable
    ro.reset_from_ordinals(np.arange(ro.plan.n_games))
    with pytest.raises(L.CoevoError):
        ro.run(1)
    ro.obs_host = good
    ro.reset_from_ordinals(np.arange(ro.plan.n_games))
    ro.run(1)                                       # ... and the context still works afterwards
    ro.check_status()
    ro.close()
