"""Whole Co-ES generations on the MI355X against the reference fixtures (host_reference RNG) and against the oracle
port run with the same counter-based noise (device_philox)."""
import numpy as np
import pytest
import torch

from coevonet_amd import evolutionary_strategy as es
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import ROLES
from oracle import ref_port as rp
from tests.util import SAFE_MARGIN, Bag, load_golden, sha

pytestmark = pytest.mark.gpu
FILES = {"agent_0": "agent_0.pth", "agent_1": "agent_1.pth", "adversary_0": "adversary.pth"}


def _run(cfg, rng, env_mode="device"):
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    env = initialize_env(args)
    env.max_cycles = cfg.get("max_cycles", 25)
    agents, res = es.evolution_strategy_train(env, args, None, rng=rng, env_mode=env_mode, return_result=True)
    return args, env, agents, res


@pytest.mark.parametrize("name", ["es_small.json", "es_fs.json"])
@pytest.mark.parametrize("env_mode", ["device", "host"])
def test_es_matches_reference_fixture(name, env_mode):
    """host_reference RNG.  Generation 0 depends only on the seeded init and the numpy noise stream: per-game rewards
    equal the reference's bit for bit.  The ES update is an fp32 BLAS GEMV (evolutionary_strategy.py:144) whose
    summation order is the host BLAS's business, so from the first update on the comparison with the FIXTURE (minted on
    another CPU) is a tolerance on the weights; the comparison with the ORACLE PORT run here on the same host (same
    numpy call) stays exact: every game, every evaluation reward, the trained weights."""
    fx = load_golden(name)
    cfg = fx["config"]
    args, env, agents, res = _run(cfg, "host_reference", env_mode)
    pop = args.population
    got = res.game_rewards[0]
    n_safe = 0
    for i, rg in enumerate(fx["generations"][0]["games"][:3 * pop]):
        if rg["min_margin"] > SAFE_MARGIN:
            assert list(got[i]) == rg["rewards"], i
            n_safe += 1
    assert n_safe >= 0.8 * 3 * pop
    for g, ref in enumerate(fx["generations"]):
        assert res.sigma_after[g] == ref["sigma_after"]
        if ref["diversity"][0] is not None and g == 0:
            np.testing.assert_allclose(res.diversity[g], ref["diversity"], rtol=1e-5, atol=2e-7)
    first = {s["file"]: s for s in fx["generations"][0]["saves"]}
    assert env.n_resets == fx["env_resets"]
    # the oracle port on this host
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    want = rp.es_train(Bag(algorithm="ES", **cfg["args"]), max_cycles=cfg.get("max_cycles", 25))
    for g, w in enumerate(want):
        for i in range(3 * pop):
            assert list(res.game_rewards[g][i]) == w["games"][i]["rewards"], (g, i)
        assert [res.rewards[r][g] for r in ROLES] == w["eval_rewards"]
    for a, r in zip(agents, ROLES):
        assert sha(a.model.flat()) == sha(want[-1]["base"][r])
    # and the fixture, within the fp32 GEMV's rounding, after the first update
    for r in ROLES:
        np.testing.assert_allclose(rp.perturbable(want[0]["base"][r], rp.ROLE_D[r])[:6], first[FILES[r]]["perturbable"],
                                   rtol=1e-5, atol=1e-7)


def test_es_device_philox_matches_oracle_port():
    cfg = {"seed": 9, "args": dict(generations=3, population=7, hof_size=1, learning_rate=0.1, fitness_sharing=True,
                                   max_timesteps_per_episode=40, max_evaluation_steps=60)}
    args, env, agents, res = _run(cfg, "device_philox")
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    want = rp.es_train(Bag(algorithm="ES", **cfg["args"]), noise="philox", philox_seed=0)
    pop = args.population
    for g, w in enumerate(want):
        got = res.game_rewards[g]
        for i in range(3 * pop):
            assert list(got[i]) == w["games"][i]["rewards"], (g, i)
        assert [res.rewards[r][g] for r in ROLES] == w["eval_rewards"]
        np.testing.assert_allclose(res.diversity[g], w["diversity"], rtol=1e-5, atol=1e-6)  # n terms of fp32 eps
    for a, r in zip(agents, ROLES):
        # the sharing score enters the update through an fp32 division: distances are fp64-summed here, BLAS in numpy
        np.testing.assert_allclose(a.model.flat(), want[-1]["base"][r], rtol=1e-4, atol=5e-6)


def test_es_device_philox_without_sharing_is_bit_exact():
    cfg = {"seed": 2, "args": dict(generations=2, population=5, hof_size=1, learning_rate=0.1,
                                   max_timesteps_per_episode=30, max_evaluation_steps=30)}
    args, env, agents, res = _run(cfg, "device_philox")
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    want = rp.es_train(Bag(algorithm="ES", **cfg["args"]), noise="philox", philox_seed=0)
    for a, r in zip(agents, ROLES):
        assert sha(a.model.flat()) == sha(want[-1]["base"][r])
    assert [res.rewards[r][-1] for r in ROLES] == want[-1]["eval_rewards"]


def test_es_extension_mode_matches_oracle_port():
    """BASELINE configs[2] wording (antithetic pairs + centered ranks, on-device rank): NOT the reference's algorithm,
    a labelled extension - bit for bit the oracle port's statement of it"""
    cfg = {"seed": 6, "args": dict(generations=3, population=8, hof_size=1, learning_rate=0.1,
                                   max_timesteps_per_episode=30, max_evaluation_steps=30, coevo_antithetic=True,
                                   coevo_centered_rank=True)}
    args, env, agents, res = _run(cfg, "device_philox")
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    oa = {k: v for k, v in cfg["args"].items() if not k.startswith("coevo_")}
    want = rp.es_train(Bag(algorithm="ES", **oa), noise="philox", philox_seed=0, antithetic=True, centered_rank=True)
    for g, w in enumerate(want):
        for i in range(3 * args.population):
            assert list(res.game_rewards[g][i]) == w["games"][i]["rewards"], (g, i)
        assert [res.rewards[r][g] for r in ROLES] == w["eval_rewards"]
    for a, r in zip(agents, ROLES):
        assert sha(a.model.flat()) == sha(want[-1]["base"][r])
    # and it is a different algorithm from reference_exact
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    plain = rp.es_train(Bag(algorithm="ES", **oa), noise="philox", philox_seed=0)
    assert sha(plain[-1]["base"]["agent_0"]) != sha(want[-1]["base"]["agent_0"])


def test_reference_step_functions_reproduce_generation0():
    """the reference's per-call helpers under their own names (mutate_weights, evaluate_current_weights:
    evolutionary_strategy.py:22-116) driven in the reference's loop order (:222-265) reproduce generation 0 of the fixture
    the reference's own loop minted: every margin-safe game's reward in the role's slot"""
    fx = load_golden("es_small.json")
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    env = initialize_env(args)
    env.max_cycles = cfg.get("max_cycles", 25)
    from coevonet_amd.game_logic import create_agent
    from coevonet_amd.genetic_algorithm import RET_SLOT
    a0, a1, adv = (create_agent(env, args, role=r) for r in ROLES)
    logs = ([], [], [])
    games = fx["generations"][0]["games"]
    n_safe = 0
    for j in range(args.population):
        for ri, r in enumerate(ROLES):
            rew, noise, w = es.mutate_weights(env, a0, a1, adv, args, r, j, *logs)
            ref = games[3 * j + ri]
            assert noise.dtype == np.float32 and w.ndim == 1 and len(noise) == len(w)   # (both: the three Linear layers)
            if ref["min_margin"] > SAFE_MARGIN:
                assert rew == ref["rewards"][RET_SLOT[r]], (j, r)
                n_safe += 1
    assert n_safe >= 0.8 * 3 * args.population and len(logs[0]) == args.population
    ev = es.evaluate_current_weights(a0, a1, adv, env, args)     # (the un-updated trio: ten more games of the reset stream)
    assert len(ev) == 3 and all(np.isfinite(ev))
