"""The CPU oracle (oracle/) against the golden fixtures minted from the reference itself."""
import numpy as np
import pytest
import torch

from oracle import ref_port as rp
from tests.util import Bag, SAFE_MARGIN, load_golden, sha


def test_fc_forward_vectors():
    for case in load_golden("fc_forward.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        D = case["D"]
        w = rp.init_net(D)
        if case["mutated"]:
            w = rp.mutate_torch(w, D, 0.05)
        assert sha(w) == case["weights"]["sha256"]
        obs = np.array(case["obs"], dtype=np.float32)
        for r in range(obs.shape[0]):
            a, logits, st = rp.fc_forward(w, D, obs[r])
            ref = np.array(case["logits"][r], dtype=np.float32)
            assert st == 0
            # tolerance: fp32 summation-order noise of a 512-term dot product
            np.testing.assert_allclose(logits, ref, rtol=2e-5, atol=2e-6)
            srt = np.sort(ref)[::-1]
            if srt[0] - srt[1] > SAFE_MARGIN:
                assert a == case["actions"][r]


def test_play_game_vectors():
    for case in load_golden("play_game.json")["cases"]:
        rp.seed = None
        torch.manual_seed(case["torch_seed"])
        np.random.seed(case["torch_seed"])
        a0, a1, adv = rp.init_net(10), rp.init_net(10), rp.init_net(8)
        assert [sha(a0), sha(a1), sha(adv)] == [w["sha256"] for w in case["weights"]]
        stream = rp.Stream()
        for g in case["games"]:
            got = rp.play_game(stream, a0, a1, adv, case["limit"], case["max_cycles"])
            assert got["steps"] == g["steps"]
            if g["min_margin"] > SAFE_MARGIN:
                assert got["actions"] == g["actions"]
                assert got["rewards"] == g["rewards"]  # fp64, bit for bit
            assert abs(got["min_margin"] - g["min_margin"]) < 1e-5


def _check_games(got_games, ref_games):
    n_safe = 0
    for got, ref in zip(got_games, ref_games):
        assert got["steps"] == ref["steps"]
        if ref["min_margin"] > SAFE_MARGIN:
            assert got["actions"] == ref["actions"]
            assert got["rewards"] == ref["rewards"]
            n_safe += 1
    return n_safe


@pytest.mark.parametrize("name", ["ga_cfg1.json", "ga_hof2.json"])
def test_ga_generations(name):
    fx = load_golden(name)
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    out = rp.ga_train(args, max_cycles=cfg.get("max_cycles", 25))
    role_files = {"agent_0": ("hall_of_fame_agent_0.pth", "elite_weights_agent_0.pth"),
                  "agent_1": ("hall_of_fame_agent_1.pth", "elite_weights_agent_1.pth"),
                  "adversary_0": ("hall_of_fame_adversary.pth", "elite_weights_adversary.pth")}
    for g, (got, ref) in enumerate(zip(out, fx["generations"])):
        n_safe = _check_games(got["games"], ref["games"])
        assert n_safe >= 0.9 * len(ref["games"])
        assert got["elite_ids"] == ref["elite_ids"], f"gen {g}"
        np.testing.assert_allclose(got["diversity"], ref["diversity"], rtol=1e-5)
        for ph in range(3):
            np.testing.assert_allclose(got["fitness"][ph], ref["fitness"][ph], rtol=1e-5, atol=1e-7)
        saves = {s["file"]: s["agents"] for s in ref["saves"]}
        for role, (hf, ef) in role_files.items():
            assert [sha(w) for w in got["hof"][role]] == [a["sha256"] for a in saves[hf]]
            assert [sha(w) for w in got["elites"][role]] == [a["sha256"] for a in saves[ef]]
        np.testing.assert_allclose(got["eval_rewards"], ref["eval_rewards"], rtol=1e-12)
        assert got["sigma_after"] == ref["sigma_after"]


@pytest.mark.parametrize("name", ["es_small.json", "es_fs.json"])
def test_es_generations(name):
    fx = load_golden(name)
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    out = rp.es_train(args, max_cycles=cfg.get("max_cycles", 25))
    files = {"agent_0": "agent_0.pth", "agent_1": "agent_1.pth", "adversary_0": "adversary.pth"}
    for g, (got, ref) in enumerate(zip(out, fx["generations"])):
        if g == 0:
            # generation 0 games depend only on seeded init + numpy noise: exact where margin-safe
            _check_games(got["games"][:3 * args.population], ref["games"][:3 * args.population])
        saves = {s["file"]: s for s in ref["saves"]}
        for role, f in files.items():
            w = got["base"][role]
            # the ES update is an fp32 GEMV whose order differs (numpy BLAS vs numpy BLAS is the same
            # here, so this is usually exact; tolerance stated for other hosts)
            np.testing.assert_allclose(np.sum(w.astype(np.float64)), saves[f]["agent"]["sum"], rtol=1e-6)
            np.testing.assert_allclose(rp.perturbable(w, rp.ROLE_D[role])[:6], saves[f]["perturbable"],
                                       rtol=1e-5, atol=1e-7)
        assert got["sigma_after"] == ref["sigma_after"]
        if ref["diversity"][0] is not None:
            np.testing.assert_allclose(got["diversity"], ref["diversity"], rtol=1e-5)


def _safe_rewards(got_games, ref_games):
    """compact fixtures (no action lists): fp64 reward triples bit for bit on margin-safe games"""
    n_safe = 0
    for got, ref in zip(got_games, ref_games):
        assert got["steps"] == ref["steps"]
        if ref["min_margin"] > SAFE_MARGIN:
            assert got["rewards"] == ref["rewards"]
            n_safe += 1
    return n_safe


def test_ga_long_horizon():
    """26 generations at tiny size: the gen > 10 branches of the adaptive mutation power, quirk Q5 included, elite ids
    and the final HoF / elite weights (genetic_algorithm.py:323-345)"""
    fx = load_golden("ga_long.json")
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    out = rp.ga_train(args)
    ups = 0
    prev = None
    for g, (got, ref) in enumerate(zip(out, fx["generations"])):
        assert _safe_rewards(got["games"], ref["games"]) >= 0.9 * len(ref["games"])
        assert got["elite_ids"] == ref["elite_ids"], f"gen {g}"
        np.testing.assert_allclose(got["eval_rewards"], ref["eval_rewards"], rtol=1e-12)
        assert got["sigma_after"] == ref["sigma_after"], f"gen {g}"
        if prev is not None:
            ups += sum(a > b for a, b in zip(ref["sigma_after"], prev))
        prev = ref["sigma_after"]
    assert len(out) == 26 and ups >= 10          # the fixture does take the increase branch
    saves = {s["file"]: s["agents"] for s in fx["generations"][-1]["saves"]}
    assert [sha(w) for w in out[-1]["hof"]["agent_0"]] == [a["sha256"] for a in saves["hall_of_fame_agent_0.pth"]]
    assert [sha(w) for w in out[-1]["elites"]["adversary_0"]] == [a["sha256"] for a in saves["elite_weights_adversary.pth"]]


@pytest.mark.parametrize("name", ["es_long.json", "es_stop.json"])
def test_es_long_horizon_and_early_stopping(name):
    """every generation's games against the reference fixture (exact on margin-safe games: the fp32 GEMV order of the
    update moves weights by ~1 ulp, actions only at near-ties), sigma past generation 10, early stopping
    (evolutionary_strategy.py:292-354)"""
    fx = load_golden(name)
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    np.random.seed(cfg["seed"])
    args = Bag(algorithm="ES", **cfg["args"])
    out = rp.es_train(args)
    assert len(out) == len(fx["generations"])
    assert (fx["stopped_at"] is not None) == bool(out[-1].get("stopped"))
    if name == "es_stop.json":
        assert fx["stopped_at"] == len(out) - 1 < cfg["args"]["generations"] - 1
    exact = True  # until an evaluation game at a near-tie could have gone the other way on this host's BLAS
    for g, (got, ref) in enumerate(zip(out, fx["generations"])):
        assert _safe_rewards(got["games"], ref["games"]) >= 0.8 * len(ref["games"]), f"gen {g}"
        exact = exact and all(x["min_margin"] > SAFE_MARGIN for x in ref["games"][-10:])
        if exact:
            np.testing.assert_allclose(got["eval_rewards"], ref["eval_rewards"], rtol=1e-12)
            assert got["sigma_after"] == ref["sigma_after"], f"gen {g}"
    assert exact or name == "es_long.json"


def test_deepqn_forward_vectors():
    """oracle DeepQN.forward vs the logits the reference's DeepQN produced (train-mode BatchNorm at batch 1 =
    per-sample spatial statistics).  Tolerance = fp32 summation-order noise of 3136-term dot products."""
    import hashlib
    for case in load_golden("deepqn_forward.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        C, n = case["C"], case["n_actions"]
        flat, shapes = rp.dqn_init(C, n)
        flat = rp.dqn_mutate_torch(flat, shapes, case["mutate_std"])
        assert len(flat) == rp.lib().oracle_dqn_param_count(C, n)
        assert sha(flat) == case["weights_sha256"]  # same parameters as the reference's DeepQN, byte for byte
        g = np.random.Generator(np.random.PCG64(case["frame_pcg_seed"]))
        frames = g.integers(0, 256, size=(2, 84, 84, C), dtype=np.uint8)
        assert hashlib.sha256(frames.tobytes()).hexdigest() == case["frame_sha256"]
        for r in range(2):
            a, logits = rp.dqn_forward(flat, C, n, frames[r])
            ref = np.array(case["logits"][r], dtype=np.float32)
            np.testing.assert_allclose(logits, ref, rtol=1e-4, atol=2e-5)
            srt = np.sort(ref)[::-1]
            if srt[0] - srt[1] > 1e-3:
                assert a == int(np.argmax(ref))
