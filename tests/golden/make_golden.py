#!/usr/bin/env python3
"""Mint golden fixtures by running the REFERENCE's own Python (imported from /root/reference,
never copied) against this repo's AEC env.  Run in the build container only:

    python tests/golden/make_golden.py            # rewrites tests/golden/*.json

What is recorded (data only - inputs and expected outputs):
  fc_forward.json   FCNetwork.forward / determine_action on seeded nets + observations
  deepqn_forward.json  DeepQN.forward on seeded nets + synthetic frames (logits only)
  deepqn_weights.json  DeepQN's weight accessors (flat orders, perturbable subset, partial sets, state_dict keys)
  play_game.json    play_game() reward triples, action sequences, min top-2 logit margins
  ga_*.json         genetic_algorithm_train: per-game rewards, diversity, fitness, elite ids,
                    HoF / elite weight checksums, eval rewards, adaptive sigma trajectory
  es_*.json         evolution_strategy_train: per-game rewards, base-weight checksums, sigma

The reference imports ``supersuit`` unconditionally (utils/game_logic_functions.py:8) but only
uses it for Atari env construction, which is never reached here: an inert module object with the
four names is registered so the import statement succeeds.  ``pettingzoo`` is only imported inside
``initialize_env`` (:45), which is bypassed by handing the loop this repo's env object.
"""
import hashlib
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

_ss = types.ModuleType("supersuit")
for _n in ("frame_stack_v1", "resize_v1", "frame_skip_v0", "agent_indicator_v0"):
    setattr(_ss, _n, lambda env, *a, **k: env)
sys.modules["supersuit"] = _ss

import genetic_algorithm as ref_ga  # noqa: E402
import evolutionary_strategy as ref_es  # noqa: E402
import utils.game_logic_functions as ref_glf  # noqa: E402
from MPE.fcnetwork import FCNetwork  # noqa: E402
from Atari.deepqn import DeepQN  # noqa: E402

from coevonet_amd.mpe.simple_adversary import SimpleAdversaryAEC, ENV_SEED  # noqa: E402

PARAM_ORDER = ["fc1.weight", "fc1.bias", "ln1.weight", "ln1.bias", "fc2.weight", "fc2.bias",
               "ln2.weight", "ln2.bias", "output.weight", "output.bias"]


class Bag:
    """Duck-typed args, the attribute set of main.py:95-142."""

    def __init__(self, **kw):
        d = dict(algorithm="GA", generations=2, population=16, hof_size=1, game="simple_adversary_v3",
                 mutation_power_agent_0=0.05, mutation_power_agent_1=0.05, mutation_power_adversary=0.05,
                 learning_rate=0.1, max_timesteps_per_episode=None, max_evaluation_steps=None,
                 elites_number=2, adaptive=True, max_mutation_power=0.2, min_mutation_power=0.001,
                 fitness_sharing=False, early_stopping=False, patience=300, min_delta=0.1, debug=False,
                 train=True, test=False, render=False, env_mode="AEC", precision="float32", save=True,
                 average_window=50, play_against_yourself=False)
        d.update(kw)
        self.__dict__.update(d)


def flat_all_params(model):
    sd = model.state_dict()
    return np.concatenate([sd[k].detach().cpu().numpy().ravel() for k in PARAM_ORDER]).astype(np.float32)


def wsum(model):
    w = flat_all_params(model)
    return {"sha256": hashlib.sha256(w.tobytes()).hexdigest(), "n": int(w.size),
            "sum": float(np.sum(w.astype(np.float64))), "head": [float(x) for x in w[:4]]}


# ------------------------------------------------------------------ instrumentation ----
class GameLog:
    def __init__(self):
        self.games = []
        self.cur = None

    def begin(self):
        self.cur = {"actions": [], "min_margin": float("inf")}

    def end(self, ret):
        self.cur["rewards"] = [float(x) for x in ret]
        self.cur["steps"] = len(self.cur["actions"])
        self.games.append(self.cur)
        self.cur = None


LOG = GameLog()
_orig_forward = FCNetwork.forward


def _logged_forward(self, x, args):
    out = _orig_forward(self, x, args)
    if LOG.cur is not None:
        o = out.detach().to(torch.float64).numpy()
        srt = np.sort(o)[::-1]
        LOG.cur["min_margin"] = min(LOG.cur["min_margin"], float(srt[0] - srt[1]))
        LOG.cur["actions"].append(int(np.argmax(o)))
    return out


FCNetwork.forward = _logged_forward
_orig_play_game = ref_glf.play_game


def _logged_play_game(*a, **k):
    LOG.begin()
    ret = _orig_play_game(*a, **k)
    LOG.end(ret)
    return ret


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)


def make_env(max_cycles=25):
    env = SimpleAdversaryAEC(max_cycles=max_cycles)
    env.reset(seed=ENV_SEED)  # what initialize_env does (utils/game_logic_functions.py:54)
    return env


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", name, os.path.getsize(path), "bytes")


# ------------------------------------------------------------------ K1 forward vectors ----
def mint_fc_forward():
    args = Bag()
    cases = []
    for seed, D in [(0, 10), (1, 8), (2, 10), (3, 8)]:
        torch.manual_seed(seed)
        net = FCNetwork(D, 5, "float32")
        if seed >= 2:  # GA-style mutation touches the LayerNorm affine params too (agent.py:27)
            for p in net.parameters():
                p.data += torch.normal(0, 0.05, size=p.size())
        g = np.random.Generator(np.random.PCG64(100 + seed))
        obs = g.uniform(-2, 2, size=(6, D)).astype(np.float32)
        logits, actions = [], []
        for r in range(obs.shape[0]):
            x = torch.from_numpy(obs[r])
            out = _orig_forward(net, x, args)
            logits.append([float(v) for v in out.detach().numpy()])
            actions.append(int(net.determine_action(x, args)))
        cases.append({"torch_seed": seed, "mutated": seed >= 2, "D": D, "obs": obs.tolist(),
                      "logits": logits, "actions": actions, "weights": wsum(net)})
    dump("fc_forward.json", {"cases": cases})


def mint_deepqn_forward():
    """DeepQN.forward (Atari/deepqn.py:39-48) of six mutated nets covering C = 3, 4, 5, 6 frame planes and n = 6 / 18 actions on
    eight frames each: five random, all-0, all-255, constant planes (tests/util.py:dqn_golden_frames)"""
    from tests.util import DQN_FRAME_KINDS, dqn_golden_frames
    cases = []
    for seed, C, n in [(0, 4, 6), (1, 6, 18), (2, 3, 6), (3, 5, 18), (4, 4, 18), (5, 6, 6)]:
        torch.manual_seed(seed)
        net = DeepQN(C, n, "float32")
        # make the BatchNorm affine non-trivial, as a GA mutation would
        for p in net.parameters():
            p.data += torch.normal(0, 0.02, size=p.size())
        frames = dqn_golden_frames(C, 200 + seed)
        logits = []
        for r in range(frames.shape[0]):
            x = torch.from_numpy(frames[r]).to(torch.float32).permute(2, 0, 1).unsqueeze(0)
            out = net.forward(x)
            logits.append([float(v) for v in out.detach().numpy()[0]])
        sd = net.state_dict()
        w = np.concatenate([sd[k].detach().numpy().ravel() for k in sd
                            if not k.endswith("num_batches_tracked")
                            and "running" not in k]).astype(np.float32)
        cases.append({"torch_seed": seed, "C": C, "n_actions": n, "frame_pcg_seed": 200 + seed,
                      "frame_kinds": DQN_FRAME_KINDS, "frame_sha256": hashlib.sha256(frames.tobytes()).hexdigest(),
                      "logits": logits, "mutate_std": 0.02,
                      "weights_sha256": hashlib.sha256(w.tobytes()).hexdigest()})
    dump("deepqn_forward.json", {"cases": cases})


def mint_deepqn_weights():
    """the weight accessors of Atari/deepqn.py:63-231 (what the ES branch and --save go through): flat orders, the
    perturbable subset (no BatchNorm), partial sets, the state_dict key set"""
    def h(a):
        return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()
    cases = []
    for seed, C, n in [(0, 4, 6), (1, 6, 18)]:
        torch.manual_seed(seed)
        net = DeepQN(C, n, "float32")
        for p in net.parameters():
            p.data += torch.normal(0, 0.02, size=p.size())
        a = Bag(precision="float32")
        case = {"torch_seed": seed, "C": C, "n_actions": n, "mutate_std": 0.02,
                "state_dict_keys": list(net.state_dict().keys()),
                "perturbable_layers": [nm for nm, m in net.named_modules() if m in net.get_perturbable_layers()],
                "all_len": int(net.get_weights_ES().size), "all_sha256": h(net.get_weights_ES()),
                "perturbable_len": int(net.get_perturbable_weights().size),
                "perturbable_sha256": h(net.get_perturbable_weights()),
                "fc1_vbn2_sha256": h(net.get_weights_ES([net.fc1, net.vbn2])),
                "get_weights_fc1_vbn1_keys": list(net.get_weights(["fc1", "vbn1"]).keys())}
        v = (net.get_perturbable_weights() * np.float32(0.5) + np.float32(0.01)).astype(np.float32)
        net.set_perturbable_weights(v, a)
        case["after_set_perturbable_sha256"] = h(net.get_weights_ES())
        k = int(net.fc1.weight.numel() + net.fc1.bias.numel() + 2 * 64)
        u = (np.arange(k, dtype=np.float32) % np.float32(97.0)) * np.float32(1e-3)
        net.set_weights_ES(u, a, [net.fc1, net.vbn2])
        case["after_set_fc1_vbn2_sha256"] = h(net.get_weights_ES())
        new = {key: val * 2 for key, val in net.get_weights(["output"]).items()}
        net.set_weights(new, layers=["output"])
        case["after_set_weights_output_sha256"] = h(net.get_weights_ES())
        cases.append(case)
    dump("deepqn_weights.json", {"cases": cases})


# ------------------------------------------------------------------ play_game ----
def mint_play_game():
    out = []
    for seed, limit, max_cycles in [(10, None, 25), (11, 50, 25), (12, 200, 70), (13, 7, 25)]:
        seed_all(seed)
        env = make_env(max_cycles)
        args = Bag(max_timesteps_per_episode=limit, max_evaluation_steps=limit)
        a0 = ref_glf.create_agent(env, args, "agent_0")
        a1 = ref_glf.create_agent(env, args, "agent_1")
        adv = ref_glf.create_agent(env, args, "adversary_0")
        LOG.games = []
        ref_glf.play_game = _logged_play_game
        for _ in range(3):
            _logged_play_game(env=env, player1=a0.model, player2=a1.model, adversary=adv.model,
                              args=args, eval=False)
        out.append({"torch_seed": seed, "limit": limit, "max_cycles": max_cycles,
                    "weights": [wsum(a0.model), wsum(a1.model), wsum(adv.model)],
                    "games": LOG.games})
    dump("play_game.json", {"cases": out})


# ------------------------------------------------------------------ GA ----
def run_ga(cfg):
    seed_all(cfg["seed"])
    env = make_env(cfg.get("max_cycles", 25))
    args = Bag(algorithm="GA", **cfg["args"])
    LOG.games = []
    rec = {"diversity": [], "saves": [], "plots": [], "argsort_in": [], "argsort_out": []}

    class NpProxy:
        """Forwards to numpy; logs what the trainer hands to np.argsort (genetic_algorithm.py:223-225),
        i.e. the population fitness lists exactly as the reference computed them."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def argsort(a, *x, **k):
            out = np.argsort(a, *x, **k)
            rec["argsort_in"].append([(float(v), type(v).__name__) for v in a])
            rec["argsort_out"].append([int(v) for v in out])
            return out

    def div_hook(individual_weights, population_weights, args, sigma=None):
        d = ref_glf.diversity_penalty(individual_weights, population_weights, args, sigma)
        rec["diversity"].append(float(d))
        return d

    def save_hook(obj, path):
        rec["saves"].append({"file": os.path.basename(path), "agents": [wsum(a.model) for a in obj]})

    def plot_hook(rewards, mutation_power_history, fitness, diversity, file_path, args):
        rec["plots"].append({"file": os.path.basename(file_path), "rewards": [float(r) for r in rewards],
                             "mutation_power_history": None if mutation_power_history is None
                             else [float(m) for m in mutation_power_history]})

    ref_ga.play_game = _logged_play_game
    ref_ga.diversity_penalty = div_hook
    ref_ga.save_model = save_hook
    ref_ga.plot_experiment_metrics = plot_hook
    ref_ga.np = NpProxy()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            ref_ga.genetic_algorithm_train(env, env.agents[0], args, td)
        finally:
            os.chdir(cwd)
            ref_ga.np = np
    pop, hof, gens = args.population, args.hof_size, args.generations
    per_gen_games = 3 * pop * hof + 10
    gens_out = []
    for g in range(gens):
        games = LOG.games[g * per_gen_games:(g + 1) * per_gen_games]
        div = rec["diversity"][g * 3 * pop:(g + 1) * 3 * pop]
        fit, fit_type, elites = [], [], []
        for ph in range(3):  # phases: agent_0, agent_1, adversary
            logged = rec["argsort_in"][g * 3 + ph]
            fit.append([v for v, _ in logged])
            fit_type.append(sorted(set(t for _, t in logged)))
            elites.append(rec["argsort_out"][g * 3 + ph][::-1][:args.elites_number])
        saves = rec["saves"][g * 6:(g + 1) * 6]
        plots = rec["plots"][g * 3:(g + 1) * 3]
        gens_out.append({
            "games": games, "diversity": [div[0], div[pop], div[2 * pop]],
            "fitness": fit, "fitness_scalar_type": fit_type, "elite_ids": elites, "saves": saves,
            "eval_rewards": [plots[0]["rewards"][-1], plots[1]["rewards"][-1], plots[2]["rewards"][-1]],
            "sigma_after": [plots[0]["mutation_power_history"][-1], plots[1]["mutation_power_history"][-1],
                            plots[2]["mutation_power_history"][-1]],
        })
    if cfg.get("compact"):  # long runs: rewards + margins only, weight checksums of the last generation only
        for g, rec_g in enumerate(gens_out):
            rec_g["games"] = [{"rewards": x["rewards"], "min_margin": x["min_margin"], "steps": x["steps"]}
                              for x in rec_g["games"]]
            if g < gens - 1:
                rec_g["saves"] = []
    return {"config": cfg, "env_resets": env.n_resets, "generations": gens_out,
            "final_args": {"mutation_power_agent_0": args.mutation_power_agent_0,
                           "mutation_power_agent_1": args.mutation_power_agent_1,
                           "mutation_power_adversary": args.mutation_power_adversary}}


# ------------------------------------------------------------------ ES ----
def run_es(cfg):
    seed_all(cfg["seed"])
    env = make_env(cfg.get("max_cycles", 25))
    args = Bag(algorithm="ES", **cfg["args"])
    LOG.games = []
    rec = {"saves": [], "plots": []}

    def save_hook(obj, path):
        rec["saves"].append({"file": os.path.basename(path), "agent": wsum(obj.model),
                             "perturbable": [float(x) for x in obj.model.get_perturbable_weights()[:6]]})

    def plot_hook(rewards, mutation_power_history, fitness, diversity, file_path, args):
        rec["plots"].append({"file": os.path.basename(file_path), "rewards": [float(r) for r in rewards],
                             "diversity": None if diversity is None else [float(d) for d in diversity],
                             "mutation_power_history": None if mutation_power_history is None
                             else [float(m) for m in mutation_power_history]})

    evals = []
    _orig_eval = ref_es.evaluate_current_weights

    def eval_hook(*a, **k):
        out = _orig_eval(*a, **k)
        evals.append([float(x) for x in out])
        return out

    ref_es.evaluate_current_weights = eval_hook
    ref_es.play_game = _logged_play_game
    ref_es.save_model = save_hook
    ref_es.plot_experiment_metrics = plot_hook
    ref_es.plot_weights_logging = lambda *a, **k: None
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            agents = ref_es.evolution_strategy_train(env, args, td)
        finally:
            os.chdir(cwd)
            ref_es.evaluate_current_weights = _orig_eval
    pop = args.population
    per_gen = 3 * pop + 10
    gens = len(evals)                      # generations that ran (early stopping breaks before save / plot)
    stopped_at = gens - 1 if len(rec["plots"]) // 3 < gens else None
    gens_out = []
    for g in range(gens):
        plots = rec["plots"][g * 3:(g + 1) * 3]
        games = LOG.games[g * per_gen:(g + 1) * per_gen]
        if cfg.get("compact"):
            games = [{"rewards": x["rewards"], "min_margin": x["min_margin"], "steps": x["steps"]} for x in games]
        gens_out.append({
            "games": games,
            "saves": [] if cfg.get("compact") and g < gens - 1 else rec["saves"][g * 3:(g + 1) * 3],
            "eval_rewards": evals[g],
            "diversity": [None if p["diversity"] is None else p["diversity"][-1] for p in plots] if plots else None,
            # the stopping generation is never plotted: its sigma is what the args bag holds at the end
            "sigma_after": [p["mutation_power_history"][-1] for p in plots] if plots else
            [args.mutation_power_agent_0, args.mutation_power_agent_1, args.mutation_power_adversary],
        })
    final = [flat_all_params(a.model) for a in agents]
    return {"config": cfg, "env_resets": env.n_resets, "stopped_at": stopped_at, "generations": gens_out,
            "final_weights": [{"sha256": hashlib.sha256(w.tobytes()).hexdigest(),
                               "sum": float(np.sum(w.astype(np.float64))),
                               "l2": float(np.sqrt(np.sum(w.astype(np.float64) ** 2)))} for w in final]}


GA_CONFIGS = {
    # BASELINE.json configs[0]: pop=16, HoF=1, T=50 (SURVEY 8d "config 1")
    "ga_cfg1.json": {"seed": 0, "args": dict(generations=2, population=16, hof_size=1, elites_number=2,
                                              max_timesteps_per_episode=50, max_evaluation_steps=50)},
    # HoF>1, elites 3, fitness sharing on, no step limit (75-step cap), 3 generations
    "ga_hof2.json": {"seed": 7, "args": dict(generations=3, population=6, hof_size=2, elites_number=3,
                                              fitness_sharing=True, mutation_power_agent_0=0.005)},
    # long horizon at tiny size: the gen > 10 branches of the adaptive mutation power (genetic_algorithm.py:323-345,
    # quirk Q5: agent_0's increase starts from agent_1's sigma) both fire; h[-20:-10] is partial for gens 11..18
    "ga_long.json": {"seed": 21, "compact": True,
                     "args": dict(generations=26, population=4, hof_size=1, elites_number=2,
                                  max_timesteps_per_episode=9, max_evaluation_steps=9, fitness_sharing=True,
                                  mutation_power_agent_0=0.05, mutation_power_agent_1=0.08,
                                  mutation_power_adversary=0.03, max_mutation_power=0.1, min_mutation_power=0.02)},
}
ES_CONFIGS = {
    "es_small.json": {"seed": 3, "args": dict(generations=2, population=6, hof_size=1, learning_rate=0.1,
                                               max_timesteps_per_episode=400, max_evaluation_steps=400)},
    "es_fs.json": {"seed": 4, "args": dict(generations=2, population=5, hof_size=1, learning_rate=0.1,
                                            fitness_sharing=True, max_timesteps_per_episode=30,
                                            max_evaluation_steps=45)},
    # long horizon: adaptive sigma past generation 10 (evolutionary_strategy.py:292-316)
    "es_long.json": {"seed": 22, "compact": True,
                     "args": dict(generations=26, population=4, hof_size=1, learning_rate=0.1,
                                  max_timesteps_per_episode=9, max_evaluation_steps=9,
                                  mutation_power_agent_0=0.05, mutation_power_agent_1=0.08,
                                  mutation_power_adversary=0.03, max_mutation_power=0.1, min_mutation_power=0.02)},
    # early stopping (evolutionary_strategy.py:320-354): breaks before save_model of the stopping generation
    "es_stop.json": {"seed": 23, "compact": True,
                     "args": dict(generations=12, population=4, hof_size=1, learning_rate=0.1,
                                  max_timesteps_per_episode=9, max_evaluation_steps=9, early_stopping=True,
                                  patience=3, min_delta=0.05)},
}

if __name__ == "__main__":
    which = set(sys.argv[1:])
    if not which or "fc" in which:
        mint_fc_forward()
    if not which or "dqn" in which:
        mint_deepqn_forward()
    if not which or "dqnw" in which:
        mint_deepqn_weights()
    if not which or "play" in which:
        mint_play_game()
    if not which or "ga" in which:
        for name, cfg in GA_CONFIGS.items():
            dump(name, run_ga(cfg))
    if not which or "es" in which:
        for name, cfg in ES_CONFIGS.items():
            dump(name, run_es(cfg))
