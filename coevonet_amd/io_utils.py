"""Checkpoint export and metrics.

* ``save_model(obj, path)`` = utils/utils_pth_and_plots.py:78-80 (``torch.save`` of python objects).  The trainers call it
  with lists of ``Agent`` objects under the reference's file names (genetic_algorithm.py:53-61, 293-299;
  evolutionary_strategy.py:154-156, 357-360), so ``load_agent_for_testing`` (utils/utils_pth_and_plots.py:8-74) +
  ``main.py --test`` keep working on what this package writes (the pickles reference ``coevonet_amd.agent.MPEAgent``;
  pass ``weights_only=False`` to ``torch.load`` on torch >= 2.6, which the reference does not).
* ``save_state_dicts`` writes, beside every pickle, ``<name>.state_dict.pth``: plain ``OrderedDict``s of tensors (one
  ``state_dict()`` per agent, keys = the reference's parameter names) that ``torch.load(..., weights_only=True)`` accepts
  - no code object is unpickled; ``agents_from_state_dicts`` rebuilds the ``Agent`` objects from them.
* ``MetricsWriter`` replaces the three matplotlib savefigs per generation (utils/utils_pth_and_plots.py:153-262, 2.6 s per
  generation in the reference, and a crash without --adaptive at :235) with one JSON line per generation.
"""
from __future__ import annotations

import json
import os

import torch

GA_FILES = {"agent_0": ("hall_of_fame_agent_0.pth", "elite_weights_agent_0.pth"),
            "agent_1": ("hall_of_fame_agent_1.pth", "elite_weights_agent_1.pth"),
            "adversary_0": ("hall_of_fame_adversary.pth", "elite_weights_adversary.pth")}
ES_FILES = {"agent_0": "agent_0.pth", "agent_1": "agent_1.pth", "adversary_0": "adversary.pth"}


def save_model(obj, file_path):
    torch.save(obj, file_path)


def load_agents(file_path):
    return torch.load(file_path, weights_only=False)


_TEST_ARGS = {   # the attributes main.py --test names the checkpoints with (main.py:60-75), per algorithm and role
    "GA": (("GA_hof_to_test_agent_0", "agent_0"), ("GA_hof_to_test_agent_1", "agent_1"),
           ("GA_hof_to_test_adversary", "adversary_0")),
    "ES": (("ES_model_to_test_agent_0", "agent_0"), ("ES_model_to_test_agent_1", "agent_1"),
           ("ES_model_to_test_adversary_0", "adversary_0")),
}


def load_agent_for_testing(args, env=None):
    """utils/utils_pth_and_plots.py:8-74: the three agents `main.py --test` plays (main.py:204-243).  GA checkpoints are
    Hall-of-Fame lists (the newest member is taken), ES checkpoints single agents.  Same ValueErrors: a path that is not
    given, then a path that does not exist, checked for all three roles before anything is loaded."""
    if args.algorithm not in _TEST_ARGS:
        return None
    spec = _TEST_ARGS[args.algorithm]
    for attr, role in spec:
        if getattr(args, attr, None) is None:
            raise ValueError(f"Error: Model file for {role} not specified. Please specify the agent to test")
    for attr, _ in spec:
        if not os.path.exists(getattr(args, attr)):
            raise ValueError(f"Error: Model file {getattr(args, attr)} not found.")
    loaded = [load_agents(getattr(args, attr)) for attr, _ in spec]
    return tuple(x[-1] for x in loaded) if args.algorithm == "GA" else tuple(loaded)


def create_output_dir(args):
    """utils/utils_pth_and_plots.py:83-96: the run's directory name from its hyper-parameters"""
    name = (f"{args.algorithm}_models/gens{args.generations}_pop{args.population}_hof{args.hof_size}_game{args.game}"
            f"_tslimit{args.max_timesteps_per_episode}_fitness-sharing{args.fitness_sharing}_adaptive{args.adaptive}")
    if args.adaptive:
        name += f"max_mutation{args.max_mutation_power}_min_mutation{args.min_mutation_power}"
    if args.algorithm == "ES":
        name += f"_lr{args.learning_rate}"
    os.makedirs(name, exist_ok=True)
    return name


STATE_DICT_SUFFIX = ".state_dict.pth"


def state_dict_path(file_path):
    root = file_path[:-4] if file_path.endswith(".pth") else file_path
    return root + STATE_DICT_SUFFIX


def save_state_dicts(agents, file_path, role=None):
    """the `weights_only`-safe variant of save_model: {"format", "role", "agents": [state_dict, ...]}
    with tensors only.  `agents`: an Agent or a list of Agents.  Returns the path written."""
    single = not isinstance(agents, (list, tuple))
    lst = [agents] if single else list(agents)
    from . import lib as L
    payload = {"format": "coevonet_amd.state_dict.v1", "role": role or "", "single": single,
               # what a resumed device_philox run must match: other rounds are other offspring for the same seed
               "coevo_version": int(L.load().coevo_version()), "noise": f"philox4x32-{int(L.load().coevo_noise_rounds())}",
               "agents": [{k: v.detach().clone() for k, v in a.model.state_dict().items()} for a in lst]}
    path = state_dict_path(file_path)
    torch.save(payload, path)
    return path


def load_state_dicts(path, strict_noise=False):
    """-> (list of state_dicts, role, was_single_agent); never unpickles code.

    The file records the offspring-noise contract it was written under ("noise": "philox4x32-<rounds>"): a device_philox
    run resumed under another contract breeds OTHER offspring from the same seed.  A mismatch warns (the weights themselves
    are valid under any contract); strict_noise=True refuses."""
    payload = torch.load(path, weights_only=True)
    if payload.get("format") != "coevonet_amd.state_dict.v1":
        raise ValueError(f"{path} is not a coevonet_amd state_dict checkpoint")
    saved = payload.get("noise")
    if saved is not None:
        from . import lib as L
        now = f"philox4x32-{int(L.load().coevo_noise_rounds())}"
        if saved != now:
            msg = (f"{path} was written under offspring noise {saved!r} (coevo_version {payload.get('coevo_version')}), this "
                   f"library breeds with {now!r}: a resumed device_philox run will not continue the saved run's offspring")
            if strict_noise:
                raise ValueError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return payload["agents"], payload["role"], bool(payload["single"])


def agents_from_state_dicts(env, args, role, path):
    """Agent objects (what load_agent_for_testing returns, utils/utils_pth_and_plots.py:8-74) from the safe file"""
    from .game_logic import create_agent
    sds, saved_role, single = load_state_dicts(path)
    state = torch.random.get_rng_state()
    out = []
    for sd in sds:
        a = create_agent(env, args, role or saved_role or None)
        a.model.load_state_dict(sd)
        out.append(a)
    torch.random.set_rng_state(state)
    return out[0] if single else out


def agents_from_flat(env, args, role, flats):
    """[n][P] flat parameter rows -> list of MPEAgent (the objects the reference pickles)"""
    from .game_logic import create_agent
    state = torch.random.get_rng_state()  # constructing agents draws from the generator: leave the stream untouched
    out = []
    for w in flats:
        a = create_agent(env, args, role)
        a.model.set_flat(w)
        out.append(a)
    torch.random.set_rng_state(state)
    return out


class MetricsWriter:
    def __init__(self, output_dir, name="metrics.jsonl"):
        self.path = None
        if output_dir:
            os.makedirs(output_dir, exist_ok=True)
            self.path = os.path.join(output_dir, name)
            open(self.path, "w").close()

    def write(self, **rec):
        if self.path:
            with open(self.path, "a") as f:
                f.write(json.dumps(rec) + "\n")
