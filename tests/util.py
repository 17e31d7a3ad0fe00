import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sha(flat):
    return hashlib.sha256(np.ascontiguousarray(flat, dtype=np.float32).tobytes()).hexdigest()


class Bag:
    """Duck-typed args bag with the attribute set of the reference's main.py:95-142."""

    def __init__(self, **kw):
        d = dict(algorithm="GA", generations=2, population=16, hof_size=1, game="simple_adversary_v3",
                 mutation_power_agent_0=0.05, mutation_power_agent_1=0.05, mutation_power_adversary=0.05,
                 learning_rate=0.1, max_timesteps_per_episode=None, max_evaluation_steps=None,
                 elites_number=2, adaptive=True, max_mutation_power=0.2, min_mutation_power=0.001,
                 fitness_sharing=False, early_stopping=False, patience=300, min_delta=0.1, debug=False,
                 train=True, test=False, render=False, env_mode="AEC", precision="float32", save=False,
                 average_window=50, play_against_yourself=False)
        d.update(kw)
        self.__dict__.update(d)


# top-2 logit margin below which an action of the reference's torch-CPU forward may legitimately differ
# from the canonical-order forward (fp32 summation-order noise on |logit| ~ 1 is ~1e-6)
SAFE_MARGIN = 1e-4


# the frames of tests/golden/deepqn_forward.json: five random ones, then three structured ones on which every BatchNorm
# channel of conv1 sees ONE value (variance -> 0: the normalised activation is rounding noise times 1/sqrt(1e-5))
DQN_FRAME_KINDS = ["random"] * 5 + ["all_0", "all_255", "constant_planes"]


def dqn_golden_frames(C, pcg_seed):
    g = np.random.Generator(np.random.PCG64(pcg_seed))
    frames = g.integers(0, 256, size=(len(DQN_FRAME_KINDS), 84, 84, C), dtype=np.uint8)
    frames[5] = 0
    frames[6] = 255
    frames[7] = (37 * (np.arange(C) + 1) % 256).astype(np.uint8)[None, None, :]
    return frames
