#!/usr/bin/env python3
"""Where a generation of the host-cores env mode goes (cfg 2): wall time of rollout / evaluation read-back / selection /
breeding, each closed with a device synchronize (so the sum exceeds the pipelined generation).  python tools/host_gen_phases.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args   # noqa: E402
from coevonet_amd.game_logic import initialize_env   # noqa: E402
from coevonet_amd.genetic_algorithm import GATrainer, ROLES, SIGMA_ATTR, _finish_generation   # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
args = make_args(200, 5, 2, 200)
args.generations = 40
env = initialize_env(args)
tr = GATrainer(env, args, rng="device_philox", env_mode="host", collect=False)
eng = tr.eng
for _ in range(3):
    tr.step()
acc = {k: 0.0 for k in ("rollout", "eval_readback", "select", "breed")}
n = 20
t_all = time.perf_counter()
for _ in range(n):
    gen = tr.gen
    t0 = time.perf_counter()
    eng.rollout(gen, with_prev_eval=gen > 0)
    t1 = time.perf_counter()
    _finish_generation(args, gen - 1, eng.eval_rewards(), tr.res)
    t2 = time.perf_counter()
    eng.select()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    eng.breed_device(gen, {r: getattr(args, SIGMA_ATTR[r]) for r in ROLES})
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    tr.gen += 1
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k] += v
tot = time.perf_counter() - t_all
print({k: round(1e3 * v / n, 3) for k, v in acc.items()}, "ms per generation; total", round(1e3 * tot / n, 3))
