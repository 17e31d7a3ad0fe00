#!/usr/bin/env python3
"""Soak of the host-cores rollouts: many generations, contexts created and destroyed repeatedly, every (cohorts, threads,
copy mode, signal) form; each run's final evaluation history is compared with the first form's (any race in the thread pool,
the completion words or the buffer hand-over would show as a different number).  python tools/host_soak.py [generations]"""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args   # noqa: E402
from coevonet_amd.game_logic import initialize_env   # noqa: E402
from coevonet_amd.genetic_algorithm import GATrainer, ROLES   # noqa: E402

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ref = None
t_all = time.time()
for rep, (K, T, Z, S) in enumerate([(4, 2, 1, "flag"), (4, 4, 1, "flag"), (3, 1, 1, "flag"), (2, 2, 0, "event"), (4, 2, 0, "flag"),
                                    (4, 3, 1, "event"), (1, 1, 1, "flag"), (4, 2, 1, "flag")]):
    os.environ.update(COEVO_HOST_COHORTS=str(K), COEVO_HOST_THREADS=str(T), COEVO_HOST_ZERO_COPY=str(Z), COEVO_HOST_SIGNAL=S)
    torch.manual_seed(0)
    np.random.seed(0)
    args = make_args(60, 3, 2, 200)
    args.generations = gens
    env = initialize_env(args)
    tr = GATrainer(env, args, rng="device_philox", env_mode="host", collect=False)
    t0 = time.time()
    for _ in range(gens):
        tr.step()
    res = tr.finish()
    hist = [tuple(res.rewards[r]) for r in ROLES]
    sig = (args.mutation_power_agent_0, args.mutation_power_agent_1, args.mutation_power_adversary)
    print(f"K={K} T={T} zero_copy={Z} signal={S}: {gens / (time.time() - t0):7.1f} generations/s, last eval "
          f"{[h[-1] for h in hist]}, sigma {sig}", flush=True)
    if ref is None:
        ref = (hist, sig)
    assert (hist, sig) == ref, "a different number: race"
    tr.eng.ro.close()
    del tr, env
    gc.collect()
print(f"soak ok: {8 * gens} generations in {time.time() - t_all:.0f} s, all forms identical")
