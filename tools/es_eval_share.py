#!/usr/bin/env python3
"""How much of a Co-ES generation is the evaluation of the updated base nets (10 sequential games, a chain of tiny
launches)?  cfg 3 (MPE, pop 1000) and cfg 5 shard (DeepQN, pop 250): whole generation vs the evaluation alone."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from coevonet_amd.game_logic import initialize_env  # noqa: E402


def wall(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    torch.manual_seed(0)
    from coevonet_amd.dqn_population import DQNESTrainer
    args = bench.make_args(250, 1, 2, 200)
    args.algorithm, args.game, args.coevo_channels, args.fitness_sharing, args.generations = "ES", "boxing_v2", 4, False, 8
    tr = DQNESTrainer(initialize_env(args), args, collect=False)
    for _ in range(2):
        tr.step()
    gen = wall(tr.step, 3)
    eng = tr.eng
    ev = wall(lambda: eng.eval_ro.enqueue(eng.T_eval, eng.gen_dev), 3)
    main_ro = wall(lambda: eng.ro.enqueue(eng.T_train, eng.gen_dev), 3)
    print(f"cfg5 shard: generation {gen:.1f} ms | main rollout alone {main_ro:.1f} ms | evaluation (10 games x {eng.T_eval} steps) "
          f"{ev:.1f} ms = {ev / gen:.2f} of the generation")
    tr.close()
    from coevonet_amd.evolutionary_strategy import ESTrainer
    args = bench.make_args(1000, 1, 2, 200)
    args.algorithm, args.fitness_sharing = "ES", False
    args.coevo_antithetic = args.coevo_centered_rank = False
    tr = ESTrainer(initialize_env(args), args, rng="device_philox", env_mode="device", collect=False)
    for _ in range(2):
        tr.step()
    gen = wall(tr.step, 10)
    ev = wall(lambda: tr.eng.evaluate(3), 10)
    print(f"cfg3: generation {gen:.2f} ms | evaluation (10 games) {ev:.2f} ms = {ev / gen:.2f} of the generation")


main()
