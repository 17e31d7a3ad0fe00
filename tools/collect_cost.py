"""diagnostic: generations/s of the trainer with and without per-generation result collection"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer
for collect in (False, True):
    torch.manual_seed(0); np.random.seed(0)
    args = bench.make_args(200, 5, 2, 200); args.generations = 200
    env = initialize_env(args)
    tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=collect)
    for _ in range(5): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): tr.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"collect={collect}: {100 / dt:.1f} generations/s")
