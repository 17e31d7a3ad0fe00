#!/usr/bin/env python3
"""cfg3 (BASELINE.json configs[2]): Co-ES pop=1000, sigma=0.05, device perturb + update.  Not the headline bench."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args
from coevonet_amd.evolutionary_strategy import ESTrainer
from coevonet_amd.game_logic import initialize_env

ap = argparse.ArgumentParser()
ap.add_argument("--pop", type=int, default=1000)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
torch.manual_seed(0); np.random.seed(0)
args = make_args(a.pop, 1, 2, 200)
args.algorithm = "ES"; args.fitness_sharing = False
env = initialize_env(args)
tr = ESTrainer(env, args, rng="device_philox", env_mode="device", collect=False)
for _ in range(2):
    tr.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    tr.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
P = {"agent_0": 559124, "agent_1": 559124, "adversary_0": 555028}
cyc = tr.eng.plan and (tr.eng.T_train + 2) // 3
gb = sum(a.pop * (1 + cyc) * b for b in P.values()) / 1e9   # materialise-once model of SURVEY 8d: write n*4P, read C*n*4P
print(f"  materialise-once bytes {gb:.1f} GB/generation -> {gb / dt / 1e3:.2f} TB/s = {gb / dt / 8000:.2f} of the 8 TB/s peak")
print(f"Co-ES pop={a.pop}: {1/dt:.2f} generations/s, {dt*1e3:.2f} ms/generation, "
      f"{tr.eng.steps_per_generation/dt/1e6:.1f} M agent-steps/s; tasks light {len(tr.eng.plan.light_np)} heavy {len(tr.eng.plan.heavy_np)}")
