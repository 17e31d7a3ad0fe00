"""diagnostic: host time to enqueue one generation (no sync) vs GPU time per generation, host-free GA loop"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer

L.load()
torch.manual_seed(0); np.random.seed(0)
args = bench.make_args(200, 5, 2, 200)
args.generations = 400
if len(sys.argv) > 1:
    args.coevo_cohorts = int(sys.argv[1])
env = initialize_env(args)
tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
for i in range(5):
    tr.step()
torch.cuda.synchronize()
n = 100
t0 = time.perf_counter()
for i in range(n):
    tr.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"cohorts={tr.eng.ro.n_cohorts}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/gen, total {1e3 * (t2 - t0) / n:.3f} ms/gen "
      f"({n / (t2 - t0):.1f} gen/s)")
