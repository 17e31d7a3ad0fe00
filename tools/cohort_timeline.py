"""diagnostic: when do the per-individual launches of each cohort run inside one replayed generation graph?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer

K = int(sys.argv[1])
L.load()
torch.manual_seed(0); np.random.seed(0)
args = bench.make_args(200, 5, 2, 200)
args.coevo_cohorts = K
env = initialize_env(args)
tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
ro = tr.eng.ro
ro.time_light = True
for i in range(4):
    tr.step()
torch.cuda.synchronize()
n = tr.eng.n_cycles
st = ro.stamps[:n * ro.n_cohorts].cpu().numpy().astype(np.int64)   # [k*n + c][slot][2]
t0 = st[:, :, 0].min()
for c in range(min(n, 6)):
    line = []
    for k in range(ro.n_cohorts):
        s = st[k * n + c]
        line.append(f"k{k}: {(s[:, 0].min() - t0) / 100:8.1f} .. {(s[:, 1].max() - t0) / 100:8.1f} us")
    print(f"cycle {c}: " + "   ".join(line))
print("span of all light launches:", (st[:, :, 1].max() - t0) / 100, "us")
