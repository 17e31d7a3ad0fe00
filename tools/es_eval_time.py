"""diagnostic: what do the parts of one Co-ES generation cost (cfg3)?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import make_args
from coevonet_amd.evolutionary_strategy import ESTrainer
from coevonet_amd.game_logic import initialize_env

torch.manual_seed(0); np.random.seed(0)
args = make_args(1000, 1, 2, 200); args.algorithm = "ES"; args.fitness_sharing = False
env = initialize_env(args)
tr = ESTrainer(env, args, rng="device_philox", env_mode="device", collect=False)
eng = tr.eng
for _ in range(3): tr.step()
def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
sig = {r: 0.05 for r in ("agent_0", "agent_1", "adversary_0")}
print("perturb   %.3f ms" % timed(lambda: eng.perturb_device(5, sig)))
print("rollout   %.3f ms" % timed(lambda: eng.rollout(5)))
print("update    %.3f ms" % timed(lambda: eng.update_device(5, 0.1, False)))
print("evaluate  %.3f ms" % timed(lambda: eng.evaluate(5)))
print("eval plan tasks: light", len(eng.eval_plan.light_np), "heavy", len(eng.eval_plan.heavy_np))
