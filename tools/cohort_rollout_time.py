"""diagnostic: time of one generation's rollout (26 cycles) alone, eager vs graph, for K cohorts"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer

L.load()
for K in [int(x) for x in sys.argv[1:]]:
    torch.manual_seed(0); np.random.seed(0)
    args = bench.make_args(200, 5, 2, 200)
    args.coevo_cohorts = K
    args.coevo_device_loop = False
    env = initialize_env(args)
    tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
    eng, ro = tr.eng, tr.eng.ro
    tr.step()
    n = eng.n_cycles
    for graph in (False, True):
        ro.use_graph = graph
        for i in range(3):
            ro.run(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for i in range(reps):
            ro.run(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"K={K} graph={graph}: rollout {dt * 1e3:.3f} ms = {dt * 1e6 / n:.1f} us/cycle", flush=True)
    del tr, eng, ro
