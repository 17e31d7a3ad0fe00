"""diagnostic: 600 distinct nets, one game per trio -> only per-individual (light) tasks; rollout time vs cohorts"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from coevonet_amd import lib as L
from coevonet_amd.rollout import RolloutPlan, DeviceRollout

L.load()
dev = "cuda"
s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
N = 200
off, D, o = [], [], 0
for i in range(N): off.append(o); D.append(8); o += s8
for i in range(2 * N): off.append(o); D.append(10); o += s10
slab = torch.randn(o, device=dev) * 0.05
games = [(i, N + 2 * i, N + 2 * i + 1) for i in range(N)]
ncyc = 25
for K in [int(x) for x in sys.argv[1:]]:
    plan = RolloutPlan(np.array(games), off, D, device=dev, n_cohorts=K)
    ro = DeviceRollout(plan, slab)
    ro.set_limits(np.full(plan.n_games, 75))
    ro.reset(0, plan.n_games, 1)
    for i in range(3):
        ro.run(ncyc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        ro.run(ncyc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    gb = plan.distinct_weight_bytes_per_cycle() / 1e9
    print(f"K={plan.n_cohorts}: {dt * 1e6 / ncyc:.1f} us/cycle  -> {gb / (dt / ncyc):.0f} GB/s aggregate", flush=True)
