"""probe: eager vs graph rollout with cohorts (diagnostic)"""
import sys, faulthandler
import numpy as np, torch
sys.path.insert(0, ".")
faulthandler.enable()
from coevonet_amd import lib as L
from coevonet_amd.rollout import RolloutPlan, DeviceRollout

L.load()
dev = "cuda"
s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
npop, nh = 24, 2
# nets: pop agent_0 [0..npop), hof agent_1 [npop..npop+nh), hof adv
off, D = [], []
o = 0
for i in range(npop): off.append(o); D.append(10); o += s10
for i in range(nh): off.append(o); D.append(10); o += s10
for i in range(nh): off.append(o); D.append(8); o += s8
slab = (torch.randn(o, device=dev) * 0.1)
games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]
mode, K, ncyc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
res = {}
for kk in (1, K):
    plan = RolloutPlan(np.array(games), off, D, device=dev, n_cohorts=kk)
    ro = DeviceRollout(plan, slab)
    ro.use_graph = (mode == "graph")
    ro.set_limits(np.full(plan.n_games, 75))
    ro.reset(0, plan.n_games, 1)
    torch.cuda.synchronize()
    print("cohorts", plan.n_cohorts, plan.heavy_begin_np, plan.light_begin_np, flush=True)
    ro.run(ncyc)
    torch.cuda.synchronize()
    res[kk] = ro.rewards.cpu().numpy().copy()
    print("ran", kk, flush=True)
print("equal:", np.array_equal(res[1], res[K]))
