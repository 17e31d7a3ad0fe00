"""diagnostic (COEVO_PHASE_STAMPS build): begin/end of every workgroup of one merged cycle launch"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, ".")
os.environ["COEVO_MERGED"] = "1"
import bench
from coevonet_amd import lib as L
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import GATrainer

dll = L.load()
torch.manual_seed(0); np.random.seed(0)
args = bench.make_args(200, 5, 2, 200)
args.coevo_cohorts = 1
args.coevo_device_loop = False
env = initialize_env(args)
shard = int(os.environ.get("COEVO_PROBE_SHARD", "1"))   # N: rank 0 of pop 200 over N GPUs (dist.ShardRehearsal)
if shard > 1:
    from coevonet_amd.dist import ShardRehearsal
    args.coevo_device_loop = True
    tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False, dist_ctx=ShardRehearsal(0, shard))
else:
    tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
eng, ro = tr.eng, tr.eng.ro
tr.step()
ro.use_graph = False
for i in range(2):
    ro.run(eng.n_cycles)
torch.cuda.synchronize()
nh, nl = len(eng.plan.heavy_np), len(eng.plan.light_np)
lean = eng.plan.heavy_max <= 16 and nh + nl <= 1024
n = nh + nl if lean else (nh + (nl + 1) // 2 if nh + nl > 512 else nh + nl)
dll.coevo_debug_read_phase_stamps.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * (n * 16))()
assert dll.coevo_debug_read_phase_stamps(buf, n * 16) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.int64)
t0 = st[:, 0].min()
b, e = (st[:, 0] - t0) / 100.0, (st[:, 6] - t0) / 100.0   # us (100 MHz ticks)
def desc(name, idx):
    print(f"{name:24s} n={len(idx):4d} begin {b[idx].min():7.1f}..{b[idx].max():7.1f}  end {e[idx].min():7.1f}..{e[idx].max():7.1f}  life {np.mean(e[idx]-b[idx]):6.1f} us")
heavy = np.arange(nh); light = np.arange(nh, n)
desc("heavy", heavy)
first = light[b[light] < 5]; later = light[b[light] >= 5]
desc("light, first round", first)
if len(later): desc("light, later", later)
ph = np.diff(st[:, :7], axis=1) / 100.0
for nm, idx in (("heavy", heavy), ("light first", first), ("light later", later)):
    if not len(idx): continue
    print(nm, "phases us:", np.round(ph[idx].mean(axis=0), 1))
print("launch span us:", e.max())

order = np.argsort(e)
print("last 10 to finish:", [(int(i), "H" if i < nh else "L", round(float(b[i]), 1), round(float(e[i]), 1)) for i in order[-10:]])
hist, edges = np.histogram(b[light], bins=12)
print("light begin histogram (us):", list(zip(np.round(edges[:-1], 0).tolist(), hist.tolist())))

if len(light):
    f = st[light]
    names = ["entry barrier -> fc1 done", "fc1 done -> sums written", "sums barrier", "mean/var/normalise (2 barriers)", "h1q write + barrier"]
    idx = [(1, 8), (8, 9), (9, 10), (10, 11), (11, 12)]
    for nm, (i, j) in zip(names, idx):
        print(f"  light {nm:34s} {np.mean(f[:, j] - f[:, i]) / 100.0:6.2f} us")

# per-wave stamps of the compact per-individual body: arrival at / release from each barrier of fc1 + LayerNorm(512)
try:
    dll.coevo_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_int]
    wb = (C.c_ulonglong * (n * 4 * 16))()
    assert dll.coevo_debug_read_wave_stamps(wb, n * 4 * 16) == 0
    ws = np.frombuffer(wb, dtype=np.uint64).reshape(n, 4, 16).astype(np.int64)[light]
    ws = (ws - st[light][:, None, 0:1]) / 100.0       # us since the workgroup's start
    names = ["arrive B0 (params, obs)", "leave B0", "arrive B1 (loop1 sums)", "leave B1", "arrive B2 (loop2)", "leave B2",
             "arrive B3 (loop3)", "leave B3", "stream done"]
    for i, nm in enumerate(names):
        v = ws[:, :, i]
        print(f"  {nm:26s} mean {v.mean():6.2f}  first wave {v.min(axis=1).mean():6.2f}  last wave {v.max(axis=1).mean():6.2f}  per wave {np.round(v.mean(axis=0), 2)}")
    raw = np.frombuffer(wb, dtype=np.uint64).reshape(n, 4, 16).astype(np.int64)[light]
    for nm, (c0, c1, r0, r1) in (("fc1+LN1 section", (10, 11, 1, 2)), ("section + stream", (10, 12, 1, 8))):
        dc = raw[:, :, c1] - raw[:, :, c0]
        dr = (raw[:, :, r1] - raw[:, :, r0]) / 100.0
        print(f"  {nm}: shader cycles per wave {np.round(dc.mean(axis=0))}, us {np.round(dr.mean(axis=0), 2)}, "
              f"effective clock {dc.sum() / dr.sum():.0f} MHz")
except Exception as e:
    print("no wave stamps:", e)
