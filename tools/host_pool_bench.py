"""Host share of one cohort-cycle (coevo_host_rollout_step: world step + observations of n games on T host cores), without
the GPU: p10 / p50 / p90 microseconds per call for T threads x n games.  python tools/host_pool_bench.py [gap_us]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coevonet_amd import lib as L   # noqa: E402
from coevonet_amd.mpe import simple_adversary as sa   # noqa: E402

gap = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
lib = L.load()
n = 3010
rng = np.random.default_rng(0)
st = np.zeros((L.MPE_STATE_DOUBLES, n))
lib.coevo_mpe_host_reset(st.ctypes.data, n, L.PCG64State.from_seed(sa.ENV_SEED), np.arange(n, dtype=np.int64).ctypes.data)
game_rows = np.arange(3 * n, dtype=np.int32).reshape(n, 3).copy()
limits = np.full(n, 75, np.int32)
acts = rng.integers(0, 5, size=3 * n).astype(np.int32)
obs = np.zeros((3 * n, L.OBS_STRIDE), np.float32)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "gap_us", gap)
for T in (1, 2, 4, 8, 12, 16):
    ctx = lib.coevo_host_rollout_create(T, 2)
    for ng in (753, 1003, 1505, 3010):
        games = np.arange(ng, dtype=np.int32)
        ts = []
        for c in range(500):
            t0 = time.perf_counter()
            lib.coevo_host_rollout_step(ctx, st.ctypes.data, n, game_rows.ctypes.data, acts.ctypes.data, 3 * n, c % 25,
                                        limits.ctypes.data, 1, games.ctypes.data, ng, 1, obs.ctypes.data)
            t1 = time.perf_counter()
            ts.append(t1 - t0)
            while time.perf_counter() - t1 < gap * 1e-6:   # what the caller does between two cohort-cycles
                pass
        ts = np.array(ts[100:]) * 1e6
        print(f"T {T:2d} games {ng:4d}  p10 {np.percentile(ts, 10):6.1f}  p50 {np.percentile(ts, 50):6.1f}  "
              f"p90 {np.percentile(ts, 90):6.1f} us", flush=True)
    lib.coevo_host_rollout_destroy(ctx)
