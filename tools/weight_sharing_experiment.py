#!/usr/bin/env python3
"""How much of the headline is memory at all?  TIMING EXPERIMENT, results discarded: every per-individual task of the
headline workload is pointed at one of k weight sets per role (k = 4: L2-resident; k = 64: beyond the L2s, inside the 256 MiB
Infinity Cache; k = 200: the real thing, HBM), everything else unchanged.  (DESIGN.md section 4 argues from these figures; round 2
measured them ad hoc, this tool reproduces them.)

    python tools/weight_sharing_experiment.py [--steps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from coevonet_amd.game_logic import initialize_env  # noqa: E402
from coevonet_amd.genetic_algorithm import GATrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()


def run(k):
    torch.manual_seed(0)
    np.random.seed(0)
    args = bench.make_args(200, 5, 2, 200)
    args.generations = a.steps + 5
    tr = GATrainer(initialize_env(args), args, rng="device_philox", env_mode="device", collect=False)
    p = tr.eng.plan
    if k < 200:
        t = p.light_np.copy()
        for D in np.unique(t["D"]):
            sel = t["D"] == D
            offs = np.unique(t["net_off"][sel])                    # the role(s) of this width, individuals ascending
            roles = max(len(offs) // 200, 1)
            keep = np.concatenate([offs[r * 200:r * 200 + k] for r in range(roles)])
            idx = np.searchsorted(offs, t["net_off"][sel])
            t["net_off"][sel] = keep[(idx // 200) * k + (idx % 200) % k]
        p.light.copy_(torch.from_numpy(t.view(np.uint8).copy()).to(p.light.device))
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nets = len(np.unique(p.light_np["net_off"])) if k >= 200 else 3 * k
    print(f"k = {k:3d} weight sets per role ({nets} per-individual nets, {nets * 0.558:.0f} MB): {a.steps / dt:6.1f} generations/s", flush=True)


for k in (4, 64, 200):
    run(k)
