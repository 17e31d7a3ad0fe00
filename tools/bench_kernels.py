#!/usr/bin/env python3
"""Kernel-level timing on the real cfg2 task tables (pop=200, HoF=5): each launch class alone, HIP events, N reps.
    python tools/bench_kernels.py [--reps 50]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args  # noqa: E402
from coevonet_amd import lib as L  # noqa: E402
from coevonet_amd.game_logic import initialize_env  # noqa: E402
from coevonet_amd.genetic_algorithm import GATrainer  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    d = sorted(x.elapsed_time(y) * 1e3 for x, y in ev)
    return d[len(d) // 2], d[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--pop", type=int, default=200)
    ap.add_argument("--hof", type=int, default=5)
    a = ap.parse_args()
    torch.manual_seed(0)
    args = make_args(a.pop, a.hof, 2, 200)
    env = initialize_env(args)
    tr = GATrainer(env, args, rng="device_philox", env_mode="device", collect=False)
    tr.step()
    eng, ro, p = tr.eng, tr.eng.ro, tr.eng.plan

    def heavy():
        L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.heavy), len(p.heavy_np), p.heavy_max,
               L._p(ro.state), p.n_games, L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))

    def light():
        L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.light), len(p.light_np), p.light_max,
               L._p(ro.state), p.n_games, L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))

    def step():
        L.call("coevo_mpe_step", L._p(ro.state), p.n_games, L._p(p.game_rows), L._p(ro.actions), 0,
               L._p(ro.limits), ro.pos_first)

    stamps = torch.zeros(L.STAMP_SLOTS, 2, dtype=torch.int64, device="cuda")

    def light_stamped():
        L.call("coevo_mpe_policy_cycle_stamped", L._p(ro.slab), L._p(p.light), len(p.light_np), p.light_max,
               L._p(ro.state), p.n_games, L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status),
               L._p(stamps))

    med, mn = timeit(light_stamped, a.reps)
    print(f"light with clock stamps: median {med:7.1f} us  min {mn:7.1f} us")
    light_bytes = sum({int(t["net_off"]): L.fc_param_count(int(t["D"])) * 4 for t in p.light_np}.values())
    heavy_bytes = sum(L.fc_param_count(int(t["D"])) * 4 for t in p.heavy_np)
    for name, fn, nbytes in (("light (VALU, per-individual nets)", light, light_bytes),
                             ("heavy (MFMA, shared opponents)", heavy, heavy_bytes), ("env step", step, 0)):
        med, mn = timeit(fn, a.reps)
        extra = f"  {nbytes / med / 1e3:8.1f} GB/s of weight reads" if nbytes else ""
        print(f"{name:36s} median {med:8.1f} us  min {mn:8.1f} us{extra}")
    for n in (128, 256, 384, 512, 600):
        def light_n(n=n):
            L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.light), n, p.light_max, L._p(ro.state),
                   p.n_games, L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))
        med, mn = timeit(light_n, a.reps)
        print(f"light, first {n:4d} tasks: median {med:7.1f} us  min {mn:7.1f} us  {n * 559124 / mn / 1e3:7.1f} GB/s")
    for n in (64, 128, 196):
        def heavy_n(n=n):
            L.call("coevo_mpe_policy_cycle", L._p(ro.slab), L._p(p.heavy), n, p.heavy_max, L._p(ro.state),
                   p.n_games, L._p(p.row_game), L._p(p.row_slot), L._p(ro.actions), L._p(ro.status))
        med, mn = timeit(heavy_n, a.reps)
        print(f"heavy, first {n:4d} tasks: median {med:7.1f} us  min {mn:7.1f} us")
    print(f"tasks: light {len(p.light_np)} (max rows {p.light_max}), heavy {len(p.heavy_np)} (max rows {p.heavy_max})")


if __name__ == "__main__":
    main()
