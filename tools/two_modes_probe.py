#!/usr/bin/env python3
"""One process, both modes: host-cores env mode before and after the device-resident loop has run in the same process
(profiles/r04_experiments.md section 1: 450 -> 240 generations/s).  Prints, per phase, generations/s, the per-cohort-cycle
breakdown and where the caller's thread runs.  python tools/two_modes_probe.py [order]   (order: e.g. hdh = host, device, host)"""
import gc
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_args   # noqa: E402
from coevonet_amd.game_logic import initialize_env   # noqa: E402
from coevonet_amd.genetic_algorithm import GATrainer   # noqa: E402


def run(mode, gens=25, **kw):
    torch.manual_seed(0)
    np.random.seed(0)
    args = make_args(200, 5, 2, 200)
    args.generations = gens + 8
    for k, v in kw.items():
        setattr(args, k, v)
    env = initialize_env(args)
    tr = GATrainer(env, args, rng="device_philox", env_mode=mode, collect=False)
    for _ in range(4):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(gens):
        tr.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ph = None
    if mode == "host":
        tr.eng.ro.phase_us = np.zeros(6)
        tr.step()
        ph = tr.eng.ro.phase_us.round(1).tolist()
    cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1
    print(f"{mode:7s} {kw or ''} {gens / dt:7.1f} generations/s  cpu {cpu}  phases {ph}", flush=True)
    tr.finish()
    if hasattr(tr.eng.ro, "close"):
        tr.eng.ro.close()
    del tr, env
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


order = sys.argv[1] if len(sys.argv) > 1 else "hdh"
for c in order:
    if c == "h":
        run("host")
    elif c == "d":
        run("device")
    elif c == "n":
        run("device", coevo_device_loop=False)
    elif c == "1":
        run("device", coevo_cohorts=1)
    elif c == "c":      # only the rollout context of the device mode: side stream (high priority), 4096 timing event pairs
        from coevonet_amd import lib as L
        ctx = L.load().coevo_rollout_ctx_create(4096)
        L.load().coevo_rollout_ctx_reserve_cohorts(ctx, 2)
        torch.cuda.synchronize()
        L.load().coevo_rollout_ctx_destroy(ctx)
        print("ctx created + destroyed", flush=True)
    elif c == "C":      # ... without the timing events
        from coevonet_amd import lib as L
        ctx = L.load().coevo_rollout_ctx_create(0)
        L.load().coevo_rollout_ctx_reserve_cohorts(ctx, 2)
        torch.cuda.synchronize()
        L.load().coevo_rollout_ctx_destroy(ctx)
        print("ctx(0) created + destroyed", flush=True)
    elif c == "g":      # only a captured + replayed hipGraph of a trivial torch kernel
        x = torch.zeros(1024, device="cuda")
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            x += 1
        for _ in range(10):
            gr.replay()
        torch.cuda.synchronize()
        del gr
        print("graph captured + replayed", flush=True)
    elif c == "e":      # only many timing events
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8192)]
        for e in ev[:64]:
            e.record()
        torch.cuda.synchronize()
        del ev
        print("8192 timing events", flush=True)
    elif c == "p":      # only a high-priority stream
        s = torch.cuda.Stream(priority=-1)
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
        torch.cuda.synchronize()
        del s
        print("priority stream", flush=True)
