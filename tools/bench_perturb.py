#!/usr/bin/env python3
"""fc_perturb_kernel timing: 100 children of one role (cfg2 cohort shape), with and without the fused distance"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coevonet_amd import lib as L
dev = "cuda"
D, n, E = 10, 100, 2
stride = L.fc_slab_stride(D)
nb = int(L.load().coevo_fc_perturb_blocks(D))
elite = torch.randn(E, stride, device=dev) * 0.05
pop = torch.zeros(n + 1, stride, device=dev)
stale = torch.randn(stride, device=dev) * 0.05
pidx = torch.arange(n, dtype=torch.int32, device=dev) % E
sigma = torch.full((1,), 0.05, device=dev)
part = torch.zeros(n * nb, dtype=torch.float64, device=dev)


def run(dist):
    L.call("coevo_fc_perturb_dist", L._p(elite), L._p(pidx), L._p(pop), 1, n, D, L._p(sigma), 0, 0, 3, 0, None,
           L._p(stale) if dist else None, L._p(part) if dist else None)


for dist in (True, False):
    for _ in range(3):
        run(dist)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record(); run(dist); e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)[10]
    P = L.fc_param_count(D)
    print(f"perturb {n} nets (dist={dist}): {ms * 1e3:.1f} us, {n * P / ms / 1e6:.1f} G params/s, "
          f"{2 * n * stride * 4 / ms / 1e6:.0f} GB/s read+write")
