#!/usr/bin/env python3
"""Headline benchmark: Co-GA population evaluation on MPE simple_adversary_v3, pop=200, HoF=5, T=200 (BASELINE.json
configs[1] / `metric`: "pop=200 HoF=5 ... 1/2/4/8 GPU").  One "step" = one full generation: 3*pop*hof training games + the
10 evaluation games of the best trio, fitness sharing, ranking, HoF update and (pop-1) mutated offspring per role, all on
the device.  `--gpus N` shards THAT population over N ranks (strong scaling, the metric's own split; `--scaling weak`
keeps 200 individuals per GPU instead).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the contract in DESIGN.md "Measurement").
Synthetic data: random-init nets (torch default init, seed 0) and the env's own seeded reset stream.
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto this many hardware queues round robin (default 4): after the headline engine's cohort streams the
# cohort stream of a later workload in the same process landed on the caller's queue and its launches serialised (cfg5 in
# the `extra` block: 8.3 instead of 10.1 generations/s).  Read by the runtime when it starts: set before torch loads it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def compact(o, digits=4):
    """floats to `digits` significant digits, recursively: the driver keeps 8 KB of this line"""
    if isinstance(o, float):
        return float(f"{o:.{digits}g}")
    if isinstance(o, dict):
        return {k: compact(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [compact(v, digits) for v in o]
    return o


def host_cores():
    """cores this process may use: the affinity mask cut by the cgroup CPU quota (a GPU box hands a share of a larger host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    return {"nproc": os.cpu_count(), "affinity": n, "cgroup_quota": quota, "usable": min(n, quota) if quota else n}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def make_args(pop, hof, elites, limit):
    return Bag(algorithm="GA", generations=0, population=pop, hof_size=hof, game="simple_adversary_v3",
               mutation_power_agent_0=0.05, mutation_power_agent_1=0.05, mutation_power_adversary=0.05,
               learning_rate=0.1, max_timesteps_per_episode=limit, max_evaluation_steps=limit, elites_number=elites,
               adaptive=True, max_mutation_power=0.2, min_mutation_power=0.001, fitness_sharing=True,
               early_stopping=False, patience=300, min_delta=0.1, debug=False, train=True, test=False, render=False,
               env_mode="AEC", precision="float32", save=False, average_window=50, play_against_yourself=False)


def cpu_baseline(pop, hof, limit, budget_s=15.0):
    """The oracle port (sequential, batch-1 forward per agent-step, one AEC episode per game: how the reference runs)
    timed on this host, single thread, on a bounded sample of generation 0's games of the same workload."""
    from oracle import ref_port as rp
    torch.manual_seed(0)
    nets10 = [rp.init_net(10) for _ in range(8)]
    nets8 = [rp.init_net(8) for _ in range(4)]
    stream = rp.Stream()
    rp.play_game(stream, nets10[0], nets10[1], nets8[0], limit, 25)  # warm
    t0 = time.perf_counter()
    games = steps = 0
    while time.perf_counter() - t0 < budget_s:
        g = rp.play_game(stream, nets10[games % 8], nets10[(games + 3) % 8], nets8[games % 4], limit, 25)
        games += 1
        steps += g["steps"]
    dt = time.perf_counter() - t0
    games_per_gen = 3 * pop * hof + 10
    return {"value": (games / dt) / games_per_gen, "unit": "generations/s", "env_steps_per_sec": steps / dt,
            "cores": 1, "kind": "port",
            "sample": f"{games} sequential games ({steps} agent-steps) of the same workload in {dt:.1f} s, "
                      f"extrapolated to {games_per_gen} games/generation; selection+mutation not included"}


def _cpu_worker(job):
    """one process of the multi-core CPU baseline: plays sequential games with the oracle for `budget_s` seconds"""
    seed, limit, budget_s = job
    from oracle import ref_port as rp
    torch.manual_seed(seed)
    nets10 = [rp.init_net(10) for _ in range(4)]
    nets8 = [rp.init_net(8) for _ in range(2)]
    stream = rp.Stream()
    stream.ordinal = 1 + 100000 * seed
    t0 = time.perf_counter()
    games = steps = 0
    while time.perf_counter() - t0 < budget_s:
        g = rp.play_game(stream, nets10[games % 4], nets10[(games + 1) % 4], nets8[games % 2], limit, 25)
        games += 1
        steps += g["steps"]
    return games, steps, time.perf_counter() - t0


def cpu_baseline_all_cores(pop, hof, limit, workers=16, budget_s=10.0):
    """the same port, one independent process per host core (games are independent): the fair 'all host cores' figure"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")  # never fork a process that has initialised the GPU
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(i, limit, budget_s) for i in range(workers)])
    games = sum(r[0] for r in res)
    steps = sum(r[1] for r in res)
    dt = max(r[2] for r in res)
    games_per_gen = 3 * pop * hof + 10
    return {"value": (games / dt) / games_per_gen, "unit": "generations/s", "env_steps_per_sec": steps / dt,
            "cores": workers, "kind": "port", "host": host_cores(), "cpu_model": cpu_model(),
            "torch_threads_per_process": torch.get_num_threads(),
            "sample": f"{games} games ({steps} agent-steps) over {workers} processes in {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle-ms", type=float, default=120.0, help="after the W warm-up steps keep stepping (untimed) until the "
                    "device has been under this workload for this long in total: a generation of the headline is 1.7 ms, and "
                    "the part needs ~100 ms of load before its clocks settle (W = 5: 573-590 generations/s, W = 60 or more: "
                    "600-605 on the same box; profiles/r05_experiments.md section 7).  0 = exactly W steps.  The line "
                    "carries settle_steps")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None, help="ga / es under --gpus N: strong (default) = "
                    "BASELINE's metric, ONE population (--pop, 200) sharded over the N ranks; weak = --pop-per-gpu "
                    "individuals on every GPU.  (dqn-ga / dqn-es are the per-GPU shards of configs[3] / [4]: always weak)")
    ap.add_argument("--pop", type=int, default=None, help="total population under --scaling strong (default 200; es: 1000)")
    ap.add_argument("--pop-per-gpu", type=int, default=None, help="per-GPU population under --scaling weak (default 200); "
                    "given alone it selects weak scaling")
    ap.add_argument("--shard-of", type=int, default=0, metavar="N", help="ga on ONE GPU: run rank 0 of an N-rank strong-"
                    "scaling split of the population (its games, its breeding, the elite rebuild; the all-gather replaced "
                    "by tiling its own shard) - the per-GPU half of the scaling curve")
    ap.add_argument("--hof", type=int, default=5)
    ap.add_argument("--elites", type=int, default=2)
    ap.add_argument("--limit", type=int, default=200)
    ap.add_argument("--env", default="device", choices=["device", "host"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every launch eagerly instead of replaying the "
                    "captured hipGraph of a generation's rollout")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--stamp-every", type=int, default=0, help="ga: read the dominant kernel's launch stamps back every N-th "
                    "generation inside the timed region (a host sync each time); 0 = the last timed generation only")
    ap.add_argument("--no-device-loop", action="store_true", help="drive every generation from the host (the code path "
                    "the sharded multi-GPU run uses), also on one GPU")
    ap.add_argument("--max-cycles", type=int, default=25, help="world steps before the env truncates a game; 25 is what "
                    "the reference constructs (75 agent-steps per game); >= 67 lets T=200 bind (SURVEY 8d, cfg 2-T200)")
    ap.add_argument("--sharded-path", action="store_true", help="run the population-sharded loop (what --gpus N > 1 "
                    "uses) even on one GPU: its per-GPU cost without the all-gather")
    ap.add_argument("--workload", default="ga", choices=["ga", "es", "dqn-ga", "dqn-es"],
                    help="ga = the headline (BASELINE configs[1]); es = configs[2]; dqn-ga / dqn-es = the per-GPU shards of "
                         "configs[3] / configs[4] over the synthetic Atari-shaped env")
    ap.add_argument("--extension", action="store_true", help="es / dqn-es: antithetic pairs + centered ranks (BASELINE's "
                    "wording of configs[2]; NOT the reference's algorithm)")
    ap.add_argument("--es-pop-per-gpu", type=int, default=None)
    ap.add_argument("--dqn-pop-per-gpu", type=int, default=None, help="default 50 (dqn-ga) / 250 (dqn-es)")
    ap.add_argument("--dqn-hof", type=int, default=10)
    ap.add_argument("--channels", type=int, default=4, help="frame channels: 4 = BASELINE's 84x84x4; the reference's "
                    "wrapper stack would yield 6")
    ap.add_argument("--frames", default="device", choices=["device", "host"], help="dqn-ga / dqn-es: host = the env in host "
                    "memory (frames rendered by the host cores, copied up every agent-step, actions copied down): the "
                    "PCIe-inclusive form; device = frames synthesised in HBM (the measured form of cfg 4 / cfg 5)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configs carried in "
                    "the headline JSON's `extra` block")
    ap.add_argument("--cohorts", type=int, default=None, help="independent game cohorts per rollout (default: the "
                    "engine's DEFAULT_COHORTS)")
    a = ap.parse_args()
    if a.scaling is None:
        a.scaling = "weak" if (a.pop_per_gpu is not None or a.es_pop_per_gpu is not None) else "strong"
    if a.scaling == "strong" and a.workload in ("ga", "es"):
        total = a.pop or (200 if a.workload == "ga" else 1000)
        if total % a.gpus:
            raise SystemExit(f"--scaling strong: population {total} is not divisible by --gpus {a.gpus}")

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as fresh child processes (one per GPU, torch.distributed.run)
        # BEFORE this process touches the GPU, forward their output (rank 0 prints the JSON line) and their exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from coevonet_amd import lib as L
    from coevonet_amd.dist import DistContext

    ctx = DistContext()
    if a.shard_of:
        if ctx.world != 1 or a.workload != "ga" or a.shard_of < 2:
            raise SystemExit("--shard-of N (N >= 2) is a one-GPU rehearsal of the ga workload")
        from coevonet_amd.dist import ShardRehearsal
        ctx = ShardRehearsal(0, a.shard_of)
        a.gpus = 1
    elif ctx.world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run")
    dev_index = ctx.local_rank % max(torch.cuda.device_count(), 1)  # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    L.load()
    torch.manual_seed(0)
    np.random.seed(0)
    if a.workload == "ga":
        out = run_ga(a, ctx, dev)
    elif a.workload == "es":
        out = run_es(a, ctx, dev)
    else:
        out = run_dqn(a, ctx, dev, a.workload[4:])
    scaling = a.scaling if a.workload in ("ga", "es") else "weak"
    rehearsal = bool(getattr(ctx, "rehearsal", False))
    real_world = 1 if rehearsal else ctx.world
    out.update({"n_gpus": real_world, "steps": a.steps, "warmup": a.warmup, "settle_steps": getattr(a, "settle_steps", 0),
                "settle_ms": a.settle_ms, "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "noise": f"philox4x32-{L.load().coevo_noise_rounds()}", "coevo_version": L.load().coevo_version(),
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "hip_dynamic_queues": os.environ.get("DEBUG_HIP_DYNAMIC_QUEUES"),
                "dist_backend": (torch.distributed.get_backend() if real_world > 1 else None),
                "rccl_ranks": (torch.distributed.get_world_size() if real_world > 1 else 1)})
    if rehearsal:
        out["rehearsal"] = (f"rank 0 of {ctx.world} of a strong-scaling split on ONE GPU: its games, breeding and elite "
                            "rebuild; the all-gather replaced by tiling its own shard (no collective)")
    if real_world > 1:
        us = ctx.gather_times_us()
        # collectives of the timed region on this rank's stream (HIP events): separates exchange from compute in a scaling
        # record.  RCCL: the stream-ordered all-gather itself; gloo (ranks sharing a device): host-staged, not xGMI.
        out["allgather_us"] = {"per_generation": float(np.sum(us)) / max(a.steps, 1), "calls_per_generation":
                               len(us) / max(a.steps, 1), "max_call": float(np.max(us)) if us else 0.0,
                               "transport": "rccl" if torch.distributed.get_backend() == "nccl" else "gloo (host-staged)"}
    if real_world > 1 and a.workload in ("ga", "es") and scaling == "strong" and not a.no_extra:
        # the other reading of "N GPUs": per-GPU work fixed (every rank takes part; short run)
        import copy
        w = copy.copy(a)
        w.scaling, w.steps, w.warmup, w.no_cpu_baseline, w.pop_per_gpu, w.es_pop_per_gpu = "weak", 5, 2, True, None, None
        r = run_ga(w, ctx, dev) if a.workload == "ga" else run_es(w, ctx, dev)
        out["weak_scaling"] = {"population": r["config"]["population"], "gens_per_sec": r["gens_per_sec"],
                               "value": r["value"], "ms_per_step": r["ms_per_step"]}
    if ctx.rank == 0:
        if a.workload == "ga" and ctx.world == 1 and not a.no_extra and not a.shard_of:
            out["extra"] = extras(a, ctx, dev)
        print(json.dumps(compact(out)), flush=True)
    ctx.shutdown()


MAX_SETTLE_STEPS = 256


def _warm_up(step, a, ctx, dev, cold_probe=False):
    """the W untimed warm-up steps, then - untimed as well - as many more as it takes until the device has been under this
    workload for --settle-ms in total (every rank runs the same number: it comes from times reduced over the ranks).
    -> the number of extra steps.  cold_probe: the first K of them are timed on the side, bracketed like the real region
    (a.cold_gens_per_sec = what `--settle-ms 0` would have reported, for the record; they count as settling steps)"""
    t0 = time.perf_counter()
    for _ in range(a.warmup):
        step()
    extra = 0
    want = getattr(a, "settle_ms", 0.0) * 1e-3
    a.cold_gens_per_sec = None
    if cold_probe and want > 0 and a.warmup > 0:
        ctx.barrier()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        ctx.barrier()
        a.cold_gens_per_sec = a.steps / ctx.max_over_ranks(time.perf_counter() - tc, dev)
        extra = a.steps
    if want > 0 and a.warmup > 0:
        for _ in range(4):   # (the warm-up steps hold graph captures: their mean overestimates a step)
            torch.cuda.synchronize()
            done = ctx.max_over_ranks(time.perf_counter() - t0, dev)
            if done >= want or extra >= MAX_SETTLE_STEPS:
                break
            per = done / (a.warmup + extra)
            n = int(min(MAX_SETTLE_STEPS - extra, max(1, np.ceil((want - done) / per))))
            for _ in range(n):
                step()
            extra += n
    a.settle_steps = extra
    return extra


def _timed_steps(step, a, ctx, dev, before_timed=None):
    """W untimed warm-up steps (+ the settling steps, _warm_up), then exactly K steps bracketed by barrier + synchronize;
    max over ranks"""
    _warm_up(step, a, ctx, dev)
    if before_timed:
        before_timed()
    if ctx.world > 1:
        ctx.start_gather_timing()   # HIP events around every collective of the timed region -> "allgather_us"
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ctx.barrier()
    return ctx.max_over_ranks(time.perf_counter() - t0, dev)


def run_es(a, ctx, dev, pop_per_gpu=None, extension=None):
    """BASELINE configs[2]: Co-ES on simple_adversary_v3, pop 1000 per GPU, sigma 0.05, on-device perturb + update.
    reference_exact (iid noise, raw rewards: what the reference runs) unless --extension (antithetic pairs + centered
    ranks: BASELINE's wording, NOT in the reference - labelled in the output)."""
    from coevonet_amd.evolutionary_strategy import ESTrainer
    from coevonet_amd.game_logic import initialize_env
    ext = a.extension if extension is None else extension
    if pop_per_gpu is None and a.scaling == "strong":
        pop = a.pop or 1000
        ppg = pop // ctx.world
    else:
        ppg = pop_per_gpu or a.es_pop_per_gpu or 1000
        pop = ppg * ctx.world
    args = make_args(pop, 1, 2, a.limit)
    args.algorithm, args.fitness_sharing = "ES", False
    args.coevo_antithetic = args.coevo_centered_rank = bool(ext)
    env = initialize_env(args)
    env.max_cycles = a.max_cycles
    tr = ESTrainer(env, args, rng="device_philox", env_mode=a.env, collect=False, dist_ctx=ctx)
    eng = tr.eng
    dt = _timed_steps(tr.step, a, ctx, dev)
    gens = a.steps / dt
    cyc = (eng.T_train + 2) // 3
    P4 = {10: 559124, 8: 555028}
    gb = pop * (1 + cyc) * (2 * P4[10] + P4[8])   # materialise-once model of SURVEY 8d: write n*4P, read C*n*4P
    # the dominant launch of a Co-ES generation: one env-cycle of one cohort, every individual's weight set streamed once
    es_traffic, es_src = pmc_traffic("cfg3_es", "fc_cycle_kernel")   # (both modes launch the same kernel on the same bytes)
    return {"metric": "env-steps/sec (agent-steps of the whole job; generations/sec in gens_per_sec), Co-ES "
                      f"simple_adversary_v3 pop={pop}",
            "value": gens * eng.steps_per_generation, "unit": "env-steps/s", "gens_per_sec": gens,
            "ms_per_step": 1e3 * dt / a.steps,
            "config": {"workload": f"Co-ES simple_adversary_v3 pop={pop} ({ppg}/GPU) sigma=0.05 lr=0.1 T={a.limit} "
                                   f"(env max_cycles={a.max_cycles}), " +
                                   ("EXTENSION mode: antithetic pairs + centered ranks (not in the reference)" if ext else
                                    "reference_exact: iid noise, raw rewards"),
                       "population": pop, "agent_steps_per_generation": eng.steps_per_generation,
                       "offspring": "device_philox", "update": f"{eng.chunks} chunk partial sums",
                       "parallelism": f"population shard x{ctx.world}" if ctx.world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "whole generation (materialise-once model: write n*4P, read C*n*4P)",
                         "achieved": gb * gens / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gb * gens / 1e9 / HBM_PEAK_GBS, "traffic": es_traffic, "traffic_source": es_src,
                         "traffic_note": "HBM bytes per launch of fc_cycle_kernel (one env-cycle of one cohort), not per "
                                         "generation", "algorithmic_bytes_per_generation": gb}}


MFMA_F32_PEAK_TFLOPS = 157.3  # dense f32-input MFMA = the f32 vector rate (/opt/skills/guides/MI355X_MICROARCH.md)
DQN_CONV_MAC = 3276800 + 2654208 + 1806336   # conv1 + conv2 + conv3 MACs per frame at C = 4 (3.28/2.65/1.81 M)


def run_dqn(a, ctx, dev, algo, pop_per_gpu=None, T=None):
    """BASELINE configs[3] / [4]: Co-GA (pop 200 over 4 GPUs = 50 per GPU, HoF 10, pong: 6 actions) or Co-ES (pop 2000
    over 8 GPUs = 250 per GPU, boxing: 18 actions) over DeepQN policies on 84x84x4 frames, T=200 agent-steps per game.
    No ALE in the image: the synthetic Atari-shaped env (coevonet_amd.atari_synthetic) - this measures the DeepQN forward
    over (population x HoF x env copies), the on-device offspring and the all-gather, not game dynamics."""
    from coevonet_amd.dqn_population import DQNESTrainer, DQNGATrainer
    from coevonet_amd.game_logic import initialize_env
    ga = algo == "ga"
    ppg = pop_per_gpu or (a.dqn_pop_per_gpu or (50 if ga else 250))
    pop = ppg * ctx.world
    T = T or a.limit
    args = make_args(pop, a.dqn_hof if ga else 1, a.elites, T)
    args.algorithm = "GA" if ga else "ES"
    args.game = "pong_v3" if ga else "boxing_v2"
    args.coevo_channels = a.channels
    args.fitness_sharing = ga          # the GA always computes its diversity (Q3); ES without, as cfg 3
    args.generations = a.steps + a.warmup + MAX_SETTLE_STEPS
    args.coevo_graph = False           # eager enqueue: HIP events bracket sampled launches of the dominant kernel
    args.coevo_frames = a.frames
    env = initialize_env(args)
    tr = (DQNGATrainer if ga else DQNESTrainer)(env, args, collect=False, dist_ctx=ctx)
    eng = tr.eng
    host_frames = a.frames == "host"
    if not host_frames:
        eng.ro.start_timing(pairs=2048, every=7)
    dt = _timed_steps(tr.step, a, ctx, dev, before_timed=lambda: (torch.cuda.synchronize(), eng.ro.reset_timing()))
    conv_ms, fc1_ms = eng.ro.times_ms(0), eng.ro.times_ms(1)
    frame_phase = None
    if host_frames:   # one more generation with HIP events around every cohort-step's copies and forward
        eng.ro.phase_us = np.zeros(5)
        tr.step()
        torch.cuda.synchronize()
        frame_phase = eng.ro.phase_us.copy()
        eng.ro.phase_us = None
    if ga:
        tr.finish()
    gens = a.steps / dt
    lane0 = eng.ro.lanes[0]
    n_frames = lane0["n"]                       # frames per sampled launch (the first cohort's)
    mac = DQN_CONV_MAC + (a.channels - 4) * 64 * 32 * 400
    out = {"metric": f"env-steps/sec (agent-steps of the whole job), Co-{'GA' if ga else 'ES'} over DeepQN on "
                     f"{args.game}-shaped SYNTHETIC frames",
           "value": gens * eng.steps_per_generation, "unit": "env-steps/s", "gens_per_sec": gens,
           "ms_per_step": 1e3 * dt / a.steps,
           "config": {"workload": (f"Co-GA DeepQN pop={pop} ({ppg}/GPU) HoF={args.hof_size} elites={a.elites}" if ga else
                                   f"Co-ES DeepQN pop={pop} ({ppg}/GPU) sigma=0.05 lr=0.1") +
                                  f" T={T} frames 84x84x{a.channels} actions={eng.n_actions}, synthetic env (no ALE in "
                                  "the image; frames keyed by game, step and the previous action), build-defined 2-role loop",
                      "population": pop, "agent_steps_per_generation": eng.steps_per_generation,
                      "games_per_launch": n_frames, "cohorts": len(eng.ro.lanes), "offspring": "device_philox",
                      "parallelism": f"population shard x{ctx.world}" if ctx.world > 1 else "single GPU"}}
    if frame_phase is not None:
        K = len(eng.ro.lanes)
        per_cohort = eng.ro.n_games / K * 84 * 84 * a.channels
        out["host_frames"] = {"host_cores": eng.ro.threads, "cohorts": K, "frame_bytes_per_agent_step": eng.ro.n_games * 84 * 84 * a.channels,
                              "per_cohort_step_us": {"host_wait_for_actions": float(frame_phase[0]),
                                                     "host_book_and_render": float(frame_phase[1]),
                                                     "h2d_frames": float(frame_phase[2]), "forward": float(frame_phase[3]),
                                                     "d2h_actions": float(frame_phase[4])},
                              "pcie_h2d_GBps": per_cohort / max(float(frame_phase[2]), 1e-9) / 1e3,
                              "agent_step_ms": 1e3 * dt / a.steps / max(T, 1),
                              "copy_hidden_under_other_cohort": bool(K > 1 and frame_phase[2] <= frame_phase[1] + frame_phase[0]),
                              "note": "SYNTHETIC env in host memory; the frames of a step depend on the previous step's "
                                      "actions, so a cohort's copy can only hide under ANOTHER cohort's host / GPU work"}
    if conv_ms and fc1_ms:
        c_avg, f_avg = float(np.mean(conv_ms)), float(np.mean(fc1_ms))
        tf = n_frames * 2 * mac / (c_avg * 1e-3) / 1e12
        # fc1: every distinct acting net's 512 x 3136 fp32 matrix once + the activations in / out (mean over both parities)
        nets = np.mean([eng.ro.distinct_nets_per_step(p) for p in (0, 1)])
        fc1_bytes = nets * 512 * 3136 * 4 + n_frames * (3136 + 512) * 4
        fc1_gbs = fc1_bytes / (f_avg * 1e-3) / 1e9
        rounds = (T + 1) // 2
        gen_bytes = eng.ro.weight_bytes_per_round() * rounds
        wl = ("cfg4_dqn_ga" if ga else "cfg5_dqn_es") + ("" if a.channels == 4 else f"_c{a.channels}")
        conv_traffic, conv_src = pmc_traffic(wl, "dqn_conv_kernel")
        tiled = bool(getattr(eng, "fc1_tiled", False))
        fc1_traffic, fc1_src = pmc_traffic(wl, "dqn_fc1_tiled_kernel" if tiled else "dqn_fc1_kernel")
        if fc1_traffic is None:
            fc1_traffic, fc1_src = pmc_traffic(wl, "dqn_fc1_kernel")
        conv_rl = {"bound": "mfma", "kernel": "dqn_conv_kernel (conv stack + per-sample BatchNorm of every frame of one "
                   "agent-step of one cohort on v_mfma_f32_16x16x4_f32; exact f32 = the reference's arithmetic)",
                   "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                   "traffic": conv_traffic, "traffic_source": conv_src, "flops_per_launch": n_frames * 2 * mac,
                   "avg_launch_ms": c_avg,
                   "launches_timed": len(conv_ms)}
        fc1_rl = {"bound": "hbm", "kernel": ("dqn_fc1_tiled_kernel (every acting net's 6.4 MB fc1 matrix, kept tiled for "
                  "v_mfma_f32_16x16x4, streamed once for its <= 16 frames per task)" if tiled else
                  "dqn_fc1_kernel (every acting net's 6.4 MB fc1 matrix streamed once for its <= 16 frames per task, "
                  "v_mfma_f32_4x4x1)"), "achieved": fc1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": fc1_gbs / HBM_PEAK_GBS, "traffic": fc1_traffic, "traffic_source": fc1_src,
                  "algorithmic_bytes_per_launch": fc1_bytes,
                  "avg_launch_ms": f_avg, "launches_timed": len(fc1_ms)}
        dom, other = (conv_rl, fc1_rl) if c_avg >= f_avg else (fc1_rl, conv_rl)
        out["roofline"] = dict(dom)
        out["roofline"]["timing"] = "HIP events around sampled launches of the first cohort on its stream (eager enqueue)"
        out["roofline"]["second_kernel"] = other
        out["roofline"]["concurrent_cohorts"] = len(eng.ro.lanes)
        if len(eng.ro.lanes) > 1:
            out["roofline"]["note"] = ("the sampled launches run beside the other cohort's conv / fc1 launches (separate "
                                       "streams), so a launch's own duration includes the HBM and CU time it shares; "
                                       "`generation` is the whole-rollout rate")
        out["roofline"]["generation"] = {"bound": "hbm", "algorithmic_bytes": gen_bytes,
                                         "achieved": gen_bytes * gens / 1e9, "unit": "GB/s",
                                         "frac": gen_bytes * gens / 1e9 / HBM_PEAK_GBS,
                                         "note": "SURVEY 8d cfg 4/5 byte model: every distinct acting weight set once "
                                                 "per agent-step + frames"}
    tr.close()   # cohort streams + timing events back to the library
    return out


def L_load():
    from coevonet_amd import lib as L
    return L.load()


def pmc_traffic(workload, kernel_substr):
    """HBM bytes per launch of a kernel from the tracked PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter
    per pass, gfx950 correction applied: tools/pmc_traffic_all.py), newest round first: (bytes, source) or (None, None)"""
    for tag in ("r05", "r04", "r03"):
        path = os.path.join(REPO, "profiles", f"{tag}_pmc_hbm_traffic.json")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            j = json.load(f)
        for name, e in j.get("workloads", {}).get(workload, {}).get("kernels", {}).items():
            if kernel_substr in name and "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"], f"profiles/{tag}_pmc_hbm_traffic.json [{workload}] {name}"
    return None, None


def extras(a, ctx, dev):
    """short, bounded runs of the other BASELINE configs on the same GPU, carried in the headline JSON so that the
    driver's record holds them (each is also its own --workload / flag set; the prose - kernel, timing, workload - is in
    that run's own line, here only numbers: the driver keeps 8 KB of this line)"""
    import copy
    import gc
    from coevonet_amd.dist import ShardRehearsal
    ex = {}

    def slim_rl(rl):
        o = {k: rl[k] for k in ("bound", "frac", "avg_launch_ms", "traffic") if rl.get(k) is not None}
        if "generation" in rl:
            o["generation_frac"] = rl["generation"]["frac"]
        if rl.get("rollout_aggregate"):
            o["rollout_frac"] = rl["rollout_aggregate"]["frac"]
            o["rollout_span_ms"] = rl["rollout_aggregate"]["rollout_span_ms"]
        if "second_kernel" in rl:
            o["second"] = {k: rl["second_kernel"][k] for k in ("bound", "frac", "avg_launch_ms", "traffic")
                           if rl["second_kernel"].get(k) is not None}
        return o

    only = [x for x in os.environ.get("COEVO_BENCH_LEGS", "").split(",") if x]   # (diagnostics: a subset of the legs)

    def leg(name, fn):
        if only and name not in only:
            return
        try:
            r = fn()
            ex[name] = {"gens_per_sec": r["gens_per_sec"], "ms_per_step": r["ms_per_step"], "env_steps_per_sec": r["value"]}
            if r.get("roofline"):
                ex[name]["roofline"] = slim_rl(r["roofline"])
            if r.get("host_env"):
                ex[name]["host_env"] = r["host_env"]
            if r.get("host_frames"):
                ex[name]["host_frames"] = {k: v for k, v in r["host_frames"].items() if k != "note"}
        except Exception as e:
            ex[name] = {"error": repr(e)[:200]}
        gc.collect()
        torch.cuda.empty_cache()

    b = copy.copy(a)
    b.steps, b.warmup, b.no_cpu_baseline = 5, 2, True
    # SURVEY 8d "two variants, both reported": the env built with max_cycles >= 67 so that T = 200 binds
    t200 = copy.copy(b)
    t200.max_cycles = 67
    leg("cfg2_T200", lambda: run_ga(t200, ctx, dev))
    # north_star's literal first configuration: the env vectorised on the host cores, observations up / actions down over
    # PCIe every cycle - the PCIe-inclusive rate, never the headline
    # Its own process: after the device-resident loop has run in a process (lane streams + a replayed tail graph), every
    # operation on the host rollout's cohort streams takes ~9 us longer and the leg reads 240 instead of 370 generations/s
    # (profiles/r04_experiments.md); `python bench.py --env host` is how the mode is run anyway.
    def host_leg(*more):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--env", "host", "--no-extra", "--no-cpu-baseline"] + list(more)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
        return json.loads(p.stdout.decode().strip().splitlines()[-1])
    leg("cfg2_host_env", lambda: host_leg("--steps", "40", "--warmup", "3"))
    leg("cfg3_host_env", lambda: host_leg("--workload", "es", "--steps", "5", "--warmup", "2"))
    # the metric's own split (pop 200 over 2 / 4 / 8 GPUs): what ONE rank of it does on this GPU (dist.ShardRehearsal)
    for n in (2, 4, 8):
        sh = copy.copy(b)
        sh.steps, sh.warmup = 20, 3
        leg(f"cfg2_shard_1_of_{n}", lambda: run_ga(sh, ShardRehearsal(0, n), dev))
    leg("cfg3_reference_exact", lambda: run_es(b, ctx, dev, extension=False))
    leg("cfg3_extension", lambda: run_es(b, ctx, dev, extension=True))
    # (two timed generations under-reported cfg5 by 15 %: the first ones carry one-off host work)
    leg("cfg4_shard", lambda: run_dqn(b, ctx, dev, "ga"))
    leg("cfg5_shard", lambda: run_dqn(b, ctx, dev, "es"))
    # six frame planes: what the reference's wrapper stack yields (frame_stack_v1(4) + agent_indicator_v0,
    # utils/game_logic_functions.py:50-53; Atari/atari_agent.py:20); BASELINE.json words the configs as 84x84x4
    c6 = copy.copy(b)
    c6.channels = 6
    leg("cfg4_shard_6_planes", lambda: run_dqn(c6, ctx, dev, "ga"))
    leg("cfg5_shard_6_planes", lambda: run_dqn(c6, ctx, dev, "es"))
    # PCIe-inclusive: the (synthetic) env in host memory, frames up / actions down every agent-step (last: the legs that
    # follow a host-stepped rollout in the same process read a few per cent low, profiles/r04_experiments.md section 1)
    hf = copy.copy(b)
    hf.frames, hf.steps, hf.warmup = "host", 2, 1
    leg("cfg4_shard_host_frames", lambda: run_dqn(hf, ctx, dev, "ga"))
    ex["legend"] = ("cfg2_T200: env max_cycles 67; cfg2_host_env / cfg3_host_env: env on the host cores (PCIe-inclusive); cfg2_shard_1_of_N: "
                    "rank 0 of pop 200 over N GPUs rehearsed on this GPU, no collective; cfg3: Co-ES pop 1000 (extension = "
                    "antithetic + centered ranks, not in the reference); cfg4 / cfg5: per-GPU shards (pop 50, HoF 10 / pop 250) "
                    "over DeepQN on SYNTHETIC 84x84x4 frames (6_planes: 84x84x6; host_frames: the env in host memory, frames over PCIe "
                    "every agent-step)")
    return ex


def run_ga(a, ctx, dev):
    from coevonet_amd import lib as L
    from coevonet_amd.game_logic import initialize_env
    from coevonet_amd.genetic_algorithm import GATrainer

    if a.scaling == "strong":   # BASELINE's metric: ONE population, sharded by index over the ranks
        pop = a.pop or 200
        ppg = pop // ctx.world
    else:                       # weak: per-GPU work fixed
        ppg = a.pop_per_gpu or 200
        pop = ppg * ctx.world
    args = make_args(pop, a.hof, a.elites, a.limit)
    args.generations = 2 * a.steps + a.warmup + MAX_SETTLE_STEPS  # sizes the device-resident evaluation / sigma histories
    if a.cohorts is not None:
        args.coevo_cohorts = a.cohorts
    if a.no_device_loop:
        args.coevo_device_loop = False
    if a.sharded_path:
        args.coevo_force_sharded_loop = True
    env = initialize_env(args)
    env.max_cycles = a.max_cycles
    tr = GATrainer(env, args, rng="device_philox", env_mode=a.env, collect=False, dist_ctx=ctx)
    eng = tr.eng

    timed = a.env == "device"
    if timed:
        eng.ro.use_graph = not a.no_graph
        eng.ro.overlap = not a.no_overlap
        eng.ro.time_light = True  # before the warm-up, so the (timed) graph is captured outside the timed region
    _warm_up(tr.step, a, ctx, dev, cold_probe=True)
    if timed:  # duration of every launch of the dominant kernel in the timed region from here on
        torch.cuda.synchronize()
        eng.ro.collect_stamps()
        eng.ro.reset_timing()
    if ctx.world > 1:
        ctx.start_gather_timing()
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if timed and not a.no_graph:
        # the kernels stamp every launch, but reading the stamps back needs a host sync that drains the enqueued generations
        # (2 % at every 8th generation): only the LAST timed generation's launches (2 cohorts x 25 cycles) are read, after the
        # closing synchronize - `--stamp-every N` samples every N-th generation as well, at that price
        tr.stamp_every = a.stamp_every if a.stamp_every > 0 else 1 << 30
    for i in range(a.steps):
        tr.step()
    torch.cuda.synchronize()
    ctx.barrier()
    dt = ctx.max_over_ranks(time.perf_counter() - t0, dev)
    light_ms = eng.ro.light_times_ms() if timed else []
    if timed:
        eng.ro.time_light = False
    host_phase = None
    if a.env == "host" and getattr(eng.ro, "ctx", None):
        # two more generations with HIP events around every cohort-cycle's copies and launch + host clocks around the
        # waits and the env (outside the timed region: the events cost a few microseconds per cycle)
        eng.ro.phase_us = np.zeros(6)
        ph = []
        for _ in range(2):
            tr.step()
            ph.append(eng.ro.phase_us.copy())
        eng.ro.phase_us = None
        host_phase = np.mean(ph, axis=0)
    tr.finish()

    steps_per_gen = eng.steps_per_generation  # agent-steps of the whole job per generation
    gens_per_s = a.steps / dt
    out = {
        "metric": "env-steps/sec (agent-steps of the whole job; generations/sec in gens_per_sec), Co-GA "
                  f"simple_adversary_v3 pop={pop} HoF={a.hof}",
        "value": gens_per_s * steps_per_gen, "unit": "env-steps/s", "gens_per_sec": gens_per_s,
        "ms_per_step": 1e3 * dt / a.steps,
        # host time of a step (enqueue only: the loop never waits for the device): well below ms_per_step = the host runs ahead
        # the same K steps timed straight after the W warm-up steps, before the settling steps (what --settle-ms 0 reports)
        "cold_gens_per_sec": getattr(a, "cold_gens_per_sec", None),
        "host_enqueue_ms_per_step": 1e3 * float(np.mean(tr.res.seconds[-a.steps:])) if getattr(tr, "res", None) and tr.res.seconds else None,
        "config": {"workload": f"Co-GA simple_adversary_v3 pop={pop} ({ppg}/GPU) HoF={a.hof} "
                               f"elites={a.elites} T={a.limit} (env max_cycles={a.max_cycles} caps a game at "
                               f"{3 * a.max_cycles} agent-steps" + (", as in the reference" if a.max_cycles == 25 else
                                                                     ": the T=200 variant, SURVEY 8d cfg 2-T200") +
                               "), fitness sharing, adaptive sigma",
                   "population": pop, "hof": a.hof, "games_per_generation": 3 * pop * a.hof + 10,
                   "agent_steps_per_generation": steps_per_gen, "env": a.env, "offspring": "device_philox",
                   "parallelism": f"population shard x{ctx.world}" if ctx.world > 1 else "single GPU"},
    }
    if ctx.rank == 0 and host_phase is not None:
        K = eng.plan.n_cohorts
        nets = {}
        for arr in (eng.plan.heavy_np, eng.plan.light_np):
            for t in arr:
                nets[int(t["net_off"])] = L.fc_param_count(int(t["D"])) * 4
        cycle_bytes = sum(nets.values()) + eng.plan.n_rows * (4 * 10 + 4)
        launch_ms = float(host_phase[4]) * 1e-3
        ach = cycle_bytes / K / max(launch_ms, 1e-9) / 1e6
        out["host_env"] = {"host_cores": eng.ro.threads, "cohorts": K, "zero_copy": bool(eng.ro.zero_copy),
                           "per_cohort_cycle_us": {"host_wait_for_actions": float(host_phase[0]),
                                                   "host_step_observe": float(host_phase[1]),
                                                   "host_enqueue": float(host_phase[2]), "h2d_obs": float(host_phase[3]),
                                                   "launch": float(host_phase[4]), "d2h_actions": float(host_phase[5])},
                           "pcie_bytes_per_cycle": eng.plan.n_rows * (4 * L.OBS_STRIDE + 4),
                           # where the host cores run (csrc/host_placement.hip): the caller's thread is pinned to cpus[0] for
                           # the rollout, the workers to the rest; COEVO_HOST_PIN=far reproduces the far-socket placement
                           "placement": getattr(eng.ro, "placement", None)}
        out["roofline"] = {"bound": "hbm", "kernel": "fc_cycle16_kernel<R, MODE_OBS> (observations given; one launch per "
                           "cohort and env-cycle)", "timing": "HIP events on the cohort's stream (2 extra generations)",
                           "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": cycle_bytes / K, "avg_launch_ms": launch_ms,
                           "concurrent_launches": K,
                           "generation": {"algorithmic_bytes": cycle_bytes * eng.n_cycles,
                                          "achieved": cycle_bytes * eng.n_cycles * gens_per_s / 1e9,
                                          "frac": cycle_bytes * eng.n_cycles * gens_per_s / 1e9 / HBM_PEAK_GBS}}
    if ctx.rank == 0:
        if light_ms:
            d = light_ms
            avg_ms = float(np.mean(d))
            # algorithmic bytes of one launch of the dominant kernel: every distinct weight set among its tasks once
            # + observations in / actions out; state reads are the observations' fp64 sources
            merged = bool(eng.ro.desc.merged) and len(eng.plan.heavy_np) > 0
            nets, rows = {}, 0
            for arr in ((eng.plan.heavy_np, eng.plan.light_np) if merged else (eng.plan.light_np,)):
                for t in arr:
                    nets[int(t["net_off"])] = L.fc_param_count(int(t["D"])) * 4
                    rows += int(t["n_rows"])
            cycle_bytes = sum(nets.values()) + rows * (4 * 10 + 4)   # all cohorts, one env-cycle
            alg_bytes = cycle_bytes / max(eng.ro.n_cohorts, 1)    # a launch = one cohort's env-cycle
            achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            K = max(eng.ro.n_cohorts, 1)
            aggregate = None
            if getattr(eng.ro, "_span_ms", None):
                # all policy launches of a rollout together: bytes of every launch / (first start .. last end); with one
                # cohort this is the per-launch figure minus the inter-launch gaps, with K cohorts it accounts for the
                # launches that run side by side
                span_ms = float(np.mean(eng.ro._span_ms))
                tot = alg_bytes * K * eng.ro._span_cycles
                aggregate = {"achieved": tot / (span_ms * 1e-3) / 1e9, "frac": tot / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "rollout_span_ms": span_ms, "launches": K * eng.ro._span_cycles,
                             "note": "algorithmic bytes of every policy launch of one rollout / (first workgroup start "
                                     ".. last workgroup end of the rollout)"}
            # which instantiation the C side picks (coevo_mpe_policy_cycle_merged -> coevo_mpe_cycle_kernel_form): row-count
            # template; the small-launch kernel when no CU holds more than one workgroup, lean 16-row tiles when both
            # cohorts' workgroups fit four per CU, else 32-row tiles (two nets per streaming workgroup when one per
            # workgroup would not fit two per CU)
            Kc = max(eng.ro.n_cohorts, 1)
            R = next(r for r in (1, 2, 5, 8, 32) if eng.plan.light_max <= r)
            if merged:
                hb, lb = eng.plan.heavy_begin_np, eng.plan.light_begin_np
                form = L.load().coevo_mpe_cycle_kernel_form(int(hb[1] - hb[0]), int(lb[1] - lb[0]), eng.plan.heavy_max,
                                                            eng.plan.light_max, Kc)
                Rs = next(r for r in (1, 2, 5, 8) if max(eng.plan.light_max, min(eng.plan.heavy_max, 8)) <= r)
                kernel_id = {0: f"fc_cycle_kernel<{R}, 1>", 1: f"fc_cycle_kernel<{R}, 2>", 2: f"fc_cycle16_kernel<{R}, 2>",
                             3: f"fc_cycle_small_kernel<{Rs}, 2>"}[form]
                persistent = (getattr(eng.ro, "sync_words", None) is not None
                              and os.environ.get("COEVO_PERSISTENT", "1") != "0"
                              and L.load().coevo_mpe_persistent_fits(int(hb[1] - hb[0]), int(lb[1] - lb[0]), eng.plan.heavy_max,
                                                                     eng.plan.light_max, Kc) == 1)
                if persistent and aggregate:
                    # ONE launch plays the whole rollout (coevo_mpe_rollout_persistent): a "launch" of the roofline block is
                    # one env-cycle of it = the rollout's span / its cycles (the per-cycle stamps - earliest start after the
                    # wait .. latest action posted - overlap from cycle to cycle and would overstate it)
                    kernel_id = f"fc_rollout_small_kernel<{Rs}>"
                    avg_ms = aggregate["rollout_span_ms"] / max(eng.ro._span_cycles, 1)
                    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
            else:
                kernel_id = f"fc_policy_kernel<{R}, 2>"
            traffic, traffic_note = None, None
            if pop == 200 and ctx.world == 1 and a.hof == 5 and a.max_cycles == 25:
                # HBM bytes per launch of this kernel from the PMC counters (FETCH_SIZE/WRITE_SIZE, separate rocprofv3
                # passes of this same command, gfx950 correction applied) - collected offline, see the file
                traffic, traffic_note = pmc_traffic("headline", kernel_id)
                pmc = os.path.join(REPO, "profiles", "r02_pmc_hbm_traffic.json")
                if traffic is None and os.path.exists(pmc):   # (last round's passes, same kernel)
                    with open(pmc) as f:
                        j = json.load(f)
                    if kernel_id in j.get("dominant_kernel", ""):
                        traffic = j["dominant_kernel_hbm_bytes_per_launch"]
                        traffic_note = "profiles/r02_pmc_hbm_traffic.json (rocprofv3 --pmc passes of this command)"
            if a.shard_of and kernel_id.startswith("fc_rollout_small") and pop == 200 and a.hof == 5 and a.max_cycles == 25:
                # the persistent launch plays the whole rollout: its PMC bytes / its cycles = one env-cycle, like avg_launch_ms
                t_all, note = pmc_traffic(f"cfg2_shard_1_of_{a.shard_of}", "fc_rollout_small_kernel")
                if t_all:
                    traffic, traffic_note = t_all / max(eng.ro._span_cycles, 1), note + " / the launch's env-cycles"
            kname = (kernel_id + " (one env-cycle of the persistent whole-rollout launch: every task's weight set streamed "
                     "once per cycle, fused env step, rows exchange tagged action words)"
                     if kernel_id.startswith("fc_rollout_small") else
                     kernel_id +
                     " (one env-cycle of one cohort: per-individual weight sets streamed once + shared-opponent tasks "
                     "on the matrix cores, fused env step)" if merged else
                     kernel_id + " (per-individual weight sets, fused env step)")
            out["roofline"] = {"bound": "hbm", "kernel": kname,
                               "timing": ("in-kernel 100 MHz clock stamps: span of the rollout (first cycle's earliest start .. last "
                                          "cycle's latest action) / cycles" if kernel_id.startswith("fc_rollout_small") else
                                          "HIP events around each launch on its stream" if a.no_graph else
                                          "in-kernel 100 MHz clock stamps, first workgroup start to last workgroup "
                                          "end (HIP events cannot be read back from replayed hipGraphs; "
                                          "--no-graph uses events)"),
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                               "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                               "launches_timed": len(d), "concurrent_launches": K, "rollout_aggregate": aggregate}
            # SURVEY 8d's per-generation form: every distinct weight set that acts, once per env-cycle, over the whole
            # generation (selection, breeding and the perturb kernel's own 0.67 GB of writes are not in the numerator)
            gen_bytes = cycle_bytes * eng.n_cycles
            out["roofline"]["generation"] = {"algorithmic_bytes": gen_bytes, "achieved": gen_bytes * gens_per_s / 1e9,
                                             "frac": gen_bytes * gens_per_s / 1e9 / HBM_PEAK_GBS}
        if not a.no_cpu_baseline and ctx.world == 1:
            out["cpu_baseline"] = cpu_baseline(pop, a.hof, a.limit)
            try:
                workers = host_cores()["usable"]   # one process per core this box gives us (count in `cores`)
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(pop, a.hof, a.limit, workers=min(workers, 128))
            except Exception as e:  # the single-core figure above is the contractual one
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
    del tr
    return out


if __name__ == "__main__":
    main()
