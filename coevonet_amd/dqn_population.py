"""Co-GA / Co-ES over DeepQN policies (BASELINE configs 4 and 5) on the MI355X.

The reference's Atari loop does not run (five independent TypeErrors) and ALE is not in the image, so
this is the BUILD's definition of it: the two-role restriction of the generation bodies of genetic_algorithm.py:119-290
and evolutionary_strategy.py:222-265 (roles first_0 / second_0 of pettingzoo.atari's two-player games; the reference's
three sigma flags: first_0 takes agent_0's part, second_0 agent_1's, quirk Q5 included) with play_atari's episode loop
(utils/game_logic_functions.py:84-119) over the synthetic env of ``atari_synthetic`` / ``coevo_synth_step``.  Quirks kept:
only the last HoF game counts (Q2), diversity against the stale agent (Q3), argsort()[::-1] elites (Q13, stable),
play_atari crediting the actor with the NEXT agent's cumulative reward.  "Loop parity unpinned, forward pinned" (SURVEY
8c): ``DeepQN.forward`` is pinned by the reference's logits, the loops are checked against ``oracle/ref_port.py``.

Per agent-step every game advances together: one ``coevo_dqn_out_synth_step`` launch (output layer + first-max action of
the previous step, books it, writes the next frames) + the conv-stack and fc1 launches of
``coevo_dqn_forward_hidden_timed`` over (weight set x frames) tasks: an individual's HoF games
share its 6.75 MB weight read, a HoF / base opponent's games are cut into 16-frame tasks.  Offspring are built on the
device from counter-based noise (``coevo_dqn_perturb``), selection / sigma rule / HoF stay on the device; one GPU replays
a whole generation as one hipGraph.  Sharded over GPUs by population index with one all-gather of (last-game reward,
distance) per individual (Co-GA: elites rebuilt from noise, no weight crosses xGMI) or of rewards + chunk partial sums
(Co-ES), as for the MPE engines.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import lib as L
from .atari_synthetic import SYNTH_SEED
from .deepqn import DeepQN
from .genetic_algorithm import N_EVAL, adapt_mutation_power

ROLES2 = ("first_0", "second_0")
SIGMA2 = ("mutation_power_agent_0", "mutation_power_agent_1")
TASK_ROWS = L.DQN_MAX_ROWS
FRAME = 84 * 84


class SynthRollout:
    """A batch of synthetic-env games: game g seats net ``game_nets[g][0]`` as first_0 and ``[g][1]`` as second_0.
    Per parity (who acts) the rows are grouped by acting net into tasks of <= 16 frames.

    ``bounds`` cuts the games into contiguous cohorts (default: one).  Games are independent, so each cohort runs its own
    chain of per-step launches on its own stream: one cohort's conv stack (matrix cores) overlaps the other's fc1 weight
    stream (HBM) instead of the whole chip alternating between the two.  Eager enqueue only (no graph capture)."""

    def __init__(self, game_nets, net_off, ordinal0, C, n_actions, slab, env_seed, ordinals_per_gen, device="cuda",
                 bounds=None, fc1_tiled=False):
        self.n_games = n = int(len(game_nets))
        self.C, self.n_actions, self.slab, self.env_seed = C, n_actions, slab, int(env_seed)
        self.Cw = C | (L.DQN_FC1_TILED if fc1_tiled else 0)   # channel argument of the forward launches: + the slab's fc1 layout
        self.ordinals_per_gen = int(ordinals_per_gen)
        self.device = device
        game_nets = np.asarray(game_nets, dtype=np.int64).reshape(n, 2)
        bounds = [0, n] if bounds is None else [int(x) for x in bounds]
        assert bounds[0] == 0 and bounds[-1] == n and all(b1 > b0 for b0, b1 in zip(bounds, bounds[1:]))
        self.ordinal0 = torch.from_numpy(np.asarray(ordinal0, dtype=np.int64)).to(device)
        self.gstate = torch.zeros(n, 4, dtype=torch.int32, device=device)
        self.acc = torch.zeros(n, 3, dtype=torch.float64, device=device)   # play_game returns (first_0, second_0), 0
        self.limit = torch.zeros(n, dtype=torch.int32, device=device)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self.lanes = []
        for g0, g1 in zip(bounds, bounds[1:]):
            m = g1 - g0
            lane = {"g0": g0, "n": m, "tasks": [], "tasks_np": [], "rows": [], "n_tasks": [], "max_rows": []}
            for parity in range(2):
                by_net = {}
                for g in range(g0, g1):
                    by_net.setdefault(int(game_nets[g, parity]), []).append(g - g0)
                tasks, row_of_game = [], np.zeros(m, dtype=np.int32)
                row = 0
                for net, games in by_net.items():
                    for i in range(0, len(games), TASK_ROWS):
                        chunk = games[i:i + TASK_ROWS]
                        tasks.append((int(net_off[net]), row, len(chunk)))
                        for g in chunk:
                            row_of_game[g] = row
                            row += 1
                t_np = np.array(tasks, dtype=L.DQN_TASK_DTYPE)
                lane["tasks_np"].append(t_np)
                lane["tasks"].append(L.tasks_to_device(t_np, device))
                lane["rows"].append(torch.from_numpy(row_of_game).to(device))
                lane["n_tasks"].append(len(tasks))
                lane["max_rows"].append(int(max(t[2] for t in tasks)))
            lane["frames"] = torch.zeros(m * FRAME * C, dtype=torch.uint8, device=device)
            lane["actions"] = torch.zeros(2, m, dtype=torch.int32, device=device)
            lane["ws"] = torch.zeros(int(L.load().coevo_dqn_workspace_bytes(m)) // 4, dtype=torch.float32, device=device)
            lane["stream"] = None
            lane["done"] = torch.cuda.Event() if self.lanes else None
            self.lanes.append(lane)
        if len(self.lanes) > 1:
            # the cohorts' own streams: created by the library with hipStreamNonBlocking (streams from torch's pool landed
            # on the caller's hardware queue here - the kernel trace showed both cohorts' launches back to back on it)
            self._lane_ctx = L.load().coevo_rollout_ctx_create(0)
            L._check(L.load().coevo_rollout_ctx_reserve_cohorts(self._lane_ctx, len(self.lanes)), "coevo_rollout_ctx_reserve_cohorts")
            for k in range(1, len(self.lanes)):
                self.lanes[k]["stream"] = torch.cuda.ExternalStream(
                    L.load().coevo_rollout_ctx_cohort_stream(self._lane_ctx, k), device=device)
        else:
            self._lane_ctx = None
        self.tasks_np = [np.concatenate([ln["tasks_np"][p] for ln in self.lanes]) for p in range(2)]
        self._fork = torch.cuda.Event() if len(self.lanes) > 1 else None
        self.timing_ctx = None      # set by start_timing(): HIP events around sampled launches (eager only)
        self.timing_every = 1

    def start_timing(self, pairs=512, every=7):
        """sample the conv-stack launch (and, on other steps, the fc1 launch) of the first cohort with HIP events"""
        self._destroy_timing()
        self.timing_ctx = [L.load().coevo_rollout_ctx_create(int(pairs)) for _ in range(2)]
        self.timing_every = int(every)

    def _destroy_timing(self):
        for c in self.timing_ctx or []:
            L.load().coevo_rollout_ctx_destroy(c)
        self.timing_ctx = None

    def close(self):
        """gives the cohort streams and the timing events back (library contexts); idempotent"""
        if getattr(self, "_lane_ctx", None) or getattr(self, "timing_ctx", None):
            if torch.cuda.is_available():
                torch.cuda.synchronize()   # nothing may still run on a stream that is about to be destroyed
            self._destroy_timing()
            if getattr(self, "_lane_ctx", None):
                for ln in self.lanes[1:]:
                    ln["stream"] = None
                L.load().coevo_rollout_ctx_destroy(self._lane_ctx)
                self._lane_ctx = None

    def __del__(self):
        # No device-wide synchronize from a finaliser: the collector may run it while ANOTHER engine is capturing a hipGraph
        # (bench.py builds many engines in one process), and a device sync invalidates the capture.  Owners call close()
        # (DQNGATrainer.close / DQNESTrainer.close); a rollout dropped without it only gives its contexts back when nothing
        # is capturing, after waiting for its OWN streams.
        try:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            for ln in getattr(self, "lanes", [])[1:]:
                if ln.get("stream") is not None:
                    ln["stream"].synchronize()
            self._destroy_timing()
            if getattr(self, "_lane_ctx", None):
                L.load().coevo_rollout_ctx_destroy(self._lane_ctx)
                self._lane_ctx = None
        except Exception:
            pass

    def reset_timing(self):
        for c in self.timing_ctx or []:
            L.load().coevo_rollout_ctx_reset_timing(c)

    def times_ms(self, which, max_out=100000):
        """durations of the sampled launches: which = 0 conv stack, 1 fc1"""
        if not self.timing_ctx:
            return []
        buf = (L.C.c_float * max_out)()
        n = L.load().coevo_rollout_ctx_light_times(self.timing_ctx[which], buf, max_out)
        return [buf[i] for i in range(max(n, 0))]

    def conv_times_ms(self):
        return self.times_ms(0)

    def set_limits(self, limits):
        self.limit.copy_(torch.from_numpy(np.asarray(limits, dtype=np.int32)))

    def _enqueue_lane(self, ln, T, g, stream, timed):
        g0, m = ln["g0"], ln["n"]
        gs, ac = self.gstate.data_ptr() + 16 * g0, self.acc.data_ptr() + 24 * g0
        o0, lim = self.ordinal0.data_ptr() + 8 * g0, self.limit.data_ptr() + 4 * g0
        lib = L.load()
        for t in range(T + 1):
            p, q = t & 1, (t - 1) & 1
            if t == 0:
                L._check(lib.coevo_synth_step(gs, ac, m, o0, g, self.ordinals_per_gen, t, lim, None, None,
                                              L._p(ln["rows"][p]), L._p(ln["frames"]), self.C, self.n_actions,
                                              self.env_seed, stream), "coevo_synth_step")
            else:   # the output layer of step t-1 rides in the env-step launch
                L._check(lib.coevo_dqn_out_synth_step(gs, ac, m, o0, g, self.ordinals_per_gen, t, lim, L._p(ln["rows"][q]),
                                                      ln["actions"][q].data_ptr(),
                                                      L._p(ln["rows"][p]) if t < T else None,
                                                      L._p(ln["frames"]) if t < T else None, self.C, self.n_actions,
                                                      self.env_seed, L._p(self.slab), L._p(ln["tasks"][q]),
                                                      ln["n_tasks"][q], m, L._p(ln["ws"]), L._p(self.status), stream),
                         "coevo_dqn_out_synth_step")
            if t < T:
                tc, which = None, 0
                # (HIP events cannot be recorded inside a capture: timing is for eager enqueues only)
                if timed and self.timing_ctx and t % self.timing_every == 0 and not torch.cuda.is_current_stream_capturing():
                    which = (t // self.timing_every) & 1
                    tc = self.timing_ctx[which]
                L._check(lib.coevo_dqn_forward_hidden_timed(L._p(self.slab), L._p(ln["tasks"][p]), ln["n_tasks"][p],
                                                            ln["max_rows"][p], m, self.Cw, self.n_actions,
                                                            L._p(ln["frames"]), L._p(ln["ws"]), tc, which, stream),
                         "coevo_dqn_forward_hidden_timed")

    def enqueue(self, T, gen_dev):
        """T agent-steps of every game + the closing bookkeeping call; cohort 0 on the current stream, the others on
        their own streams (forked from / joined to the current one)"""
        g = L._p(gen_dev) if gen_dev is not None else None
        main = torch.cuda.current_stream()
        if len(self.lanes) > 1:
            self._fork.record(main)
        for k, ln in reversed(list(enumerate(self.lanes))):   # the caller's stream last
            if k:
                ln["stream"].wait_event(self._fork)
                self._enqueue_lane(ln, T, g, ln["stream"].cuda_stream, False)
                ln["done"].record(ln["stream"])
            else:
                self._enqueue_lane(ln, T, g, main.cuda_stream, True)
        for ln in self.lanes[1:]:
            main.wait_event(ln["done"])

    def weight_bytes_per_round(self):
        """algorithmic bytes of one round (two agent-steps): every distinct acting weight set once per agent-step +
        the frames (SURVEY 8d cfg 4/5 model)"""
        P4 = int(L.load().coevo_dqn_param_count(self.C, self.n_actions)) * 4
        nets = sum(len({int(t["net_off"]) for t in tn}) for tn in self.tasks_np)
        return nets * P4 + 2 * self.n_games * FRAME * self.C

    def distinct_nets_per_step(self, parity, lane=0):
        return len({int(t["net_off"]) for t in self.lanes[lane]["tasks_np"][parity]})


class HostFrameRollout(SynthRollout):
    """The same games with the synthetic env on the HOST (PCIe-inclusive: what an ALE env in host memory imposes,
    utils/game_logic_functions.py:47-53,84-120): per agent-step and cohort the host cores book the previous action and render
    the next frames into page-locked memory (the bytes coevo_synth_step writes on the device, so every reward is identical),
    the cohort's stream copies them up (28 224 B per game and step at C = 4), runs conv stack + fc1 + output layer and copies
    the actions down; the cohorts alternate.  One blocking C-ABI call per rollout (coevo_dqn_host_frames_rollout); the
    bookkeeping (hits, fp64 returns) lives in host memory and is mirrored into the device tensors the engines read."""

    def __init__(self, *a, threads=None, **k):
        super().__init__(*a, **k)
        n, K = self.n_games, len(self.lanes)
        if threads is None:
            # (the cores this thread may run on, not the machine's: a cgroup / taskset-limited rank gets its own share)
            threads = int(os.environ.get("COEVO_FRAME_THREADS", "0")) or min(16, len(os.sched_getaffinity(0)) or 1)
        self.h_gstate = np.zeros((n, 4), dtype=np.int32)
        self.h_acc = np.zeros((n, 3), dtype=np.float64)
        self.h_ordinal0 = np.ascontiguousarray(self.ordinal0.cpu().numpy(), dtype=np.int64)
        self.h_limit = np.zeros(n, dtype=np.int32)
        self.frame_ctx = L.load().coevo_host_rollout_create(int(threads), K)
        if not self.frame_ctx:
            raise L.CoevoError("coevo_host_rollout_create failed")
        self.threads = int(L.load().coevo_host_rollout_threads(self.frame_ctx))
        self.placement = L.host_placement(self.frame_ctx)
        self.phase_us = None          # a float64[5] array to collect the per-cohort-step breakdown
        self._cohorts = (L.FrameCohort * K)()
        self._keep = []
        for k_, ln in enumerate(self.lanes):
            m = ln["n"]
            # page-locked, first touched on the NUMA node the context's cores run on (the GPU's: csrc/host_placement.hip)
            fh = L.host_tensor(self.frame_ctx, (m * FRAME * self.C,), np.uint8)
            ah = L.host_tensor(self.frame_ctx, (m,), np.int32)
            rows = [np.ascontiguousarray(ln["rows"][p].cpu().numpy(), dtype=np.int32) for p in range(2)]
            self._keep.append((fh, ah, rows))
            c = self._cohorts[k_]
            for p in range(2):
                c.tasks[p] = ln["tasks"][p].data_ptr()
                c.rows[p] = rows[p].ctypes.data
                c.n_tasks[p] = ln["n_tasks"][p]
                c.max_rows[p] = ln["max_rows"][p]
            c.frames_host, c.frames_dev = fh.data_ptr(), ln["frames"].data_ptr()
            c.actions_host, c.actions_dev = ah.data_ptr(), ln["actions"][0].data_ptr()
            c.workspace = ln["ws"].data_ptr()
            c.game_first, c.n_games = ln["g0"], m

    def set_limits(self, limits):
        super().set_limits(limits)
        self.h_limit[:] = np.asarray(limits, dtype=np.int32)

    def enqueue(self, T, gen_dev):
        """blocking: the whole rollout runs inside the C call; afterwards the device copies of the books are current"""
        gen = int(gen_dev.item()) if gen_dev is not None else 0
        d = L.FramesRolloutDesc(
            slab=L._p(self.slab), status=L._p(self.status), game_state=self.h_gstate.ctypes.data, acc=self.h_acc.ctypes.data,
            game_ordinal0=self.h_ordinal0.ctypes.data, limit=self.h_limit.ctypes.data,
            cohorts=L.C.cast(self._cohorts, L.C.c_void_p),
            phase_us=(self.phase_us.ctypes.data if self.phase_us is not None else None), generation=gen,
            ordinals_per_gen=self.ordinals_per_gen, seed=self.env_seed, n_games=self.n_games, n_cohorts=len(self.lanes),
            C=self.Cw, n_actions=self.n_actions, T=int(T), reserved=0)
        L._check(L.load().coevo_dqn_host_frames_rollout(self.frame_ctx, L.C.byref(d), L._stream()),
                 "coevo_dqn_host_frames_rollout")
        self.acc.copy_(torch.from_numpy(self.h_acc))
        self.gstate.copy_(torch.from_numpy(self.h_gstate))

    def close(self):
        super().close()
        if getattr(self, "frame_ctx", None):
            self._keep = []   # (the page-locked frame / action buffers belong to the context)
            L.load().coevo_host_rollout_destroy(self.frame_ctx)
            self.frame_ctx = None

    def __del__(self):
        # the destroy releases streams, events and page-locked memory (hipHostFree synchronises the device): not from a
        # finaliser that runs while another engine captures a hipGraph - the context then lives until close() / exit
        try:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            if getattr(self, "frame_ctx", None):
                self._keep = []
                L.load().coevo_host_rollout_destroy(self.frame_ctx)
                self.frame_ctx = None
        except Exception:
            pass
        super().__del__()


def frames_mode(args=None):
    """"device" (frames synthesised in HBM: the measured form of cfg 4 / cfg 5) or "host" (args.coevo_frames /
    COEVO_DQN_FRAMES): the env in host memory, frames over PCIe every agent-step"""
    m = (getattr(args, "coevo_frames", None) if args is not None else None) or os.environ.get("COEVO_DQN_FRAMES", "device")
    if m not in ("device", "host"):
        raise ValueError(f"frames mode {m!r}: 'device' or 'host'")
    return m


def even_bounds(n, K):
    K = max(1, min(int(K), n))
    return [k * n // K for k in range(K + 1)]



def dqn_init_flat(C, n_actions):
    """one randomly initialised DeepQN in canonical flat order (consumes the torch generator like DeepQN.__init__)"""
    return DeepQN(C, n_actions, "float32").flat().copy()


class _SlabMixin:
    def _ptr(self, role, region, i=0):
        return self.slab.data_ptr() + 4 * (self.base[role][region] + i * self.stride)

    def upload(self, role, region, first, flat_np):
        flat = torch.from_numpy(np.ascontiguousarray(flat_np, dtype=np.float32)).to(self.device)
        L.call("coevo_dqn_pack", L._p(flat), self._ptr(role, region, first), flat.shape[0], self.Cw, self.n_actions)
        torch.cuda.current_stream().synchronize()

    def download(self, role, region, first, n):
        out = torch.zeros(n, self.P, dtype=torch.float32, device=self.device)
        L.call("coevo_dqn_unpack", self._ptr(role, region, first), L._p(out), n, self.Cw, self.n_actions)
        return out.cpu().numpy()


class DQNGAEngine(_SlabMixin):
    def __init__(self, pop, hof, elites, C, n_actions, T_train, T_eval, device="cuda", env_seed=SYNTH_SEED,
                 philox_seed=0, shard=(0, 1), gather=None, first_ordinal=1, capacity=1024, sigmas=(0.05, 0.05),
                 sig_min=0.001, sig_max=0.2, adaptive=True, frames="device"):
        assert 1 <= elites <= pop and hof >= 1 and frames in ("device", "host")
        self.frames = frames
        self.pop, self.hof, self.E, self.C, self.n_actions = pop, hof, elites, C, n_actions
        # the engine's slab keeps fc1 TILED for v_mfma_f32_16x16x4 (its tasks carry 10 / 16 frames; include/coevo.h
        # COEVO_DQN_FC1_TILED; COEVO_DQN_FC1_LAYOUT=streamed keeps the Co-ES layout for A/B runs): every layout-dependent
        # call takes self.Cw
        self.fc1_tiled = os.environ.get("COEVO_DQN_FC1_LAYOUT", "tiled") != "streamed"
        self.Cw = C | (L.DQN_FC1_TILED if self.fc1_tiled else 0)
        self.T_train, self.T_eval = int(T_train), int(T_eval)
        self.T = max(self.T_train, self.T_eval)
        self.device, self.philox_seed = device, int(philox_seed)
        self.rank, self.world = shard
        self.gather = gather
        if self.world > 1 and pop % self.world:
            raise ValueError(f"population {pop} is not divisible by the number of ranks {self.world}")
        self.lo, self.hi = self.rank * pop // self.world, (self.rank + 1) * pop // self.world
        self.n_local = self.hi - self.lo
        lib = L.load()
        self.stride = int(lib.coevo_dqn_slab_stride(C, n_actions))
        self.P = int(lib.coevo_dqn_param_count(C, n_actions))
        self.base, off = {}, 0
        for r in ROLES2:
            self.base[r] = {}
            for region, count in (("pop", pop), ("hof", hof), ("elite", elites), ("stale", 1), ("hof_tmp", hof),
                                  ("elite_prev", elites)):
                self.base[r][region] = off
                off += count * self.stride
        self.slab = torch.zeros(off, dtype=torch.float32, device=device)
        # ---- games of one generation launch (this rank's individuals) + the evaluation games of the previous one
        net_off, ids = [], {}

        def net(region, role, i):
            key = (region, role, i)
            if key not in ids:
                ids[key] = len(net_off)
                net_off.append(self.base[role][region] + i * self.stride)
            return ids[key]

        h, M = hof, 2 * pop * hof
        self.per_gen = M + N_EVAL
        games, ordinal0 = [], []
        for ph, role in enumerate(ROLES2):
            for i in range(self.lo, self.hi):
                for k in range(h):
                    opp = net("hof", ROLES2[1 - ph], h - 1 - k)
                    games.append((net("pop", role, i), opp) if ph == 0 else (opp, net("pop", role, i)))
                    ordinal0.append(first_ordinal + ph * pop * hof + i * hof + k)
        self.n_main = len(games)
        for j in range(N_EVAL):  # the best pair = the newest HoF members; generation g-1's games ride in g's launch
            games.append((net("hof", "first_0", h - 1), net("hof", "second_0", h - 1)))
            ordinal0.append(first_ordinal - self.per_gen + M + j)
        # optional (COEVO_DQN_COHORTS=2): two cohorts = the two role phases (contiguous game ranges; the evaluation games
        # go with the second) on two streams.  Measured on the cfg 4 shard: 12.2 vs 12.6 generations/s for one chain - conv
        # stack and fc1 both live on the matrix pipe at these row counts, there is nothing complementary to overlap
        K = int(os.environ.get("COEVO_DQN_COHORTS", "1"))
        bounds = [0, self.n_main // 2, len(games)] if (K > 1 and self.n_main >= 2) else None
        if frames == "host":   # env in host memory: the cohorts alternate between the host cores and the GPU
            bounds = even_bounds(len(games), int(os.environ.get("COEVO_FRAME_COHORTS", "3")))
        self.ro = (HostFrameRollout if frames == "host" else SynthRollout)(
            games, net_off, ordinal0, C, n_actions, self.slab, env_seed, self.per_gen, device, bounds=bounds,
            fc1_tiled=self.fc1_tiled)
        self.cohorts = len(self.ro.lanes)
        # ---- device-resident loop state ------------------------------------------------------------------------
        f32 = dict(dtype=torch.float32, device=device)
        i32 = dict(dtype=torch.int32, device=device)
        self.gen_dev = torch.zeros(1, **i32)
        self.sigma64 = torch.tensor([sigmas[0], sigmas[1], 0.0], dtype=torch.float64, device=device)
        self.sigma32 = self.sigma64.to(torch.float32)
        self.sigma32_prev = self.sigma32.clone()
        self.cap = int(capacity)
        self.hist = torch.zeros(3, self.cap, dtype=torch.float64, device=device)
        self.sig_hist = torch.zeros(3, self.cap, dtype=torch.float64, device=device)
        self.loop_args = (float(sig_min), float(sig_max), 1 if adaptive else 0)
        self.dist_all = torch.zeros(2, pop, **f32)
        self.div = [torch.zeros(1, **f32) for _ in ROLES2]
        self.fitness = [torch.zeros(pop, **f32) for _ in ROLES2]
        self.order = [torch.zeros(pop, **i32) for _ in ROLES2]
        self.best_dist = [torch.zeros(1, **f32) for _ in ROLES2]
        self.last_reward = torch.zeros(2, pop, 3, dtype=torch.float64, device=device)
        self.pblocks = int(lib.coevo_dqn_perturb_blocks(self.Cw, n_actions))
        self.dist_partial = torch.zeros(max(pop, 1) * self.pblocks, dtype=torch.float64, device=device)
        self.parent_idx = torch.tensor([c % elites for c in range(max(pop - 1, 1))], **i32)
        self.iota = torch.arange(max(pop, hof, elites, 2), **i32)
        self.hof_shift_idx = torch.arange(1, max(hof, 2), **i32)
        one = torch.arange(self.n_local, device=device) * hof + hof - 1
        self.last_game_idx = torch.cat([ph * self.n_local * hof + one for ph in range(2)])
        self._graph = None
        self.generation = 0
        self.steps_per_generation = 2 * pop * hof * self.T_train + N_EVAL * self.T_eval

    def load_initial(self, pop_flat, hof_flat):
        for r in ROLES2:
            self.upload(r, "pop", 0, pop_flat[r])
            self.upload(r, "hof", 0, hof_flat[r])
            self.upload(r, "stale", 0, pop_flat[r][self.pop - 1:self.pop])   # Q3: the object left over from the init loop
        for ri, r in enumerate(ROLES2):  # distances of the initial population to the stale agent (later: fused into breeding)
            L.call("coevo_dqn_perturb", self._ptr(r, "pop"), L._p(self.iota), None, 0, self.pop, self.Cw, self.n_actions,
                   None, 0, 0, 0, 8, 1, None, 0, self._ptr(r, "stale"), L._p(self.dist_partial))
            L.call("coevo_fc_distance_finalize", L._p(self.dist_partial), self.pblocks, self.pop,
                   self.dist_all[ri].data_ptr(), 0, None)
        torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------------------------------------ one generation
    def _tail(self, gen):
        """selection -> sigma rule -> elites / HoF / best -> this rank's children -> generation counter tick"""
        ro, g = self.ro, L._p(self.gen_dev)
        sharded = self.world > 1
        if sharded:  # last HoF game of every local individual (Q2) + its distance -> every rank, one all-gather
            self.last_reward[:, self.lo:self.hi] = ro.acc[self.last_game_idx].view(2, self.n_local, 3)
            self.gather(self)
        roles = (L.GaSelectRole * 3)()
        for ri in range(2):
            roles[ri] = L.GaSelectRole(self.dist_all[ri].data_ptr(),
                                       self.last_reward[ri].data_ptr() if sharded else L._p(ro.acc),
                                       L._p(self.div[ri]), L._p(self.fitness[ri]), L._p(self.order[ri]),
                                       L._p(self.best_dist[ri]), 0 if sharded else ri * self.pop * self.hof, ri)
        L.call("coevo_ga_select", roles, 2, self.pop, 1 if sharded else self.hof, self.hof)
        mn, mx, adaptive = self.loop_args
        self.sigma32_prev.copy_(self.sigma32)
        L.call("coevo_ga_adapt_sigma", L._p(ro.acc), self.n_main, g, L._p(self.hist), L._p(self.sig_hist), self.cap,
               L._p(self.sigma64), L._p(self.sigma32), mn, mx, adaptive)
        c_lo, c_hi = (max(self.lo, 1) - 1, self.hi - 1) if sharded else (0, self.pop - 1)  # child c = individual c + 1
        for ri, r in enumerate(ROLES2):
            if sharded and gen > 0:
                # a rank holds only its own children: rebuild the new elites from last generation's elites + noise
                L.call("coevo_net_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "elite_prev"), 0, self.E,
                       self.stride)
                L.call("coevo_dqn_perturb", self._ptr(r, "elite_prev"), L._p(self.order[ri]), self._ptr(r, "elite"), 0,
                       self.E, self.Cw, self.n_actions, self.sigma32_prev.data_ptr() + 4 * ri, self.philox_seed, 0, ri, 4,
                       self.E, g, -1, None, None)
            else:
                L.call("coevo_net_gather", self._ptr(r, "pop"), L._p(self.order[ri]), self._ptr(r, "elite"), 0, self.E,
                       self.stride)
            if self.hof > 1:  # hof.pop(0); hof.append(best)
                L.call("coevo_net_gather", self._ptr(r, "hof"), L._p(self.hof_shift_idx), self._ptr(r, "hof_tmp"), 0,
                       self.hof - 1, self.stride)
                L.call("coevo_net_gather", self._ptr(r, "hof_tmp"), L._p(self.iota), self._ptr(r, "hof"), 0,
                       self.hof - 1, self.stride)
            L.call("coevo_net_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "hof"), self.hof - 1, 1,
                   self.stride)
            if self.lo == 0:
                L.call("coevo_net_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "pop"), 0, 1, self.stride)
            if c_hi > c_lo:
                L.call("coevo_dqn_perturb", self._ptr(r, "elite"), self.parent_idx.data_ptr() + 4 * c_lo,
                       self._ptr(r, "pop"), 1 + c_lo, c_hi - c_lo, self.Cw, self.n_actions,
                       self.sigma32.data_ptr() + 4 * ri, self.philox_seed, c_lo, ri, 0, self.E, g, 0,
                       self._ptr(r, "stale"), L._p(self.dist_partial))
                L.call("coevo_fc_distance_finalize", L._p(self.dist_partial), self.pblocks, c_hi - c_lo,
                       self.dist_all[ri].data_ptr(), 1 + c_lo, L._p(self.best_dist[ri]) if c_lo == 0 else None)
            elif self.lo == 0:
                self.dist_all[ri][0:1].copy_(self.best_dist[ri])
        L.call("coevo_counter_add", g, 1)

    def step(self, use_graph=True):
        gen = self.generation
        if gen >= self.cap:
            raise RuntimeError(f"generation {gen} exceeds the device history capacity ({self.cap})")
        if gen <= 1:  # the evaluation games of "generation -1" do not exist: disabled in generation 0 only
            limits = np.full(self.ro.n_games, self.T_train, dtype=np.int32)
            limits[self.n_main:] = self.T_eval if gen == 1 else 0
            self.ro.set_limits(limits)
        if self.world == 1 and use_graph and self.cohorts == 1 and self.frames == "device":   # (cohort chains: eager)
            if self._graph is None:
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    self.ro.enqueue(self.T, self.gen_dev)
                    self._tail(1)   # (gen only matters to the sharded path)
                self._graph = gr
            self._graph.replay()
        else:
            self.ro.enqueue(self.T, self.gen_dev)
            self._tail(gen)
        self.generation += 1

    def eval_only(self):
        """the evaluation games of the last generation (they would ride in the next one): main games disabled"""
        ro = self.ro
        limits = np.zeros(ro.n_games, dtype=np.int32)
        limits[self.n_main:] = self.T_eval
        ro.set_limits(limits)
        ro.enqueue(self.T_eval, self.gen_dev)
        torch.cuda.synchronize()
        L.raise_on_status(ro.status)
        r = ro.acc[self.n_main:].cpu().numpy()
        tot = [0.0, 0.0]
        for j in range(N_EVAL):
            for s in range(2):
                tot[s] += float(r[j, s])
        limits[:self.n_main] = self.T_train
        ro.set_limits(limits)
        return [t / 10 for t in tot]


class DQNResult:
    def __init__(self):
        self.rewards = {r: [] for r in ROLES2}
        self.fitness, self.elite_ids, self.diversity, self.sigma_after, self.game_rewards, self.seconds = [], [], [], [], [], []


def dqn_initial_population(pop, hof, C, n_actions):
    """creation order of the 2-role restriction: HoF second_0, HoF first_0, then the population interleaved"""
    hof_flat = {"second_0": np.stack([dqn_init_flat(C, n_actions) for _ in range(hof)])}
    hof_flat["first_0"] = np.stack([dqn_init_flat(C, n_actions) for _ in range(hof)])
    popu = {r: [] for r in ROLES2}
    for _ in range(pop):
        for r in ROLES2:
            popu[r].append(dqn_init_flat(C, n_actions))
    return {r: np.stack(popu[r]) for r in ROLES2}, hof_flat


def _env_shape(env, args):
    C = int(getattr(env, "C", None) or env.observation_space(env.agents[0]).shape[-1])
    n = int(getattr(env, "n_actions", None) or env.action_space(env.agents[0]).n)
    return C, n


class DQNGATrainer:
    def __init__(self, env, args, collect=True, dist_ctx=None):
        self.env, self.args, self.collect = env, args, collect
        C, n = _env_shape(env, args)
        pop_flat, hof_flat = dqn_initial_population(args.population, args.hof_size, C, n)
        shard, gather = (0, 1), None
        if dist_ctx is not None and dist_ctx.world > 1:
            shard, gather = (dist_ctx.rank, dist_ctx.world), dist_ctx.gather_ga2
        self.first_ordinal = getattr(env, "n_resets", 1)
        self.eng = DQNGAEngine(args.population, args.hof_size, args.elites_number, C, n,
                               args.max_timesteps_per_episode, args.max_evaluation_steps,
                               env_seed=getattr(env, "seed_value", None) or SYNTH_SEED,
                               philox_seed=getattr(args, "coevo_seed", 0), shard=shard, gather=gather,
                               first_ordinal=self.first_ordinal, capacity=max(getattr(args, "generations", 0), 1) + 64,
                               sigmas=(args.mutation_power_agent_0, args.mutation_power_agent_1),
                               sig_min=args.min_mutation_power, sig_max=args.max_mutation_power, adaptive=args.adaptive,
                               frames=frames_mode(args))
        self.eng.load_initial(pop_flat, hof_flat)
        self.res = DQNResult()
        self.res.engine = self.eng
        self.gen = 0

    def step(self):
        eng, res = self.eng, self.res
        t0 = time.perf_counter()
        eng.step(use_graph=getattr(self.args, "coevo_graph", True))
        if self.collect:
            torch.cuda.synchronize()
            L.raise_on_status(eng.ro.status)
            res.game_rewards.append(eng.ro.acc[:eng.n_main, :2].cpu().numpy().copy())
            res.fitness.append([eng.fitness[ri].cpu().numpy().tolist() for ri in range(2)])
            res.diversity.append([float(eng.div[ri].item()) for ri in range(2)])
            res.elite_ids.append([eng.order[ri][:eng.E].cpu().numpy().astype(int).tolist() for ri in range(2)])
        res.seconds.append(time.perf_counter() - t0)
        self.gen += 1

    def finish(self):
        eng, res, args = self.eng, self.res, self.args
        if self.gen > 0:
            torch.cuda.synchronize()
            L.raise_on_status(eng.ro.status)
            upto = self.gen - 1
            hist = eng.hist[:, :upto].cpu().numpy()
            sig = eng.sig_hist[:, :upto].cpu().numpy()
            for s, r in enumerate(ROLES2):
                res.rewards[r] = [float(x) for x in hist[s]]
            res.sigma_after = [[float(sig[0, e]), float(sig[1, e])] for e in range(upto)]
            cur = eng.sigma64.cpu().numpy()
            args.mutation_power_agent_0, args.mutation_power_agent_1 = float(cur[0]), float(cur[1])
            ev = eng.eval_only()
            for s, r in enumerate(ROLES2):
                res.rewards[r].append(ev[s])
            if args.adaptive:  # the last generation's rule on the host (its evaluation never rode in a next launch)
                keep = getattr(args, "mutation_power_adversary", 0.0)
                h = {"agent_0": res.rewards["first_0"], "agent_1": res.rewards["second_0"],
                     "adversary_0": [0.0] * len(res.rewards["first_0"])}
                args.mutation_power_adversary = 0.0
                adapt_mutation_power(args, self.gen - 1, h)
                args.mutation_power_adversary = keep
            res.sigma_after.append([args.mutation_power_agent_0, args.mutation_power_agent_1])
        if hasattr(self.env, "n_resets"):
            self.env.n_resets = self.first_ordinal + self.gen * eng.per_gen
        return res

    def close(self):
        self.eng.ro.close()


def dqn_genetic_algorithm_train(env, agent, args, output_dir, collect=True, dist_ctx=None):
    """genetic_algorithm_train for the two-player Atari games (main.py:181 with --game pong_v3 / boxing_v2)"""
    from .io_utils import MetricsWriter
    tr = DQNGATrainer(env, args, collect=collect, dist_ctx=dist_ctx)
    for _ in range(args.generations):
        tr.step()
    res = tr.finish()
    mw = MetricsWriter(output_dir)
    for g in range(len(res.rewards["first_0"])):
        mw.write(generation=g, eval_rewards={r: res.rewards[r][g] for r in ROLES2},
                 mutation_power=res.sigma_after[g] if g < len(res.sigma_after) else None,
                 elite_ids=res.elite_ids[g] if g < len(res.elite_ids) else None, data="synthetic env")
    return res


# ---------------------------------------------------------------------------------------------------- Co-ES
def es_cohort_bounds(n_local, K):
    """game bounds of K rollout cohorts (None = one cohort): games are individual-major, two per individual; cohort k =
    individuals [k n / K, (k + 1) n / K)"""
    K = max(1, min(int(K), n_local))
    return [2 * (k * n_local // K) for k in range(K)] + [2 * n_local] if K > 1 else None


class DQNESEngine(_SlabMixin):
    def __init__(self, pop, C, n_actions, T_train, T_eval, device="cuda", env_seed=SYNTH_SEED, philox_seed=0,
                 shard=(0, 1), gather=None, first_ordinal=1, antithetic=False, centered_rank=False, chunks=8,
                 frames="device"):
        assert frames in ("device", "host")
        self.frames = frames
        self.pop, self.C, self.n_actions, self.device = pop, C, n_actions, device
        self.Cw = C   # the streamed fc1 layout (one frame per task: v_mfma_f32_4x4x1, lane = output)
        self.T_train, self.T_eval = int(T_train), int(T_eval)
        self.philox_seed = int(philox_seed)
        self.rank, self.world = shard
        self.gather = gather
        self.antithetic, self.centered_rank, self.chunks = bool(antithetic), bool(centered_rank), int(chunks)
        if self.world > 1 and (pop % self.world or self.chunks % self.world):
            raise ValueError(f"population {pop} and the {self.chunks} update chunks must both be divisible by the "
                             f"number of ranks {self.world}")
        if self.antithetic and pop % 2:
            raise ValueError("antithetic pairs need an even population")
        self.lo, self.hi = self.rank * pop // self.world, (self.rank + 1) * pop // self.world
        self.n_local = self.hi - self.lo
        lib = L.load()
        self.stride = int(lib.coevo_dqn_slab_stride(C, n_actions))
        self.P = int(lib.coevo_dqn_param_count(C, n_actions))
        self.base, off = {}, 0
        for r in ROLES2:
            self.base[r] = {"base": off, "pert": off + self.stride}
            off += (1 + self.n_local) * self.stride
        self.slab = torch.zeros(off, dtype=torch.float32, device=device)
        self.per_gen = 2 * pop + N_EVAL
        net_off = [self.base["first_0"]["base"], self.base["second_0"]["base"]]
        games, ordinal0 = [], []
        for j in range(self.n_local):
            for ri, r in enumerate(ROLES2):
                net_off.append(self.base[r]["pert"] + j * self.stride)
                me = len(net_off) - 1
                games.append((me, 1) if ri == 0 else (0, me))
                ordinal0.append(first_ordinal + 2 * (self.lo + j) + ri)
        self.n_main = len(games)
        # two cohorts by default: one-frame tasks make this rollout fc1-bound (6.4 MB of weights per frame), and one
        # cohort's conv launch (matrix pipe) then runs under the other's fc1 stream (HBM): cfg 5 shard 9.5 vs 9.1
        # generations/s.  (Co-GA's 10-frame tasks are conv-bound: one cohort is faster there, 16.4 vs 15.6.)
        bounds = es_cohort_bounds(self.n_local, int(os.environ.get("COEVO_DQN_COHORTS", "2")))
        cls = HostFrameRollout if frames == "host" else SynthRollout
        if frames == "host":
            bounds = even_bounds(len(games), int(os.environ.get("COEVO_FRAME_COHORTS", "3")))
            bounds = [b - (b & 1) for b in bounds[:-1]] + [bounds[-1]]   # an individual's two games stay in one cohort
        self.ro = cls(games, net_off, ordinal0, C, n_actions, self.slab, env_seed, self.per_gen, device, bounds=bounds)
        self.ro.set_limits(np.full(self.n_main, self.T_train, dtype=np.int32))
        # the ten evaluation games of the updated base nets (one 10-frame task per agent-step) play on TILED twins of the two
        # base nets (fc1 laid out for v_mfma_f32_16x16x4: the narrow fc1 launch then has no cross-lane operand moves, 25 ->
        # ~16 us of a 68 us evaluation step); the engine's own slab stays streamed for the one-frame tasks of the rollout.
        # COEVO_DQN_EVAL_TILED=0: evaluate on the slab itself (A/B)
        self.eval_tiled = os.environ.get("COEVO_DQN_EVAL_TILED", "1") != "0"
        self.eval_slab = torch.zeros(2 * self.stride, dtype=torch.float32, device=device) if self.eval_tiled else self.slab
        self.eval_ro = cls([(0, 1)] * N_EVAL, [0, self.stride] if self.eval_tiled else net_off[:2],
                           [first_ordinal + 2 * pop + j for j in range(N_EVAL)], C, n_actions, self.eval_slab, env_seed,
                           self.per_gen, device, fc1_tiled=self.eval_tiled)
        self.eval_ro.set_limits(np.full(N_EVAL, self.T_eval, dtype=np.int32))
        f32 = dict(dtype=torch.float32, device=device)
        self.gen_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.sigma = torch.zeros(2, **f32)
        self.zero_idx = torch.zeros(max(self.n_local, 1), dtype=torch.int32, device=device)
        self.fitness = [torch.zeros(pop, **f32) for _ in ROLES2]
        self.raw = [torch.zeros(pop, **f32) for _ in ROLES2]
        self.div = [torch.zeros(1, **f32) for _ in ROLES2]
        self.dist_local = torch.zeros(max(self.n_local, 1), **f32)
        self.stats = torch.zeros(2, pop, 2, dtype=torch.float64, device=device)
        self.pblocks = int(lib.coevo_dqn_perturb_blocks(C, n_actions))
        self.dist_partial = torch.zeros(max(self.n_local, 1) * self.pblocks, dtype=torch.float64, device=device)
        self.chunks_local = self.chunks // self.world
        self.part_off = {r: ri * self.chunks_local * self.stride for ri, r in enumerate(ROLES2)}
        self.part_block = 2 * self.chunks_local * self.stride
        self.partials = torch.zeros(self.world * self.part_block, **f32)
        self.game_idx = torch.stack([torch.arange(self.n_local, device=device) * 2 + ri for ri in range(2)])
        self.steps_per_generation = 2 * pop * self.T_train + N_EVAL * self.T_eval
        # the evaluation rollout as a replayed hipGraph; with several ranks (an RCCL process group alive beside the capture)
        # that combination has never run on hardware, so it is opt-in there until a multi-GPU box has passed
        # tests/test_dist_gpu.py with COEVO_DQN_EVAL_GRAPH=1
        self.eval_graph = (os.environ.get("COEVO_DQN_EVAL_GRAPH", "1" if shard[1] == 1 else "0") != "0"
                           and frames == "device")   # (a host-stepped rollout is a blocking call, not capturable)
        self._eval_graph = None

    def generation(self, gen, sigmas, lr, fitness_sharing):
        """perturb -> this rank's 2*n_local games -> (rewards, distances) gathered -> fitness -> chunk partial sums
        gathered -> identical update on every rank -> 10 evaluation games -> mean evaluation rewards (host)"""
        self.gen_dev.fill_(gen)
        self.sigma.copy_(torch.tensor([float(sigmas[0]), float(sigmas[1])], dtype=torch.float32))
        flags = 1 | (2 if self.antithetic else 0)
        lo, hi = self.lo, self.hi
        for ri, r in enumerate(ROLES2):
            L.call("coevo_dqn_perturb", self._ptr(r, "base"), L._p(self.zero_idx), self._ptr(r, "pert"), 0, self.n_local,
                   self.C, self.n_actions, self.sigma.data_ptr() + 4 * ri, self.philox_seed, lo, gen * 4 + ri, flags, 1,
                   None, 0, self._ptr(r, "base") if fitness_sharing else None,
                   L._p(self.dist_partial) if fitness_sharing else None)
            if fitness_sharing:
                L.call("coevo_fc_distance_finalize", L._p(self.dist_partial), self.pblocks, self.n_local,
                       L._p(self.dist_local), 0, None)
                self.stats[ri, lo:hi, 1] = self.dist_local[:self.n_local]
        self.ro.enqueue(self.T_train, self.gen_dev)
        self.stats[:, lo:hi, 0] = self.ro.acc[self.game_idx, torch.arange(2, device=self.device)[:, None]]
        if self.world > 1:
            self.gather(self, "stats")
        for ri, r in enumerate(ROLES2):
            self.raw[ri].copy_(self.stats[ri, :, 0])
            if fitness_sharing:
                d = self.stats[ri, :, 1].to(torch.float32).contiguous()
                L.call("coevo_sharing_score", L._p(d), self.pop, L._p(self.div[ri]))
                self.raw[ri].div_(1.0 + self.div[ri])
            if self.centered_rank:
                L.call("coevo_centered_ranks", L._p(self.raw[ri]), self.pop, L._p(self.fitness[ri]))
            else:
                self.fitness[ri].copy_(self.raw[ri])
            L.call("coevo_dqn_es_partial", self._ptr(r, "base"), self._ptr(r, "pert"), lo, self.C, self.n_actions,
                   L._p(self.fitness[ri]), self.pop, self.chunks, self.rank * self.chunks_local, self.chunks_local,
                   self.partials.data_ptr() + 4 * (self.rank * self.part_block + self.part_off[r]))
        if self.world > 1:
            self.gather(self, "partials")
        for ri, r in enumerate(ROLES2):
            L.call("coevo_dqn_es_apply", self._ptr(r, "base"), self.partials.data_ptr() + 4 * self.part_off[r],
                   self.chunks, self.chunks_local, self.part_block, self.C, self.n_actions, self.pop,
                   self.sigma.data_ptr() + 4 * ri, L.C.c_float(lr))
        if self.eval_tiled:   # the updated base nets -> their tiled twins (two 6.75 MB copies)
            for ri, r in enumerate(ROLES2):
                L.call("coevo_dqn_relayout", self._ptr(r, "base"), self.eval_slab.data_ptr() + 4 * ri * self.stride, 1, self.C,
                       self.C | L.DQN_FC1_TILED, self.n_actions)
        # the ten evaluation games of the updated base nets: a chain of 3 x T_eval small dependent launches (29 % of a cfg 5
        # generation when enqueued one by one) - replayed as one hipGraph (the generation index is read from the device)
        if self.eval_graph:
            if self._eval_graph is None:
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    self.eval_ro.enqueue(self.T_eval, self.gen_dev)
                self._eval_graph = gr
            self._eval_graph.replay()
        else:
            self.eval_ro.enqueue(self.T_eval, self.gen_dev)
        torch.cuda.synchronize()
        L.raise_on_status(self.ro.status)
        L.raise_on_status(self.eval_ro.status)
        r = self.eval_ro.acc.cpu().numpy()
        tot = [0.0, 0.0]
        for j in range(N_EVAL):
            for s in range(2):
                tot[s] += float(r[j, s])
        return [t / 10 for t in tot]


class DQNESTrainer:
    def __init__(self, env, args, collect=True, dist_ctx=None):
        self.env, self.args, self.collect = env, args, collect
        C, n = _env_shape(env, args)
        base = {r: dqn_init_flat(C, n) for r in ROLES2}
        shard, gather = (0, 1), None
        if dist_ctx is not None and dist_ctx.world > 1:
            shard, gather = (dist_ctx.rank, dist_ctx.world), dist_ctx.gather_es
        self.first_ordinal = getattr(env, "n_resets", 1)
        self.eng = DQNESEngine(args.population, C, n, args.max_timesteps_per_episode, args.max_evaluation_steps,
                               env_seed=getattr(env, "seed_value", None) or SYNTH_SEED,
                               philox_seed=getattr(args, "coevo_seed", 0), shard=shard, gather=gather,
                               first_ordinal=self.first_ordinal, antithetic=getattr(args, "coevo_antithetic", False),
                               centered_rank=getattr(args, "coevo_centered_rank", False),
                               chunks=getattr(args, "coevo_es_chunks", 8), frames=frames_mode(args))
        for r in ROLES2:
            self.eng.upload(r, "base", 0, base[r][None])
        self.res = DQNResult()
        self.res.engine = self.eng
        self.gen = 0

    def step(self):
        eng, args, res = self.eng, self.args, self.res
        t0 = time.perf_counter()
        ev = eng.generation(self.gen, (args.mutation_power_agent_0, args.mutation_power_agent_1), args.learning_rate,
                            args.fitness_sharing)
        if self.collect:
            res.game_rewards.append(eng.ro.acc[:eng.n_main, :2].cpu().numpy().copy())
            res.diversity.append([float(eng.div[ri].item()) if args.fitness_sharing else None for ri in range(2)])
        for s, r in enumerate(ROLES2):
            res.rewards[r].append(ev[s])
        if args.adaptive:
            keep = getattr(args, "mutation_power_adversary", 0.0)
            h = {"agent_0": res.rewards["first_0"], "agent_1": res.rewards["second_0"],
                 "adversary_0": [0.0] * len(res.rewards["first_0"])}
            adapt_mutation_power(args, self.gen, h)
            args.mutation_power_adversary = keep
        res.sigma_after.append([args.mutation_power_agent_0, args.mutation_power_agent_1])
        res.seconds.append(time.perf_counter() - t0)
        self.gen += 1

    def finish(self):
        if hasattr(self.env, "n_resets"):
            self.env.n_resets = self.first_ordinal + self.gen * self.eng.per_gen
        return self.res

    def close(self):
        self.eng.ro.close()
        self.eng.eval_ro.close()


def dqn_evolution_strategy_train(env, args, output_dir, collect=True, dist_ctx=None):
    """evolution_strategy_train for the two-player Atari games; returns (base nets [first_0, second_0] as flat parameter
    vectors, DQNResult)"""
    from .io_utils import MetricsWriter
    tr = DQNESTrainer(env, args, collect=collect, dist_ctx=dist_ctx)
    for _ in range(args.generations):
        tr.step()
    res = tr.finish()
    mw = MetricsWriter(output_dir)
    for g in range(len(res.rewards["first_0"])):
        mw.write(generation=g, eval_rewards={r: res.rewards[r][g] for r in ROLES2}, mutation_power=res.sigma_after[g],
                 data="synthetic env")
    return [tr.eng.download(r, "base", 0, 1)[0] for r in ROLES2], res
