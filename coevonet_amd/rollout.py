"""Batched rollout of many simple_adversary games on one MI355X.

The reference plays its games one after another (play_game -> play_MPE, utils/game_logic_functions.py:123-228), one
batch-1 forward per agent-step.  All games of a generation are independent given the generation's weights and each
game's ordinal in the seeded reset stream, so they are flattened here into E env copies that advance in
lock-step world cycles.  Per cycle: every distinct weight set that acts is read ONCE (a task = one net x the rows that
share it), all three agents of a cycle observe the same world state, and the world step itself is fused into the next
cycle's policy launch (each row derives its game's state from the previous buffer + actions).

``RolloutPlan``   static description of a batch: which net plays which slot of which game -> task/row tables on device,
                  optionally partitioned into independent game cohorts
``DeviceRollout`` env state + action buffers + the cycle loop, env on the device: one merged launch per env-cycle (and
                  cohort), cohort chains on their own streams
``HostEnvRollout`` the same plan with the env stepped on the host cores (north_star's first configuration)
"""
from __future__ import annotations

import ctypes as _ct

import os

import numpy as np
import torch

from . import lib as L
from .mpe import simple_adversary as sa

LIGHT_ROWS = 8      # a net with up to this many rows is ONE per-individual (streaming) task
HEAVY_ROWS = 32     # larger row sets (shared opponents) are cut into chunks of this size (GA engine: 16, lean kernel)


class RolloutPlan:
    """game_nets: int array [n_games][3] of net ids for env slots (adversary_0, agent_0, agent_1);
    net_off / net_D: per net id, float offset inside the slab and observation width."""

    def __init__(self, game_nets, net_off, net_D, device="cuda", heavy_rows=HEAVY_ROWS, n_cohorts=1,
                 game_cohort=None, split_rows=None, row_order="class"):
        game_nets = np.asarray(game_nets, dtype=np.int64)
        self.n_games = int(game_nets.shape[0])
        by_net = {}
        for g in range(self.n_games):
            for slot in range(3):
                by_net.setdefault(int(game_nets[g, slot]), []).append((g, slot))
        for net, rows in by_net.items():
            d = int(net_D[net])
            for g, slot in rows:
                assert d == (8 if slot == 0 else 10), "net width does not match the env slot it plays"
        if game_cohort is not None:
            # the caller's own partition (e.g. contiguous ranges of individuals, so that offspring can be bred cohort by
            # cohort); it must keep every per-individual net's games together
            cohort = np.asarray(game_cohort, dtype=np.int32)
            assert cohort.shape == (self.n_games,) and cohort.min() >= 0
            for net, rows in by_net.items():
                if len(rows) <= LIGHT_ROWS and len({int(cohort[g]) for g, _ in rows}) != 1:
                    raise ValueError("a net with few rows (one task) plays in two cohorts: this partition is not valid")
            if len(np.unique(cohort)) != int(cohort.max()) + 1:
                raise ValueError("empty cohort")
        else:
            cohort = self._assign_cohorts(by_net, self.n_games, max(1, int(n_cohorts)))
        self.n_cohorts = int(cohort.max()) + 1 if self.n_games else 1
        self.game_cohort_np = cohort
        light = [[] for _ in range(self.n_cohorts)]
        heavy = [[] for _ in range(self.n_cohorts)]
        for net, rows in by_net.items():
            if split_rows is not None:
                # a handful of nets with a few rows each (the 10 evaluation games of a trio): cut every net's rows into
                # streaming-body tasks of <= split_rows rows instead of one matrix-core task - a lone small launch is
                # bound by the body's latency, and the streaming body is the short one (one cohort only)
                assert self.n_cohorts == 1 and 1 <= split_rows <= LIGHT_ROWS
                for i in range(0, len(rows), split_rows):
                    light[0].append((net, rows[i:i + split_rows]))
            elif len(rows) <= LIGHT_ROWS:
                light[int(cohort[rows[0][0]])].append((net, rows))   # all its games share one cohort by construction
            else:
                for k in range(self.n_cohorts):
                    mine = [(g, slot) for g, slot in rows if cohort[g] == k]
                    for i in range(0, len(mine), heavy_rows):
                        heavy[k].append((net, mine[i:i + heavy_rows]))
        heavy = [self._xcd_order(h) for h in heavy]
        self.heavy_begin_np = np.cumsum([0] + [len(h) for h in heavy]).astype(np.int32)
        self.light_begin_np = np.cumsum([0] + [len(x) for x in light]).astype(np.int32)
        # Row numbering.  "class" (device env): all shared-opponent tasks' rows first, then the per-individual tasks'; inside
        # each class first-game order (locality of the state reads).  "cohort" (env on the host cores): cohort by cohort, so
        # that a cohort's observations / actions are ONE contiguous range of the staging buffers (one copy each way).
        # The task tables are [class][cohort] in both; a task names its first row itself.
        assert row_order in ("class", "cohort")
        row_game, row_slot = [], []
        tasks = {"heavy": [[] for _ in range(self.n_cohorts)], "light": [[] for _ in range(self.n_cohorts)]}
        cohort_rows = np.zeros(self.n_cohorts + 1, dtype=np.int64)

        def number(name, k):
            for net, rows in (heavy if name == "heavy" else light)[k]:
                tasks[name][k].append((int(net_off[net]), len(row_game), len(rows), int(net_D[net]), 0))
                for g, slot in rows:
                    row_game.append(g)
                    row_slot.append(slot)

        if row_order == "class":
            for name in ("heavy", "light"):
                for k in range(self.n_cohorts):
                    number(name, k)
            self.cohort_row_begin_np = None
        else:
            for k in range(self.n_cohorts):
                number("heavy", k)
                number("light", k)
                cohort_rows[k + 1] = len(row_game)
            self.cohort_row_begin_np = cohort_rows
        tasks = {name: [t for per in tasks[name] for t in per] for name in tasks}
        self.n_rows = len(row_game)
        assert self.n_rows == 3 * self.n_games
        game_rows = np.zeros((self.n_games, 3), dtype=np.int32)
        for r, (g, slot) in enumerate(zip(row_game, row_slot)):
            game_rows[g, slot] = r
        self.row_game_np = np.asarray(row_game, dtype=np.int32)
        self.row_slot_np = np.asarray(row_slot, dtype=np.int32)
        self.game_rows_np = game_rows
        self.heavy_np = np.array(tasks["heavy"], dtype=L.TASK_DTYPE) if tasks["heavy"] else np.zeros(0, L.TASK_DTYPE)
        self.light_np = np.array(tasks["light"], dtype=L.TASK_DTYPE) if tasks["light"] else np.zeros(0, L.TASK_DTYPE)
        self.heavy_max = int(max([t[2] for t in tasks["heavy"]], default=0))
        self.light_max = int(max([t[2] for t in tasks["light"]], default=0))
        self.device = device
        if device is not None:
            self.row_game = torch.from_numpy(self.row_game_np).to(device)
            self.row_slot = torch.from_numpy(self.row_slot_np).to(device)
            self.game_rows = torch.from_numpy(game_rows.reshape(-1).copy()).to(device)
            self.heavy = L.tasks_to_device(self.heavy_np, device) if len(self.heavy_np) else None
            self.light = L.tasks_to_device(self.light_np, device) if len(self.light_np) else None

    @staticmethod
    def _assign_cohorts(by_net, n_games, n_cohorts):
        """game -> cohort.  Games that share a per-individual (light) net must advance together (one task reads that
        net once for all of them), so cohorts are unions of the connected components of "shares a light net"; the
        components are dealt round-robin over the cohorts.  Shared-opponent (heavy) nets
        do not tie games together: their rows are simply cut per cohort.  Speed only - results do not depend on it."""
        cohort = np.zeros(n_games, dtype=np.int32)
        if n_cohorts <= 1 or n_games == 0:
            return cohort
        parent = list(range(n_games))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for rows in by_net.values():
            if len(rows) <= LIGHT_ROWS:
                r0 = find(rows[0][0])
                for g, _ in rows[1:]:
                    parent[find(g)] = r0
        comps = {}
        for g in range(n_games):
            comps.setdefault(find(g), []).append(g)
        # dealt round-robin in first-game order: every cohort gets the same share of each role's individuals
        for c, games in enumerate(sorted(comps.values(), key=lambda c: c[0])):
            cohort[games] = c % n_cohorts
        # drop empty cohorts (fewer components than cohorts)
        used = np.unique(cohort)
        remap = {int(u): i for i, u in enumerate(used)}
        return np.array([remap[int(c)] for c in cohort], dtype=np.int32)

    @staticmethod
    def _xcd_order(heavy):
        """Workgroups are dealt round-robin over the 8 XCDs (blockIdx b and b+8 share one, each XCD has its own
        4 MiB L2).  The net-sorted chunk list is cut into 8 equal runs, run x goes to block indices = x (mod 8): every
        XCD class carries the same load and sees only one or two distinct nets, which it keeps in its own L2 (with
        fewer than 8 shared nets, as in Co-ES, a net is spread over several classes rather than leaving XCDs idle).
        Speed only: the kernel is correct under any placement."""
        if len(heavy) < 16:
            return heavy
        by_net = sorted(heavy, key=lambda t: t[0])          # chunks of one net adjacent
        n = len(by_net)
        buckets = [by_net[b * n // 8:(b + 1) * n // 8] for b in range(8)]  # equal load; 1-2 nets per XCD class
        out = []
        for q in range(max(len(b) for b in buckets)):
            for x in range(8):
                if q < len(buckets[x]):
                    out.append(buckets[x][q])
        return out

    def distinct_weight_bytes_per_cycle(self):
        """algorithmic bytes of one env-cycle: every distinct weight set that acts, once."""
        seen = {}
        for arr in (self.heavy_np, self.light_np):
            for t in arr:
                seen[int(t["net_off"])] = L.fc_param_count(int(t["D"])) * 4
        return sum(seen.values())


def effective_steps(limit, max_cycles):
    """agent-steps one game runs: play_MPE breaks at the step limit or when max_cycles world steps truncate it."""
    cap = 3 * max_cycles
    return cap if limit is None else min(int(limit), cap)


class DeviceRollout:
    """Env copies resident on the GPU.  One C-ABI call (coevo_mpe_rollout) enqueues every cycle of every cohort: per
    cycle and cohort ONE merged launch (shared-opponent workgroups on the matrix cores + per-individual streaming
    workgroups, env step fused in); ``merged=False`` keeps the two-launch form (shared-opponent launch on a side stream
    beside the streaming launch) for A/B runs."""

    def __init__(self, plan: RolloutPlan, slab: torch.Tensor, env_seed=sa.ENV_SEED, timing_pairs=0, fused_step=True,
                 merged=None):
        if merged is None:
            # one launch per env-cycle (shared-opponent + per-individual workgroups together): 385 vs 347 generations/s
            # on cfg2; COEVO_MERGED=0 keeps the two launches on two streams for A/B runs
            merged = os.environ.get("COEVO_MERGED", "1") != "0"
        self.plan = plan
        self.slab = slab
        dev = plan.device
        n = plan.n_games
        # two state buffers + actions by (game, slot), double buffered: the env step is fused into the policy launches
        self.state2 = torch.zeros(2, L.MPE_STATE_DOUBLES, n, dtype=torch.float64, device=dev)
        self.state = self.state2[0]
        self.actions_by_game = torch.zeros(2, n, 3, dtype=torch.int32, device=dev)
        self.actions = torch.zeros(plan.n_rows, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.limits = torch.zeros(n, dtype=torch.int32, device=dev)
        self.rewards = torch.zeros(n, 3, dtype=torch.float64, device=dev)
        self.rng = L.PCG64State.from_seed(env_seed)
        self.pos_first = 1 if sa.INTEGRATE_POS_FIRST else 0
        self.time_light = False
        self.overlap = True
        self.use_graph = True
        self._graphs = {}
        self._timed_ms = []
        self._span_ms, self._span_cycles = [], 0
        self.ctx = L.load().coevo_rollout_ctx_create(int(timing_pairs))
        if not self.ctx:
            raise L.CoevoError("coevo_rollout_ctx_create failed")
        p = plan
        self.desc = L.RolloutDesc(
            slab=L._p(slab), heavy=L._p(p.heavy), n_heavy=len(p.heavy_np), heavy_max_rows=p.heavy_max,
            light=L._p(p.light), n_light=len(p.light_np), light_max_rows=p.light_max,
            state=L._p(self.state), n_games=n, n_cycles=0, row_game=L._p(p.row_game), row_slot=L._p(p.row_slot),
            game_rows=L._p(p.game_rows), actions=L._p(self.actions), status=L._p(self.status),
            game_limit=L._p(self.limits), rewards=L._p(self.rewards), pos_first=self.pos_first, n_cohorts=0,
            state_alt=self.state2[1].data_ptr() if fused_step else None,
            actions_by_game=L._p(self.actions_by_game) if fused_step else None, light_stamps=None)
        self.n_cohorts = p.n_cohorts if fused_step else 1
        self.desc.merged = 1 if (fused_step and merged) else 0
        if self.n_cohorts > 1:
            self._hb = np.ascontiguousarray(p.heavy_begin_np)   # host arrays the C side reads at enqueue time
            self._lb = np.ascontiguousarray(p.light_begin_np)
            self.desc.n_cohorts = self.n_cohorts
            rc = L.load().coevo_rollout_ctx_reserve_cohorts(self.ctx, self.n_cohorts)
            if rc:
                raise L.CoevoError(f"coevo_rollout_ctx_reserve_cohorts failed with code {rc}")
            self.desc.heavy_begin = self._hb.ctypes.data
            self.desc.light_begin = self._lb.ctypes.data
        self.stamp_cycles = 256
        self.stamps = torch.zeros(self.n_cohorts * self.stamp_cycles, L.STAMP_SLOTS, 2, dtype=torch.int64,
                                  device=dev)  # per (cohort, cycle), per slot
        # scratch of the persistent whole-rollout launch (coevo_mpe_rollout_persistent: a cohort whose workgroups all fit the
        # chip at once plays its n_cycles in ONE launch; the C side decides per cohort): tagged action words, per cohort
        self.sync_words_per_cohort = int(L.load().coevo_mpe_persistent_sync_words(n))
        self.sync_words = None
        if fused_step and merged:
            self.sync_words = torch.zeros(self.n_cohorts * self.sync_words_per_cohort, dtype=torch.int32, device=dev)
            self.desc.sync_words = L._p(self.sync_words)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                L.load().coevo_rollout_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def set_limits(self, limits_np):
        self.limits.copy_(torch.from_numpy(np.asarray(limits_np, dtype=np.int32)), non_blocking=False)

    def reset(self, game_first, n_games, first_ordinal):
        """games [game_first, game_first+n_games) take reset ordinals first_ordinal.. of the seeded stream"""
        L.call("coevo_mpe_reset", L._p(self.state), self.plan.n_games, int(game_first), int(n_games), self.rng,
               int(first_ordinal))

    def reset_segments(self, segs, arm=None):
        """several (game_first, n_games, first_ordinal) resets in ONE launch (<= 4 segments).  arm = (cohort k, n_cycles):
        the launch also re-arms the clock stamps of that cohort's timed chain (time_light) and zeroes the sync words of its
        persistent rollout launch, which then need no launch of their own - the caller passes armed=True to the enqueue that
        follows"""
        arr = (L.ResetSeg * len(segs))(*[L.ResetSeg(int(a), int(b), int(c)) for a, b, c in segs])
        if arm is not None and (self.time_light or self.sync_words is not None):
            # ... and, in the same launch, zeroes the sync words of that cohort's persistent rollout launch (which then needs no
            # clearing launch of its own either: desc.sync_cleared, set by the enqueue that is passed armed=True)
            k, n_cycles = arm
            stamps = (self.stamps.data_ptr() + 16 * L.STAMP_SLOTS * int(k) * int(n_cycles)) if self.time_light else None
            sync = (self.sync_words.data_ptr() + 4 * int(k) * self.sync_words_per_cohort) if self.sync_words is not None else None
            L.call("coevo_mpe_reset_multi_prep", L._p(self.state), self.plan.n_games, _ct.cast(arr, _ct.c_void_p), len(segs),
                   self.rng, stamps, int(n_cycles) * L.STAMP_SLOTS if stamps else 0, sync,
                   self.sync_words_per_cohort if sync else 0)
            return
        L.call("coevo_mpe_reset_multi", L._p(self.state), self.plan.n_games, _ct.cast(arr, _ct.c_void_p), len(segs), self.rng)

    def run(self, n_cycles):
        """enqueue n_cycles world cycles + the rewards kernel.  With use_graph the enqueue is captured once per
        (cycle count, timed?) into a hipGraph and replayed: the fork/join between the two policy launches then costs a
        graph edge instead of a cross-stream event round trip (316 vs 266 generations/s on cfg2)."""
        n_cycles = int(n_cycles)
        assert n_cycles <= self.stamp_cycles
        ctx = self.ctx if (self.overlap or self.n_cohorts > 1) else None
        timed = bool(self.time_light)
        if self.use_graph:
            # kernel timing inside a replayed graph comes from the kernel's own 100 MHz clock stamps
            self.desc.light_stamps = L._p(self.stamps) if timed else None
            key = (n_cycles, timed, bool(self.overlap))
            g = self._graphs.get(key)
            if g is None:
                self.desc.n_cycles = n_cycles
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    L.call("coevo_mpe_rollout", L.C.byref(self.desc), ctx, 0)
                self._graphs[key] = g
            g.replay()
            if timed:
                self._pending_stamps = n_cycles
            return
        self.desc.light_stamps = None
        self.desc.n_cycles = n_cycles
        L.call("coevo_mpe_rollout", L.C.byref(self.desc), ctx, 1 if timed else 0)

    def enqueue(self, n_cycles, final=True, armed=False, pack=None):
        """plain enqueue on the current stream (no graph of its own): for callers that capture a larger graph.
        final=False: without the closing step (the caller runs enqueue_final_step itself);
        pack (with final=True; see enqueue_final_step): the closing step also writes this rank's all-gather record - inside the
        persistent launch when the rollout is one (coevo_rollout_desc.pack);
        armed: the reset launch in front of it re-armed the clock stamps (reset_segments(arm=...))"""
        self.desc.light_stamps = L._p(self.stamps) if self.time_light else None
        self.desc.n_cycles = int(n_cycles)
        self.desc.stamps_armed = 1 if (armed and self.time_light) else 0
        self.desc.sync_cleared = 1 if (armed and self.sync_words is not None) else 0
        keep = self.desc.rewards
        fp = None
        if not final:
            assert self.desc.state_alt, "only the fused-step rollout can leave its books open"
            self.desc.rewards = None
        elif pack is not None:
            out, dist, n_roles, n_local, hof, pitch, first = pack
            fp = L.FinalPack(L._p(out), L._p(dist), int(n_roles), int(n_local), int(hof), int(pitch), int(first), 0)
            self.desc.pack = L.C.addressof(fp)
        try:
            L.call("coevo_mpe_rollout", L.C.byref(self.desc), self.ctx if (self.overlap or self.n_cohorts > 1) else None, 0)
        finally:
            self.desc.rewards = keep
            self.desc.pack = None
            self.desc.stamps_armed = 0
            self.desc.sync_cleared = 0
        if self.time_light:
            self._pending_stamps = int(n_cycles)

    def enqueue_cohort(self, k, n_cycles, stream, armed=False):
        """the cycle chain of cohort k alone on `stream` (a torch stream), without the closing step: callers that breed
        and reset cohort by cohort run one such call per cohort on its own stream, then enqueue_final_step() once"""
        assert self.n_cohorts > 1 and self.desc.merged
        p = self.plan
        hb, he = int(p.heavy_begin_np[k]), int(p.heavy_begin_np[k + 1])
        lb, le = int(p.light_begin_np[k]), int(p.light_begin_np[k + 1])
        tsz = L.TASK_DTYPE.itemsize
        d = L.RolloutDesc()
        L.C.memmove(L.C.byref(d), L.C.byref(self.desc), L.C.sizeof(d))
        d.heavy = (p.heavy.data_ptr() + hb * tsz) if he > hb else None
        d.n_heavy = he - hb
        d.light = (p.light.data_ptr() + lb * tsz) if le > lb else None
        d.n_light = le - lb
        d.n_cohorts, d.heavy_begin, d.light_begin = 0, None, None
        d.concurrent_hint = self.n_cohorts
        d.n_cycles = int(n_cycles)
        d.rewards = None
        d.light_stamps = (self.stamps.data_ptr() + 16 * L.STAMP_SLOTS * k * int(n_cycles)) if self.time_light else None
        d.stamps_armed = 1 if (armed and self.time_light) else 0
        if self.sync_words is not None:   # this cohort's own scratch (zeroed by the launch, or by the reset launch: armed)
            d.sync_words = self.sync_words.data_ptr() + 4 * k * self.sync_words_per_cohort
            d.sync_cleared = 1 if armed else 0
        L._check(L.load().coevo_mpe_rollout(L.C.byref(d), self.ctx, 0, stream.cuda_stream), "coevo_mpe_rollout")
        if self.time_light:
            self._pending_stamps = int(n_cycles)

    def enqueue_final_step(self, n_cycles, pack=None):
        """close the books of a rollout whose chains were enqueued with enqueue_cohort / enqueue(final=False) (current stream).
        pack = (out [n_roles][n_local][4] fp64, dist [n_roles][pitch] fp32, n_roles, n_local, hof, pitch, first): this rank's
        record of the fitness all-gather written by the same launch (coevo_mpe_final_step_pack)"""
        n, last = self.plan.n_games, int(n_cycles) - 1
        st_last = self.state2[0] if last <= 0 or (last & 1) == 0 else self.state2[1]
        act = self.actions_by_game[(last if last > 0 else 0) & 1]
        if pack is not None:
            out, dist, n_roles, n_local, hof, pitch, first = pack
            L.call("coevo_mpe_final_step_pack", L._p(st_last), n, L._p(act), last, L._p(self.limits), self.pos_first,
                   L._p(self.rewards), L._p(out), L._p(dist), n_roles, n_local, hof, pitch, first)
            return
        L.call("coevo_mpe_final_step", L._p(st_last), n, L._p(act), last, L._p(self.limits), self.pos_first,
               L._p(self.rewards))

    def collect_stamps(self):
        """after the replay has finished (the caller synchronised): fold this replay's clock stamps into the log"""
        n = getattr(self, "_pending_stamps", 0)
        if n:
            rows = n * self.n_cohorts
            st = self.stamps[:rows].cpu().numpy()  # [cohort * n + cycle][slots][2]
            dur = st[:, :, 1].max(axis=1) - st[:, :, 0].min(axis=1)  # first workgroup start .. last workgroup end
            self._timed_ms.extend((dur * 1e-5).tolist())  # 100 MHz ticks -> ms
            # first start .. last end over every policy launch of this rollout (cohorts overlap; gaps included)
            self._span_ms.append(float(st[:, :, 1].max() - st[:, :, 0].min()) * 1e-5)
            self._span_cycles = n
            self._pending_stamps = 0

    def _read_light_times(self, max_out=100000):
        buf = (L.C.c_float * max_out)()
        n = L.load().coevo_rollout_ctx_light_times(self.ctx, buf, max_out)
        if n < 0:
            raise L.CoevoError(f"coevo_rollout_ctx_light_times failed with code {n}")
        return [buf[i] for i in range(n)]

    def light_times_ms(self):
        """durations (ms) of every timed launch of the per-individual policy kernel since timing was switched on"""
        if self.use_graph:
            torch.cuda.synchronize()
            self.collect_stamps()
            return list(self._timed_ms)
        return self._read_light_times()

    def reset_timing(self):
        L.load().coevo_rollout_ctx_reset_timing(self.ctx)
        self._timed_ms = []
        self._span_ms = []

    def check_status(self):
        L.raise_on_status(self.status)


class HostEnvRollout:
    """Same plan, env stepped on the host cores (struct-of-arrays); per cycle the observations go up and the actions come
    back over PCIe.  Results are bit-identical with DeviceRollout.

    impl = "native" (default): ONE C-ABI call per rollout, coevo_mpe_host_rollout (csrc/host_rollout.hip) - the plan's
    cohorts are dealt to the host cores (COEVO_HOST_THREADS, default 2); a core drives its cohorts from the first
    observation to the last world step and alternates between them - while one cohort's policy launch is in flight on its
    own stream the core steps another - and polls a per-cohort completion word; nothing waits for a whole stream.
    impl = "numpy": coevonet_amd/mpe/simple_adversary.py's VecSimpleAdversary, the form the env fixtures are stated in, one
    blocking cycle at a time (1.5 ms of NumPy per cycle at cfg 2)."""

    def __init__(self, plan: RolloutPlan, slab: torch.Tensor, env_seed=sa.ENV_SEED, impl=None, threads=None):
        self.impl = impl or os.environ.get("COEVO_HOST_ENV", "native")
        if self.impl not in ("native", "numpy"):
            raise ValueError(f"host env implementation {self.impl!r}: 'native' or 'numpy'")
        self.plan = plan
        self.slab = slab
        dev = plan.device
        self.obs = torch.zeros(plan.n_rows, L.OBS_STRIDE, dtype=torch.float32, device=dev)
        self.actions = torch.zeros(plan.n_rows, dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.env_seed = env_seed
        self.set_limits(np.zeros(plan.n_games, dtype=np.int64))
        self.rewards = None
        self._merged = None
        self.ctx = None
        self.state = None
        self._pending_ordinals = None   # set by reset_from_ordinals: the next run() draws these resets first
        self.phase_us = None          # set to a float64[6] array to collect the per-cohort-cycle breakdown
        self.zero_copy = os.environ.get("COEVO_HOST_ZERO_COPY", "1") == "1"
        if self.impl != "native":
            self.obs_host = torch.zeros(plan.n_rows, L.OBS_STRIDE, dtype=torch.float32).pin_memory()
            self.actions_host = torch.zeros(plan.n_rows, dtype=torch.int32).pin_memory()
        if self.impl == "native":
            p = plan
            K = p.n_cohorts
            if K > 1 and p.cohort_row_begin_np is None:
                raise ValueError("a host-stepped rollout with several cohorts needs a plan built with row_order='cohort'")
            if threads is None:
                threads = int(os.environ.get("COEVO_HOST_THREADS", "0")) or 2
            threads = max(1, min(int(threads), K))   # a core drives whole cohorts: more cores than cohorts would idle
            self.ctx = L.load().coevo_host_rollout_create(int(threads), K)
            if not self.ctx:
                raise L.CoevoError("coevo_host_rollout_create failed")
            self.threads = int(L.load().coevo_host_rollout_threads(self.ctx))
            # the staging buffers come from the context: page-locked, mapped, first touched on the NUMA node its cores run on
            # (the GPU's, as far as the affinity mask allows - csrc/host_placement.hip)
            self.obs_host = L.host_tensor(self.ctx, (plan.n_rows, L.OBS_STRIDE), np.float32)
            self.actions_host = L.host_tensor(self.ctx, (plan.n_rows,), np.int32)
            self.placement = L.host_placement(self.ctx)
            self._game_rows32 = np.ascontiguousarray(p.game_rows_np, dtype=np.int32)
            rows = p.cohort_row_begin_np if p.cohort_row_begin_np is not None else np.array([0, p.n_rows])
            self._cohort_games = [np.ascontiguousarray(np.nonzero(p.game_cohort_np == k)[0], dtype=np.int32)
                                  for k in range(K)]
            tsz = L.TASK_DTYPE.itemsize
            self._cohorts = (L.HostCohort * K)()
            for k in range(K):
                hb, he = int(p.heavy_begin_np[k]), int(p.heavy_begin_np[k + 1])
                lb, le = int(p.light_begin_np[k]), int(p.light_begin_np[k + 1])
                self._cohorts[k] = L.HostCohort(
                    heavy=(p.heavy.data_ptr() + hb * tsz) if he > hb else None,
                    light=(p.light.data_ptr() + lb * tsz) if le > lb else None,
                    games=self._cohort_games[k].ctypes.data, n_heavy=he - hb,
                    heavy_max_rows=int(max(p.heavy_np["n_rows"][hb:he], default=0)), n_light=le - lb,
                    light_max_rows=int(max(p.light_np["n_rows"][lb:le], default=0)),
                    n_games=len(self._cohort_games[k]), row_first=int(rows[k]), n_rows=int(rows[k + 1] - rows[k]))

    def close(self):
        """releases the context: its worker threads, streams, events and the page-locked buffers (obs_host / actions_host
        are invalid afterwards)"""
        if getattr(self, "ctx", None):
            self.obs_host = self.actions_host = None
            L.load().coevo_host_rollout_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        # the destroy frees page-locked memory (hipHostFree synchronises the device): never from a finaliser that the
        # garbage collector runs while another engine captures a hipGraph - the context then lives until close() / exit
        try:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return
            self.close()
        except Exception:
            pass

    def set_limits(self, limits_np):
        self.limits = np.asarray(limits_np, dtype=np.int64)
        self._limits32 = np.ascontiguousarray(self.limits, dtype=np.int32)

    def reset_from_ordinals(self, ordinals):
        """ordinals[g] = reset ordinal of game g; draws every reset up to the largest one on the host."""
        ordinals = np.ascontiguousarray(ordinals, dtype=np.int64)
        n = self.plan.n_games
        if self.impl == "native":   # the device state layout (csrc/mpe_env.hip) in host memory; the resets themselves are drawn
            if getattr(self, "state", None) is None or self.state.shape[1] != n:   # inside the rollout, cohort by cohort, by
                self.state = np.zeros((L.MPE_STATE_DOUBLES, n), dtype=np.float64)   # the core that drives the cohort
            if ordinals.shape != (n,) or (n and ordinals.min() < 0):
                raise ValueError("one non-negative reset ordinal per game")
            self._pending_ordinals = ordinals
            return
        stream = sa.ResetStream(self.env_seed, skip_initial=False)
        goal, apos, lpos = stream.take(int(ordinals.max()) + 1)
        self.env = sa.VecSimpleAdversary(goal[ordinals], apos[ordinals], lpos[ordinals])
        self.acc = np.zeros((n, 3))
        self.rg_prev = np.zeros(n)

    def _forward(self):
        """(numpy env) the policy step of every row: both task tables in one launch of the lean cycle kernel when they fit
        it (<= 16-row shared-opponent tasks, <= 8-row per-individual tasks, everything resident), else one launch per table"""
        p = self.plan
        if self._merged is None:
            self._merged = (p.heavy is not None and p.light is not None and p.heavy_max <= 16 and p.light_max <= 8 and
                            len(p.heavy_np) + len(p.light_np) <= 4 * torch.cuda.get_device_properties(self.obs.device).multi_processor_count)
        if self._merged:
            rc = L.load().coevo_fc_forward_merged(L._p(self.slab), L._p(p.heavy), len(p.heavy_np), p.heavy_max, L._p(p.light),
                                                  len(p.light_np), p.light_max, L._p(self.obs), L._p(self.actions), None,
                                                  L._p(self.status), L._stream())
            if rc == 0:
                return
            if rc != -1:
                L._check(rc, "coevo_fc_forward_merged")
            self._merged = False   # COEVO_ERR_ARG: the library's own residency check disagrees - one launch per table
        for tasks, tnp, mx in ((p.heavy, p.heavy_np, p.heavy_max), (p.light, p.light_np, p.light_max)):
            if tasks is not None:
                L.call("coevo_fc_forward_argmax", L._p(self.slab), L._p(tasks), len(tnp), mx, L._p(self.obs),
                       L._p(self.actions), None, L._p(self.status))

    def cycle(self, c):
        """(numpy env) one blocking cycle"""
        p = self.plan
        adv, a0, a1 = self.env.observe()
        o = self.obs_host.numpy()
        o[:] = 0.0
        for slot, arr in ((0, adv), (1, a0), (2, a1)):
            rows = p.game_rows_np[:, slot]
            o[rows, :arr.shape[1]] = arr
        self.obs.copy_(self.obs_host, non_blocking=True)
        self._forward()
        self.actions_host.copy_(self.actions, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        acts = self.actions_host.numpy()[p.game_rows_np]  # [n_games, 3]
        t0 = 3 * c
        m0, m1, m2 = t0 < self.limits, t0 + 1 < self.limits, t0 + 2 < self.limits
        self.acc[m0, 0] += self.rg_prev[m0]
        self.acc[m1, 1] += self.rg_prev[m1]
        rg, ra = self.env.step(acts)  # games past their limit are stepped too; their credits are masked
        self.acc[m2, 2] += ra[m2]
        self.rg_prev[m2] = rg[m2]

    def run(self, n_cycles):
        if self.impl == "native":
            p = self.plan
            d = L.HostRolloutDesc(
                slab=L._p(self.slab), state=self.state.ctypes.data, game_rows=self._game_rows32.ctypes.data,
                game_limit=self._limits32.ctypes.data, obs_host=self.obs_host.data_ptr(), obs_dev=L._p(self.obs),
                actions_host=self.actions_host.data_ptr(), actions_dev=L._p(self.actions), status=L._p(self.status),
                cohorts=C_cast(self._cohorts), phase_us=(self.phase_us.ctypes.data if self.phase_us is not None else None),
                n_games=p.n_games, n_rows=p.n_rows, n_cycles=int(n_cycles), n_cohorts=p.n_cohorts,
                pos_first=1 if sa.INTEGRATE_POS_FIRST else 0, zero_copy=1 if self.zero_copy else 0,
                reset_ordinals=(self._pending_ordinals.ctypes.data if self._pending_ordinals is not None else None),
                reset_rng=L.PCG64State.from_seed(self.env_seed))
            L._check(L.load().coevo_mpe_host_rollout(self.ctx, L.C.byref(d), L._stream()), "coevo_mpe_host_rollout")
            self._pending_ordinals = None
            # play_game's triple (agent_0, agent_1, adversary_0): state rows 20, 21, 19
            self.rewards = np.ascontiguousarray(self.state[[20, 21, 19]].T)
            return
        for c in range(n_cycles):
            self.cycle(c)
        self.rewards = np.stack([self.acc[:, 1], self.acc[:, 2], self.acc[:, 0]], axis=1)

    def check_status(self):
        L.raise_on_status(self.status)


def C_cast(arr):
    return _ct.cast(arr, _ct.c_void_p)
