"""Co-GA on the MI355X: drop-in ``genetic_algorithm_train(env, agent, args, output_dir)`` (reference
genetic_algorithm.py:51) over a batched engine.

Per generation the reference plays 3*pop*hof + 10 sequential games; here all of them advance together on the device
(coevonet_amd.rollout), fitness / sharing / ranking / HoF update / offspring stay on the device; with device-built
offspring ("device_philox") the evaluation means and the adaptive mutation power do too, and a generation has no host
round trip at all (GAEngine.replay_generation_pipelined on one GPU, GAEngine.step_sharded on several).

reference_exact semantics kept: Q1 reward attribution (device step kernel), Q2 only the last
HoF game counts, Q3 diversity against the stale agent left over from the init loop, always applied, Q4 the adversary
phase takes both good opponents from hof_agent_0, Q5 agent_0's sigma increase uses agent_1's sigma, Q6 single seeded
reset stream addressed by game ordinal, Q7 placeholder elites only burn RNG, Q8 GA mutates LayerNorm affine, Q13
argsort()[::-1].

rng modes
  "host_reference"  nets are initialised and mutated on the host with the very torch calls the reference makes, in its
                    order (bit-comparable elites; used for parity) and uploaded;
  "device_philox"   offspring are built on the device from counter-based noise (performance mode); initial nets still
                    come from torch's default init.

The 10 evaluation games of generation g depend only on g's selection, and nothing before the end of generation g+1
depends on them (they feed the sigma used by g+1's mutation), so they ride along in generation g+1's rollout launch
instead of costing a second sequential rollout; the last generation's evaluation is flushed on its own.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import lib as L
from .fcnetwork import FCNetwork
from .game_logic import create_agent
from .mpe.simple_adversary import ENV_SEED
from .rollout import DeviceRollout, HostEnvRollout, RolloutPlan, effective_steps

ROLES = ("agent_0", "agent_1", "adversary_0")
ROLE_D = {"agent_0": 10, "agent_1": 10, "adversary_0": 8}
ROLE_SLOT = {"agent_0": 1, "agent_1": 2, "adversary_0": 0}   # env slot the role acts in
RET_SLOT = {"agent_0": 0, "agent_1": 1, "adversary_0": 2}    # position in play_game's return triple
SIGMA_ATTR = {"agent_0": "mutation_power_agent_0", "agent_1": "mutation_power_agent_1",
              "adversary_0": "mutation_power_adversary"}
N_EVAL = 10


# ---- the reference's per-call helpers under their own names (genetic_algorithm.py:12-48): the sequential forms a caller of
# the reference's module may use directly.  The trainers below do the same work batched on the device.
def evaluate_current_weights(agent_0, agent_1, adversary, env, args):
    """mean rewards of 10 evaluation games of one trio (genetic_algorithm.py:12-29, evolutionary_strategy.py:22-59)"""
    from .game_logic import play_game
    tot = [0.0, 0.0, 0.0]
    for _ in range(N_EVAL):
        r = play_game(env=env, player1=agent_0.model, player2=agent_1.model, adversary=adversary.model, args=args, eval=True)
        for s in range(3):
            tot[s] += r[s]
    return tot[0] / 10, tot[1] / 10, tot[2] / 10


def mutate_elites(env, elites, args, role):
    """population - 1 children: child i = clone of elites[i % elites_number] with every parameter += N(0, sigma_role)
    (genetic_algorithm.py:32-48; any role name other than agent_0 / agent_1 takes the adversary's sigma, as there)"""
    sigma = {"agent_0": args.mutation_power_agent_0, "agent_1": args.mutation_power_agent_1}.get(
        role, args.mutation_power_adversary)
    children = []
    for i in range(args.population - 1):
        child = elites[i % args.elites_number].clone(env, args, role)
        child.mutate(sigma)
        children.append(child)
    return children


def adapt_mutation_power(args, gen, hist):
    """genetic_algorithm.py:323-345 (evolutionary_strategy.py:292-316 is identical), quirk Q5 included."""
    def worse(h):
        return gen > 10 and np.mean(h[-10:]) < np.mean(h[-20:-10])
    if worse(hist["agent_0"]):
        args.mutation_power_agent_0 = min(args.mutation_power_agent_1 * 1.2, args.max_mutation_power)
    else:
        args.mutation_power_agent_0 = max(args.mutation_power_agent_0 * 0.95, args.min_mutation_power)
    if worse(hist["agent_1"]):
        args.mutation_power_agent_1 = min(args.mutation_power_agent_1 * 1.2, args.max_mutation_power)
    else:
        args.mutation_power_agent_1 = max(args.mutation_power_agent_1 * 0.95, args.min_mutation_power)
    if worse(hist["adversary_0"]):
        args.mutation_power_adversary = min(args.mutation_power_adversary * 1.2, args.max_mutation_power)
    else:
        args.mutation_power_adversary = max(args.mutation_power_adversary * 0.95, args.min_mutation_power)


def small_shard_rows(n_local, hof, cus, pop):
    """Rows per shared-opponent task of a rank that holds n_local individuals per role.  One env-cycle of it is 3 n_local
    per-individual workgroups + 6 hof ceil(n_local / rows) shared-opponent ones (3 phases x hof games x 2 opponent seats, each
    opponent against all of the rank's individuals).  While they are all resident at once - two per CU: a rank of pop 200 over 8
    GPUs, 75 + 150, or over 4, 150 + 300 - tasks are cut to hof rows (25 = 5 x 5: no padding) and the rollout is ONE persistent
    launch of the small-launch body (<= 8 rows, vector-ALU fc2: csrc/fc_forward.hip fc_rollout_small_kernel,
    coevo_mpe_persistent_fits); else the 16-row tiles of the lean kernel.  COEVO_PERSISTENT=0 (per-cycle launches, A/B): hof rows
    only while every workgroup has a CU to itself (two small-launch workgroups per CU lose against the lean kernel)."""
    if n_local >= pop or not 1 <= hof <= 8:   # the whole population on this GPU: the full launch
        return 16
    per_cu = 2 if os.environ.get("COEVO_PERSISTENT", "1") != "0" else 1
    if 3 * n_local + 6 * hof * -(-n_local // hof) <= per_cu * cus:
        return hof
    return 16


def cohort_partition(n_local, K):
    """The ONE partition of a rank's individuals into K contiguous cohorts that the rollout plan, the breeding and the
    resets all use: -> (bounds [K+1], cohort of each individual [n_local])."""
    bounds = np.array([k * n_local // K for k in range(K + 1)], dtype=np.int64)
    return bounds, np.searchsorted(bounds[1:], np.arange(n_local), side="right").astype(np.int32)


DEFAULT_COHORTS = 1       # independent game cohorts per rollout (rollout.RolloutPlan._assign_cohorts); see DESIGN.md
DEVICE_LOOP_COHORTS = 2   # ... in the host-free loop, where their launches are enqueued eagerly and do overlap
SMALL_SHARD = 80          # ... unless a rank holds fewer individuals per role than this: then one chain


class GAEngine:
    """Device-resident population / HoF / elites of the three roles and the per-generation pipeline.

    shard = (rank, world): this process evaluates individuals [lo, hi) of every role (contiguous ranges, all HoF games
    of an individual on the same GPU); fitness and distances are all-gathered by the caller-provided ``gather``."""

    def __init__(self, pop, hof, elites, limit_train=None, limit_eval=None, max_cycles=25, device="cuda",
                 env_seed=ENV_SEED, rng="device_philox", philox_seed=0, env="device", first_ordinal=1,
                 shard=(0, 1), gather=None, timing_pairs=4096, cohorts=1, gather_packed=None):
        assert 1 <= elites <= pop and hof >= 1
        self.pop, self.hof, self.E = pop, hof, elites
        self.rng_mode, self.philox_seed, self.env_mode = rng, int(philox_seed), env
        self.device = device
        self.rank, self.world = shard
        self.gather = gather
        self.gather_packed = gather_packed   # (eng) -> eng.pack_all from every rank's eng.pack_local: ONE collective
        if self.world > 1 and pop % self.world:
            # the fitness all-gather (dist.allgather_shards) moves equal shards: unequal ones would hang RCCL
            raise ValueError(f"population {pop} is not divisible by the number of ranks {self.world}")
        self.lo = self.rank * pop // self.world
        self.hi = (self.rank + 1) * pop // self.world
        self.n_local = self.hi - self.lo
        self.T_train = effective_steps(limit_train, max_cycles)
        self.T_eval = effective_steps(limit_eval, max_cycles)
        self.n_cycles = (max(self.T_train, self.T_eval) + 2) // 3
        self.first_ordinal = first_ordinal
        self.env_seed = env_seed
        # ---- slab layout: per role [pop | hof | elite | stale | hof_tmp] -----------------------------------
        self.stride = {r: L.fc_slab_stride(ROLE_D[r]) for r in ROLES}
        self.P = {r: L.fc_param_count(ROLE_D[r]) for r in ROLES}
        self.base, off = {}, 0
        for r in ROLES:
            self.base[r] = {}
            for region, count in (("pop", pop), ("hof", hof), ("elite", elites), ("stale", 1), ("hof_tmp", hof),
                                  ("elite_prev", elites)):
                self.base[r][region] = off
                off += count * self.stride[r]
        self.slab = torch.zeros(off, dtype=torch.float32, device=device)
        # ---- game table of one generation launch -----------------------------------------------------------
        net_off, net_D, ids = [], [], {}

        def net(region, role, i):
            key = (region, role, i)
            if key not in ids:
                ids[key] = len(net_off)
                net_off.append(self.base[role][region] + i * self.stride[role])
                net_D.append(ROLE_D[role])
            return ids[key]

        games = []
        h = hof
        for role in ROLES:
            for i in range(self.lo, self.hi):
                for k in range(h):
                    if role == "agent_0":      # genetic_algorithm.py:136-142
                        a0, a1, adv = net("pop", role, i), net("hof", "agent_1", h - 1 - k), net("hof", "adversary_0", h - 1 - k)
                    elif role == "agent_1":    # :168-174
                        a0, a1, adv = net("hof", "agent_0", h - 1 - k), net("pop", role, i), net("hof", "adversary_0", h - 1 - k)
                    else:                      # :201-207, Q4: agent_1's seat is also filled from hof_agent_0
                        a0, a1, adv = net("hof", "agent_0", h - 1 - k), net("hof", "agent_0", h - 1 - k), net("pop", role, i)
                    games.append((adv, a0, a1))
        self.n_main = len(games)
        for _ in range(N_EVAL):  # evaluate_current_weights(best trio) = newest HoF members (:12-29, :301)
            games.append((net("hof", "adversary_0", h - 1), net("hof", "agent_0", h - 1), net("hof", "agent_1", h - 1)))
        # 16-row shared-opponent tasks select the lean merged cycle kernel (four workgroups per CU); COEVO_HEAVY_ROWS=32
        # keeps the 32-row tiles for A/B runs
        # (host-stepped env: the same 16-row tasks, for the observation-fed merged launch coevo_fc_forward_merged)
        heavy_rows = int(os.environ.get("COEVO_HEAVY_ROWS", "0"))
        if heavy_rows <= 0:
            cus = torch.cuda.get_device_properties(device).multi_processor_count if (env == "device" and str(device) != "cpu") else 0
            heavy_rows = small_shard_rows(self.n_local, hof, cus, pop) if (cus and cohorts <= 1) else 16
        # cohorts = contiguous ranges of this rank's individuals (so that offspring can be bred cohort by cohort and a
        # cohort's chain can start while the next cohort is still being bred); the evaluation games go with the last
        # (env on the host cores: the cohorts alternate between the cores and the GPU - COEVO_HOST_COHORTS, default 4: the
        # chain launch -> actions -> host step -> next launch of ONE cohort is ~90 us of latency however many cores step it;
        # four chains in flight hide most of it, six or more share hardware queues and serialise - profiles/r04_experiments.md)
        # ... a small batch fewer: ~700 games per cohort (550 games: one cohort 792 generations/s, four 671: profiles/r04_experiments.md)
        host_k = int(os.environ.get("COEVO_HOST_COHORTS", "0")) or max(1, min(4, len(games) // 700))
        self.K = max(1, min(int(cohorts) if env == "device" else host_k, self.n_local))
        row_order = "class" if env == "device" else "cohort"
        self._set_cohort_bounds()
        game_cohort = None
        if self.K > 1:
            # ONE partition for the rollout plan, the breeding and the resets: individual i of this rank belongs to the
            # cohort whose [bounds[k], bounds[k+1]) holds it
            per_ind = np.repeat(cohort_partition(self.n_local, self.K)[1], self.hof)
            game_cohort = np.concatenate([per_ind, per_ind, per_ind, np.full(N_EVAL, self.K - 1)]).astype(np.int32)
        try:
            self.plan = RolloutPlan(np.array(games), net_off, net_D, device=device, heavy_rows=heavy_rows,
                                    n_cohorts=self.K, game_cohort=game_cohort, row_order=row_order)
        except ValueError:  # tiny populations: the shared opponents have so few rows that they tie all games together
            self.K = 1
            self._set_cohort_bounds()
            self.plan = RolloutPlan(np.array(games), net_off, net_D, device=device, heavy_rows=heavy_rows,
                                    row_order=row_order)
        if env == "device":
            self.ro = DeviceRollout(self.plan, self.slab, env_seed=env_seed, timing_pairs=timing_pairs)
        else:
            self.ro = HostEnvRollout(self.plan, self.slab, env_seed=env_seed)
        # ---- small device buffers ---------------------------------------------------------------------------
        f32 = dict(dtype=torch.float32, device=device)
        self.dist_all = torch.zeros(3, pop, **f32)                      # [role][individual], one tensor: the fitness
        self.dist = {r: self.dist_all[i] for i, r in enumerate(ROLES)}  # all-gather packs / unpacks it in one copy
        self.div = {r: torch.zeros(1, **f32) for r in ROLES}
        self.fitness = {r: torch.zeros(pop, **f32) for r in ROLES}
        self.order = {r: torch.zeros(pop, dtype=torch.int32, device=device) for r in ROLES}
        self.sigma = {r: torch.zeros(1, **f32) for r in ROLES}
        self.sigma_prev = {r: torch.zeros(1, **f32) for r in ROLES}
        self.last_reward = torch.zeros(3, pop, 3, dtype=torch.float64, device=device)  # [role][i][triple]
        # the fused form of the exchange: this rank's record [role][j] = {triple of the last HoF game, distance}, written by
        # the rollout's closing launch, and the all-gathered [rank][role][j][4] buffer the selection reads as it is
        self.pack_local = torch.zeros(3, self.n_local, 4, dtype=torch.float64, device=device)
        self.pack_all = torch.zeros(self.world, 3, self.n_local, 4, dtype=torch.float64, device=device)
        self.parent_idx = torch.tensor([c % elites for c in range(max(pop - 1, 1))], dtype=torch.int32, device=device)
        self.hof_shift_idx = torch.arange(1, max(hof, 2), dtype=torch.int32, device=device)
        self.iota = torch.arange(max(hof, elites, 2), dtype=torch.int32, device=device)
        self.generation = 0
        self.steps_per_generation = 3 * pop * hof * self.T_train + N_EVAL * self.T_eval

    # ------------------------------------------------------------------ loading weights
    def _ptr(self, role, region, i=0):
        return self.slab.data_ptr() + 4 * (self.base[role][region] + i * self.stride[role])

    def upload(self, role, region, first, flat_np):
        """flat_np [n][P] (parameters() order) -> nets first.. of a region"""
        flat = torch.from_numpy(np.ascontiguousarray(flat_np, dtype=np.float32)).to(self.device)
        L.call("coevo_fc_pack", L._p(flat), self._ptr(role, region, first), flat.shape[0], ROLE_D[role])
        if region in ("pop", "stale"):
            self._dist_current = False   # (the stale-agent distances breed_device() left behind no longer describe the slab)
        return flat  # keep alive until the stream has consumed it

    def download(self, role, region, first, n):
        if region == "pop":
            self.flush_breeding()
        out = torch.zeros(n, self.P[role], dtype=torch.float32, device=self.device)
        L.call("coevo_fc_unpack", self._ptr(role, region, first), L._p(out), n, ROLE_D[role])
        return out.cpu().numpy()

    def load_initial(self, pop_flat, hof_flat):
        """pop_flat[role] [pop][P], hof_flat[role] [hof][P]; the stale agent of Q3 is the initial pop[pop-1]"""
        keep = []
        for r in ROLES:
            keep.append(self.upload(r, "pop", 0, pop_flat[r]))
            keep.append(self.upload(r, "hof", 0, hof_flat[r]))
            keep.append(self.upload(r, "stale", 0, pop_flat[r][self.pop - 1:self.pop]))
        torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------------ one generation
    def _ordinal_base(self, gen):
        return self.first_ordinal + gen * (3 * self.pop * self.hof + N_EVAL)

    def rollout(self, gen, with_prev_eval):
        """plays generation `gen`'s 3*n_local*hof games and, riding along, the 10 evaluation games of gen-1"""
        ro, M = self.ro, 3 * self.pop * self.hof
        limits = np.zeros(self.plan.n_games, dtype=np.int32)
        limits[:self.n_main] = self.T_train
        if with_prev_eval:
            limits[self.n_main:] = self.T_eval
        ro.set_limits(limits)
        base = self._ordinal_base(gen)
        per_phase = self.n_local * self.hof
        if self.env_mode == "device":
            for ph in range(3):
                ro.reset(ph * per_phase, per_phase, base + ph * self.pop * self.hof + self.lo * self.hof)
            if with_prev_eval:
                ro.reset(self.n_main, N_EVAL, self._ordinal_base(gen - 1) + M)
        else:
            ords = np.zeros(self.plan.n_games, dtype=np.int64)
            for ph in range(3):
                ords[ph * per_phase:(ph + 1) * per_phase] = (base + ph * self.pop * self.hof + self.lo * self.hof
                                                             + np.arange(per_phase))
            ords[self.n_main:] = (self._ordinal_base(gen - 1) + M + np.arange(N_EVAL)) if with_prev_eval else 0
            ro.reset_from_ordinals(ords)
        ro.run(self.n_cycles)

    def eval_only(self, gen):
        """flush: the evaluation games of generation `gen` alone (main games disabled)"""
        ro, M = self.ro, 3 * self.pop * self.hof
        limits = np.zeros(self.plan.n_games, dtype=np.int32)
        limits[self.n_main:] = self.T_eval
        ro.set_limits(limits)
        if self.env_mode == "device":
            ro.reset(0, self.n_main, 0)
            ro.reset(self.n_main, N_EVAL, self._ordinal_base(gen) + M)
        else:
            ords = np.zeros(self.plan.n_games, dtype=np.int64)
            ords[self.n_main:] = self._ordinal_base(gen) + M + np.arange(N_EVAL)
            ro.reset_from_ordinals(ords)
        ro.run((self.T_eval + 2) // 3)
        return self.eval_rewards()

    def rewards_host(self):
        r = self.ro.rewards
        return r.cpu().numpy() if torch.is_tensor(r) else r

    def eval_rewards(self):
        """mean reward triple (agent_0, agent_1, adversary_0) of the 10 evaluation games in the last rollout"""
        self.ro.check_status()
        if hasattr(self.ro, "collect_stamps"):
            self.ro.collect_stamps()  # check_status synchronised: this replay's kernel clock stamps are final
        r = self.rewards_host()[self.n_main:]
        tot = [0.0, 0.0, 0.0]
        for g in range(N_EVAL):  # python-float accumulation order of evaluate_current_weights
            for s in range(3):
                tot[s] += float(r[g, s])
        return [t / 10 for t in tot]

    def select(self):
        """fitness sharing + fitness + ranking of all three roles on the device -> elite ids stay on the device"""
        if torch.is_tensor(self.ro.rewards):
            dev_rewards = self.ro.rewards
        else:   # env on the host cores: the play_game triples come up once per generation (72 KB at cfg 2)
            if getattr(self, "_rewards_dev", None) is None:
                self._rewards_pin = torch.zeros(self.plan.n_games, 3, dtype=torch.float64).pin_memory()
                self._rewards_dev = torch.zeros(self.plan.n_games, 3, dtype=torch.float64, device=self.device)
            self._rewards_pin.numpy()[:] = self.ro.rewards
            self._rewards_dev.copy_(self._rewards_pin, non_blocking=True)
            dev_rewards = self._rewards_dev
        per_phase = self.n_local * self.hof
        if self._fused_host_tail():
            self._ensure_breed_buffers()
            if not getattr(self, "_dist_current", False):   # (generation 0, or a population loaded from the host)
                for r in ROLES:
                    L.call("coevo_fc_distance", self._ptr(r, "stale"), self._ptr(r, "pop"), self.pop, ROLE_D[r],
                           L._p(self.dist[r]))
            self._select_roles(lambda ri: L._p(dev_rewards), lambda ri: ri * per_phase, self.hof)
            return
        for ph, r in enumerate(ROLES):
            D = ROLE_D[r]
            # distances of the local shard to the stale agent (Q3), scores need every rank's distances
            if not getattr(self, "_dist_current", False):
                L.call("coevo_fc_distance", self._ptr(r, "stale"), self._ptr(r, "pop", self.lo), self.n_local, D,
                       self.dist[r].data_ptr() + 4 * self.lo)
            # last HoF game of every local individual (Q2)
            idx = ph * per_phase + torch.arange(self.n_local, device=self.device) * self.hof + self.hof - 1
            self.last_reward[ph, self.lo:self.hi] = dev_rewards[idx]
        if self.world > 1:
            self.gather(self)  # all-gather of dist[role][lo:hi] and last_reward[:, lo:hi]
        for ph, r in enumerate(ROLES):
            L.call("coevo_sharing_score", L._p(self.dist[r]), self.pop, L._p(self.div[r]))
            L.call("coevo_ga_fitness", self.last_reward[ph].data_ptr(), 0, self.pop, 1, self.hof, RET_SLOT[r],
                   L._p(self.div[r]), L._p(self.fitness[r]))
            L.call("coevo_rank_desc", L._p(self.fitness[r]), self.pop, L._p(self.order[r]))

    def breed_device(self, gen, sigmas):
        """elites -> elite buffer, HoF FIFO, population := [best] + (pop-1) mutated clones, all on the device.

        One GPU: the elites are gathered out of the population slab.  Several GPUs: a rank holds only its own shard of
        the population, so (from generation 1 on) every rank REBUILDS the elites from last generation's elites and the
        counter-based noise (coevo_fc_rebuild_elites) and materialises only the children of its own shard - breeding
        cost per GPU does not grow with the number of GPUs and no weight crosses xGMI."""
        sharded = self.world > 1
        if self._fused_host_tail():
            # (select() ran coevo_ga_select: best_dist holds the distance of each role's best individual)
            for r in ROLES:
                self.sigma[r].fill_(float(sigmas[r]))
                self.sigma_prev[r].fill_(float(sigmas[r]))
            self._promote_roles(elites_from_pop=True, best_to_pop0=True)
            for ri, r in enumerate(ROLES):
                # the children's stale-agent distances (Q3) are accumulated while they are written; individual 0 is the
                # unchanged best, whose distance is the one it had
                L.call("coevo_fc_perturb_dist", self._ptr(r, "elite"), L._p(self.parent_idx), self._ptr(r, "pop"), 1,
                       self.pop - 1, ROLE_D[r], L._p(self.sigma[r]), self.philox_seed, 0, gen * 4 + ri, 0, None,
                       self._ptr(r, "stale"), L._p(self.dist_partial[r]))
                L.call("coevo_fc_distance_finalize", L._p(self.dist_partial[r]), self.pblocks[r], self.pop - 1,
                       L._p(self.dist[r]), 1, L._p(self.best_dist[r]))
            self._dist_current = True
            return
        for ri, r in enumerate(ROLES):
            D = ROLE_D[r]
            if sharded and gen > 0:
                L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "elite_prev"), 0, self.E, D)
                L.call("coevo_fc_rebuild_elites", self._ptr(r, "elite_prev"), L._p(self.order[r]), self._ptr(r, "elite"),
                       self.E, D, L._p(self.sigma_prev[r]), self.philox_seed, (gen - 1) * 4 + ri, None)
            else:  # generation 0's population is the host-initialised one, present on every rank
                L.call("coevo_fc_gather", self._ptr(r, "pop"), L._p(self.order[r]), self._ptr(r, "elite"), 0, self.E, D)
            self.sigma[r].fill_(float(sigmas[r]))
            self.sigma_prev[r].fill_(float(sigmas[r]))
            self._hof_push(r)
            L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "pop"), 0, 1, D)
            c_lo, c_hi = (max(self.lo, 1) - 1, self.hi - 1) if sharded else (0, self.pop - 1)  # child c = individual c+1
            if c_hi > c_lo:
                L.call("coevo_fc_perturb", self._ptr(r, "elite"), self.parent_idx.data_ptr() + 4 * c_lo,
                       self._ptr(r, "pop"), 1 + c_lo, c_hi - c_lo, D, L._p(self.sigma[r]), self.philox_seed, c_lo,
                       gen * 4 + ri, 0)

    def _hof_push(self, r):
        """hof.append(best); hof.pop(0)  (genetic_algorithm.py:270-275)"""
        D = ROLE_D[r]
        if self.hof > 1:
            L.call("coevo_fc_gather", self._ptr(r, "hof"), L._p(self.hof_shift_idx), self._ptr(r, "hof_tmp"), 0,
                   self.hof - 1, D)
            L.call("coevo_fc_gather", self._ptr(r, "hof_tmp"), L._p(self.iota), self._ptr(r, "hof"), 0,
                   self.hof - 1, D)
        L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "hof"), self.hof - 1, 1, D)

    def breed_host_reference(self, env, args, sigmas):
        """mutate_elites as the reference runs it (genetic_algorithm.py:32-48): per role, per child a fresh net is
        constructed (burning the torch generator) and torch.normal noise is added to every parameter."""
        keep = []
        self._dist_current = False
        for r in ROLES:
            D = ROLE_D[r]
            L.call("coevo_fc_gather", self._ptr(r, "pop"), L._p(self.order[r]), self._ptr(r, "elite"), 0, self.E, D)
            self._hof_push(r)
            L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "pop"), 0, 1, D)
        elite_flat = {r: self.download(r, "elite", 0, self.E) for r in ROLES}
        for r in ROLES:
            children = np.empty((self.pop - 1, self.P[r]), dtype=np.float32)
            for i in range(self.pop - 1):
                child = FCNetwork(ROLE_D[r], 5, "float32")          # clone(): fresh net first
                child.set_flat(elite_flat[r][i % self.E])            # load_state_dict
                for param in child.parameters():                     # Agent.mutate
                    param.data += torch.normal(0, sigmas[r], size=param.size())
                children[i] = child.flat()
            if self.pop > 1:
                keep.append(self.upload(r, "pop", 1, children))
        torch.cuda.current_stream().synchronize()

    def elite_ids(self):
        return {r: self.order[r][:self.E].cpu().numpy().astype(int).tolist() for r in ROLES}

    # ------------------------------------------------------------------ whole generation on the device, one graph
    def setup_device_loop(self, args, capacity):
        """state for the host-free generation loop: generation counter, float64 sigma master copy + float32 copy for
        the perturb kernel, evaluation-reward and sigma histories (what the reference keeps in python lists)"""
        dev = self.device
        self.gen_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.sigma64 = torch.tensor([getattr(args, SIGMA_ATTR[r]) for r in ROLES], dtype=torch.float64, device=dev)
        self.sigma32 = self.sigma64.to(torch.float32)
        self.sigma32_prev = self.sigma32.clone()
        self.cap = int(max(capacity, 1))
        self.hist = torch.zeros(3, self.cap, dtype=torch.float64, device=dev)
        self.sig_hist = torch.zeros(3, self.cap, dtype=torch.float64, device=dev)
        self.loop_args = (float(args.min_mutation_power), float(args.max_mutation_power), 1 if args.adaptive else 0)
        self._gen_graph = None
        # distances to the stale agent: computed once for the initial population, then accumulated by the perturb
        # kernel while it writes each new child (no second pass over the 336 MB population)
        self._ensure_breed_buffers()
        for r in ROLES:
            L.call("coevo_fc_distance", self._ptr(r, "stale"), self._ptr(r, "pop"), self.pop, ROLE_D[r],
                   L._p(self.dist[r]))
        # argument blocks of the fused selection / promotion launches (one launch for the three roles)
        self.fused_tail = self.E <= 8 and self.hof <= 16 and self.pop <= 4096
        self.pipelined = os.environ.get("COEVO_PIPELINED", "1") != "0"   # breed / reset / roll out cohort by cohort

    def _ensure_breed_buffers(self):
        """partial sums of the children's stale-agent distances (fused into the perturb kernel) + the best's own distance"""
        if getattr(self, "pblocks", None) is None:
            dev = self.device
            self.pblocks = {r: int(L.load().coevo_fc_perturb_blocks(ROLE_D[r])) for r in ROLES}
            self.dist_partial = {r: torch.zeros(max(self.pop - 1, 1) * self.pblocks[r], dtype=torch.float64, device=dev)
                                 for r in ROLES}
            self.best_dist = {r: torch.zeros(1, dtype=torch.float32, device=dev) for r in ROLES}

    def _fused_host_tail(self):
        """the host-driven loop on one GPU with device-built offspring takes the device-resident loop's fused launches:
        selection of the three roles in one (coevo_ga_select), promotion in one (coevo_ga_promote), children with their
        distances fused in - 8 launches instead of ~40 per generation"""
        return (self.world == 1 and self.rng_mode == "device_philox" and self.pop > 1 and self.E <= 8 and self.hof <= 16
                and self.pop <= 4096)

    def _select_roles(self, rewards_ptr_of, game_first_of, games_per_individual):
        roles = (L.GaSelectRole * 3)()
        for ri, r in enumerate(ROLES):
            roles[ri] = L.GaSelectRole(L._p(self.dist[r]), rewards_ptr_of(ri), L._p(self.div[r]), L._p(self.fitness[r]),
                                       L._p(self.order[r]), L._p(self.best_dist[r]), game_first_of(ri), RET_SLOT[r])
        L.call("coevo_ga_select", roles, 3, self.pop, games_per_individual, self.hof)

    def _adapt_args(self, with_prev=False):
        mn, mx, adaptive = self.loop_args
        return L.GaAdaptArgs(L._p(self.ro.rewards), L._p(self.gen_dev), L._p(self.hist), L._p(self.sig_hist),
                             L._p(self.sigma64), L._p(self.sigma32), L._p(self.sigma32_prev) if with_prev else None, mn, mx,
                             self.n_main, self.cap, adaptive, 0)

    def _select_adapt_roles(self, rewards_ptr_of, game_first_of, games_per_individual, gathered=None, with_prev=False):
        """selection of the three roles + the sigma rule (evaluation means, adaptive mutation power) in ONE launch"""
        roles = (L.GaSelectRole * 3)()
        for ri, r in enumerate(ROLES):
            roles[ri] = L.GaSelectRole(None if gathered is not None else L._p(self.dist[r]),
                                       None if gathered is not None else rewards_ptr_of(ri), L._p(self.div[r]),
                                       L._p(self.fitness[r]), L._p(self.order[r]), L._p(self.best_dist[r]),
                                       0 if gathered is not None else game_first_of(ri), RET_SLOT[r])
        ad = self._adapt_args(with_prev)
        L.call("coevo_ga_select_adapt", roles, 3, self.pop, games_per_individual, self.hof,
               L._p(gathered) if gathered is not None else None, self.n_local if gathered is not None else 0, L.C.byref(ad))

    def _promote_roles(self, elites_from_pop, best_to_pop0, tick=False):
        roles = (L.GaPromoteRole * 3)()
        for ri, r in enumerate(ROLES):
            roles[ri] = L.GaPromoteRole(self._ptr(r, "pop"), self._ptr(r, "hof"), self._ptr(r, "elite"),
                                        L._p(self.order[r]), ROLE_D[r], 1 if elites_from_pop else 0,
                                        1 if best_to_pop0 else 0, 0)
        if tick:   # + the generation counter's increment: the last launch of the generation's tail
            L.call("coevo_ga_promote_tick", roles, 3, self.E, self.hof, L._p(self.gen_dev))
            return
        L.call("coevo_ga_promote", roles, 3, self.E, self.hof)

    def enqueue_generation(self):
        """reset -> 25 cycles -> rewards -> sharing/fitness/rank -> evaluation means + adaptive sigma -> HoF push and
        offspring -> generation counter tick; every launch takes the generation from the device counter, so the
        captured graph is replayed unchanged generation after generation"""
        self._enqueue_resets()
        self.ro.enqueue(self.n_cycles)
        self._enqueue_selection_and_breeding()

    def _enqueue_resets(self):
        assert self.world == 1 and self.env_mode == "device"
        ro, M = self.ro, 3 * self.pop * self.hof
        per_gen, per_phase = M + N_EVAL, self.pop * self.hof
        g = L._p(self.gen_dev)
        for ph in range(3):
            L.call("coevo_mpe_reset_gen", L._p(ro.state), self.plan.n_games, ph * per_phase, per_phase, ro.rng,
                   self.first_ordinal + ph * per_phase, g, per_gen)
        L.call("coevo_mpe_reset_gen", L._p(ro.state), self.plan.n_games, self.n_main, N_EVAL, ro.rng,
               self.first_ordinal - per_gen + M, g, per_gen)

    def _enqueue_selection_and_breeding(self, breed=True):
        """(Measured: running the three roles' chains as parallel graph branches gains 1 % in the split loop and costs
        30 % inside the single whole-generation graph - this runtime schedules branched graphs badly; kept serial.)"""
        ro, M = self.ro, 3 * self.pop * self.hof
        per_gen, per_phase = M + N_EVAL, self.pop * self.hof
        g = L._p(self.gen_dev)
        mn, mx, adaptive = self.loop_args
        if self.fused_tail:
            # selection + sigma rule in one launch, promotion (+ the counter's tick when nothing follows) in one: the tail of a
            # pipelined generation is closing step -> these two (round 4: five launches)
            self._select_adapt_roles(lambda ri: L._p(ro.rewards), lambda ri: ri * per_phase, self.hof)
            self._promote_roles(elites_from_pop=True, best_to_pop0=True, tick=not (self.pop > 1 and breed))
            if not (self.pop > 1 and breed):
                return
        else:
            for ph, r in enumerate(ROLES):
                L.call("coevo_sharing_score", L._p(self.dist[r]), self.pop, L._p(self.div[r]))
                L.call("coevo_ga_fitness", L._p(ro.rewards), ph * per_phase, self.pop, self.hof, self.hof, RET_SLOT[r],
                       L._p(self.div[r]), L._p(self.fitness[r]))
                L.call("coevo_rank_desc", L._p(self.fitness[r]), self.pop, L._p(self.order[r]))
            L.call("coevo_ga_adapt_sigma", L._p(ro.rewards), self.n_main, g, L._p(self.hist), L._p(self.sig_hist), self.cap,
                   L._p(self.sigma64), L._p(self.sigma32), mn, mx, adaptive)
        for ri, r in enumerate(ROLES):
            D = ROLE_D[r]
            if not self.fused_tail:
                L.call("coevo_fc_gather", self._ptr(r, "pop"), L._p(self.order[r]), self._ptr(r, "elite"), 0, self.E, D)
                self._hof_push(r)
                L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "pop"), 0, 1, D)
                L.call("coevo_gather_f32", L._p(self.best_dist[r]), L._p(self.dist[r]), L._p(self.order[r]), 1)
            if self.pop > 1 and breed:
                L.call("coevo_fc_perturb_dist", self._ptr(r, "elite"), L._p(self.parent_idx), self._ptr(r, "pop"), 1,
                       self.pop - 1, D, self.sigma32.data_ptr() + 4 * ri, self.philox_seed, 0, ri, 0, g,
                       self._ptr(r, "stale"), L._p(self.dist_partial[r]))
                L.call("coevo_fc_distance_finalize", L._p(self.dist_partial[r]), self.pblocks[r], self.pop - 1,
                       L._p(self.dist[r]), 1, L._p(self.best_dist[r]))
        L.call("coevo_counter_add", g, 1)

    # ------------------------------------------------------------------ pipelined generation (cohort by cohort)
    def _set_cohort_bounds(self):
        self.cohort_bounds = cohort_partition(self.n_local, self.K)[0]

    def _cohort_individuals(self, k):
        return self.lo + int(self.cohort_bounds[k]), self.lo + int(self.cohort_bounds[k + 1])

    def _breed_cohort(self, k, noise_gen, gen_dev=None, tick=False):
        """children of the individuals of cohort k (child c = individual c + 1; individual 0 is the unchanged best),
        bred from the elites of generation `noise_gen` with that generation's noise streams: the three roles in ONE
        perturb launch and one distance reduction (as six launches they queued in front of the cohort's chain).
        gen_dev: the generation comes from the device counter instead (replayable graph); tick: the counter's increment
        rides in the distance reduction (the generation's last launch)"""
        lo_k, hi_k = self._cohort_individuals(k)
        c_lo, c_hi = max(lo_k, 1) - 1, hi_k - 1
        if c_hi > c_lo:
            import ctypes as ct
            pj = (L.PerturbJob * 3)()
            fj = (L.FinalizeJob * 3)()
            for ri, r in enumerate(ROLES):
                part = self.dist_partial[r].data_ptr() + 8 * c_lo * self.pblocks[r]
                pj[ri] = L.PerturbJob(self._ptr(r, "elite"), self.parent_idx.data_ptr() + 4 * c_lo, self._ptr(r, "pop"),
                                      self.sigma32.data_ptr() + 4 * ri, self._ptr(r, "stale"), part, 1 + c_lo,
                                      c_hi - c_lo, ROLE_D[r], c_lo, ri if gen_dev is not None else noise_gen * 4 + ri, 0)
                fj[ri] = L.FinalizeJob(part, self.dist[r].data_ptr(),
                                       self.best_dist[r].data_ptr() if c_lo == 0 else None, self.pblocks[r],
                                       c_hi - c_lo, 1 + c_lo, 0)
            L.call("coevo_fc_perturb_dist_multi", ct.cast(pj, ct.c_void_p), 3, self.philox_seed, 0, gen_dev)
            if tick:
                L.call("coevo_fc_distance_finalize_multi_tick", ct.cast(fj, ct.c_void_p), 3, L._p(self.gen_dev))
            else:
                L.call("coevo_fc_distance_finalize_multi", ct.cast(fj, ct.c_void_p), 3)
            return
        if lo_k == 0:
            for r in ROLES:
                self.dist[r][0:1].copy_(self.best_dist[r])
        if tick:
            L.call("coevo_counter_add", L._p(self.gen_dev), 1)

    def _reset_cohort(self, k, gen):
        lo_k, hi_k = self._cohort_individuals(k)
        base, M = self._ordinal_base(gen), 3 * self.pop * self.hof
        per_phase = self.n_local * self.hof
        segs = [(ph * per_phase + (lo_k - self.lo) * self.hof, (hi_k - lo_k) * self.hof,
                 base + ph * self.pop * self.hof + lo_k * self.hof) for ph in range(3)]
        if k == self.K - 1 and gen > 0:
            segs.append((self.n_main, N_EVAL, self._ordinal_base(gen - 1) + M))
        self.ro.reset_segments(segs, arm=(k, self.n_cycles))   # (+ the cohort's clock stamps, when launches are timed)

    def flush_breeding(self):
        """the pipelined loop breeds generation g+1's population at the start of call g+1; anything that reads the
        population slab before that (export, tests) asks for it here"""
        if getattr(self, "_breeding_pending", False):
            torch.cuda.synchronize()
            for k in range(self.K):
                self._breed_cohort(k, self.generation_enqueued - 1)
            torch.cuda.synchronize()
            self._breeding_pending = False

    def replay_generation_pipelined(self, gen):
        """Generation `gen`, cohort by cohort: cohort k's stream breeds ITS children (deferred from the previous
        generation's selection), resets its games and runs its 25-cycle chain; the caller's stream then closes the
        rollout and replays the selection graph.  Cohort k+1 is still breeding (Philox / Box-Muller: compute-bound)
        while cohort k's chain already streams weights, and the chains start a breeding time apart, which is the stagger
        they want anyway.  The host knows `gen`; sigma, ranks and distances stay on the device."""
        assert self.world == 1 and self.env_mode == "device" and self.fused_tail and self.K > 1
        self._enqueue_cohort_chains(gen)
        if self._tail_graph is None:
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                self._enqueue_selection_and_breeding(breed=False)
            self._tail_graph = gr
        self._tail_graph.replay()
        self._breeding_pending = self.pop > 1
        self.generation_enqueued = gen + 1

    def _enqueue_cohort_chains(self, gen):
        """per cohort stream: deferred breeding of its children -> reset of its games -> its chain of merged launches;
        then, on the caller's stream, the closing step of the rollout"""
        ro = self.ro
        if gen <= 1:  # the evaluation games of "generation -1" do not exist: disabled in generation 0 only
            limits = np.zeros(self.plan.n_games, dtype=np.int32)
            limits[:self.n_main] = self.T_train
            if gen == 1:
                limits[self.n_main:] = self.T_eval
            ro.set_limits(limits)
        main = torch.cuda.current_stream()
        if getattr(self, "_cohort_streams", None) is None:
            # cohort 0 on the caller's stream, the others on the rollout context's own lane streams (streams from
            # torch's pool did not overlap with each other here: 356 vs 510 generations/s)
            self._cohort_streams = [main] + [
                torch.cuda.ExternalStream(L.load().coevo_rollout_ctx_cohort_stream(ro.ctx, k), device=self.device)
                for k in range(1, self.K)]
            self._cohort_done = [torch.cuda.Event() for _ in range(self.K)]
            self._tail_done = torch.cuda.Event()
            self._tail_graph = None
            self._breeding_pending = False
        self._tail_done.record(main)  # everything enqueued so far (the previous tail, or the initial uploads)
        for k, s in reversed(list(enumerate(self._cohort_streams))):  # the caller's stream (cohort 0) last
            if k:
                s.wait_event(self._tail_done)
            with torch.cuda.stream(s):
                if self._breeding_pending:
                    self._breed_cohort(k, gen - 1)
                self._reset_cohort(k, gen)
                ro.enqueue_cohort(k, self.n_cycles, s, armed=True)
            if k:
                self._cohort_done[k].record(s)
        self._breeding_pending = False
        for ev in self._cohort_done[1:]:
            main.wait_event(ev)
        ro.enqueue_final_step(self.n_cycles, pack=self._pack_args() if self._packed_exchange() else None)

    def _packed_exchange(self):
        """the fused exchange of a sharded generation: the rollout's closing launch writes this rank's all-gather record,
        the selection reads the gathered buffer as it is, elites are rebuilt inside the promotion launch"""
        return (getattr(self, "sharded_run", False) and self.fused_tail and self.gather_packed is not None
                and os.environ.get("COEVO_PACKED_EXCHANGE", "1") != "0")

    def _pack_args(self):
        return (self.pack_local, self.dist_all, 3, self.n_local, self.hof, self.pop, self.lo)

    def step_sharded(self, gen):
        """One generation of the population-sharded run (world > 1) without a host round trip: the same launches as
        enqueue_generation, restricted to this rank's individuals, with the fitness all-gather (stream-ordered on RCCL)
        between the rollout and the selection.  Evaluation means and the adaptive sigma rule run on the device on every
        rank alike (the 10 evaluation games are replicated), elites are rebuilt from last generation's elites, children
        are bred with their stale-agent distance fused in.  Enqueued eagerly - the host knows `gen`, only data-dependent
        values (sigma, ranks, distances) have to stay on the device.

        Launches per generation outside the rollout's n_cycles (round 5, the packed exchange): one reset, the closing step
        (+ this rank's all-gather record), ONE collective, selection, promotion (+ elite rebuild), sigma rule, children,
        their distances (+ counter tick) = 7 + the collective, where the separate form took ~32 (a rank of 8: 0.16 ms of a
        0.72 ms generation - profiles/r04_cfg2_shard_1_of_4_kernel_stats.csv)."""
        assert self.env_mode == "device" and self.rng_mode == "device_philox"
        self.sharded_run = True
        ro, M = self.ro, 3 * self.pop * self.hof
        if self.K > 1 and self.fused_tail and self.pipelined and ro.desc.merged:
            # as on one GPU: each cohort stream breeds its children (deferred), resets its games, runs its chain
            self._enqueue_cohort_chains(gen)
            self._sharded_tail(gen, breed=False)
            self._breeding_pending = self.n_local > 0 and self.pop > 1
            self.generation_enqueued = gen + 1
            return
        if gen <= 1:
            limits = np.zeros(self.plan.n_games, dtype=np.int32)
            limits[:self.n_main] = self.T_train
            if gen == 1:
                limits[self.n_main:] = self.T_eval
            ro.set_limits(limits)
        base = self._ordinal_base(gen)
        per_phase = self.n_local * self.hof
        segs = [(ph * per_phase, per_phase, base + ph * self.pop * self.hof + self.lo * self.hof) for ph in range(3)]
        if gen > 0:
            segs.append((self.n_main, N_EVAL, self._ordinal_base(gen - 1) + M))
        ro.reset_segments(segs, arm=(0, self.n_cycles))   # one launch (three phases + the evaluation games + clock stamps)
        if self._packed_exchange():   # (the closing step + this rank's all-gather record: in the persistent launch, if it is one)
            ro.enqueue(self.n_cycles, armed=True, pack=self._pack_args())
        else:
            ro.enqueue(self.n_cycles, armed=True)
        self._sharded_tail(gen, breed=True)

    def _sharded_tail(self, gen, breed):
        """all-gather of the fitness inputs -> selection -> elites (rebuilt) / HoF / best -> sigma rule -> [children]"""
        ro = self.ro
        per_phase = self.n_local * self.hof
        packed = self._packed_exchange()
        if packed:
            self.gather_packed(self)   # pack_local (written by the closing step) -> pack_all on every rank
        else:
            # last HoF game of every local individual (Q2) + its stale-agent distance -> every rank (one fused all-gather)
            if getattr(self, "_last_game_idx", None) is None:  # built once: one gather launch per generation
                one = torch.arange(self.n_local, device=self.device) * self.hof + self.hof - 1
                self._last_game_idx = torch.cat([ph * per_phase + one for ph in range(3)])
            self.last_reward[:, self.lo:self.hi] = ro.rewards[self._last_game_idx].view(3, self.n_local, 3)
            if self.world > 1:
                self.gather(self)
        # everything after the all-gather replays as one graph from generation 1 on (the elite rebuild and the children then
        # take the generation from the device counter); generation 0 (elites out of the initial population) runs eagerly
        if gen > 0 and self.fused_tail and self.ro.use_graph and (packed or not breed):
            key = (bool(breed), packed)
            graphs = self.__dict__.setdefault("_sharded_tail_graphs", {})
            if key not in graphs:
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    self._sharded_tail_post(gen, breed=breed, gen_from_device=True, packed=packed)
                graphs[key] = gr
            graphs[key].replay()
            return
        self._sharded_tail_post(gen, breed, gen_from_device=False, packed=packed)

    def _sharded_tail_post(self, gen, breed, gen_from_device, packed=False):
        ro = self.ro
        g = L._p(self.gen_dev)
        mn, mx, adaptive = self.loop_args
        if packed:
            # selection off the gathered buffer + the sigma rule (which also saves the sigma the evaluated children were bred
            # with) in one launch; promotion with the elites rebuilt from that sigma in one; then the rank's children and
            # their distances (+ the counter's tick): four launches behind the collective
            self._select_adapt_roles(None, None, 1, gathered=self.pack_all, with_prev=True)
            pr = (L.GaPromoteRole * 3)()
            for ri, r in enumerate(ROLES):
                pr[ri] = L.GaPromoteRole(self._ptr(r, "pop"), self._ptr(r, "hof"), self._ptr(r, "elite"), L._p(self.order[r]),
                                         ROLE_D[r], 1 if gen == 0 else 0, 1 if self.lo == 0 else 0, 0)
            if gen == 0:   # generation 0's population is the host-initialised one, present on every rank
                L.call("coevo_ga_promote", pr, 3, self.E, self.hof)
            else:
                L.call("coevo_ga_promote_rebuild", pr, 3, self.E, self.hof, L._p(self.sigma32_prev), self.philox_seed,
                       0 if gen_from_device else (gen - 1) * 4, g if gen_from_device else None)
            if breed:   # K = 1: the rank's children now, their distances + the counter's tick in the last launch
                self._breed_cohort(0, gen, gen_dev=g if gen_from_device else None, tick=True)
            else:
                L.call("coevo_counter_add", g, 1)
            return
        if self.fused_tail:
            self._select_roles(lambda ri: self.last_reward[ri].data_ptr(), lambda ri: 0, 1)
        else:
            for ph, r in enumerate(ROLES):
                L.call("coevo_sharing_score", L._p(self.dist[r]), self.pop, L._p(self.div[r]))
                L.call("coevo_ga_fitness", self.last_reward[ph].data_ptr(), 0, self.pop, 1, self.hof, RET_SLOT[r],
                       L._p(self.div[r]), L._p(self.fitness[r]))
                L.call("coevo_rank_desc", L._p(self.fitness[r]), self.pop, L._p(self.order[r]))
        self.sigma32_prev.copy_(self.sigma32)  # what last generation's children were bred with (elite rebuild)
        L.call("coevo_ga_adapt_sigma", L._p(ro.rewards), self.n_main, g, L._p(self.hist), L._p(self.sig_hist), self.cap,
               L._p(self.sigma64), L._p(self.sigma32), mn, mx, adaptive)
        c_lo, c_hi = max(self.lo, 1) - 1, self.hi - 1  # child c = individual c + 1
        for ri, r in enumerate(ROLES):
            D = ROLE_D[r]
            if gen > 0:
                L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "elite_prev"), 0, self.E, D)
                L.call("coevo_fc_rebuild_elites", self._ptr(r, "elite_prev"), L._p(self.order[r]), self._ptr(r, "elite"),
                       self.E, D, self.sigma32_prev.data_ptr() + 4 * ri, self.philox_seed,
                       ri if gen_from_device else (gen - 1) * 4 + ri, g if gen_from_device else None)
            elif not self.fused_tail:  # generation 0's population is the host-initialised one, present on every rank
                L.call("coevo_fc_gather", self._ptr(r, "pop"), L._p(self.order[r]), self._ptr(r, "elite"), 0, self.E, D)
            if not self.fused_tail:
                self._hof_push(r)
                L.call("coevo_gather_f32", L._p(self.best_dist[r]), L._p(self.dist[r]), L._p(self.order[r]), 1)
                if self.lo == 0:
                    L.call("coevo_fc_gather", self._ptr(r, "elite"), L._p(self.iota), self._ptr(r, "pop"), 0, 1, D)
        if self.fused_tail:
            self._promote_roles(elites_from_pop=(gen == 0), best_to_pop0=(self.lo == 0))
        for ri, r in enumerate(ROLES):
            D = ROLE_D[r]
            if not breed:
                break
            if c_hi > c_lo:
                L.call("coevo_fc_perturb_dist", self._ptr(r, "elite"), self.parent_idx.data_ptr() + 4 * c_lo,
                       self._ptr(r, "pop"), 1 + c_lo, c_hi - c_lo, D, self.sigma32.data_ptr() + 4 * ri,
                       self.philox_seed, c_lo, gen * 4 + ri, 0, None, self._ptr(r, "stale"),
                       L._p(self.dist_partial[r]))
                L.call("coevo_fc_distance_finalize", L._p(self.dist_partial[r]), self.pblocks[r], c_hi - c_lo,
                       L._p(self.dist[r]), 1 + c_lo, L._p(self.best_dist[r]) if c_lo == 0 else None)
            elif self.lo == 0:  # a shard that holds only the unchanged best
                self.dist[r][0:1].copy_(self.best_dist[r])
        L.call("coevo_counter_add", g, 1)

    def replay_generation(self, gen):
        if gen <= 1:  # the evaluation games of "generation -1" do not exist: disabled in generation 0 only
            limits = np.zeros(self.plan.n_games, dtype=np.int32)
            limits[:self.n_main] = self.T_train
            if gen == 1:
                limits[self.n_main:] = self.T_eval
            self.ro.set_limits(limits)
        if self.ro.use_graph and self.ro.n_cohorts > 1 and self.fused_tail and self.pipelined:
            return self.replay_generation_pipelined(gen)
        if self.ro.use_graph and self.ro.n_cohorts > 1:
            # cohort chains only run side by side when their launches are enqueued eagerly on their own streams (inside a
            # captured graph this runtime schedules them no better than one chain): the resets and the selection /
            # breeding tail are two small graphs, the rollout between them is one C call
            if self._gen_graph is None:
                torch.cuda.synchronize()
                parts = []
                for fn in (self._enqueue_resets, self._enqueue_selection_and_breeding):
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                        fn()
                    parts.append(gr)
                self._gen_graph = parts
            self._gen_graph[0].replay()
            self.ro.enqueue(self.n_cycles)
            self._gen_graph[1].replay()
        elif self.ro.use_graph:
            key = bool(self.ro.time_light)
            if self._gen_graph is None:
                self._gen_graph = {}
            if key not in self._gen_graph:
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    self.enqueue_generation()
                self._gen_graph[key] = gr
                # the capture did not execute anything: the counter still holds `gen`
            self._gen_graph[key].replay()
            if self.ro.time_light:
                self.ro._pending_stamps = self.n_cycles  # this replay re-armed and re-wrote the clock stamps
        else:
            self.enqueue_generation()


# --------------------------------------------------------------------------------------------- trainer
class GAResult:
    """what the reference only plots: per-generation evaluation rewards, fitness, elite ids, sigma history"""

    def __init__(self):
        self.rewards = {r: [] for r in ROLES}
        self.fitness, self.elite_ids, self.diversity, self.sigma_after = [], [], [], []
        self.game_rewards = []   # per generation: play_game triples of the 3*pop*hof training games, reference order
        self.seconds = []


def initial_population(env, args):
    """creation order of genetic_algorithm.py:63-68 and :110-117 (torch RNG order matters for parity)"""
    hof_n, pop = args.hof_size, args.population
    hof = {"agent_1": [create_agent(env, args, "agent_1") for _ in range(hof_n)]}
    hof["agent_0"] = [create_agent(env, args, "agent_0") for _ in range(hof_n)]
    hof["adversary_0"] = [create_agent(env, args, "adversary_0") for _ in range(hof_n)]
    for role in ("agent_1", "agent_0", "adversary_0"):  # placeholder elites, overwritten in generation 0 (Q7)
        for _ in range(hof_n):
            create_agent(env, args, role)
    popu = {r: [] for r in ROLES}
    for _ in range(pop):
        for r in ROLES:
            popu[r].append(create_agent(env, args, r))
    pop_flat = {r: np.stack([a.model.flat() for a in popu[r]]) for r in ROLES}
    hof_flat = {r: np.stack([a.model.flat() for a in hof[r]]) for r in ROLES}
    return pop_flat, hof_flat


class GATrainer:
    """The generation loop of genetic_algorithm_train as an object, one ``step()`` per generation (bench.py times
    exactly these steps)."""

    def __init__(self, env, args, rng=None, env_mode=None, collect=True, dist_ctx=None):
        self.env, self.args, self.collect = env, args, collect
        self.rng = rng or getattr(args, "coevo_rng", "host_reference")
        env_mode = env_mode or getattr(args, "coevo_env", "device")
        self.first_ordinal = getattr(env, "n_resets", 1)
        pop_flat, hof_flat = initial_population(env, args)
        shard, gather, gather_packed = (0, 1), None, None
        if dist_ctx is not None and dist_ctx.world > 1:
            shard, gather = (dist_ctx.rank, dist_ctx.world), dist_ctx.gather_ga
            gather_packed = getattr(dist_ctx, "gather_ga_packed", None)
        # the host-free generation loop needs device-built offspring, the device env and a single rank
        self.device_loop = (self.rng == "device_philox" and env_mode == "device" and shard == (0, 1)
                            and getattr(args, "coevo_device_loop", True))
        # cohort chains overlap only when enqueued eagerly (GAEngine.replay_generation): worth it in the host-free loop
        # ... and its population-sharded counterpart (GAEngine.step_sharded)
        force = bool(getattr(args, "coevo_force_sharded_loop", False))  # measurement: the N>1 code path on one GPU
        self.sharded_loop = (self.rng == "device_philox" and env_mode == "device" and (shard != (0, 1) or force)
                             and getattr(args, "coevo_device_loop", True))
        if self.sharded_loop:
            self.device_loop = False
        cohorts = getattr(args, "coevo_cohorts", None)
        if cohorts is None:
            cohorts = DEVICE_LOOP_COHORTS if (self.device_loop or self.sharded_loop) else DEFAULT_COHORTS
            # a small shard (pop 200 over 4 / 8 GPUs: 50 / 25 individuals per role) is ONE launch per env-cycle: its few
            # hundred workgroups are all resident anyway, and a second chain only adds launches (one rank of 4: 1312 vs 1120,
            # of 8: 1392 vs 1167 generations/s; 100 individuals per role: 850 vs 890 - profiles/r04_experiments.md)
            if args.population // max(shard[1], 1) < SMALL_SHARD:
                cohorts = 1
        self.eng = GAEngine(args.population, args.hof_size, args.elites_number, args.max_timesteps_per_episode,
                            args.max_evaluation_steps, max_cycles=getattr(env, "max_cycles", 25), rng=self.rng,
                            philox_seed=getattr(args, "coevo_seed", 0), env=env_mode,
                            first_ordinal=self.first_ordinal,
                            env_seed=getattr(env, "seed_value", ENV_SEED) or ENV_SEED, shard=shard, gather=gather,
                            cohorts=int(cohorts), gather_packed=gather_packed)
        self.eng.load_initial(pop_flat, hof_flat)
        self.res = GAResult()
        self.res.engine = self.eng
        self.gen = 0
        self.stamp_every = 1
        if self.device_loop or self.sharded_loop:
            self.eng.setup_device_loop(args, capacity=max(getattr(args, "generations", 0), 1) + 64)

    def step(self):
        """generation self.gen: play its games (+ the previous generation's 10 evaluation games), select, breed"""
        eng, args, res, gen = self.eng, self.args, self.res, self.gen
        t0 = time.perf_counter()
        if self.device_loop or self.sharded_loop:
            if gen > eng.cap:
                raise RuntimeError(f"generation {gen} exceeds the device history capacity ({eng.cap}) this trainer was "
                                   f"set up with (args.generations + 64): the adaptive mutation power could no longer "
                                   f"be updated on the device")
            # no host round trip: evaluation means, adaptive sigma, selection and offspring all stay on the device;
            # the host only enqueues (replays) the generation and may run ahead of the GPU
            if self.sharded_loop:
                eng.step_sharded(gen)
            else:
                eng.replay_generation(gen)
            if self.collect:
                self._collect_device_loop(gen)
            elif eng.ro.time_light and gen % self.stamp_every == 0:
                # kernel-timing sample: the stamps are written by every replay (switching between a stamped and a plain
                # graph costs more than the stamps), but reading them back needs a host sync: sampled generations only
                torch.cuda.current_stream().synchronize()
                eng.ro.collect_stamps()
            res.seconds.append(time.perf_counter() - t0)
            self.gen += 1
            return
        eng.rollout(gen, with_prev_eval=gen > 0)
        if gen > 0:
            _finish_generation(args, gen - 1, eng.eval_rewards(), res)
        eng.select()
        sigmas = {r: getattr(args, SIGMA_ATTR[r]) for r in ROLES}
        if self.rng == "host_reference":
            eng.breed_host_reference(self.env, args, sigmas)
        else:
            eng.breed_device(gen, sigmas)
        if self.collect:
            res.game_rewards.append(eng.rewards_host()[:eng.n_main].copy())
            res.fitness.append([eng.fitness[r].cpu().numpy().tolist() for r in ROLES])
            res.diversity.append([float(eng.div[r].item()) for r in ROLES])
            res.elite_ids.append([eng.elite_ids()[r] for r in ROLES])
        res.seconds.append(time.perf_counter() - t0)
        self.gen += 1

    def _collect_device_loop(self, gen):
        """Per-generation results (game rewards, fitness, diversity, elite ids, status word) without draining the
        pipeline: asynchronous copies into one of two pinned host slots, enqueued behind generation `gen` on the caller's
        stream (so before anything of generation gen+1 can overwrite the sources), and the PREVIOUS generation's slot -
        long complete - is harvested into the result.  The result therefore lags one generation until finish()."""
        eng = self.eng
        if getattr(self, "_slots", None) is None:
            pin = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype).pin_memory()
            self._slots = [{"rewards": pin(eng.n_main, 3, dtype=torch.float64),
                            "fitness": pin(3, eng.pop, dtype=torch.float32), "div": pin(3, dtype=torch.float32),
                            "elite": pin(3, eng.E, dtype=torch.int32), "status": pin(1, dtype=torch.int32),
                            "event": torch.cuda.Event(), "gen": -1} for _ in range(2)]
        slot = self._slots[gen % 2]
        slot["rewards"].copy_(eng.ro.rewards[:eng.n_main], non_blocking=True)
        for i, r in enumerate(ROLES):
            slot["fitness"][i].copy_(eng.fitness[r], non_blocking=True)
            slot["div"][i:i + 1].copy_(eng.div[r], non_blocking=True)
            slot["elite"][i].copy_(eng.order[r][:eng.E], non_blocking=True)
        slot["status"].copy_(eng.ro.status, non_blocking=True)
        slot["event"].record()
        slot["gen"] = gen
        if gen > 0:
            self._harvest(self._slots[(gen - 1) % 2])

    def _harvest(self, slot):
        if slot["gen"] < 0:
            return
        slot["event"].synchronize()
        res = self.res
        st = int(slot["status"][0])
        if st:
            L.raise_on_status(slot["status"])
        res.game_rewards.append(slot["rewards"].numpy().copy())
        res.fitness.append([slot["fitness"][i].numpy().tolist() for i in range(3)])
        res.diversity.append([float(slot["div"][i]) for i in range(3)])
        res.elite_ids.append([slot["elite"][i].numpy().astype(int).tolist() for i in range(3)])
        slot["gen"] = -1

    def _sync_history_from_device(self, upto):
        """mirror the device-resident histories into the result and the (mutable, reference-style) args bag"""
        eng, res, args = self.eng, self.res, self.args
        hist = eng.hist[:, :upto].cpu().numpy()
        sig = eng.sig_hist[:, :upto].cpu().numpy()
        for s, r in enumerate(ROLES):
            res.rewards[r] = [float(x) for x in hist[s]]
        res.sigma_after = [[float(sig[0, e]), float(sig[1, e]), float(sig[2, e])] for e in range(upto)]
        cur = eng.sigma64.cpu().numpy()
        args.mutation_power_agent_0, args.mutation_power_agent_1, args.mutation_power_adversary = (
            float(cur[0]), float(cur[1]), float(cur[2]))

    def finish(self):
        """flush the last generation's evaluation games and leave the env's reset counter where the reference would"""
        if self.gen > 0:
            if self.device_loop or self.sharded_loop:
                self.eng.flush_breeding()
                torch.cuda.synchronize()
                self.eng.ro.check_status()
                if self.collect and getattr(self, "_slots", None):  # the lagging last generation
                    self._harvest(self._slots[(self.gen - 1) % 2])
                self.eng.ro.collect_stamps()
                self._sync_history_from_device(upto=self.gen - 1)
            _finish_generation(self.args, self.gen - 1, self.eng.eval_only(self.gen - 1), self.res)
        if hasattr(self.env, "n_resets"):
            self.env.n_resets = self.first_ordinal + self.gen * (3 * self.args.population * self.args.hof_size + N_EVAL)
        return self.res


def genetic_algorithm_train(env, agent, args, output_dir, rng=None, env_mode=None, collect=True, dist_ctx=None):
    """Drop-in for genetic_algorithm.py:51 (``agent`` is unused there too).  Returns a GAResult (the reference returns
    None and plots instead).

    rng: "host_reference" (default) reproduces the reference's torch RNG stream bit for bit;
         "device_philox" builds offspring on the device (args.coevo_rng may select it as well)."""
    if getattr(args, "game", "simple_adversary_v3") != "simple_adversary_v3":   # the two-player Atari games
        from .dqn_population import dqn_genetic_algorithm_train
        return dqn_genetic_algorithm_train(env, agent, args, output_dir, collect=collect, dist_ctx=dist_ctx)
    from .io_utils import GA_FILES, MetricsWriter, agents_from_flat, save_model, save_state_dicts
    tr = GATrainer(env, args, rng=rng, env_mode=env_mode, collect=collect, dist_ctx=dist_ctx)
    save = bool(getattr(args, "save", False)) and output_dir is not None
    mw = MetricsWriter(output_dir)
    written = 0

    def flush_metrics(res, upto):
        """one line per generation as soon as its numbers exist on the host (a crash keeps what was written): the
        host-driven loop knows generation g-1's evaluation after step g; the host-free device loop harvests a
        generation's results one step later and keeps the evaluation / sigma histories on the device until finish()"""
        nonlocal written
        while written < upto:
            g = written
            have = lambda lst: lst[g] if g < len(lst) else None
            mw.write(generation=g,
                     eval_rewards={r: res.rewards[r][g] for r in ROLES} if g < len(res.rewards["agent_0"]) else None,
                     mutation_power=have(res.sigma_after), diversity=have(res.diversity), elite_ids=have(res.elite_ids),
                     fitness_best=[max(f) for f in res.fitness[g]] if g < len(res.fitness) else None,
                     seconds=have(res.seconds))
            written += 1

    for _ in range(args.generations):
        tr.step()
        if save:  # genetic_algorithm.py:293-299: HoF and elites of the three roles, every generation
            for r in ROLES:
                hof_file, elite_file = GA_FILES[r]
                for region, n, fname in (("hof", args.hof_size, hof_file), ("elite", args.elites_number, elite_file)):
                    agents = agents_from_flat(env, args, r, tr.eng.download(r, region, 0, n))
                    save_model(agents, os.path.join(output_dir, fname))
                    save_state_dicts(agents, os.path.join(output_dir, fname), role=r)   # the weights_only-safe twin
        if tr.device_loop or tr.sharded_loop:
            flush_metrics(tr.res, min(len(tr.res.elite_ids), tr.gen - 1))   # harvested so far; final values at finish()
        else:
            flush_metrics(tr.res, len(tr.res.rewards["agent_0"]))
    res = tr.finish()
    if tr.device_loop or tr.sharded_loop:  # the histories only now left the device: rewrite the file with them filled in
        mw = MetricsWriter(output_dir)
        written = 0
    flush_metrics(res, len(res.rewards["agent_0"]))
    return res


def _finish_generation(args, gen, eval_triple, res):
    for s, r in enumerate(ROLES):
        res.rewards[r].append(eval_triple[s])
    if args.adaptive:
        adapt_mutation_power(args, gen, res.rewards)
    res.sigma_after.append([args.mutation_power_agent_0, args.mutation_power_agent_1, args.mutation_power_adversary])
