"""Co-ES on the MI355X: drop-in ``evolution_strategy_train(env, args, output_dir)`` (reference
evolutionary_strategy.py:151) over the batched rollout engine.

Reference semantics kept: per generation ``pop`` iterations x 3 roles, each one
perturbed copy of the role's base net (Linear weights/biases only, LayerNorm untouched) playing ONE game against the two
current base nets, in the interleaved order agent_0, agent_1, adversary_0 (game ordinal 3j+role); the role's reward slot
(subject to quirk Q1) is the raw fitness - no rank transform, no antithetic pairs; update theta += lr/(n*sigma) *
sum_i f_i eps_i; optional fitness sharing f/(1+D); 10 evaluation games of the updated trio; adaptive sigma (Q5);
early stopping.  hof_size is unused, as in the reference (Q12).

Two things beyond the reference, both optional and labelled wherever they are reported:
  * the cfg 3 EXTENSION mode of BASELINE.json configs[2] (``args.coevo_antithetic``, ``args.coevo_centered_rank``;
    device_philox only): antithetic pairs (individuals 2m / 2m+1 share one noise stream with opposite signs) and the
    centered-rank transform where the reference has its own normalisation commented out (:133-135).  OFF by default.
  * population sharding over several GPUs: rank r perturbs, plays and partially sums individuals
    [lo, hi); one all-gather of (reward, distance) per individual, one all-gather of the chunk partial sums (the
    update is defined as ES_CHUNKS chunk sums added left to right, so N ranks reproduce one rank bit for bit); the 10
    evaluation games are replicated.

rng modes
  "host_reference"  noise from the global numpy generator in the reference's call order, perturbed nets and the update
                    computed on the host exactly as the reference does (fp64 noise, fp32 GEMV) and uploaded - parity;
  "device_philox"   perturbed nets are materialised once per generation on the device from counter-based noise and the
                    update regenerates the same noise (no noise matrix is ever stored: the reference keeps n x P).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import lib as L
from .game_logic import create_agent
from .genetic_algorithm import N_EVAL, RET_SLOT, ROLE_D, ROLES, SIGMA_ATTR, adapt_mutation_power
from .mpe.simple_adversary import ENV_SEED
from .rollout import DeviceRollout, HostEnvRollout, RolloutPlan, effective_steps


# ---- the reference's per-call helpers under their own names (evolutionary_strategy.py:11-148): sequential, on the host
# except for the games.  The trainer below does the same work batched on the device.
def get_numpy_dtype(precision):
    if precision == "float32":
        return np.float32
    raise ValueError(f"Unsupported precision: {precision}")   # (float16 is out of the parity scope)


def evaluate_current_weights(agent_0, agent_1, adversary, env, args):
    from .genetic_algorithm import evaluate_current_weights as ga_eval
    return ga_eval(agent_0, agent_1, adversary, env, args)


def mutate_weights(env, agent_0, agent_1, adversary, args, role, step, weights_logging_agent_0, weights_logging_agent_1,
                   weights_logging_adversary):
    """one ES individual of `role`: clone the role's base agent, perturb its Linear layers (Agent.mutate_ES), play ONE game
    against the two other base nets -> (the role's slot of play_game's triple, noise, get_weights_ES() of the mutated net)
    (evolutionary_strategy.py:63-116)"""
    from .game_logic import play_game
    base = {"agent_0": agent_0, "agent_1": agent_1, "adversary_0": adversary}
    if role not in base:
        return None
    mutated = base[role].clone(env, args, role=role)
    noise = mutated.mutate_ES(args, role=role, step=step, weights_logging_agent_0=weights_logging_agent_0,
                              weights_logging_agent_1=weights_logging_agent_1,
                              weights_logging_adversary=weights_logging_adversary).astype(get_numpy_dtype(args.precision))
    trio = dict(base)
    trio[role] = mutated
    r = play_game(env=env, player1=trio["agent_0"].model, player2=trio["agent_1"].model,
                  adversary=trio["adversary_0"].model, args=args)
    return r[RET_SLOT[role]], noise, mutated.model.get_weights_ES()


def compute_weight_update(noises, rewards, args, role, individual_weights=None, population_weights=None):
    """(lr / (n sigma_role)) * noises^T fitness, fitness = rewards or rewards / (1 + diversity) -> (update, diversity)
    (evolutionary_strategy.py:120-148: raw rewards, the reference's own normalisation is commented out there)"""
    from .game_logic import diversity_penalty
    dt = get_numpy_dtype(args.precision)
    noises, rewards = np.array(noises, dtype=dt), np.array(rewards, dtype=dt)
    diversity = None
    fitness = rewards
    if args.fitness_sharing:
        diversity = diversity_penalty(individual_weights=individual_weights, population_weights=population_weights, args=args)
        fitness = rewards / (1 + diversity)
    sigma = getattr(args, SIGMA_ATTR[role])
    update = (args.learning_rate / (len(noises) * sigma)) * np.dot(noises.T, fitness)
    return update.astype(dt), diversity


ES_CHUNKS = 8   # the canonical ES summation: this many chunk sums, added left to right (include/coevo.h, K5)


class ESEngine:
    def __init__(self, pop, limit_train=None, limit_eval=None, max_cycles=25, device="cuda", env_seed=ENV_SEED,
                 rng="device_philox", philox_seed=0, env="device", first_ordinal=1, shard=(0, 1), gather=None,
                 antithetic=False, centered_rank=False, chunks=ES_CHUNKS):
        self.pop, self.rng_mode, self.philox_seed, self.env_mode, self.device = pop, rng, int(philox_seed), env, device
        self.rank, self.world = shard
        self.gather = gather
        self.antithetic, self.centered_rank, self.chunks = bool(antithetic), bool(centered_rank), int(chunks)
        if self.world > 1 and (pop % self.world or self.chunks % self.world):
            raise ValueError(f"population {pop} and the {self.chunks} update chunks must both be divisible by the "
                             f"number of ranks {self.world}")
        if (self.antithetic or self.centered_rank or self.world > 1) and rng != "device_philox":
            raise ValueError("the extension mode and the sharded run need device_philox offspring")
        if self.antithetic and pop % 2:
            raise ValueError("antithetic pairs need an even population")
        self.lo, self.hi = self.rank * pop // self.world, (self.rank + 1) * pop // self.world
        self.n_local = self.hi - self.lo
        self.T_train = effective_steps(limit_train, max_cycles)
        self.T_eval = effective_steps(limit_eval, max_cycles)
        self.n_cycles = (max(self.T_train, self.T_eval) + 2) // 3
        self.first_ordinal = first_ordinal
        self.stride = {r: L.fc_slab_stride(ROLE_D[r]) for r in ROLES}
        self.P = {r: L.fc_param_count(ROLE_D[r]) for r in ROLES}
        self.base, off = {}, 0
        for r in ROLES:
            self.base[r] = {"base": off, "pert": off + self.stride[r]}
            off += (1 + self.n_local) * self.stride[r]
        self.slab = torch.zeros(off, dtype=torch.float32, device=device)
        net_off, net_D, ids = [], [], {}

        def net(region, role, i=0):
            key = (region, role, i)
            if key not in ids:
                ids[key] = len(net_off)
                net_off.append(self.base[role][region] + i * self.stride[role])
                net_D.append(ROLE_D[role])
            return ids[key]

        games = []
        for j in range(self.n_local):  # evolutionary_strategy.py:236-251: mutate_weights for agent_0, agent_1, adversary_0
            for r in ROLES:
                seat = {q: net("base", q) for q in ROLES}
                seat[r] = net("pert", r, j)
                games.append((seat["adversary_0"], seat["agent_0"], seat["agent_1"]))
        self.n_main = len(games)
        # Unlike Co-GA, the evaluation games cannot ride in the next generation's launch: generation g+1 perturbs with
        # sigma_{g+1}, which the adaptive rule derives from generation g's evaluation (evolutionary_strategy.py:272-316).
        # They get their own 10-game rollout after each update.
        eval_games = [(net("base", "adversary_0"), net("base", "agent_0"), net("base", "agent_1"))] * N_EVAL
        cls = DeviceRollout if env == "device" else HostEnvRollout
        heavy_rows = int(os.environ.get("COEVO_HEAVY_ROWS", "32"))
        # device env: 2 cohorts (114 vs 109 generations/s at cfg3); env on the host cores: COEVO_HOST_COHORTS alternating
        # cohorts (default 4), rows numbered cohort by cohort (one contiguous observation / action range each)
        es_cohorts = (int(os.environ.get("COEVO_ES_COHORTS", "2")) if env == "device"
                      else int(os.environ.get("COEVO_HOST_COHORTS", "4")))
        game_cohort = None
        if env != "device" and es_cohorts > 1 and len(games) >= es_cohorts:
            # contiguous game ranges (a core then owns whole cache lines of the struct-of-arrays game state)
            game_cohort = (np.arange(len(games)) * es_cohorts // len(games)).astype(np.int32)
        self.plan = RolloutPlan(np.array(games), net_off, net_D, device=device, heavy_rows=heavy_rows,
                                n_cohorts=es_cohorts, game_cohort=game_cohort,
                                row_order="class" if env == "device" else "cohort")
        self.ro = cls(self.plan, self.slab, env_seed=env_seed)
        # the 10 evaluation games: three nets x 10 rows, as two 5-row streaming tasks per net (1.0 -> 0.5 ms)
        self.eval_plan = RolloutPlan(np.array(eval_games), net_off, net_D, device=device, split_rows=5)
        self.eval_ro = cls(self.eval_plan, self.slab, env_seed=env_seed)
        f32 = dict(dtype=torch.float32, device=device)
        self.fitness = {r: torch.zeros(pop, **f32) for r in ROLES}
        self.raw = {r: torch.zeros(pop, **f32) for r in ROLES}
        self.dist_local = {r: torch.zeros(max(self.n_local, 1), **f32) for r in ROLES}
        self.div = {r: torch.zeros(1, **f32) for r in ROLES}
        self.sigma = {r: torch.zeros(1, **f32) for r in ROLES}
        self.zero_idx = torch.zeros(max(self.n_local, 1), dtype=torch.int32, device=device)
        self.game_idx = torch.stack([torch.arange(self.n_local, device=device) * 3 + ri for ri in range(3)])
        self.ret_slot = torch.tensor([RET_SLOT[r] for r in ROLES], device=device)
        # per individual (reward in the role's slot, distance to the base net), all roles: what the ranks exchange
        self.stats = torch.zeros(3, pop, 2, dtype=torch.float64, device=device)
        # chunk partial sums of the update, rank-major: [world][role][chunks/world][stride_role]
        self.chunks_local = self.chunks // self.world
        self.part_off, o = {}, 0
        for r in ROLES:
            self.part_off[r] = o
            o += self.chunks_local * self.stride[r]
        self.part_block = o
        self.partials = torch.zeros(self.world * self.part_block, **f32)
        self.steps_per_generation = 3 * pop * self.T_train + N_EVAL * self.T_eval

    def _ptr(self, role, region, i=0):
        return self.slab.data_ptr() + 4 * (self.base[role][region] + i * self.stride[role])

    def upload(self, role, region, first, flat_np):
        flat = torch.from_numpy(np.ascontiguousarray(flat_np, dtype=np.float32)).to(self.device)
        L.call("coevo_fc_pack", L._p(flat), self._ptr(role, region, first), flat.shape[0], ROLE_D[role])
        return flat

    def download(self, role, region, first, n):
        out = torch.zeros(n, self.P[role], dtype=torch.float32, device=self.device)
        L.call("coevo_fc_unpack", self._ptr(role, region, first), L._p(out), n, ROLE_D[role])
        return out.cpu().numpy()

    def _ordinal_base(self, gen):
        return self.first_ordinal + gen * (3 * self.pop + N_EVAL)

    def perturb_device(self, gen, sigmas):
        """this rank's perturbed nets: individual j (global index) of role ri takes noise stream (j, 4*gen + ri)"""
        flags = 1 | (2 if self.antithetic else 0)   # LayerNorm untouched; antithetic pairs in the extension mode
        for ri, r in enumerate(ROLES):
            self.sigma[r].fill_(float(sigmas[r]))
            L.call("coevo_fc_perturb_flags", self._ptr(r, "base"), L._p(self.zero_idx), self._ptr(r, "pert"), 0,
                   self.n_local, ROLE_D[r], L._p(self.sigma[r]), self.philox_seed, self.lo, gen * 4 + ri, flags)

    def rollout(self, gen):
        """this rank's 3*n_local training games of generation `gen` (game ordinal 3j + role in the seeded stream)"""
        ro = self.ro
        ro.set_limits(np.full(self.plan.n_games, self.T_train, dtype=np.int32))
        first = self._ordinal_base(gen) + 3 * self.lo
        if self.env_mode == "device":
            ro.reset(0, self.n_main, first)
        else:
            ro.reset_from_ordinals(first + np.arange(self.n_main))
        if getattr(ro, "n_cohorts", 1) > 1:
            ro.enqueue((self.T_train + 2) // 3)  # cohort chains overlap only when enqueued eagerly
        else:
            ro.run((self.T_train + 2) // 3)

    def evaluate(self, gen):
        """evaluate_current_weights: 10 games of the current base trio -> mean reward triple (:22-59, :272)"""
        ro = self.eval_ro
        ro.set_limits(np.full(N_EVAL, self.T_eval, dtype=np.int32))
        first = self._ordinal_base(gen) + 3 * self.pop
        if self.env_mode == "device":
            ro.reset(0, N_EVAL, first)
        else:
            ro.reset_from_ordinals(first + np.arange(N_EVAL))
        ro.run((self.T_eval + 2) // 3)
        ro.check_status()
        r = ro.rewards.cpu().numpy() if torch.is_tensor(ro.rewards) else ro.rewards
        tot = [0.0, 0.0, 0.0]
        for g in range(N_EVAL):
            for s in range(3):
                tot[s] += float(r[g, s])
        return [t / 10 for t in tot]

    def rewards_host(self):
        r = self.ro.rewards
        return r.cpu().numpy() if torch.is_tensor(r) else r

    def update_device(self, gen, lr, fitness_sharing):
        """compute_weight_update (evolutionary_strategy.py:120-148) + base += update, on the device.  Sharded: the
        (reward, distance) pairs and then the chunk partial sums are all-gathered; every rank applies the same update."""
        rew = self.ro.rewards if torch.is_tensor(self.ro.rewards) else torch.from_numpy(self.ro.rewards).to(self.device)
        lo, hi = self.lo, self.hi
        self.stats[:, lo:hi, 0] = rew[self.game_idx, self.ret_slot[:, None]]
        if fitness_sharing:
            for ri, r in enumerate(ROLES):
                L.call("coevo_fc_distance", self._ptr(r, "base"), self._ptr(r, "pert"), self.n_local, ROLE_D[r],
                       L._p(self.dist_local[r]))
                self.stats[ri, lo:hi, 1] = self.dist_local[r][:self.n_local]
        if self.world > 1:
            self.gather(self, "stats")
        for ri, r in enumerate(ROLES):
            self.raw[r].copy_(self.stats[ri, :, 0])                  # np.array(rewards, dtype=float32)
            if fitness_sharing:
                d = self.stats[ri, :, 1].to(torch.float32).contiguous()
                L.call("coevo_sharing_score", L._p(d), self.pop, L._p(self.div[r]))
                self.raw[r].div_(1.0 + self.div[r])
            if self.centered_rank:
                L.call("coevo_centered_ranks", L._p(self.raw[r]), self.pop, L._p(self.fitness[r]))
            else:
                self.fitness[r].copy_(self.raw[r])
            L.call("coevo_es_partial", self._ptr(r, "base"), self._ptr(r, "pert"), lo, ROLE_D[r],
                   L._p(self.fitness[r]), self.pop, self.chunks, self.rank * self.chunks_local, self.chunks_local,
                   self.partials.data_ptr() + 4 * (self.rank * self.part_block + self.part_off[r]))
        if self.world > 1:
            self.gather(self, "partials")
        for ri, r in enumerate(ROLES):
            L.call("coevo_es_apply", self._ptr(r, "base"), self.partials.data_ptr() + 4 * self.part_off[r], self.chunks,
                   self.chunks_local, self.part_block, ROLE_D[r], self.pop, L._p(self.sigma[r]), L.C.c_float(lr))


class ESResult:
    def __init__(self):
        self.rewards = {r: [] for r in ROLES}
        self.diversity, self.sigma_after, self.game_rewards, self.seconds = [], [], [], []
        self.stopped_at = None


def _linear_mask(D):
    """boolean mask over the flat parameter vector: True on Linear weights/biases (the perturbable entries)"""
    from .fcnetwork import param_shapes
    m = []
    for name, shp in param_shapes(D):
        m.append(np.full(int(np.prod(shp)), not name.startswith("ln"), dtype=bool))
    return np.concatenate(m)


class ESTrainer:
    def __init__(self, env, args, rng=None, env_mode=None, collect=True, dist_ctx=None):
        self.env, self.args, self.collect = env, args, collect
        self.rng = rng or getattr(args, "coevo_rng", "host_reference")
        env_mode = env_mode or getattr(args, "coevo_env", "device")
        self.first_ordinal = getattr(env, "n_resets", 1)
        agents = {r: create_agent(env, args, role=r) for r in ROLES}  # evolutionary_strategy.py:163-165
        shard, gather = (0, 1), None
        if dist_ctx is not None and dist_ctx.world > 1:
            shard, gather = (dist_ctx.rank, dist_ctx.world), dist_ctx.gather_es
        self.eng = ESEngine(args.population, args.max_timesteps_per_episode, args.max_evaluation_steps,
                            max_cycles=getattr(env, "max_cycles", 25), rng=self.rng,
                            philox_seed=getattr(args, "coevo_seed", 0), env=env_mode,
                            first_ordinal=self.first_ordinal,
                            env_seed=getattr(env, "seed_value", ENV_SEED) or ENV_SEED, shard=shard, gather=gather,
                            antithetic=getattr(args, "coevo_antithetic", False),
                            centered_rank=getattr(args, "coevo_centered_rank", False),
                            chunks=getattr(args, "coevo_es_chunks", ES_CHUNKS))
        keep = [self.eng.upload(r, "base", 0, agents[r].model.flat()[None]) for r in ROLES]
        torch.cuda.current_stream().synchronize()
        self.base_flat = {r: agents[r].model.flat().copy() for r in ROLES}  # host copy (host_reference mode)
        self.mask = {r: _linear_mask(ROLE_D[r]) for r in ROLES}
        self.res = ESResult()
        self.gen = 0
        self.best = {r: -float("inf") for r in ROLES}
        self.no_improve = {r: 0 for r in ROLES}

    # -- host_reference: the reference's own numpy calls, in its order ------------------------------------------
    def _perturb_host(self):
        args, eng, pop = self.args, self.eng, self.args.population
        self.noises = {r: np.empty((pop, int(self.mask[r].sum())), dtype=np.float32) for r in ROLES}
        pert = {r: np.repeat(self.base_flat[r][None], pop, axis=0) for r in ROLES}
        for j in range(pop):
            for r in ROLES:
                w = self.base_flat[r][self.mask[r]]
                noise = np.random.normal(loc=0.0, scale=getattr(args, SIGMA_ATTR[r]), size=len(w))  # agent.py:52
                pert[r][j, self.mask[r]] = (w + noise).astype(np.float32)   # torch.tensor(..., float32)
                self.noises[r][j] = noise.astype(np.float32)                # evolutionary_strategy.py:74
        keep = [eng.upload(r, "pert", 0, pert[r]) for r in ROLES]
        torch.cuda.current_stream().synchronize()
        self.pert_host = pert

    def _update_host(self):
        args, eng = self.args, self.eng
        rew = eng.rewards_host()
        divs = []
        for ri, r in enumerate(ROLES):
            f = np.array([rew[3 * j + ri, RET_SLOT[r]] for j in range(args.population)], dtype=np.float32)
            div = None
            if args.fitness_sharing:
                base_w = self.base_flat[r][self.mask[r]]
                d = np.array([np.linalg.norm(p[self.mask[r]] - base_w) for p in self.pert_host[r]])
                div = np.sum(np.maximum(0, 1 - d / np.mean(d)))
                f = f / (1 + div)
            sigma = getattr(args, SIGMA_ATTR[r])
            upd = (args.learning_rate / (len(self.noises[r]) * sigma)) * np.dot(self.noises[r].T, f)
            new = self.base_flat[r].copy()
            new[self.mask[r]] = self.base_flat[r][self.mask[r]] + upd.astype(np.float32)
            self.base_flat[r] = new
            divs.append(None if div is None else float(div))
        keep = [eng.upload(r, "base", 0, self.base_flat[r][None]) for r in ROLES]
        torch.cuda.current_stream().synchronize()
        return divs

    def step(self):
        """one generation in the reference's order: perturb (sigma_g) -> 3*pop games -> update -> 10 evaluation games
        -> adaptive sigma / early stopping.  Returns False when early stopping fired."""
        eng, args, res, gen = self.eng, self.args, self.res, self.gen
        t0 = time.perf_counter()
        sigmas = {r: getattr(args, SIGMA_ATTR[r]) for r in ROLES}
        if self.rng == "host_reference":
            self._perturb_host()
        else:
            eng.perturb_device(gen, sigmas)
        eng.rollout(gen)
        if self.rng == "host_reference":
            eng.ro.check_status()
            divs = self._update_host()
        else:
            eng.update_device(gen, args.learning_rate, args.fitness_sharing)
            divs = [float(eng.div[r].item()) if args.fitness_sharing else None for r in ROLES] if self.collect else None
        if self.collect:
            res.game_rewards.append(eng.rewards_host()[:eng.n_main].copy())
            res.diversity.append(divs)
        stop = self._finish_generation(gen, eng.evaluate(gen))
        res.seconds.append(time.perf_counter() - t0)
        self.gen += 1
        return not stop

    def _finish_generation(self, gen, ev):
        """bookkeeping after generation `gen`'s evaluation games; returns True when early stopping fires"""
        args, res = self.args, self.res
        for s, r in enumerate(ROLES):
            res.rewards[r].append(ev[s])
        if args.adaptive:
            adapt_mutation_power(args, gen, res.rewards)
        res.sigma_after.append([args.mutation_power_agent_0, args.mutation_power_agent_1, args.mutation_power_adversary])
        if getattr(args, "early_stopping", False):  # evolutionary_strategy.py:320-354
            for s, r in enumerate(ROLES):
                if ev[s] > self.best[r] + args.min_delta:
                    self.best[r], self.no_improve[r] = ev[s], 0
                else:
                    self.no_improve[r] += 1
            for r in ROLES:
                if self.no_improve[r] >= args.patience:
                    res.stopped_at = gen
                    return True
        return False

    def finish(self):
        self.eng.ro.check_status()
        if hasattr(self.env, "n_resets"):
            self.env.n_resets = self.first_ordinal + self.gen * (3 * self.args.population + N_EVAL)
        return self.res

    def base_agents(self):
        """the three trained agents, as the reference returns them (evolutionary_strategy.py:393)"""
        out = []
        state = torch.random.get_rng_state()  # building the agent objects must not advance the reference's RNG stream
        for r in ROLES:
            a = create_agent(self.env, self.args, role=r)
            a.model.set_flat(self.eng.download(r, "base", 0, 1)[0])
            out.append(a)
        torch.random.set_rng_state(state)
        return tuple(out)


def evolution_strategy_train(env, args, output_dir, rng=None, env_mode=None, collect=True, return_result=False,
                             dist_ctx=None):
    """Drop-in for evolutionary_strategy.py:151: returns (agent_0, agent_1, adversary) like the reference; with
    return_result=True also the ESResult history (what the reference only plots)."""
    import os
    if getattr(args, "game", "simple_adversary_v3") != "simple_adversary_v3":   # the two-player Atari games
        from .dqn_population import dqn_evolution_strategy_train
        return dqn_evolution_strategy_train(env, args, output_dir, collect=collect, dist_ctx=dist_ctx)
    from .io_utils import ES_FILES, MetricsWriter, save_model, save_state_dicts
    tr = ESTrainer(env, args, rng=rng, env_mode=env_mode, collect=collect, dist_ctx=dist_ctx)
    save = bool(getattr(args, "save", False)) and output_dir is not None
    mw = MetricsWriter(output_dir)
    for _ in range(args.generations):
        go_on = tr.step()
        res, g = tr.res, tr.gen - 1   # one metrics line per generation, written as it finishes
        mw.write(generation=g, eval_rewards={r: res.rewards[r][g] for r in ROLES}, mutation_power=res.sigma_after[g],
                 diversity=res.diversity[g] if g < len(res.diversity) else None,
                 seconds=res.seconds[g] if g < len(res.seconds) else None, stopped=not go_on)
        if not go_on:  # evolutionary_strategy.py:343-354: the reference breaks BEFORE save_model (:357-360), so the
            break      # stopping generation's update is never written
        if save:
            for a, r in zip(tr.base_agents(), ROLES):
                save_model(a, os.path.join(output_dir, ES_FILES[r]))
                save_state_dicts(a, os.path.join(output_dir, ES_FILES[r]), role=r)   # the weights_only-safe twin
    res = tr.finish()
    res.engine = tr.eng
    agents = tr.base_agents()
    return (agents, res) if return_result else agents
