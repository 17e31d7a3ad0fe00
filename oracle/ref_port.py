"""TEST INFRASTRUCTURE ONLY - sequential CPU port of the reference's Co-GA / Co-ES generation loop.

Drives oracle/coevo_oracle.c (one batch-1 forward per agent-step, one AEC episode per game) in exactly
the reference's call order, quirks Q1-Q14 of SURVEY.md Appendix A included, with the reference's own
host RNG calls (global torch generator for net init + GA mutation, global numpy generator for ES noise)
so that results are comparable with the golden fixtures bit for bit wherever the argmax margins allow.

Follows (file:line under /root/reference):
  init_net / create order      MPE/fcnetwork.py:11-22, MPE/mpe_agent.py:11-21, genetic_algorithm.py:63-68,110-117
  mutate                       agent.py:25-29
  ga_train                     genetic_algorithm.py:51-345
  es_train                     evolutionary_strategy.py:151-316, agent.py:31-70
  diversity                    utils/game_logic_functions.py:12-37

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: PINNED by tests/golden/{ga_cfg1,ga_hof2,ga_long,es_small,es_fs,es_long,es_stop,play_game,fc_forward}.json.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_ct = C   # (functions below that take the channel count name it C, like the reference's DeepQN)
HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
ENV_SEED = 1870300
ROLES = ("agent_0", "agent_1", "adversary_0")
ROLE_D = {"agent_0": 10, "agent_1": 10, "adversary_0": 8}
H1, H2, NACT = 512, 256, 5


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "coevo_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.oracle_fc_param_count.restype = C.c_int
        L.oracle_fc_forward.restype = C.c_int
        L.oracle_fc_forward.argtypes = [fp, C.c_int, fp, fp, ip]
        L.oracle_play_game.restype = C.c_int
        L.oracle_play_game.argtypes = [fp, fp, fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double), ip, fp, ip]
        L.oracle_mpe_reset.argtypes = [C.c_uint64] * 5 + [C.c_void_p]
        L.oracle_mpe_observe.argtypes = [C.c_void_p, C.c_int, fp]
        L.oracle_mpe_world_step.argtypes = [C.c_void_p, ip, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_mpe_set_pos_first.argtypes = [C.c_int]
        L.oracle_diversity.restype = C.c_double
        L.oracle_diversity.argtypes = [fp, fp, C.c_int, C.c_size_t, ip, ip, C.c_int, fp]
        L.oracle_philox_normal4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, fp]
        L.oracle_philox4x32.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_noise_rounds.restype = C.c_int
        L.oracle_perturb_philox.argtypes = [fp, fp, C.c_int, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32,
                                            ip, ip, C.c_int, C.c_int]
        L.oracle_es_update_from_pert.argtypes = [fp, C.c_int, fp, fp, C.c_int, C.c_float, ip, ip, C.c_int, C.c_int]
        L.oracle_dqn_param_count.restype = C.c_long
        L.oracle_dqn_forward.restype = C.c_int
        L.oracle_dqn_forward.argtypes = [fp, C.c_int, C.c_int, C.c_void_p, fp]
        L.oracle_synth_target.restype = C.c_uint32
        L.oracle_synth_target.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int]
        L.oracle_synth_frame.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_dqn_play_game.restype = C.c_int
        L.oracle_dqn_play_game.argtypes = [fp, fp, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int,
                                           C.POINTER(C.c_double), ip]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


# ------------------------------------------------------------------ parameter layout
def param_shapes(D):
    return [("fc1.weight", (H1, D)), ("fc1.bias", (H1,)), ("ln1.weight", (H1,)), ("ln1.bias", (H1,)),
            ("fc2.weight", (H2, H1)), ("fc2.bias", (H2,)), ("ln2.weight", (H2,)), ("ln2.bias", (H2,)),
            ("output.weight", (NACT, H2)), ("output.bias", (NACT,))]


def param_count(D):
    return sum(int(np.prod(s)) for _, s in param_shapes(D))


def linear_segments(D):
    """(offset, length) of the Linear W,b entries in the flat vector = get_weights_ES() content."""
    segs, off = [], 0
    for name, shp in param_shapes(D):
        n = int(np.prod(shp))
        if not name.startswith("ln"):
            segs.append((off, n))
        off += n
    return segs


def ln_segments(D):
    segs, off = [], 0
    for name, shp in param_shapes(D):
        n = int(np.prod(shp))
        if name.startswith("ln"):
            segs.append((off, n))
        off += n
    return segs


def weights_es(flat, D):
    return np.concatenate([flat[o:o + n] for o, n in linear_segments(D)])


def init_net(D):
    """Same torch-generator consumption as FCNetwork.__init__ (three nn.Linear in module order)."""
    fc1 = torch.nn.Linear(D, H1)
    fc2 = torch.nn.Linear(H1, H2)
    out = torch.nn.Linear(H2, NACT)
    parts = [fc1.weight, fc1.bias, torch.ones(H1), torch.zeros(H1), fc2.weight, fc2.bias,
             torch.ones(H2), torch.zeros(H2), out.weight, out.bias]
    return np.concatenate([p.detach().numpy().ravel() for p in parts]).astype(np.float32)


def mutate_torch(flat, D, sigma):
    """Agent.mutate (agent.py:25-29): torch.normal per parameter tensor, in parameters() order."""
    out = flat.copy()
    off = 0
    for _, shp in param_shapes(D):
        n = int(np.prod(shp))
        noise = torch.normal(0, sigma, size=shp).numpy().ravel()
        out[off:off + n] = out[off:off + n] + noise
        off += n
    return out


# ------------------------------------------------------------------ env stream + games
class Stream:
    """The seeded reset stream; ordinal 0 was consumed by initialize_env."""

    def __init__(self, seed=ENV_SEED):
        st = np.random.PCG64(seed).state["state"]
        self.st = (st["state"] >> 64, st["state"] & (2 ** 64 - 1), st["inc"] >> 64, st["inc"] & (2 ** 64 - 1))
        self.ordinal = 1

    def next_ordinal(self):
        o = self.ordinal
        self.ordinal += 1
        return o


def fc_forward(flat, D, obs):
    logits = np.zeros(NACT, dtype=np.float32)
    st = C.c_int(0)
    a = lib().oracle_fc_forward(_fp(flat), D, _fp(np.ascontiguousarray(obs, dtype=np.float32)), _fp(logits),
                                C.byref(st))
    return a, logits, st.value


def play_game(stream, net_a0, net_a1, net_adv, limit=None, max_cycles=25, ordinal=None):
    """-> dict(rewards=(agent_0, agent_1, adversary_0), steps, actions, min_margin, status)."""
    o = stream.next_ordinal() if ordinal is None else ordinal
    rewards = (C.c_double * 3)()
    actions = np.zeros(3 * max_cycles + 3, dtype=np.int32)
    mm = C.c_float(0)
    st = C.c_int(0)
    steps = lib().oracle_play_game(_fp(net_adv), _fp(net_a0), _fp(net_a1), *stream.st, o,
                                   -1 if limit is None else int(limit), max_cycles, rewards, _ip(actions),
                                   C.byref(mm), C.byref(st))
    if st.value:
        raise ValueError(f"oracle forward status {st.value} (NaN/inf or no action)")
    return {"rewards": [rewards[0], rewards[1], rewards[2]], "steps": steps,
            "actions": actions[:steps].tolist(), "min_margin": float(mm.value), "ordinal": o}


def diversity(individual_es, population_es):
    """diversity_penalty on get_weights_ES() vectors, numpy exactly as the reference writes it."""
    distances = np.array([np.linalg.norm(w - individual_es) for w in population_es])
    sigma = np.mean(distances)
    return np.sum(np.maximum(0, 1 - distances / sigma))


# ------------------------------------------------------------------ Co-GA
def perturb_philox_flat(flat, sigma, seed, stream_lo, stream_hi, skip_segments=(), negate=False):
    """child = parent +- sigma*eps(seed, stream, canonical index) over any flat parameter vector (FCNetwork or DeepQN)"""
    flat = np.ascontiguousarray(flat, dtype=np.float32)
    out = np.zeros(len(flat), dtype=np.float32)
    so = np.array([s[0] for s in skip_segments], dtype=np.int32)
    sl = np.array([s[1] for s in skip_segments], dtype=np.int32)
    lib().oracle_perturb_philox(_fp(flat), _fp(out), len(flat), float(sigma), int(seed), int(stream_lo),
                                int(stream_hi), _ip(so), _ip(sl), len(skip_segments), 1 if negate else 0)
    return out


def mutate_philox(flat, D, sigma, seed, stream_lo, stream_hi, skip_layernorm=False, negate=False):
    """child = parent + sigma*eps with the build's counter-based noise (device_philox mode)"""
    assert len(flat) == param_count(D)
    return perturb_philox_flat(flat, sigma, seed, stream_lo, stream_hi, ln_segments(D) if skip_layernorm else (),
                               negate)


def ga_initial(pop, hof_n):
    """initial HoF and population in the reference's creation order (the torch generator is consumed in it)"""
    D = ROLE_D
    # genetic_algorithm.py:63-68 - creation order and (swapped) roles matter for the torch RNG
    hof = {"agent_1": [init_net(10) for _ in range(hof_n)]}
    hof["agent_0"] = [init_net(10) for _ in range(hof_n)]
    hof["adversary_0"] = [init_net(8) for _ in range(hof_n)]
    for role_dim in (10, 10, 8):  # placeholder elites (Q7), overwritten in generation 0
        for _ in range(hof_n):
            init_net(role_dim)
    popu = {r: [] for r in ROLES}
    for _ in range(pop):  # :110-117 interleaved
        for r in ROLES:
            popu[r].append(init_net(D[r]))
    return hof, popu


def ga_game_nets(role, popu_i, hof, k, hof_n):
    """(agent_0, agent_1, adversary) nets of HoF game k of an individual of `role` (genetic_algorithm.py:136-142,
    168-174, 201-207; Q4: the adversary phase seats hof_agent_0 twice)"""
    j = hof_n - 1 - k
    if role == "agent_0":
        return popu_i, hof["agent_1"][j], hof["adversary_0"][j]
    if role == "agent_1":
        return hof["agent_0"][j], popu_i, hof["adversary_0"][j]
    return hof["agent_0"][j], hof["agent_0"][j], popu_i


def ga_train(args, max_cycles=25, log=None, noise="torch", philox_seed=0):
    """genetic_algorithm_train restated. args: attribute bag (mutated in place when adaptive).
    noise="torch": the reference's own RNG calls; noise="philox": the build's device_philox offspring rule
    (child c of role ri in generation g uses stream (c, 4g+ri); no torch draws after initialisation)."""
    stream = Stream()
    pop, hof_n, E = args.population, args.hof_size, args.elites_number
    D = ROLE_D
    hof, popu = ga_initial(pop, hof_n)
    stale = {r: popu[r][-1] for r in ROLES}  # Q3: objects left over from the init loop
    sig_attr = {"agent_0": "mutation_power_agent_0", "agent_1": "mutation_power_agent_1",
                "adversary_0": "mutation_power_adversary"}
    rewards_hist = {r: [] for r in ROLES}
    out = []
    for gen in range(args.generations):
        rec = {"games": [], "fitness": [], "diversity": [], "elite_ids": []}
        fitness = {}
        for ph, role in enumerate(ROLES):
            fit = []
            pop_es = [weights_es(w, D[role]) for w in popu[role]]
            div = diversity(weights_es(stale[role], D[role]), pop_es)
            for i in range(pop):
                last = None
                for k in range(hof_n):
                    if role == "agent_0":
                        g = play_game(stream, popu[role][i], hof["agent_1"][hof_n - 1 - k],
                                      hof["adversary_0"][hof_n - 1 - k], args.max_timesteps_per_episode, max_cycles)
                    elif role == "agent_1":
                        g = play_game(stream, hof["agent_0"][hof_n - 1 - k], popu[role][i],
                                      hof["adversary_0"][hof_n - 1 - k], args.max_timesteps_per_episode, max_cycles)
                    else:  # Q4: both good opponents come from hof_agent_0
                        g = play_game(stream, hof["agent_0"][hof_n - 1 - k], hof["agent_0"][hof_n - 1 - k],
                                      popu[role][i], args.max_timesteps_per_episode, max_cycles)
                    rec["games"].append(g)
                    last = g["rewards"][ph]  # Q2: overwritten each k
                fit.append(last / hof_n / (1 + div))
            fitness[role] = fit
            rec["fitness"].append([float(f) for f in fit])
            rec["diversity"].append(float(div))
        elites = {}
        for role in ROLES:
            order = np.argsort(fitness[role])[::-1]  # Q13
            ids = [int(x) for x in order[:E]]
            rec["elite_ids"].append(ids)
            elites[role] = [popu[role][i] for i in ids]
        best = {r: elites[r][0] for r in ROLES}
        for r in ROLES:
            hof[r].append(best[r])
            hof[r].pop(0)
        new_pop = {}
        rec["sigma_before"] = [getattr(args, sig_attr[r]) for r in ROLES]   # what this generation's children get
        for ri, r in enumerate(ROLES):  # mutate_elites: clone() builds a fresh net first (burns init draws)
            sigma = getattr(args, sig_attr[r])
            children = []
            for i in range(pop - 1):
                if noise == "torch":
                    init_net(D[r])
                    children.append(mutate_torch(elites[r][i % E], D[r], sigma))
                else:
                    children.append(mutate_philox(elites[r][i % E], D[r], np.float32(sigma), philox_seed, i,
                                                  gen * 4 + ri))
            new_pop[r] = [best[r]] + children
        popu = new_pop
        rec["hof"] = {r: [w for w in hof[r]] for r in ROLES}
        rec["elites"] = {r: [w for w in elites[r]] for r in ROLES}
        ev = [0.0, 0.0, 0.0]
        for _ in range(10):
            g = play_game(stream, best["agent_0"], best["agent_1"], best["adversary_0"],
                          args.max_evaluation_steps, max_cycles)
            rec["games"].append(g)
            for s in range(3):
                ev[s] += g["rewards"][s]
        ev = [e / 10 for e in ev]
        rec["eval_rewards"] = ev
        for s, r in enumerate(ROLES):
            rewards_hist[r].append(ev[s])
        if args.adaptive:
            adapt_sigma(args, gen, rewards_hist)
        rec["sigma_after"] = [args.mutation_power_agent_0, args.mutation_power_agent_1,
                              args.mutation_power_adversary]
        out.append(rec)
        if log:
            log(gen, rec)
    return out


def adapt_sigma(args, gen, hist):
    """genetic_algorithm.py:323-345 / evolutionary_strategy.py:292-316 (Q5 included)."""
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


# ------------------------------------------------------------------ Co-ES
def perturbable(flat, D):
    return weights_es(flat, D)


def set_perturbable(flat, D, vec):
    out = flat.copy()
    i = 0
    for o, n in linear_segments(D):
        out[o:o + n] = vec[i:i + n]
        i += n
    return out


ES_CHUNKS = 8   # the build's canonical ES summation: 8 chunk sums added left to right (coevo_es_partial / _apply)


def centered_ranks(f):
    """cfg 3 extension mode (not in the reference): u_i = rank_i/(n-1) - 0.5, stable ascending rank"""
    f = np.asarray(f, dtype=np.float32)
    n = len(f)
    ranks = np.empty(n, dtype=np.int64)
    ranks[np.argsort(f, kind="stable")] = np.arange(n)
    if n == 1:
        return np.zeros(1, dtype=np.float32)
    return (ranks.astype(np.float32) / np.float32(n - 1) - np.float32(0.5)).astype(np.float32)


def es_update_from_pert(theta, D, pert, fitness, sigma, lr, chunks=ES_CHUNKS):
    """theta += lr/(n*sigma) * sum_i f_i * (pert_i - theta) over the Linear entries (device_philox mode)"""
    P = param_count(D)
    out = np.ascontiguousarray(theta, dtype=np.float32).copy()
    segs = ln_segments(D)
    so = np.array([s[0] for s in segs], dtype=np.int32)
    sl = np.array([s[1] for s in segs], dtype=np.int32)
    f = np.ascontiguousarray(fitness, dtype=np.float32)
    pert = np.ascontiguousarray(pert, dtype=np.float32)
    n = len(f)
    scale = np.float32(lr) / (np.float32(n) * np.float32(sigma))
    lib().oracle_es_update_from_pert(_fp(out), P, _fp(pert), _fp(f), n, float(scale), _ip(so), _ip(sl), len(segs),
                                     int(chunks))
    return out


def es_train(args, max_cycles=25, noise="numpy", philox_seed=0, antithetic=False, centered_rank=False):
    """evolution_strategy_train restated; noise="numpy": the reference's own RNG calls; noise="philox": the build's
    device_philox rule (perturbed net j of role ri in generation g uses stream (j, 4g+ri), fp32 noise, fp32 update).
    antithetic / centered_rank (philox only): the cfg 3 EXTENSION mode of BASELINE.json configs[2], not in the reference:
    individuals 2m, 2m+1 share stream m with opposite signs; fitness -> centered ranks where the reference has its
    normalisation commented out (evolutionary_strategy.py:133-135)."""
    assert noise == "philox" or not (antithetic or centered_rank)
    mode = noise
    stream = Stream()
    D = ROLE_D
    base = {r: init_net(D[r]) for r in ROLES}  # evolutionary_strategy.py:163-165
    base_w = {r: perturbable(base[r], D[r]).astype(np.float32) for r in ROLES}
    sig_attr = {"agent_0": "mutation_power_agent_0", "agent_1": "mutation_power_agent_1",
                "adversary_0": "mutation_power_adversary"}
    hist = {r: [] for r in ROLES}
    best = {r: -float("inf") for r in ROLES}       # evolutionary_strategy.py:211-219
    no_improve = {r: 0 for r in ROLES}
    out = []
    for gen in range(args.generations):
        rec = {"games": []}
        noises = {r: [] for r in ROLES}
        rewards = {r: [] for r in ROLES}
        pop_w = {r: [] for r in ROLES}
        pert_full = {r: [] for r in ROLES}
        for _ in range(args.population):
            for s, r in enumerate(ROLES):
                sigma = getattr(args, sig_attr[r])
                if mode == "philox":
                    mutated = mutate_philox(base[r], D[r], np.float32(sigma), philox_seed, (_ >> 1) if antithetic else _,
                                            gen * 4 + s, skip_layernorm=True, negate=bool(antithetic and (_ & 1)))
                    nets = {q: base[q] for q in ROLES}
                    nets[r] = mutated
                    g = play_game(stream, nets["agent_0"], nets["agent_1"], nets["adversary_0"],
                                  args.max_timesteps_per_episode, max_cycles)
                    rec["games"].append(g)
                    rewards[r].append(g["rewards"][s])
                    pop_w[r].append(weights_es(mutated, D[r]))
                    pert_full[r].append(mutated)
                    continue
                init_net(D[r])  # clone() constructs a fresh net (torch RNG only)
                w = perturbable(base[r], D[r])
                noise = np.random.normal(loc=0.0, scale=sigma, size=len(w))  # agent.py:52, fp64
                mutated = set_perturbable(base[r], D[r], (w + noise).astype(np.float32))
                nets = {q: base[q] for q in ROLES}
                nets[r] = mutated
                g = play_game(stream, nets["agent_0"], nets["agent_1"], nets["adversary_0"],
                              args.max_timesteps_per_episode, max_cycles)
                rec["games"].append(g)
                noises[r].append(noise.astype(np.float32))
                rewards[r].append(g["rewards"][s])
                pop_w[r].append(weights_es(mutated, D[r]))
        rec["diversity"] = []
        for ri, r in enumerate(ROLES):  # compute_weight_update :120-148
            f = np.array(rewards[r], dtype=np.float32)
            div = None
            if args.fitness_sharing:
                div = diversity(base_w[r], pop_w[r])
                f = f / (1 + div)
            sigma = getattr(args, sig_attr[r])
            if centered_rank:
                f = centered_ranks(f)
            if mode == "philox":
                base[r] = es_update_from_pert(base[r], D[r], np.stack(pert_full[r]), f, sigma, args.learning_rate)
                base_w[r] = perturbable(base[r], D[r])
                rec["diversity"].append(None if div is None else float(div))
                rec.setdefault("fitness", []).append([float(x) for x in f])
                continue
            n_arr = np.array(noises[r], dtype=np.float32)
            upd = (args.learning_rate / (len(n_arr) * sigma)) * np.dot(n_arr.T, f)
            base_w[r] = base_w[r] + upd.astype(np.float32)
            base[r] = set_perturbable(base[r], D[r], base_w[r])
            rec["diversity"].append(None if div is None else float(div))
        ev = [0.0, 0.0, 0.0]
        for _ in range(10):
            g = play_game(stream, base["agent_0"], base["agent_1"], base["adversary_0"],
                          args.max_evaluation_steps, max_cycles)
            rec["games"].append(g)
            for s in range(3):
                ev[s] += g["rewards"][s]
        ev = [e / 10 for e in ev]
        rec["eval_rewards"] = ev
        for s, r in enumerate(ROLES):
            hist[r].append(ev[s])
        if args.adaptive:
            adapt_sigma(args, gen, hist)
        rec["sigma_after"] = [args.mutation_power_agent_0, args.mutation_power_agent_1,
                              args.mutation_power_adversary]
        rec["base"] = {r: base[r].copy() for r in ROLES}
        out.append(rec)
        if getattr(args, "early_stopping", False):  # evolutionary_strategy.py:320-354: all three counters first,
            for s, r in enumerate(ROLES):           # then the patience checks in role order; break before saving
                if ev[s] > best[r] + args.min_delta:
                    best[r], no_improve[r] = ev[s], 0
                else:
                    no_improve[r] += 1
            if any(no_improve[r] >= args.patience for r in ROLES):
                rec["stopped"] = True
                break
    return out


# ------------------------------------------------------------------ DeepQN (Atari/deepqn.py)
def dqn_init(C, n_actions):
    """flat parameters in parameters() order, consuming the torch generator like DeepQN.__init__ (:16-37)"""
    conv1 = torch.nn.Conv2d(C, 32, kernel_size=8, stride=4)
    conv2 = torch.nn.Conv2d(32, 64, kernel_size=4, stride=2)
    conv3 = torch.nn.Conv2d(64, 64, kernel_size=3, stride=1)
    fc1 = torch.nn.Linear(64 * 7 * 7, 512)
    out = torch.nn.Linear(512, n_actions)
    parts = [conv1.weight, conv1.bias, conv2.weight, conv2.bias, conv3.weight, conv3.bias, fc1.weight, fc1.bias,
             out.weight, out.bias, torch.ones(32), torch.zeros(32), torch.ones(64), torch.zeros(64), torch.ones(64),
             torch.zeros(64)]
    return np.concatenate([p.detach().numpy().ravel() for p in parts]).astype(np.float32), [tuple(p.shape) for p in parts]


def dqn_mutate_torch(flat, shapes, sigma):
    out, off = flat.copy(), 0
    for shp in shapes:
        n = int(np.prod(shp))
        out[off:off + n] = out[off:off + n] + torch.normal(0, sigma, size=shp).numpy().ravel()
        off += n
    return out


def dqn_forward(flat, C, n_actions, frame_u8):
    logits = np.zeros(n_actions, dtype=np.float32)
    f = np.ascontiguousarray(frame_u8, dtype=np.uint8)
    a = lib().oracle_dqn_forward(_fp(np.ascontiguousarray(flat, dtype=np.float32)), C, n_actions, f.ctypes.data, _fp(logits))
    return a, logits


# ------------------------------------------------------------------ DeepQN population loops (cfg 4 / cfg 5)
# The reference's Atari loop does not run, so this is the BUILD's definition: the 2-role restriction of the
# Co-GA / Co-ES generation bodies (genetic_algorithm.py:119-290, evolutionary_strategy.py:222-265) over the synthetic env
# of include/coevo.h.  "Loop parity unpinned, forward pinned" (SURVEY 8c): DeepQN.forward is pinned by
# tests/golden/deepqn_forward.json, everything below is checked HIP-vs-this-port only.
DQN_ROLES = ("first_0", "second_0")
DQN_BN_SEGMENT_NAMES = ("vbn1.w", "vbn1.b", "vbn2.w", "vbn2.b", "vbn3.w", "vbn3.b")


def dqn_bn_segments(C, n_actions):
    """(offset, length) of the BatchNorm affine entries in the canonical flat order (they come last)"""
    P = lib().oracle_dqn_param_count(C, n_actions)
    return [(P - 320, 320)]


def synth_frame(seed, ordinal, t, last_action, C):
    out = np.zeros((84, 84, C), dtype=np.uint8)
    lib().oracle_synth_frame(int(seed), int(ordinal), int(t), int(last_action), int(C), out.ctypes.data)
    return out


def dqn_play_game(net_first, net_second, C, n_actions, seed, ordinal, limit):
    """play_atari over the synthetic env -> dict(rewards=[first_0, second_0], steps, actions)"""
    rewards = (_ct.c_double * 2)()
    actions = np.zeros(max(int(limit), 1), dtype=np.int32)
    steps = lib().oracle_dqn_play_game(_fp(np.ascontiguousarray(net_first, dtype=np.float32)),
                                       _fp(np.ascontiguousarray(net_second, dtype=np.float32)), C, n_actions, int(seed),
                                       int(ordinal), int(limit), rewards, _ip(actions))
    return {"rewards": [rewards[0], rewards[1]], "steps": steps, "actions": actions[:steps].tolist()}


def dqn_diversity(individual, population):
    """diversity_penalty on DeepQN.get_weights_ES() = ALL parameters (self.layers includes the BatchNorm layers)"""
    return diversity(individual, population)


def dqn_ga_train(args, C, n_actions, env_seed=0, philox_seed=0):
    """2-role Co-GA: per role phase (first_0, second_0), individual i, k < hof: one game against hof_other[hof-1-k];
    fitness = last game's reward in the role's slot / hof / (1 + diversity vs the stale agent) (Q2, Q3 kept);
    argsort()[::-1] elites; HoF FIFO; population = [best] + children(elite[c % E] + sigma*eps(stream (c, 4g+ri)));
    10 evaluation games of the best pair; adaptive sigma (first_0 plays agent_0's part in Q5, second_0 agent_1's)."""
    pop, hof_n, E = args.population, args.hof_size, args.elites_number
    T, T_eval = args.max_timesteps_per_episode, args.max_evaluation_steps
    hof = {"second_0": [dqn_init(C, n_actions)[0] for _ in range(hof_n)]}   # creation order mirrors ga_initial
    hof["first_0"] = [dqn_init(C, n_actions)[0] for _ in range(hof_n)]
    popu = {r: [] for r in DQN_ROLES}
    for _ in range(pop):
        for r in DQN_ROLES:
            popu[r].append(dqn_init(C, n_actions)[0])
    stale = {r: popu[r][-1] for r in DQN_ROLES}
    sig_attr = {"first_0": "mutation_power_agent_0", "second_0": "mutation_power_agent_1"}
    hist = {"agent_0": [], "agent_1": [], "adversary_0": []}
    per_gen = 2 * pop * hof_n + 10
    out = []
    for gen in range(args.generations):
        rec = {"games": [], "fitness": [], "diversity": [], "elite_ids": []}
        fitness = {}
        for ph, role in enumerate(DQN_ROLES):
            div = dqn_diversity(stale[role], popu[role])
            fit = []
            for i in range(pop):
                last = None
                for k in range(hof_n):
                    opp = hof[DQN_ROLES[1 - ph]][hof_n - 1 - k]
                    nets = (popu[role][i], opp) if ph == 0 else (opp, popu[role][i])
                    g = dqn_play_game(nets[0], nets[1], C, n_actions, env_seed,
                                      1 + gen * per_gen + ph * pop * hof_n + i * hof_n + k, T)
                    rec["games"].append(g)
                    last = g["rewards"][ph]
                fit.append(last / hof_n / (1 + div))
            fitness[role] = fit
            rec["fitness"].append([float(f) for f in fit])
            rec["diversity"].append(float(div))
        elites = {}
        for role in DQN_ROLES:
            # integer hit counts tie often: the build's tie-break is the stable ascending sort reversed (higher index
            # first), which is what np.argsort's default gives for the small n the reference's own tests could use
            ids = [int(x) for x in np.argsort(fitness[role], kind="stable")[::-1][:E]]
            rec["elite_ids"].append(ids)
            elites[role] = [popu[role][i] for i in ids]
        for r in DQN_ROLES:
            hof[r].append(elites[r][0])
            hof[r].pop(0)
        rec["sigma_before"] = [getattr(args, sig_attr[r]) for r in DQN_ROLES]
        new_pop = {}
        for ri, r in enumerate(DQN_ROLES):
            sigma = np.float32(getattr(args, sig_attr[r]))
            new_pop[r] = [elites[r][0]] + [perturb_philox_flat(elites[r][c % E], sigma, philox_seed, c, gen * 4 + ri)
                                           for c in range(pop - 1)]
        popu = new_pop
        rec["hof"] = {r: list(hof[r]) for r in DQN_ROLES}
        rec["elites"] = elites
        rec["pop"] = popu
        ev = [0.0, 0.0]
        for j in range(10):
            g = dqn_play_game(elites["first_0"][0], elites["second_0"][0], C, n_actions, env_seed,
                              1 + gen * per_gen + 2 * pop * hof_n + j, T_eval)
            rec["games"].append(g)
            for s in range(2):
                ev[s] += g["rewards"][s]
        ev = [e / 10 for e in ev]
        rec["eval_rewards"] = ev
        hist["agent_0"].append(ev[0])
        hist["agent_1"].append(ev[1])
        hist["adversary_0"].append(0.0)
        if args.adaptive:
            adapt_sigma(args, gen, hist)
        rec["sigma_after"] = [args.mutation_power_agent_0, args.mutation_power_agent_1]
        out.append(rec)
    return out


def dqn_es_update_from_pert(theta, pert, fitness, sigma, lr, C, n_actions, chunks=ES_CHUNKS):
    P = len(theta)
    out = np.ascontiguousarray(theta, dtype=np.float32).copy()
    segs = dqn_bn_segments(C, n_actions)
    so = np.array([s[0] for s in segs], dtype=np.int32)
    sl = np.array([s[1] for s in segs], dtype=np.int32)
    f = np.ascontiguousarray(fitness, dtype=np.float32)
    pert = np.ascontiguousarray(pert, dtype=np.float32)
    scale = np.float32(lr) / (np.float32(len(f)) * np.float32(sigma))
    lib().oracle_es_update_from_pert(_fp(out), P, _fp(pert), _fp(f), len(f), float(scale), _ip(so), _ip(sl), len(segs),
                                     int(chunks))
    return out


def dqn_es_train(args, C, n_actions, env_seed=0, philox_seed=0, antithetic=False, centered_rank=False):
    """2-role Co-ES: per iteration j and role ri one game of the perturbed net (BatchNorm untouched) against the other
    role's current base net (game ordinal 2j + ri); raw reward (or centered rank) fitness, optional sharing; chunked
    canonical update; 10 evaluation games; adaptive sigma."""
    pop = args.population
    T, T_eval = args.max_timesteps_per_episode, args.max_evaluation_steps
    base = {r: dqn_init(C, n_actions)[0] for r in DQN_ROLES}
    bn = dqn_bn_segments(C, n_actions)
    sig_attr = {"first_0": "mutation_power_agent_0", "second_0": "mutation_power_agent_1"}
    hist = {"agent_0": [], "agent_1": [], "adversary_0": []}
    per_gen = 2 * pop + 10
    out = []
    for gen in range(args.generations):
        rec = {"games": [], "diversity": []}
        pert = {r: [] for r in DQN_ROLES}
        rew = {r: [] for r in DQN_ROLES}
        for j in range(pop):
            for ri, r in enumerate(DQN_ROLES):
                sigma = np.float32(getattr(args, sig_attr[r]))
                m = perturb_philox_flat(base[r], sigma, philox_seed, (j >> 1) if antithetic else j, gen * 4 + ri, bn,
                                        negate=bool(antithetic and (j & 1)))
                nets = (m, base["second_0"]) if ri == 0 else (base["first_0"], m)
                g = dqn_play_game(nets[0], nets[1], C, n_actions, env_seed, 1 + gen * per_gen + 2 * j + ri, T)
                rec["games"].append(g)
                pert[r].append(m)
                rew[r].append(g["rewards"][ri])
        new = {}
        for ri, r in enumerate(DQN_ROLES):
            f = np.array(rew[r], dtype=np.float32)
            if args.fitness_sharing:
                div = dqn_diversity(base[r], pert[r])
                f = f / (1 + div)
                rec["diversity"].append(float(div))
            else:
                rec["diversity"].append(None)
            if centered_rank:
                f = centered_ranks(f)
            new[r] = dqn_es_update_from_pert(base[r], np.stack(pert[r]), f, getattr(args, sig_attr[r]),
                                             args.learning_rate, C, n_actions)
        base = new
        ev = [0.0, 0.0]
        for j in range(10):
            g = dqn_play_game(base["first_0"], base["second_0"], C, n_actions, env_seed,
                              1 + gen * per_gen + 2 * pop + j, T_eval)
            rec["games"].append(g)
            for s in range(2):
                ev[s] += g["rewards"][s]
        ev = [e / 10 for e in ev]
        rec["eval_rewards"] = ev
        hist["agent_0"].append(ev[0])
        hist["agent_1"].append(ev[1])
        hist["adversary_0"].append(0.0)
        if args.adaptive:
            adapt_sigma(args, gen, hist)
        rec["sigma_after"] = [args.mutation_power_agent_0, args.mutation_power_agent_1]
        rec["base"] = {r: base[r].copy() for r in DQN_ROLES}
        out.append(rec)
    return out
