"""Host-side MPE ``simple_adversary`` environment (product code, NumPy, fp64).

The reference builds its env with ``pettingzoo.mpe.simple_adversary_v3.env()``
(/root/reference/utils/game_logic_functions.py:45-46) and drives it through the AEC
API (``reset / agent_iter / observe / step / last``, same file :123-212, :217).
PettingZoo is a third-party dependency that is NOT vendored in the reference
(requirements.txt:5, un-pinned) and is absent from this image, so this module is a
restatement of the published simple_adversary semantics:

* agents ``[adversary_0, agent_0, agent_1]``, two landmarks, one of them the goal;
* reset draws, in this order, from ``np.random.Generator(PCG64)``: ``choice`` of the
  goal landmark, ``uniform(-1, 1, 2)`` for each agent, then for each landmark;
* discrete actions 0 noop, 1 -x, 2 +x, 3 -y, 4 +y, force = 5.0 * u, mass 1,
  dt 0.1, damping 0.25, no collisions, no max speed;
* the world advances once per 3 agent-steps (when the last agent has acted);
* rewards after a world step: adversary ``-|p_adv - goal|``; both good agents
  ``-min_good |p - goal| + |p_adv - goal|``;
* observations (float32): good ``[goal-p, lm0-p, lm1-p, other0-p, other1-p]`` (10),
  adversary ``[lm0-p, lm1-p, other0-p, other1-p]`` (8);
* every agent is truncated once ``max_cycles`` (default 25) world steps have run.

Two classes share one set of arithmetic:

``SimpleAdversaryAEC``  a single env copy behind the AEC surface the reference calls.
``VecSimpleAdversary``  struct-of-arrays over E env copies, stepped one world-cycle at a
                        time by the batched engine (north_star: "vectorised env stepping
                        runs on the host cores").

Every operation is a single IEEE fp64 add/mul/sqrt in a fixed order, so the C oracle
(oracle/coevo_oracle.c) and the HIP device env (csrc/mpe_env.hip.h) reproduce it bit for bit.
"""
from __future__ import annotations

import numpy as np

AGENTS = ("adversary_0", "agent_0", "agent_1")
SLOT_ADVERSARY, SLOT_AGENT_0, SLOT_AGENT_1 = 0, 1, 2
OBS_DIM = {"adversary_0": 8, "agent_0": 10, "agent_1": 10}
N_ACTIONS = 5
DT = 0.1
DAMPING = 0.25
SENSITIVITY = 5.0
MASS = 1.0
DEFAULT_MAX_CYCLES = 25
ENV_SEED = 1870300  # /root/reference/utils/game_logic_functions.py:54

# PettingZoo releases differ in where ``p_pos += p_vel * dt`` sits inside
# ``World.integrate_state``.  True = position is advanced with the OLD
# velocity before damping/force are applied (PettingZoo >= 1.24); False = after.
INTEGRATE_POS_FIRST = True

# draws consumed from the PCG64 stream by one reset: 1 choice + 5 * uniform(2)
DOUBLES_PER_RESET = 10


class _Box:
    def __init__(self, n):
        self.shape = (n,)
        self.dtype = np.float32
        self.low = -np.inf
        self.high = np.inf


class _Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = np.int64


def draw_reset(rng: np.random.Generator):
    """One reset's worth of draws in PettingZoo's order -> (goal, agent_pos[3,2], lm_pos[2,2])."""
    goal = int(rng.choice(2))
    apos = np.empty((3, 2), dtype=np.float64)
    for a in range(3):
        apos[a] = rng.uniform(-1, +1, 2)
    lpos = np.empty((2, 2), dtype=np.float64)
    for l in range(2):
        lpos[l] = rng.uniform(-1, +1, 2)
    return goal, apos, lpos


def _force_from_action(action):
    """Discrete action -> force vector (u * sensitivity + 0.0 noise) / mass * dt."""
    u = np.zeros(2, dtype=np.float64)
    if action == 1:
        u[0] = -1.0
    if action == 2:
        u[0] = +1.0
    if action == 3:
        u[1] = -1.0
    if action == 4:
        u[1] = +1.0
    u *= SENSITIVITY
    return ((u + 0.0) / MASS) * DT


class SimpleAdversaryAEC:
    """AEC facade over one env copy; the object ``initialize_env`` returns."""

    metadata = {"name": "simple_adversary_v3", "is_parallelizable": True}

    def __init__(self, max_cycles=DEFAULT_MAX_CYCLES, render_mode=None):
        self.max_cycles = int(max_cycles)
        self.render_mode = render_mode
        self.possible_agents = list(AGENTS)
        self.agents = list(AGENTS)
        self._index_map = {a: i for i, a in enumerate(AGENTS)}
        self.np_random = np.random.Generator(np.random.PCG64())
        self.seed_value = None  # seed of the stream the resets are drawn from (None = OS entropy)
        self.n_resets = 0
        self._obs_spaces = {a: _Box(OBS_DIM[a]) for a in AGENTS}
        self._act_spaces = {a: _Discrete(N_ACTIONS) for a in AGENTS}
        self._init_episode_state()

    # -- spaces -----------------------------------------------------------------
    def observation_space(self, agent):
        return self._obs_spaces[agent]

    def action_space(self, agent):
        return self._act_spaces[agent]

    # -- episode state ----------------------------------------------------------
    def _init_episode_state(self):
        self.p_pos = np.zeros((3, 2), dtype=np.float64)
        self.p_vel = np.zeros((3, 2), dtype=np.float64)
        self.lm_pos = np.zeros((2, 2), dtype=np.float64)
        self.goal = 0
        self.rewards = {a: 0.0 for a in AGENTS}
        self._cumulative_rewards = {a: 0.0 for a in AGENTS}
        self.terminations = {a: False for a in AGENTS}
        self.truncations = {a: False for a in AGENTS}
        self.infos = {a: {} for a in AGENTS}
        self.agent_selection = AGENTS[0]
        self.steps = 0
        self.current_actions = [None] * 3

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.np_random = np.random.Generator(np.random.PCG64(seed))
            self.seed_value = seed
            self.n_resets = 0
        goal, apos, lpos = draw_reset(self.np_random)
        self.n_resets += 1
        self.agents = list(AGENTS)
        self._init_episode_state()
        self.goal = goal
        self.p_pos[:] = apos
        self.lm_pos[:] = lpos

    def close(self):
        pass

    def render(self):
        return None

    # -- AEC surface ------------------------------------------------------------
    def observe(self, agent):
        i = self._index_map[agent]
        me = self.p_pos[i]
        parts = []
        if i != SLOT_ADVERSARY:
            parts.append(self.lm_pos[self.goal] - me)
        parts.append(self.lm_pos[0] - me)
        parts.append(self.lm_pos[1] - me)
        for j in range(3):
            if j != i:
                parts.append(self.p_pos[j] - me)
        return np.concatenate(parts).astype(np.float32)

    def last(self, observe=True):
        agent = self.agent_selection
        obs = self.observe(agent) if observe else None
        return (obs, self._cumulative_rewards[agent], self.terminations[agent],
                self.truncations[agent], self.infos[agent])

    def agent_iter(self, max_iter=2 ** 63):
        it = 0
        while self.agents and it < max_iter:
            yield self.agent_selection
            it += 1

    def _world_step(self):
        for i in range(3):
            f = _force_from_action(self.current_actions[i])
            if INTEGRATE_POS_FIRST:
                self.p_pos[i] += self.p_vel[i] * DT
            self.p_vel[i] = self.p_vel[i] * (1 - DAMPING)
            self.p_vel[i] += f
            if not INTEGRATE_POS_FIRST:
                self.p_pos[i] += self.p_vel[i] * DT
        g = self.lm_pos[self.goal]
        d = [float(np.sqrt(np.sum(np.square(self.p_pos[i] - g)))) for i in range(3)]
        r_adv = -d[0]
        r_good = -min(d[1], d[2]) + d[0]
        self.rewards = {"adversary_0": r_adv, "agent_0": r_good, "agent_1": r_good}

    def _was_dead_step(self, action):
        if action is not None:
            raise ValueError("when an agent is dead, the only valid action is None")
        agent = self.agent_selection
        self.agents.remove(agent)
        for k in (self.terminations, self.truncations, self.rewards,
                  self._cumulative_rewards, self.infos):
            k.pop(agent, None)
        if self.agents:
            self.agent_selection = self.agents[0]

    def step(self, action):
        cur = self.agent_selection
        if self.terminations.get(cur, False) or self.truncations.get(cur, False):
            self._was_dead_step(action)
            return
        idx = self._index_map[cur]
        nxt = (idx + 1) % 3
        self.agent_selection = AGENTS[nxt]
        self.current_actions[idx] = int(action)
        if nxt == 0:
            self._world_step()
            self.steps += 1
            if self.steps >= self.max_cycles:
                for a in self.agents:
                    self.truncations[a] = True
        else:
            self.rewards = {a: 0.0 for a in AGENTS}
        self._cumulative_rewards[cur] = 0
        for a in AGENTS:
            self._cumulative_rewards[a] += self.rewards[a]


class ResetStream:
    """The reference's single seeded reset stream (quirk Q6: one ``reset(seed=1870300)`` in
    ``initialize_env``, every later ``play_game`` reset continues it), addressable by the
    game's ordinal.  ``take(n)`` returns the next n resets as arrays."""

    def __init__(self, seed=ENV_SEED, skip_initial=True):
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self.ordinal = 0
        if skip_initial:  # the reset inside initialize_env itself
            self.take(1)

    def take(self, n):
        goal = np.empty(n, dtype=np.int32)
        apos = np.empty((n, 3, 2), dtype=np.float64)
        lpos = np.empty((n, 2, 2), dtype=np.float64)
        for g in range(n):
            goal[g], apos[g], lpos[g] = draw_reset(self.rng)
        self.ordinal += n
        return goal, apos, lpos


class VecSimpleAdversary:
    """E env copies, struct-of-arrays, stepped a whole world-cycle at a time.

    ``observe()`` gives the three agents' float32 observations for the current state (all three
    agents of a cycle see the same world state, because the world only moves when the last
    agent has acted); ``step(actions[E,3])`` applies one world step and returns
    ``(r_good[E], r_adv[E])``.
    """

    def __init__(self, goal, apos, lpos):
        self.E = int(goal.shape[0])
        self.goal = goal.astype(np.int64)
        self.p_pos = apos.astype(np.float64).copy()
        self.p_vel = np.zeros_like(self.p_pos)
        self.lm_pos = lpos.astype(np.float64).copy()
        self.goal_pos = self.lm_pos[np.arange(self.E), self.goal]  # [E,2]

    def observe(self):
        p = self.p_pos
        lm0, lm1, g = self.lm_pos[:, 0], self.lm_pos[:, 1], self.goal_pos
        adv = np.concatenate([lm0 - p[:, 0], lm1 - p[:, 0], p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]],
                             axis=1).astype(np.float32)
        a0 = np.concatenate([g - p[:, 1], lm0 - p[:, 1], lm1 - p[:, 1], p[:, 0] - p[:, 1],
                             p[:, 2] - p[:, 1]], axis=1).astype(np.float32)
        a1 = np.concatenate([g - p[:, 2], lm0 - p[:, 2], lm1 - p[:, 2], p[:, 0] - p[:, 2],
                             p[:, 1] - p[:, 2]], axis=1).astype(np.float32)
        return adv, a0, a1

    def step(self, actions):
        a = np.asarray(actions)
        u = np.zeros((self.E, 3, 2), dtype=np.float64)
        u[..., 0] = np.where(a == 1, -1.0, np.where(a == 2, 1.0, 0.0))
        u[..., 1] = np.where(a == 3, -1.0, np.where(a == 4, 1.0, 0.0))
        u *= SENSITIVITY
        f = ((u + 0.0) / MASS) * DT
        if INTEGRATE_POS_FIRST:
            self.p_pos += self.p_vel * DT
        self.p_vel = self.p_vel * (1 - DAMPING)
        self.p_vel += f
        if not INTEGRATE_POS_FIRST:
            self.p_pos += self.p_vel * DT
        delta = self.p_pos - self.goal_pos[:, None, :]
        sq = np.square(delta)
        d = np.sqrt(sq[..., 0] + sq[..., 1])  # [E,3]
        r_adv = -d[:, 0]
        # python min(d1, d2) keeps the first on ties; values are equal then anyway
        r_good = -np.where(d[:, 2] < d[:, 1], d[:, 2], d[:, 1]) + d[:, 0]
        return r_good, r_adv
