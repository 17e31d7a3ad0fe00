"""Synthetic stand-in for ``pettingzoo.atari`` pong_v3 / boxing_v2 (SURVEY.md 8d cfg 4/5: ALE and its ROMs are not part
of the image, and the reference's own Atari loop does not run).

Same AEC surface as the envs ``initialize_env`` builds at utils/game_logic_functions.py:47-52 (agents ``first_0`` /
``second_0`` alternating, uint8 ``[84, 84, C]`` observations, ``Discrete(n)`` actions), no game dynamics: the frame of
agent-step t is noise keyed by (seed, the game's reset ordinal, t, the action of step t-1); a step "hits" when the action
equals a keyed target; rewards are zero-sum hits with PettingZoo's ``_cumulative_rewards`` bookkeeping.  The device twin
is ``coevo_synth_step`` (csrc/dqn_engine.hip); this host version exists for the drop-in ``play_game`` contract over an
env object and as an independent statement of the same rules.  Everything measured on it is labelled synthetic.
"""
from __future__ import annotations

import numpy as np

ATARI_GAMES = {"pong_v3": 6, "boxing_v2": 18}   # action-space sizes of the two games main.py:27 offers
SYNTH_SEED = 1870300                             # the reference's one env seed (utils/game_logic_functions.py:54)
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10 (counters: uint32 arrays or scalars) -> four uint32 arrays"""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0 & _MASK, p1 & _MASK, n2 & _MASK, p0 & _MASK
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def synth_target(seed, ordinal, t, n_actions):
    o = philox4x32_10(0xFFFFFFFF, t, ordinal & 0xFFFFFFFF, ((ordinal >> 32) & 0xFFFFFFFF) ^ 0x74617267, seed, seed >> 32)
    return int(o[0]) % n_actions


def synth_frame(seed, ordinal, t, last_action, C):
    n16 = 84 * 84 * C // 16
    o = philox4x32_10(np.arange(n16, dtype=np.uint64), t | (last_action << 16), ordinal & 0xFFFFFFFF,
                      ((ordinal >> 32) & 0xFFFFFFFF) ^ 0x66726D65, seed, seed >> 32)
    words = np.stack(o, axis=1).astype("<u4")          # 16 bytes per counter, little-endian words
    return words.view(np.uint8).reshape(84, 84, C)


class _Space:
    def __init__(self, shape=None, n=None):
        self.shape, self.n = shape, n


class SyntheticAtariAEC:
    """AEC env object with the attribute set the reference's loops touch (agents, reset, agent_iter, observe, step,
    last, observation_space, action_space, close)."""

    def __init__(self, game="pong_v3", channels=4, render_mode=None):
        if game not in ATARI_GAMES:
            raise ValueError(f"Unsupported game type: {game}")
        self.game, self.C, self.n_actions = game, int(channels), ATARI_GAMES[game]
        self.possible_agents = ["first_0", "second_0"]
        self.agents = list(self.possible_agents)
        self.seed_value = None
        self.n_resets = 0
        self.synthetic = True

    def observation_space(self, agent):
        return _Space(shape=(84, 84, self.C))

    def action_space(self, agent):
        return _Space(n=self.n_actions)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.seed_value = int(seed)
            self.n_resets = 0
        if self.seed_value is None:
            self.seed_value = SYNTH_SEED
        self.ordinal = self.n_resets      # games are addressed by their ordinal in the reset sequence (quirk Q6)
        self.n_resets += 1
        self.t, self.last_action = 0, 0xFF
        self.agent_selection = "first_0"
        self._cumulative_rewards = {a: 0.0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}

    def agent_iter(self, max_iter=2 ** 63):
        for _ in range(max_iter):
            yield self.agent_selection

    def observe(self, agent):
        return synth_frame(self.seed_value, self.ordinal, self.t, self.last_action, self.C)

    def step(self, action):
        actor = self.agent_selection
        other = "second_0" if actor == "first_0" else "first_0"
        self._cumulative_rewards[actor] = 0.0
        hit = 1.0 if int(action) == synth_target(self.seed_value, self.ordinal, self.t, self.n_actions) else 0.0
        self._cumulative_rewards[actor] += hit
        self._cumulative_rewards[other] += -hit
        self.last_action = int(action)
        self.t += 1
        self.agent_selection = other

    def last(self):
        a = self.agent_selection
        return None, self._cumulative_rewards[a], self.terminations[a], self.truncations[a], {}

    def close(self):
        pass
