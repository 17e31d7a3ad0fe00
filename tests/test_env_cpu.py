"""Host env (product) vs the oracle's C env, the AEC bookkeeping (quirk Q1) and the seeded reset stream - CPU only."""
import ctypes as C

import numpy as np

from coevonet_amd.mpe import simple_adversary as sa
from oracle import ref_port as rp


class MpeState(C.Structure):
    _fields_ = [("ppos", C.c_double * 6), ("pvel", C.c_double * 6), ("lm", C.c_double * 4), ("goal", C.c_int)]


def test_reset_stream_addressable_by_ordinal():
    """AEC env resets == ResetStream == oracle's jump-ahead PCG64 reset, for even and odd ordinals"""
    env = sa.SimpleAdversaryAEC()
    env.reset(seed=sa.ENV_SEED)
    stream = sa.ResetStream(sa.ENV_SEED, skip_initial=False)
    goal, apos, lpos = stream.take(40)
    st = rp.Stream().st
    for o in range(40):
        if o > 0:
            env.reset()
        assert env.goal == goal[o] and np.array_equal(env.p_pos, apos[o]) and np.array_equal(env.lm_pos, lpos[o])
        s = MpeState()
        rp.lib().oracle_mpe_reset(*st, o, C.byref(s))
        assert s.goal == goal[o]
        assert np.array_equal(np.array(s.ppos).reshape(3, 2), apos[o])
        assert np.array_equal(np.array(s.lm).reshape(2, 2), lpos[o])
    assert env.n_resets == 40


def test_vec_env_equals_aec_env_and_oracle_env():
    n = 17
    g = np.random.Generator(np.random.PCG64(3))
    goal, apos, lpos = sa.ResetStream().take(n)
    vec = sa.VecSimpleAdversary(goal, apos, lpos)
    aecs = []
    stream = sa.ResetStream(skip_initial=False)
    for i in range(n):
        e = sa.SimpleAdversaryAEC()
        e.reset(seed=sa.ENV_SEED)
        for _ in range(i + 1):
            e.reset()
        aecs.append(e)
    st = rp.Stream().st
    cs = []
    for i in range(n):
        s = MpeState()
        rp.lib().oracle_mpe_reset(*st, i + 1, C.byref(s))
        cs.append(s)
    for cyc in range(25):
        adv, a0, a1 = vec.observe()
        acts = g.integers(0, 5, size=(n, 3))
        for i, e in enumerate(aecs):
            assert np.array_equal(e.observe("adversary_0"), adv[i])
            assert np.array_equal(e.observe("agent_0"), a0[i])
            assert np.array_equal(e.observe("agent_1"), a1[i])
            obs = np.zeros(10, dtype=np.float32)
            rp.lib().oracle_mpe_observe(C.byref(cs[i]), 2, rp._fp(obs))
            assert np.array_equal(obs, a1[i])
            for k, name in enumerate(sa.AGENTS):
                assert e.agent_selection == name
                e.step(int(acts[i, k]))
        rg, ra = vec.step(acts)
        for i, e in enumerate(aecs):
            assert e.rewards["adversary_0"] == ra[i] and e.rewards["agent_0"] == rg[i] == e.rewards["agent_1"]
            a = np.ascontiguousarray(acts[i], dtype=np.int32)
            crg, cra = C.c_double(), C.c_double()
            rp.lib().oracle_mpe_world_step(C.byref(cs[i]), rp._ip(a), C.byref(crg), C.byref(cra))
            assert crg.value == rg[i] and cra.value == ra[i]
    assert all(e.truncations["agent_0"] for e in aecs)


def test_aec_reward_attribution_closed_form():
    """the credit rule the device step kernel uses (cycle c credits adversary/agent_0 with the good reward of world
    step c, agent_1 with the adversary reward of step c+1) == the AEC loop the reference runs (play_MPE :179-190)"""
    g = np.random.Generator(np.random.PCG64(11))
    for limit in (None, 50, 7, 75, 1):
        env = sa.SimpleAdversaryAEC()
        env.reset(seed=sa.ENV_SEED)
        env.reset()
        rewards = {"agent_0": 0, "agent_1": 0, "adversary_0": 0}
        acts, t = [], 0
        for agent in env.agent_iter():
            a = int(g.integers(0, 5))
            acts.append(a)
            env.step(a)
            _, r, term, trunc, _ = env.last()
            rewards[agent] += r
            t += 1
            if limit is not None and t >= limit:
                break
            if term or trunc:
                break
        T = 75 if limit is None else min(limit, 75)
        assert t == T
        goal, apos, lpos = sa.ResetStream().take(1)
        vec = sa.VecSimpleAdversary(goal, apos, lpos)
        acc, rg_prev = [0.0, 0.0, 0.0], 0.0
        for c in range((T + 2) // 3):
            if 3 * c < T:
                acc[0] += rg_prev
            if 3 * c + 1 < T:
                acc[1] += rg_prev
            if 3 * c + 2 < T:
                rg, ra = vec.step(np.array([acts[3 * c:3 * c + 3]]))
                acc[2] += ra[0]
                rg_prev = rg[0]
        assert [rewards["adversary_0"], rewards["agent_0"], rewards["agent_1"]] == acc
