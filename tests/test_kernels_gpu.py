"""HIP kernels (through the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact unless stated."""
import ctypes as C

import numpy as np
import pytest
import torch

from coevonet_amd import lib as L
from oracle import ref_port as rp
from tests.util import SAFE_MARGIN, load_golden, sha

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_nets(n, D, seed, mutate=True):
    torch.manual_seed(seed)
    nets = []
    for _ in range(n):
        w = rp.init_net(D)
        if mutate:
            w = rp.mutate_torch(w, D, 0.05)
        nets.append(w)
    return np.stack(nets)


def to_slab(flat_np, D):
    n = flat_np.shape[0]
    flat = torch.from_numpy(flat_np).to(DEV)
    slab = torch.zeros(n, L.fc_slab_stride(D), dtype=torch.float32, device=DEV)
    L.call("coevo_fc_pack", L._p(flat), L._p(slab), n, D)
    return slab


def test_pack_unpack_roundtrip():
    for D in (8, 10):
        flat = make_nets(3, D, seed=D)
        slab = to_slab(flat, D)
        back = torch.zeros(3, L.fc_param_count(D), dtype=torch.float32, device=DEV)
        L.call("coevo_fc_unpack", L._p(slab), L._p(back), 3, D)
        assert np.array_equal(back.cpu().numpy(), flat)
        # spot-check the tiling: W2q[jb][kq][l][c] == fc2.w[64jb+l][4kq+c]
        s = slab[1].cpu().numpy()
        o_w2 = D * 512 + 1536
        w2 = flat[1][o_w2:o_w2 + 256 * 512].reshape(256, 512)
        tile = s[o_w2:o_w2 + 256 * 512].reshape(4, 128, 64, 4)
        assert tile[2, 17, 5, 3] == w2[2 * 64 + 5, 17 * 4 + 3]
        w1 = flat[1][:512 * D].reshape(512, D)
        assert s[3 * 512 + 77] == w1[77, 3]


@pytest.mark.parametrize("D,rows_per_task", [(10, [5, 5, 1, 8, 3]), (8, [5, 2, 8]), (10, [16, 9, 12]),
                                             (8, [32, 17, 1, 29])])
def test_fc_forward_bit_exact_vs_oracle(D, rows_per_task):
    n_tasks = len(rows_per_task)
    flat = make_nets(n_tasks, D, seed=41 + D + len(rows_per_task))
    slab = to_slab(flat, D)
    rows = sum(rows_per_task)
    g = np.random.Generator(np.random.PCG64(5))
    obs = np.zeros((rows, L.OBS_STRIDE), dtype=np.float32)
    obs[:, :D] = g.uniform(-2, 2, size=(rows, D)).astype(np.float32)
    tasks = np.zeros(n_tasks, dtype=L.TASK_DTYPE)
    r0 = 0
    for t, n in enumerate(rows_per_task):
        tasks[t] = ((n_tasks - 1 - t) * L.fc_slab_stride(D), r0, n, D, 0)  # nets used in reverse order
        r0 += n
    d_tasks = L.tasks_to_device(tasks)
    d_obs = torch.from_numpy(obs).to(DEV)
    actions = torch.full((rows,), -7, dtype=torch.int32, device=DEV)
    logits = torch.zeros(rows, L.LOGIT_STRIDE, dtype=torch.float32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.call("coevo_fc_forward_argmax", L._p(slab), L._p(d_tasks), n_tasks, max(rows_per_task), L._p(d_obs),
           L._p(actions), L._p(logits), L._p(status))
    assert status.item() == 0
    got_a, got_l = actions.cpu().numpy(), logits.cpu().numpy()
    r0 = 0
    for t, n in enumerate(rows_per_task):
        w = flat[n_tasks - 1 - t]
        for r in range(r0, r0 + n):
            a, lg, st = rp.fc_forward(w, D, obs[r, :D])
            assert st == 0
            assert np.array_equal(got_l[r, :5].view(np.uint32), lg.view(np.uint32)), (t, r, got_l[r, :5], lg)
            assert got_a[r] == a
        r0 += n


@pytest.mark.parametrize("D,heavy_rows,light_rows", [(10, [16, 9, 12], [5, 1, 8, 3, 2, 7]), (8, [13], [5, 5, 5]),
                                                     (10, [16, 16, 3], [1, 1])])
def test_fc_forward_merged_bit_exact_vs_oracle(D, heavy_rows, light_rows):
    """coevo_fc_forward_merged: both task tables of a host-stepped env cycle in one launch of the lean cycle kernel
    (shared-opponent tasks of <= 16 rows + one task of <= 8 rows per individual), observations given - logits, actions and
    status against the oracle (MPE/fcnetwork.py:37-90)"""
    n_h, n_l = len(heavy_rows), len(light_rows)
    flat = make_nets(n_h + n_l, D, seed=7 + D + n_h)
    slab = to_slab(flat, D)
    rows = sum(heavy_rows) + sum(light_rows)
    g = np.random.Generator(np.random.PCG64(9))
    obs = np.zeros((rows, L.OBS_STRIDE), dtype=np.float32)
    obs[:, :D] = g.uniform(-2, 2, size=(rows, D)).astype(np.float32)
    obs[:, D:] = 123.0    # columns past the net's input width are not inputs (an 8-wide net in a 12-float row)
    stride, r0, net_of_row = L.fc_slab_stride(D), 0, []
    heavy, light = np.zeros(n_h, dtype=L.TASK_DTYPE), np.zeros(n_l, dtype=L.TASK_DTYPE)
    for t, n in enumerate(heavy_rows):
        heavy[t] = (t * stride, r0, n, D, 0)
        net_of_row += [t] * n
        r0 += n
    for t, n in enumerate(light_rows):
        light[t] = ((n_h + t) * stride, r0, n, D, 0)
        net_of_row += [n_h + t] * n
        r0 += n
    d_heavy, d_light, d_obs = L.tasks_to_device(heavy), L.tasks_to_device(light), torch.from_numpy(obs).to(DEV)
    actions = torch.full((rows,), -7, dtype=torch.int32, device=DEV)
    logits = torch.zeros(rows, L.LOGIT_STRIDE, dtype=torch.float32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    L.call("coevo_fc_forward_merged", L._p(slab), L._p(d_heavy), n_h, max(heavy_rows), L._p(d_light), n_l, max(light_rows),
           L._p(d_obs), L._p(actions), L._p(logits), L._p(status))
    assert status.item() == 0
    got_a, got_l = actions.cpu().numpy(), logits.cpu().numpy()
    for r in range(rows):
        a, lg, st = rp.fc_forward(flat[net_of_row[r]], D, obs[r, :D])
        assert st == 0
        assert np.array_equal(got_l[r, :5].view(np.uint32), lg.view(np.uint32)), (r, got_l[r, :5], lg)
        assert got_a[r] == a
    # a table the lean kernel cannot hold (a 17-row task) is refused, not launched
    with pytest.raises(L.CoevoError):
        L.call("coevo_fc_forward_merged", L._p(slab), L._p(d_heavy), n_h, 17, L._p(d_light), n_l, max(light_rows),
               L._p(d_obs), L._p(actions), L._p(logits), L._p(status))


def test_fc_forward_golden_vectors():
    """HIP forward against logits the reference's own FCNetwork produced (fixture), tolerance = fp32
    summation-order noise of a 512-term dot product; action must agree where the margin is safe."""
    for case in load_golden("fc_forward.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        D = case["D"]
        w = rp.init_net(D)
        if case["mutated"]:
            w = rp.mutate_torch(w, D, 0.05)
        assert sha(w) == case["weights"]["sha256"]
        slab = to_slab(w[None], D)
        obs_in = np.array(case["obs"], dtype=np.float32)
        rows = obs_in.shape[0]
        obs = np.zeros((rows, L.OBS_STRIDE), dtype=np.float32)
        obs[:, :D] = obs_in
        tasks = np.zeros(1, dtype=L.TASK_DTYPE)
        tasks[0] = (0, 0, rows, D, 0)
        actions = torch.zeros(rows, dtype=torch.int32, device=DEV)
        logits = torch.zeros(rows, L.LOGIT_STRIDE, dtype=torch.float32, device=DEV)
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        d_tasks, d_obs = L.tasks_to_device(tasks), torch.from_numpy(obs).to(DEV)  # keep alive across the launch
        L.call("coevo_fc_forward_argmax", L._p(slab), L._p(d_tasks), 1, rows, L._p(d_obs), L._p(actions),
               L._p(logits), L._p(status))
        ref = np.array(case["logits"], dtype=np.float32)
        np.testing.assert_allclose(logits.cpu().numpy()[:, :5], ref, rtol=2e-5, atol=2e-6)
        for r in range(rows):
            srt = np.sort(ref[r])[::-1]
            if srt[0] - srt[1] > SAFE_MARGIN:
                assert actions[r].item() == case["actions"][r]


def test_fc_forward_status_bits():
    D = 10
    flat = make_nets(1, D, seed=3)
    flat[0, 5] = np.nan  # a NaN weight in fc1 poisons LayerNorm -> "after fc1" in the reference
    slab = to_slab(flat, D)
    obs = torch.zeros(2, L.OBS_STRIDE, dtype=torch.float32, device=DEV)
    tasks = np.zeros(1, dtype=L.TASK_DTYPE)
    tasks[0] = (0, 0, 2, D, 0)
    actions = torch.zeros(2, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    d_tasks = L.tasks_to_device(tasks)
    L.call("coevo_fc_forward_argmax", L._p(slab), L._p(d_tasks), 1, 2, L._p(obs), L._p(actions), None, L._p(status))
    assert status.item() & 2
    with pytest.raises(ValueError):
        L.raise_on_status(status)
    # inf observation -> "input contains inf or NaN"
    flat = make_nets(1, D, seed=3)
    slab = to_slab(flat, D)
    obs[1, 2] = float("inf")
    status.zero_()
    L.call("coevo_fc_forward_argmax", L._p(slab), L._p(d_tasks), 1, 2, L._p(obs), L._p(actions), None, L._p(status))
    assert status.item() & 1


# ----------------------------------------------------------------------------------- MPE env
def test_mpe_reset_matches_numpy_stream():
    from coevonet_amd.mpe.simple_adversary import ResetStream
    n, first = 301, 1
    st = torch.zeros(L.MPE_STATE_DOUBLES, n, dtype=torch.float64, device=DEV)
    L.call("coevo_mpe_reset", L._p(st), n, 0, n, L.PCG64State.from_seed(1870300), first)
    goal, apos, lpos = ResetStream().take(n)
    s = st.cpu().numpy()
    assert np.array_equal(s[0:6].T.reshape(n, 3, 2), apos)
    assert np.array_equal(s[12:16].T.reshape(n, 2, 2), lpos)
    assert np.array_equal(s[22].astype(np.int32), goal)
    assert np.array_equal(s[16:18].T, lpos[np.arange(n), goal])
    assert not s[6:12].any() and not s[18:22].any()
    # a shard that starts at an odd/even ordinal deep in the stream sees the same values
    st2 = torch.zeros(L.MPE_STATE_DOUBLES, 40, dtype=torch.float64, device=DEV)
    for first2 in (200, 257):
        L.call("coevo_mpe_reset", L._p(st2), 40, 0, 40, L.PCG64State.from_seed(1870300), first2)
        assert np.array_equal(st2.cpu().numpy()[:18], s[:18, first2 - 1:first2 - 1 + 40])


@pytest.mark.parametrize("limit,cycles", [(None, 25), (50, 17), (7, 3), (200, 67)])
def test_mpe_step_observe_match_host_env(limit, cycles):
    from coevonet_amd.mpe.simple_adversary import ResetStream, VecSimpleAdversary
    n = 130
    st = torch.zeros(L.MPE_STATE_DOUBLES, n, dtype=torch.float64, device=DEV)
    L.call("coevo_mpe_reset", L._p(st), n, 0, n, L.PCG64State.from_seed(1870300), 1)
    env = VecSimpleAdversary(*ResetStream().take(n))
    row_game = torch.arange(n, dtype=torch.int32, device=DEV).repeat_interleave(3).contiguous()
    row_slot = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV).repeat(n).contiguous()
    game_rows = torch.arange(3 * n, dtype=torch.int32, device=DEV)
    lim = None if limit is None else torch.full((n,), limit, dtype=torch.int32, device=DEV)
    obs = torch.zeros(3 * n, L.OBS_STRIDE, dtype=torch.float32, device=DEV)
    g = np.random.Generator(np.random.PCG64(9))
    acc = np.zeros((n, 3))  # adv, a0, a1 slots
    rg_prev = np.zeros(n)
    T = 10 ** 9 if limit is None else limit
    for c in range(cycles):
        L.call("coevo_mpe_observe", L._p(st), n, L._p(row_game), L._p(row_slot), 3 * n, L._p(obs))
        o = obs.cpu().numpy().reshape(n, 3, L.OBS_STRIDE)
        adv, a0, a1 = env.observe()
        assert np.array_equal(o[:, 0, :8], adv) and np.array_equal(o[:, 1, :10], a0) and np.array_equal(o[:, 2, :10], a1)
        acts = g.integers(0, 5, size=(n, 3)).astype(np.int32)
        d_act = torch.from_numpy(acts.reshape(-1)).to(DEV)
        L.call("coevo_mpe_step", L._p(st), n, L._p(game_rows), L._p(d_act), c, L._p(lim), 1)
        # host model of the AEC credit rule
        if 3 * c < T:
            acc[:, 0] += rg_prev
        if 3 * c + 1 < T:
            acc[:, 1] += rg_prev
        if 3 * c + 2 < T:
            rg, ra = env.step(acts)
            acc[:, 2] += ra
            rg_prev = rg
    rew = torch.zeros(n, 3, dtype=torch.float64, device=DEV)
    L.call("coevo_mpe_rewards", L._p(st), n, L._p(rew))
    r = rew.cpu().numpy()
    assert np.array_equal(r[:, 0], acc[:, 1]) and np.array_equal(r[:, 1], acc[:, 2]) and np.array_equal(r[:, 2], acc[:, 0])


@pytest.mark.parametrize("limit,max_cycles", [(None, 25), (50, 25), (200, 70), (7, 25)])
def test_device_rollout_matches_oracle_play_game(limit, max_cycles):
    """Whole games on the device (fused observe+policy launch, then step) == oracle play_game, bit for bit."""
    n_games = 12
    nets10 = make_nets(2 * n_games, 10, seed=77, mutate=False)
    nets8 = make_nets(n_games, 8, seed=78, mutate=False)
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(nets10, 10).reshape(-1), to_slab(nets8, 8).reshape(-1)]).contiguous()
    off8 = 2 * n_games * s10
    # rows: game g -> rows 3g (adv), 3g+1 (a0), 3g+2 (a1); one task per row (R=1 tasks)
    tasks = np.zeros(3 * n_games, dtype=L.TASK_DTYPE)
    for g in range(n_games):
        tasks[3 * g] = (off8 + g * s8, 3 * g, 1, 8, 0)
        tasks[3 * g + 1] = (g * s10, 3 * g + 1, 1, 10, 0)
        tasks[3 * g + 2] = ((n_games + g) * s10, 3 * g + 2, 1, 10, 0)
    d_tasks = L.tasks_to_device(tasks)
    row_game = torch.arange(n_games, dtype=torch.int32, device=DEV).repeat_interleave(3).contiguous()
    row_slot = torch.tensor([0, 1, 2], dtype=torch.int32, device=DEV).repeat(n_games).contiguous()
    game_rows = torch.arange(3 * n_games, dtype=torch.int32, device=DEV)
    st = torch.zeros(L.MPE_STATE_DOUBLES, n_games, dtype=torch.float64, device=DEV)
    first = 5
    L.call("coevo_mpe_reset", L._p(st), n_games, 0, n_games, L.PCG64State.from_seed(1870300), first)
    actions = torch.zeros(3 * n_games, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    T = 3 * max_cycles if limit is None else min(limit, 3 * max_cycles)
    lim = torch.full((n_games,), T, dtype=torch.int32, device=DEV)
    for c in range((T + 2) // 3):
        L.call("coevo_mpe_policy_cycle", L._p(slab), L._p(d_tasks), 3 * n_games, 1, L._p(st), n_games,
               L._p(row_game), L._p(row_slot), L._p(actions), L._p(status))
        L.call("coevo_mpe_step", L._p(st), n_games, L._p(game_rows), L._p(actions), c, L._p(lim), 1)
    assert status.item() == 0
    rew = torch.zeros(n_games, 3, dtype=torch.float64, device=DEV)
    L.call("coevo_mpe_rewards", L._p(st), n_games, L._p(rew))
    r = rew.cpu().numpy()
    stream = rp.Stream()
    for g in range(n_games):
        want = rp.play_game(stream, nets10[g], nets10[n_games + g], nets8[g], limit, max_cycles, ordinal=first + g)
        assert want["steps"] == T
        assert list(r[g]) == want["rewards"], (g, r[g], want["rewards"])


@pytest.mark.parametrize("mode,nh,heavy_rows", [
    ("two_launches", 2, 32), ("merged", 2, 32), ("cohorts2", 2, 32), ("cohorts3_eager", 2, 32),
    # the lean kernel (16-row shared-opponent tiles) and every per-individual row-count instantiation (1, 2, 5, 8)
    ("merged", 1, 16), ("merged", 2, 16), ("merged", 5, 16), ("merged", 8, 16), ("cohorts2", 5, 16),
    ("merged", 5, 32), ("merged", 8, 32)])
def test_rollout_variants_match_oracle(mode, nh, heavy_rows):
    """A GA-shaped batch (per-individual nets against shared opponents) through DeviceRollout: the two-launch cycle,
    the merged one-launch cycle (32-row and lean 16-row tiles) and the cohort chains all give the oracle's rewards bit
    for bit."""
    from coevonet_amd.rollout import RolloutPlan, DeviceRollout
    npop, limit, max_cycles = 20, 40, 25
    nets10 = make_nets(npop + nh, 10, seed=91, mutate=False)      # individuals (agent_0) + opponents for agent_1
    nets8 = make_nets(nh, 8, seed=92, mutate=False)               # opponent adversaries
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(nets10, 10).reshape(-1), to_slab(nets8, 8).reshape(-1)]).contiguous()
    off = [i * s10 for i in range(npop + nh)] + [(npop + nh) * s10 + k * s8 for k in range(nh)]
    D = [10] * (npop + nh) + [8] * nh
    games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]   # (adversary, agent_0, agent_1)
    K = {"two_launches": 1, "merged": 1, "cohorts2": 2, "cohorts3_eager": 3}[mode]
    plan = RolloutPlan(np.array(games), off, D, device=DEV, n_cohorts=K, heavy_rows=heavy_rows)
    assert plan.n_cohorts == K and len(plan.heavy_np) > 0 and len(plan.light_np) == npop
    assert plan.light_max == nh and plan.heavy_max <= heavy_rows
    ro = DeviceRollout(plan, slab, merged=(mode != "two_launches"))
    ro.use_graph = mode != "cohorts3_eager"
    T = min(limit, 3 * max_cycles)
    ro.set_limits(np.full(plan.n_games, T))
    first = 3
    ro.reset(0, plan.n_games, first)
    ro.run((T + 2) // 3)
    torch.cuda.synchronize()
    ro.check_status()
    r = ro.rewards.cpu().numpy()
    stream = rp.Stream()
    for g, (adv, a0, a1) in enumerate(games):
        want = rp.play_game(stream, nets10[a0], nets10[a1], nets8[adv - npop - nh], limit, max_cycles, ordinal=first + g)
        assert list(r[g]) == want["rewards"], (mode, g, r[g], want["rewards"])


def _small_launch_setup(npop, nh, heavy_rows, cohorts):
    from coevonet_amd.rollout import RolloutPlan
    nets10 = make_nets(npop + nh, 10, seed=191, mutate=False)
    nets8 = make_nets(nh, 8, seed=192, mutate=False)
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(nets10, 10).reshape(-1), to_slab(nets8, 8).reshape(-1)]).contiguous()
    off = [i * s10 for i in range(npop + nh)] + [(npop + nh) * s10 + k * s8 for k in range(nh)]
    D = [10] * (npop + nh) + [8] * nh
    games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]   # (adversary, agent_0, agent_1)
    plan = RolloutPlan(np.array(games), off, D, device=DEV, n_cohorts=cohorts, heavy_rows=heavy_rows)
    return plan, slab, games, nets10, nets8


@pytest.mark.parametrize("limit", [40, 1, 4, 75, 200])
def test_persistent_rollout_leaves_what_the_per_cycle_launches_leave(limit):
    """coevo_mpe_rollout_persistent (ONE launch for the n_cycles of a small cohort: rows keep their games in LDS and hand each
    other tagged action words) against the chain of fc_cycle_small_kernel launches: the same rewards, and the same state buffer
    / plain action words for coevo_mpe_final_step - bit for bit; 1 cycle (nothing to wait for), 2 cycles, an odd and an even
    last cycle"""
    from coevonet_amd.rollout import DeviceRollout
    plan, slab, games, _, _ = _small_launch_setup(25, 5, 5, 1)
    got = []
    for persistent in (True, False):
        ro = DeviceRollout(plan, slab, merged=True)
        if not persistent:
            ro.sync_words, ro.desc.sync_words = None, None
        ro.use_graph = False
        T = limit   # (200: the T = 200 variant's 67 cycles, SURVEY 8d cfg 2-T200)
        ro.set_limits(np.full(plan.n_games, T))
        ro.reset(0, plan.n_games, 11)
        n_cycles = (T + 2) // 3
        ro.run(n_cycles)
        torch.cuda.synchronize()
        ro.check_status()
        last = n_cycles - 1
        st = (ro.state2[0] if last <= 0 or (last & 1) == 0 else ro.state2[1])[:22].cpu().numpy()
        act = ro.actions_by_game[(last if last > 0 else 0) & 1].cpu().numpy()
        got.append((ro.rewards.cpu().numpy(), st, act))
    for a, b in zip(*got):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


@pytest.mark.parametrize("npop,nh,heavy_rows,cohorts,limit", [(7, 1, 1, 1, 9), (13, 2, 2, 1, 30), (31, 3, 8, 1, 75), (40, 8, 8, 2, 21),
                                                              (64, 4, 3, 1, 48), (17, 5, 5, 2, 75), (100, 1, 8, 1, 12), (9, 7, 7, 1, 5)])
def test_persistent_rollout_equals_per_cycle_launches_over_shapes(npop, nh, heavy_rows, cohorts, limit):
    """every row-count instantiation (1, 2, 5, 8), ragged chunks, one and two cohorts, odd and even cycle counts: ONE persistent
    launch per cohort == the chain of launches per env-cycle - rewards, final state buffer, plain action words, bit for bit"""
    from coevonet_amd.rollout import DeviceRollout
    plan, slab, games, _, _ = _small_launch_setup(npop, nh, heavy_rows, cohorts)
    for k in range(plan.n_cohorts):
        n_h = int(plan.heavy_begin_np[k + 1] - plan.heavy_begin_np[k])
        n_l = int(plan.light_begin_np[k + 1] - plan.light_begin_np[k])
        assert L.load().coevo_mpe_persistent_fits(n_h, n_l, plan.heavy_max, plan.light_max, plan.n_cohorts) == 1
    got = []
    for persistent in (True, False):
        ro = DeviceRollout(plan, slab, merged=True)
        if not persistent:
            ro.sync_words, ro.desc.sync_words = None, None
        ro.set_limits(np.full(plan.n_games, limit))
        ro.reset(0, plan.n_games, 5)
        n_cycles = (limit + 2) // 3
        ro.run(n_cycles)
        torch.cuda.synchronize()
        ro.check_status()
        last = n_cycles - 1
        st = (ro.state2[0] if last <= 0 or (last & 1) == 0 else ro.state2[1])[:22].cpu().numpy()
        act = ro.actions_by_game[(last if last > 0 else 0) & 1].cpu().numpy()
        got.append((ro.rewards.cpu().numpy(), st, act))
    for a, b in zip(*got):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_persistent_rollout_times_out_instead_of_hanging():
    """a task list that lacks one seat of some games: their other rows wait, give up after COEVO_SYNC_SPINS polls, raise the
    abort word, every workgroup leaves and the status word carries COEVO_ST_SYNC_TIMEOUT (no reference counterpart)"""
    from coevonet_amd.rollout import DeviceRollout
    plan, slab, games, _, _ = _small_launch_setup(25, 5, 5, 1)
    ro = DeviceRollout(plan, slab, merged=True)
    ro.set_limits(np.full(plan.n_games, 40))
    ro.reset(0, plan.n_games, 3)
    lib = L.load()
    rc = lib.coevo_mpe_rollout_persistent(
        L._p(slab), L._p(plan.heavy), len(plan.heavy_np), L._p(plan.light), len(plan.light_np) - 1, plan.light_max, plan.heavy_max,
        L._p(ro.state2[0]), L._p(ro.state2[1]), plan.n_games, L._p(plan.row_game), L._p(plan.row_slot), L._p(ro.actions_by_game),
        L._p(ro.limits), 3, ro.pos_first, L._p(ro.status), None, L._p(ro.sync_words), 1, None, None, 0, L._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(ro.status.item()) & 32
    with pytest.raises(ValueError, match="SYNC_TIMEOUT"):
        ro.check_status()


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("npop,nh,heavy_rows,cohorts", [(25, 5, 5, 1), (50, 5, 8, 1), (20, 8, 8, 1), (20, 2, 5, 1), (23, 1, 7, 1),
                                                        (26, 5, 5, 2), (90, 5, 5, 1)])
def test_small_launch_kernel_matches_oracle(npop, nh, heavy_rows, cohorts, persistent):
    """fc_cycle_small_kernel (every task <= 8 rows through the per-individual body, fc2 as v_fmac_f32 with DPP row_newbcast
    activations - the launch shape of ONE RANK of a sharded population, genetic_algorithm.py:125-217 split by index: 25 / 50
    individuals per role, shared opponents cut into hof- or 8-row chunks) == the oracle's play_game, bit for bit: every row-count
    instantiation (1, 2, 5, 8), ragged last chunks, two cohorts side by side; as one launch per env-cycle and as ONE persistent
    launch per cohort (fc_rollout_small_kernel)"""
    from coevonet_amd.rollout import RolloutPlan, DeviceRollout
    limit, max_cycles = 40, 25
    nets10 = make_nets(npop + nh, 10, seed=191, mutate=False)
    nets8 = make_nets(nh, 8, seed=192, mutate=False)
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(nets10, 10).reshape(-1), to_slab(nets8, 8).reshape(-1)]).contiguous()
    off = [i * s10 for i in range(npop + nh)] + [(npop + nh) * s10 + k * s8 for k in range(nh)]
    D = [10] * (npop + nh) + [8] * nh
    games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]   # (adversary, agent_0, agent_1)
    plan = RolloutPlan(np.array(games), off, D, device=DEV, n_cohorts=cohorts, heavy_rows=heavy_rows)
    assert plan.n_cohorts == cohorts and plan.heavy_max <= heavy_rows <= 8 and plan.light_max == nh
    for k in range(cohorts):
        n_h = int(plan.heavy_begin_np[k + 1] - plan.heavy_begin_np[k])
        n_l = int(plan.light_begin_np[k + 1] - plan.light_begin_np[k])
        assert L.load().coevo_mpe_persistent_fits(n_h, n_l, plan.heavy_max, plan.light_max, cohorts) == 1
        # (90 individuals: 270 workgroups, two per CU on some - persistent only; per cycle that shape runs the lean kernel)
        form = L.load().coevo_mpe_cycle_kernel_form(n_h, n_l, plan.heavy_max, plan.light_max, cohorts)
        assert form == (3 if npop < 90 else 2)   # COEVO_CYCLE_FORM_SMALL / _LEAN16
    ro = DeviceRollout(plan, slab, merged=True)
    if not persistent:
        ro.sync_words, ro.desc.sync_words = None, None
    T = min(limit, 3 * max_cycles)
    ro.set_limits(np.full(plan.n_games, T))
    first = 3
    ro.reset(0, plan.n_games, first)
    ro.run((T + 2) // 3)
    torch.cuda.synchronize()
    ro.check_status()
    r = ro.rewards.cpu().numpy()
    stream = rp.Stream()
    for g, (adv, a0, a1) in enumerate(games):
        want = rp.play_game(stream, nets10[a0], nets10[a1], nets8[adv - npop - nh], limit, max_cycles, ordinal=first + g)
        assert list(r[g]) == want["rewards"], (g, r[g], want["rewards"])


def test_paired_streaming_workgroups_odd_count():
    """more tasks than the 32-row merged kernel has workgroup slots -> two per-individual nets per workgroup; an odd
    number of them leaves the last workgroup's second net absent (it must write nothing).  523 one-row tasks."""
    from coevonet_amd.rollout import RolloutPlan, DeviceRollout
    npop, limit, max_cycles = 523, 12, 25
    base10 = make_nets(3, 10, seed=11, mutate=False)
    base8 = make_nets(1, 8, seed=12, mutate=False)
    rng = np.random.default_rng(5)
    nets10 = [base10[0] + (rng.standard_normal(base10[0].shape) * 0.01).astype(np.float32) for _ in range(npop)]
    nets10 += [base10[1]]                                          # the shared agent_1 opponent
    s10, s8 = L.fc_slab_stride(10), L.fc_slab_stride(8)
    slab = torch.cat([to_slab(np.stack(nets10), 10).reshape(-1), to_slab(base8, 8).reshape(-1)]).contiguous()
    off = [i * s10 for i in range(npop + 1)] + [(npop + 1) * s10]
    D = [10] * (npop + 1) + [8]
    games = [(npop + 1, i, npop) for i in range(npop)]             # (adversary, agent_0, agent_1)
    plan = RolloutPlan(np.array(games), off, D, device=DEV, heavy_rows=32)
    assert len(plan.light_np) == npop and len(plan.heavy_np) + npop > 512 and npop % 2 == 1
    ro = DeviceRollout(plan, slab, merged=True)
    T = min(limit, 3 * max_cycles)
    ro.set_limits(np.full(plan.n_games, T))
    ro.reset(0, plan.n_games, 7)
    ro.run((T + 2) // 3)
    torch.cuda.synchronize()
    ro.check_status()
    r = ro.rewards.cpu().numpy()
    stream = rp.Stream()
    for g in list(range(0, npop, 37)) + [npop - 2, npop - 1]:
        want = rp.play_game(stream, nets10[g], nets10[npop], base8[0], limit, max_cycles, ordinal=7 + g)
        assert list(r[g]) == want["rewards"], (g, r[g], want["rewards"])


# ----------------------------------------------------------------------------------- offspring
@pytest.mark.parametrize("D,skip_ln", [(10, 0), (8, 1)])
def test_perturb_bit_exact_vs_oracle(D, skip_ln):
    P = L.fc_param_count(D)
    parents = make_nets(3, D, seed=11)
    slab = to_slab(parents, D)
    n_children = 5
    pidx = np.array([2, 0, 1, 2, 0], dtype=np.int32)
    child = torch.zeros(1 + n_children, L.fc_slab_stride(D), dtype=torch.float32, device=DEV)
    sigma = torch.tensor([0.05], dtype=torch.float32, device=DEV)
    seed, slo, shi = 0x1234567890ABCDEF, 1000, 7
    d_pidx = torch.from_numpy(pidx).to(DEV)
    L.call("coevo_fc_perturb", L._p(slab), L._p(d_pidx), L._p(child), 1, n_children, D,
           L._p(sigma), seed, slo, shi, skip_ln)
    back = torch.zeros(1 + n_children, P, dtype=torch.float32, device=DEV)
    L.call("coevo_fc_unpack", L._p(child), L._p(back), 1 + n_children, D)
    got = back.cpu().numpy()
    segs = rp.ln_segments(D) if skip_ln else []
    for c in range(n_children):
        want = rp.perturb_philox_flat(parents[pidx[c]], np.float32(0.05), seed, slo + c, shi, segs)
        assert np.array_equal(got[1 + c].view(np.uint32), want.view(np.uint32)), c
    noise = got[1] - parents[2]
    if not skip_ln:
        assert abs(noise.mean()) < 1e-3 and abs(noise.std() - 0.05) < 1e-3  # it is N(0, sigma)
    assert not got[0].any()  # slot 0 untouched


def test_antithetic_perturb_and_centered_ranks_vs_oracle():
    """the cfg 3 extension mode's two kernels: antithetic pairs share a stream with opposite signs; centered ranks"""
    D, n = 10, 6
    P = L.fc_param_count(D)
    parent = make_nets(1, D, seed=13)
    slab = to_slab(parent, D)
    child = torch.zeros(n, L.fc_slab_stride(D), dtype=torch.float32, device=DEV)
    sigma = torch.tensor([0.05], dtype=torch.float32, device=DEV)
    zero = torch.zeros(n, dtype=torch.int32, device=DEV)
    first = 10   # global index of the first individual of this call (a shard boundary is always even)
    L.call("coevo_fc_perturb_flags", L._p(slab), L._p(zero), L._p(child), 0, n, D, L._p(sigma), 77, first, 5, 3)
    back = torch.zeros(n, P, dtype=torch.float32, device=DEV)
    L.call("coevo_fc_unpack", L._p(child), L._p(back), n, D)
    got = back.cpu().numpy()
    for c in range(n):
        j = first + c
        want = rp.mutate_philox(parent[0], D, np.float32(0.05), 77, j >> 1, 5, skip_layernorm=True, negate=bool(j & 1))
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c
    lin = np.ones(P, dtype=bool)
    for o, ln_n in rp.ln_segments(D):
        lin[o:o + ln_n] = False
    np.testing.assert_allclose((got[0] - parent[0])[lin], -(got[1] - parent[0])[lin], atol=1e-6)  # opposite noise
    assert np.abs((got[0] - parent[0])[lin]).max() > 0.1
    g = np.random.Generator(np.random.PCG64(3))
    for m in (1, 2, 7, 1000, 5000):
        f = g.normal(size=m).astype(np.float32)
        if m > 5:
            f[3] = f[1]
            f[m - 1] = f[1]
        d_f = torch.from_numpy(f).to(DEV)
        out = torch.zeros(m, dtype=torch.float32, device=DEV)
        L.call("coevo_centered_ranks", L._p(d_f), m, L._p(out))
        assert np.array_equal(out.cpu().numpy(), rp.centered_ranks(f)), m


def test_es_update_bit_exact_vs_oracle():
    D, n = 8, 37
    P = L.fc_param_count(D)
    theta = make_nets(1, D, seed=21)
    stride = L.fc_slab_stride(D)
    slab = torch.zeros(1 + n, stride, dtype=torch.float32, device=DEV)
    slab[0] = to_slab(theta, D)[0]
    g = np.random.Generator(np.random.PCG64(4))
    fit = g.normal(size=n).astype(np.float32)
    sigma, lr, seed, shi = np.float32(0.05), np.float32(0.1), 99, 3
    d_sigma = torch.tensor([sigma], device=DEV)
    zero_idx = torch.zeros(n, dtype=torch.int32, device=DEV)
    L.call("coevo_fc_perturb", L._p(slab), L._p(zero_idx), L._p(slab), 1, n, D, L._p(d_sigma), seed, 0, shi, 1)
    pert = torch.zeros(n, P, dtype=torch.float32, device=DEV)
    L.call("coevo_fc_unpack", slab.data_ptr() + 4 * stride, L._p(pert), n, D)
    d_fit = torch.from_numpy(fit).to(DEV)
    L.call("coevo_es_update", L._p(slab), slab.data_ptr() + 4 * stride, D, L._p(d_fit), n, L._p(d_sigma), C.c_float(lr))
    back = torch.zeros(1, P, dtype=torch.float32, device=DEV)
    L.call("coevo_fc_unpack", L._p(slab), L._p(back), 1, D)
    want = rp.es_update_from_pert(theta[0], D, pert.cpu().numpy(), fit, sigma, lr, chunks=1)
    assert np.array_equal(back.cpu().numpy()[0].view(np.uint32), want.view(np.uint32))
    assert np.abs(want - theta[0]).max() > 0  # it moved
    # the chunked form (coevo_es_partial + coevo_es_apply): the canonical summation of the engine, also computed in two
    # "rank" halves into a rank-major partial buffer as the sharded run does
    for chunks, world in [(1, 1), (8, 1), (8, 2), (6, 3), (64, 1)]:
        slab[0] = to_slab(theta, D)[0]
        cl = chunks // world
        parts = torch.full((world * cl * stride + 5,), float("nan"), dtype=torch.float32, device=DEV)
        for rank in range(world):
            lo = rank * cl * n // chunks
            L.call("coevo_es_partial", L._p(slab), slab.data_ptr() + 4 * stride * (1 + lo), lo, D, L._p(d_fit), n, chunks,
                   rank * cl, cl, parts.data_ptr() + 4 * rank * cl * stride)
        L.call("coevo_es_apply", L._p(slab), L._p(parts), chunks, cl, cl * stride, D, n, L._p(d_sigma), C.c_float(lr))
        L.call("coevo_fc_unpack", L._p(slab), L._p(back), 1, D)
        want_c = rp.es_update_from_pert(theta[0], D, pert.cpu().numpy(), fit, sigma, lr, chunks=chunks)
        assert np.array_equal(back.cpu().numpy()[0].view(np.uint32), want_c.view(np.uint32)), (chunks, world)
    assert not np.array_equal(want_c, want)  # a different summation order, not a no-op
    ln = np.zeros(P, dtype=bool)
    for o, ln_n in rp.ln_segments(D):
        ln[o:o + ln_n] = True
    assert np.array_equal(want[ln], theta[0][ln])  # LayerNorm affine untouched (MPE/fcnetwork.py:185-199)


# ----------------------------------------------------------------------------------- selection
def test_diversity_fitness_rank():
    D, n = 10, 23
    pop = make_nets(n, D, seed=31)
    ref = pop[n - 1].copy()
    slab = to_slab(pop, D)
    ref_slab = to_slab(ref[None], D)
    dist = torch.zeros(n, dtype=torch.float32, device=DEV)
    score = torch.zeros(1, dtype=torch.float32, device=DEV)
    L.call("coevo_fc_diversity", L._p(ref_slab), L._p(slab), n, D, L._p(dist), L._p(score))
    segs = rp.linear_segments(D)
    so = np.array([s[0] for s in segs], dtype=np.int32)
    sl = np.array([s[1] for s in segs], dtype=np.int32)
    want_d = np.zeros(n, dtype=np.float32)
    rp.lib().oracle_diversity.restype = C.c_double
    want_s = rp.lib().oracle_diversity(rp._fp(ref), rp._fp(pop), n, pop.shape[1], rp._ip(so), rp._ip(sl), len(segs),
                                       rp._fp(want_d))
    np.testing.assert_allclose(dist.cpu().numpy(), want_d, rtol=1.2e-7)  # <= 1 ulp: fp64 sums, rounded once
    np.testing.assert_allclose(score.item(), want_s, rtol=1e-6)
    # the reference's own numpy expression
    np.testing.assert_allclose(score.item(), rp.diversity(rp.weights_es(ref, D), [rp.weights_es(w, D) for w in pop]),
                               rtol=1e-5)
    # GA fitness (quirk Q2) + ranking
    popn, hof = 9, 3
    g = np.random.Generator(np.random.PCG64(8))
    rewards = g.normal(size=(40, 3)) * 10
    d_rew = torch.from_numpy(rewards).to(DEV)
    fit = torch.zeros(popn, dtype=torch.float32, device=DEV)
    L.call("coevo_ga_fitness", L._p(d_rew), 4, popn, hof, hof, 1, L._p(score), L._p(fit))
    div = np.float32(score.item())
    want = [np.float32(rewards[4 + i * hof + hof - 1, 1] / hof) / (1 + div) for i in range(popn)]
    assert np.array_equal(fit.cpu().numpy(), np.array(want, dtype=np.float32))
    f = g.normal(size=200).astype(np.float32)
    f[17] = f[3]
    f[150] = f[3]  # ties
    order = torch.zeros(200, dtype=torch.int32, device=DEV)
    d_f = torch.from_numpy(f).to(DEV)
    L.call("coevo_rank_desc", L._p(d_f), 200, L._p(order))
    assert np.array_equal(order.cpu().numpy(), np.argsort(f, kind="stable")[::-1])


@pytest.mark.parametrize("pop", [37, 1600])
def test_fused_select_and_promote_equal_separate_launches(pop):
    """coevo_ga_select / coevo_ga_promote (one launch for the three roles) == the per-role launches they replace
    (pop 1600 = the global population of an 8-GPU weak-scaling run: several ranking slices per role)"""
    rng = np.random.default_rng(3)
    hof, E, gpi = 4, 3, 4
    n_games = 3 * pop * gpi
    rewards = torch.from_numpy(rng.normal(size=(n_games, 3)) * 10).to(DEV)
    Ds = [10, 10, 8]
    sel = (L.GaSelectRole * 3)()
    keep, want = [], []
    for ri in range(3):
        dist = torch.from_numpy(rng.random(pop).astype(np.float32) * 3).to(DEV)
        div0, fit0 = torch.zeros(1, device=DEV), torch.zeros(pop, device=DEV)
        order0 = torch.zeros(pop, dtype=torch.int32, device=DEV)
        L.call("coevo_sharing_score", L._p(dist), pop, L._p(div0))
        L.call("coevo_ga_fitness", L._p(rewards), ri * pop * gpi, pop, gpi, hof, ri, L._p(div0), L._p(fit0))
        L.call("coevo_rank_desc", L._p(fit0), pop, L._p(order0))
        div1, fit1 = torch.zeros(1, device=DEV), torch.zeros(pop, device=DEV)
        order1 = torch.zeros(pop, dtype=torch.int32, device=DEV)
        best = torch.zeros(1, device=DEV)
        sel[ri] = L.GaSelectRole(L._p(dist), L._p(rewards), L._p(div1), L._p(fit1), L._p(order1), L._p(best),
                                 ri * pop * gpi, ri)
        keep.append((dist, div1, fit1, order1, best))
        want.append((div0, fit0, order0))
    L.call("coevo_ga_select", sel, 3, pop, gpi, hof)
    torch.cuda.synchronize()
    for (dist, div1, fit1, order1, best), (div0, fit0, order0) in zip(keep, want):
        assert torch.equal(div0, div1) and torch.equal(fit0, fit1) and torch.equal(order0, order1)
        assert best.item() == dist[int(order0[0])].item()

    pro = (L.GaPromoteRole * 3)()
    checks = []
    iota = torch.arange(16, dtype=torch.int32, device=DEV)
    shift = torch.arange(1, 16, dtype=torch.int32, device=DEV)
    for ri, D in enumerate(Ds):
        stride = L.fc_slab_stride(D)
        order = torch.from_numpy(rng.permutation(pop).astype(np.int32)).to(DEV)
        if ri == 1:
            order[0] = 0   # the best already sits in pop[0]
        slabs = [torch.from_numpy(rng.normal(size=n * stride).astype(np.float32)).to(DEV) for n in (pop, hof, E)]
        ref = [x.clone() for x in slabs] + [torch.zeros(hof * stride, device=DEV)]
        rp_, rh, re_, rt = ref
        L.call("coevo_fc_gather", L._p(rp_), L._p(order), L._p(re_), 0, E, D)
        L.call("coevo_fc_gather", L._p(rh), L._p(shift), L._p(rt), 0, hof - 1, D)
        L.call("coevo_fc_gather", L._p(rt), L._p(iota), L._p(rh), 0, hof - 1, D)
        L.call("coevo_fc_gather", L._p(re_), L._p(iota), L._p(rh), hof - 1, 1, D)
        L.call("coevo_fc_gather", L._p(re_), L._p(iota), L._p(rp_), 0, 1, D)
        pro[ri] = L.GaPromoteRole(L._p(slabs[0]), L._p(slabs[1]), L._p(slabs[2]), L._p(order), D, 1, 1, 0)
        checks.append((slabs, ref, order))
    L.call("coevo_ga_promote", pro, 3, E, hof)
    torch.cuda.synchronize()
    for slabs, ref, _ in checks:
        for got, exp in zip(slabs, ref[:3]):
            assert torch.equal(got, exp)


@pytest.mark.parametrize("world,n_local,hof,E", [(8, 25, 5, 2), (4, 50, 5, 3), (2, 7, 3, 1)])
def test_packed_exchange_launches_equal_separate_launches(world, n_local, hof, E):
    """the fused exchange of a population-sharded generation (genetic_algorithm.py:125-217 by individual index, :223-252 on every
    rank) against the launches it replaces: coevo_mpe_final_step_pack == coevo_mpe_final_step + the pack copies;
    coevo_ga_select_gathered off the rank-major gathered buffer == coevo_ga_select on the unpacked arrays;
    coevo_ga_promote_rebuild == coevo_fc_gather + coevo_fc_rebuild_elites + coevo_ga_promote (host generation and device
    counter forms); coevo_fc_distance_finalize_multi_tick == finalize + counter_add - bit for bit"""
    import ctypes as ct
    rng = np.random.default_rng(world)
    pop = world * n_local
    # ---- closing step + pack: random state / actions of 3 * n_local * hof + 10 games
    n = 3 * n_local * hof + 10
    st = torch.from_numpy(rng.normal(size=(L.MPE_STATE_DOUBLES, n))).to(DEV)
    act = torch.from_numpy(rng.integers(0, 5, size=(n, 3)).astype(np.int32)).to(DEV)
    lim = torch.from_numpy(rng.integers(60, 76, size=n).astype(np.int32)).to(DEV)
    dist_all = torch.from_numpy(rng.random((3, pop)).astype(np.float32) * 3).to(DEV)
    lo = (world - 1) * n_local
    rew0, rew1 = torch.zeros(n, 3, dtype=torch.float64, device=DEV), torch.zeros(n, 3, dtype=torch.float64, device=DEV)
    pack = torch.full((3, n_local, 4), -7.0, dtype=torch.float64, device=DEV)
    L.call("coevo_mpe_final_step", L._p(st), n, L._p(act), 24, L._p(lim), 1, L._p(rew0))
    L.call("coevo_mpe_final_step_pack", L._p(st), n, L._p(act), 24, L._p(lim), 1, L._p(rew1), L._p(pack), L._p(dist_all), 3,
           n_local, hof, pop, lo)
    torch.cuda.synchronize()
    assert torch.equal(rew0, rew1)
    last = rew0[:3 * n_local * hof].view(3, n_local, hof, 3)[:, :, hof - 1]
    assert torch.equal(pack[:, :, :3], last) and torch.equal(pack[:, :, 3], dist_all[:, lo:lo + n_local].double())
    # ---- selection off the gathered buffer
    gathered = torch.from_numpy(rng.normal(size=(world, 3, n_local, 4)) * 10).to(DEV)
    gathered[..., 3] = torch.from_numpy(rng.random((world, 3, n_local)).astype(np.float32) * 3).to(DEV).double()
    unpack = gathered.permute(1, 0, 2, 3).reshape(3, pop, 4).contiguous()
    outs = []
    for form in ("separate", "gathered"):
        sel = (L.GaSelectRole * 3)()
        keep = []
        for ri in range(3):
            d32 = unpack[ri, :, 3].float().contiguous()
            r3 = unpack[ri, :, :3].contiguous()
            div, fit = torch.zeros(1, device=DEV), torch.zeros(pop, device=DEV)
            order, best = torch.zeros(pop, dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV)
            sel[ri] = L.GaSelectRole(L._p(d32) if form == "separate" else None, L._p(r3) if form == "separate" else None,
                                     L._p(div), L._p(fit), L._p(order), L._p(best), 0, [0, 1, 2][ri])
            keep.append((d32, r3, div, fit, order, best))
        if form == "separate":
            L.call("coevo_ga_select", sel, 3, pop, 1, hof)
        else:
            L.call("coevo_ga_select_gathered", sel, 3, pop, hof, L._p(gathered), n_local)
        torch.cuda.synchronize()
        outs.append([(k[2].clone(), k[3].clone(), k[4].clone(), k[5].clone()) for k in keep])
    for a, b in zip(*outs):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert L.load().coevo_ga_select_gathered(sel, 3, pop + 1, hof, L._p(gathered), n_local, None) == -1   # pop % n_local
    # ---- promotion with the elites rebuilt in the launch
    Ds = [10, 10, 8]
    iota = torch.arange(16, dtype=torch.int32, device=DEV)
    gen = 5
    sigma = torch.tensor([0.05, 0.031, 0.07], dtype=torch.float32, device=DEV)
    gen_dev = torch.tensor([gen], dtype=torch.int32, device=DEV)
    for device_gen in (False, True):
        pro = (L.GaPromoteRole * 3)()
        checks = []
        for ri, D in enumerate(Ds):
            stride = L.fc_slab_stride(D)
            order = torch.from_numpy(rng.permutation(pop).astype(np.int32)).to(DEV)
            if ri == 1:
                order[0] = 0          # the unchanged best stays the best
            if ri == 2 and E > 1:
                order[E - 1] = 0      # ... or survives as a lesser elite
            slabs = [torch.from_numpy(rng.normal(size=k * stride).astype(np.float32)).to(DEV) for k in (1, hof, E)]
            rp_, rh, re_ = [x.clone() for x in slabs]
            prev = torch.zeros(E * stride, device=DEV)
            L.call("coevo_fc_gather", L._p(re_), L._p(iota), L._p(prev), 0, E, D)
            L.call("coevo_fc_rebuild_elites", L._p(prev), L._p(order), L._p(re_), E, D, sigma.data_ptr() + 4 * ri, 77,
                   ri if device_gen else (gen - 1) * 4 + ri, L._p(gen_dev) if device_gen else None)
            one = (L.GaPromoteRole * 1)(L.GaPromoteRole(L._p(rp_), L._p(rh), L._p(re_), L._p(order), D, 0, 1, 0))
            L.call("coevo_ga_promote", one, 1, E, hof)
            pro[ri] = L.GaPromoteRole(L._p(slabs[0]), L._p(slabs[1]), L._p(slabs[2]), L._p(order), D, 0, 1, 0)
            checks.append((slabs, (rp_, rh, re_), order))
        L.call("coevo_ga_promote_rebuild", pro, 3, E, hof, L._p(sigma), 77, 0 if device_gen else (gen - 1) * 4,
               L._p(gen_dev) if device_gen else None)
        torch.cuda.synchronize()
        for slabs, ref, _ in checks:
            for got, exp in zip(slabs, ref):
                assert torch.equal(got, exp)
    # ---- distance reduction + counter tick
    nb = 7
    part = torch.from_numpy(rng.random((3, 9, nb))).to(DEV)
    d0, d1 = torch.zeros(3, 12, device=DEV), torch.zeros(3, 12, device=DEV)
    head = torch.tensor([4.5], device=DEV)
    cnt = torch.tensor([41], dtype=torch.int32, device=DEV)
    for dst, tick in ((d0, False), (d1, True)):
        fj = (L.FinalizeJob * 3)()
        for ri in range(3):
            fj[ri] = L.FinalizeJob(part[ri].data_ptr(), dst[ri].data_ptr(), L._p(head) if ri == 0 else None, nb, 9 - ri, 1, 0)
        if tick:
            L.call("coevo_fc_distance_finalize_multi_tick", ct.cast(fj, ct.c_void_p), 3, L._p(cnt))
        else:
            L.call("coevo_fc_distance_finalize_multi", ct.cast(fj, ct.c_void_p), 3)
    torch.cuda.synchronize()
    assert torch.equal(d0, d1) and cnt.item() == 42 and d0[0, 0].item() == 4.5
    # ---- selection + sigma rule in one launch == the two launches; promotion + tick; reset + stamp re-arm
    cap, genv = 40, 14
    ev = torch.from_numpy(rng.normal(size=(n_local * 3 + 10, 3))).to(DEV)
    res = []
    for fusedf in (False, True):
        hist = torch.from_numpy(np.random.default_rng(9).normal(size=(3, cap))).to(DEV)
        sigh = torch.zeros(3, cap, dtype=torch.float64, device=DEV)
        s64 = torch.tensor([0.05, 0.08, 0.03], dtype=torch.float64, device=DEV)
        s32, s32p = torch.zeros(3, device=DEV), torch.full((3,), -1.0, device=DEV)
        s32.copy_(s64.float())
        gd = torch.tensor([genv], dtype=torch.int32, device=DEV)
        sel = (L.GaSelectRole * 3)()
        keep = []
        for ri in range(3):
            div, fit = torch.zeros(1, device=DEV), torch.zeros(pop, device=DEV)
            order, best = torch.zeros(pop, dtype=torch.int32, device=DEV), torch.zeros(1, device=DEV)
            sel[ri] = L.GaSelectRole(None, None, L._p(div), L._p(fit), L._p(order), L._p(best), 0, ri)
            keep.append((div, fit, order, best))
        if fusedf:
            ad = L.GaAdaptArgs(L._p(ev), L._p(gd), L._p(hist), L._p(sigh), L._p(s64), L._p(s32), L._p(s32p), 0.02, 0.1,
                               3 * n_local, cap, 1, 0)
            L.call("coevo_ga_select_adapt", sel, 3, pop, 1, hof, L._p(gathered), n_local, L.C.byref(ad))
        else:
            L.call("coevo_ga_select_gathered", sel, 3, pop, hof, L._p(gathered), n_local)
            s32p.copy_(s32)
            L.call("coevo_ga_adapt_sigma", L._p(ev), 3 * n_local, L._p(gd), L._p(hist), L._p(sigh), cap, L._p(s64), L._p(s32),
                   0.02, 0.1, 1)
        torch.cuda.synchronize()
        res.append([t.clone() for k in keep for t in k] + [hist, sigh, s64, s32, s32p])
    assert all(torch.equal(x, y) for x, y in zip(*res))
    D = 8
    stride = L.fc_slab_stride(D)
    order = torch.from_numpy(rng.permutation(pop).astype(np.int32)).to(DEV)
    base = [torch.from_numpy(rng.normal(size=k * stride).astype(np.float32)).to(DEV) for k in (pop, hof, E)]
    a_, b_ = [x.clone() for x in base], [x.clone() for x in base]
    cnt2 = torch.tensor([7], dtype=torch.int32, device=DEV)
    L.call("coevo_ga_promote", (L.GaPromoteRole * 1)(L.GaPromoteRole(L._p(a_[0]), L._p(a_[1]), L._p(a_[2]), L._p(order), D, 1, 1, 0)),
           1, E, hof)
    L.call("coevo_ga_promote_tick", (L.GaPromoteRole * 1)(L.GaPromoteRole(L._p(b_[0]), L._p(b_[1]), L._p(b_[2]), L._p(order), D, 1, 1, 0)),
           1, E, hof, L._p(cnt2))
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a_, b_)) and cnt2.item() == 8
    st_a, st_b = torch.zeros(L.MPE_STATE_DOUBLES, n, dtype=torch.float64, device=DEV), torch.zeros(L.MPE_STATE_DOUBLES, n, dtype=torch.float64, device=DEV)
    segs = (L.ResetSeg * 2)(L.ResetSeg(0, n - 10, 1000), L.ResetSeg(n - 10, 10, 5))
    rngst = L.PCG64State.from_seed(1870300)
    stamps = torch.full((3 * L.STAMP_SLOTS, 2), 5, dtype=torch.int64, device=DEV)
    L.call("coevo_mpe_reset_multi", L._p(st_a), n, ct.cast(segs, ct.c_void_p), 2, rngst)
    L.call("coevo_mpe_reset_multi_arm", L._p(st_b), n, ct.cast(segs, ct.c_void_p), 2, rngst, L._p(stamps), 2 * L.STAMP_SLOTS)
    torch.cuda.synchronize()
    assert torch.equal(st_a, st_b)
    armed = stamps[:2 * L.STAMP_SLOTS]
    assert bool((armed[:, 0] == -1).all()) and bool((armed[:, 1] == 0).all()) and bool((stamps[2 * L.STAMP_SLOTS:] == 5).all())


def test_multi_job_launches_equal_per_role_launches():
    """coevo_fc_perturb_dist_multi / coevo_fc_distance_finalize_multi / coevo_mpe_reset_multi (one launch for the three
    roles / several game ranges) write the same bits as one coevo_fc_perturb_dist / coevo_fc_distance_finalize /
    coevo_mpe_reset per role or range"""
    roles_D, E, n, child_first = (8, 10, 10), 2, 7, 1
    jobs_p, jobs_f, keep = (L.PerturbJob * 3)(), (L.FinalizeJob * 3)(), []
    single, multi = [], []
    sigma = torch.tensor([0.05, 0.02, 0.11], dtype=torch.float32, device=DEV)
    pidx = (torch.arange(n, dtype=torch.int32, device=DEV) % E).contiguous()
    for ri, D in enumerate(roles_D):
        stride, nb = L.fc_slab_stride(D), int(L.load().coevo_fc_perturb_blocks(D))
        elite = to_slab(make_nets(E, D, seed=40 + ri), D).contiguous()
        stale = to_slab(make_nets(1, D, seed=50 + ri), D).contiguous()
        head = torch.tensor([0.25 + ri], dtype=torch.float32, device=DEV)
        out = []
        for _ in range(2):
            pop = torch.zeros(child_first + n, stride, dtype=torch.float32, device=DEV)
            part = torch.zeros(n * nb, dtype=torch.float64, device=DEV)
            dist = torch.zeros(child_first + n, dtype=torch.float32, device=DEV)
            out.append((pop, part, dist))
        (pop1, part1, dist1), (pop2, part2, dist2) = out
        L.call("coevo_fc_perturb_dist", L._p(elite), L._p(pidx), L._p(pop1), child_first, n, D, sigma.data_ptr() + 4 * ri,
               1234, 3, 8 + ri, 0, None, L._p(stale), L._p(part1))
        L.call("coevo_fc_distance_finalize", L._p(part1), nb, n, L._p(dist1), child_first, L._p(head))
        jobs_p[ri] = L.PerturbJob(L._p(elite), L._p(pidx), L._p(pop2), sigma.data_ptr() + 4 * ri, L._p(stale), L._p(part2),
                                  child_first, n, D, 3, 8 + ri, 0)
        jobs_f[ri] = L.FinalizeJob(L._p(part2), L._p(dist2), L._p(head), nb, n, child_first, 0)
        keep += [elite, stale, head]
        single.append((pop1, dist1)); multi.append((pop2, dist2))
    L.call("coevo_fc_perturb_dist_multi", C.cast(jobs_p, C.c_void_p), 3, 1234, 0, None)
    L.call("coevo_fc_distance_finalize_multi", C.cast(jobs_f, C.c_void_p), 3)
    torch.cuda.synchronize()
    for (p1, d1), (p2, d2) in zip(single, multi):
        assert torch.equal(p1, p2) and torch.equal(d1, d2) and float(d1[1:].min()) > 0.0
    # resets: three ranges with their own first ordinals
    n_games, rng = 50, L.PCG64State.from_seed(1870300)
    segs = [(0, 13, 5), (13, 20, 1000), (40, 10, 77)]
    a = torch.zeros(L.MPE_STATE_DOUBLES, n_games, dtype=torch.float64, device=DEV)
    b = torch.zeros_like(a)
    for g0, cnt, first in segs:
        L.call("coevo_mpe_reset", L._p(a), n_games, g0, cnt, rng, first)
    arr = (L.ResetSeg * len(segs))(*[L.ResetSeg(*s) for s in segs])
    L.call("coevo_mpe_reset_multi", L._p(b), n_games, C.cast(arr, C.c_void_p), len(segs), rng)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float(a[:, :33].abs().sum()) > 0.0


def test_philox_kernel_known_answers_and_normals():
    """the device generator against Random123's published vectors (7 and 10 rounds), its Gaussians against the oracle's bit
    for bit over three streams, and the moments of 4 M device normals (agent.py:27-28 / :52 in the device_philox mode)"""
    from tests.test_oracle_golden import PHILOX_KAT
    import ctypes as C
    lib = L.load()
    assert lib.coevo_noise_rounds() == 7
    for rounds in (7, 10):
        kat = [k for k in PHILOX_KAT if k[0] == rounds]
        ck = torch.from_numpy(np.array([k[1] for k in kat], dtype=np.uint32).view(np.int32)).cuda()
        out = torch.zeros(len(kat), 4, dtype=torch.int32, device="cuda")
        L.call("coevo_philox4x32", rounds, L._p(ck), len(kat), L._p(out))
        got = out.cpu().numpy().view(np.uint32)
        assert got.tolist() == [k[2] for k in kat], rounds
    olib = rp.lib()
    for seed, lo, hi, q0 in ((0, 0, 0, 0), (7, 199, 41, 1000), ((1 << 40) + 3, 2 ** 31, 3, 2 ** 32 - 600)):
        z = torch.zeros(512, 4, dtype=torch.float32, device="cuda")
        L.call("coevo_philox_normals", seed, lo, hi, q0, 512, L._p(z))
        got = z.cpu().numpy()
        want = np.empty((512, 4), np.float32)
        buf = (C.c_float * 4)()
        for i in range(512):
            olib.oracle_philox_normal4(seed, lo, hi, (q0 + i) & 0xffffffff, buf)
            want[i] = buf[:]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, lo, hi)
    n = 1 << 20
    z = torch.zeros(n, 4, dtype=torch.float32, device="cuda")
    L.call("coevo_philox_normals", 3, 17, 5, 0, n, L._p(z))
    x = z.double().reshape(-1)
    se = 5.0 / np.sqrt(x.numel())
    assert abs(float(x.mean())) < se and abs(float(x.var()) - 1.0) < se * np.sqrt(2)
    assert abs(float((x ** 4).mean()) - 3.0) < se * np.sqrt(96)
    assert abs(float((x[:-1] * x[1:]).mean())) < se
