"""The launch shapes the cfg 4 / cfg 5 legs of bench.py time, against the oracle at full population size.

The small-population DeepQN tests keep every net at <= 16 rows (one task per net).  At bench size a HoF / base net acts in
50 - 125 games and is cut into several <= 16-row tasks, the row -> task search runs over ~90 / ~270 tasks, the conv grid is
rounded up to a multiple of 8 and remapped over the XCDs, fc1 dispatches every row-group instantiation in one launch and
Co-ES runs two game cohorts on two streams.  Here those launches are compared with ``oracle/ref_port.py`` (short horizon so
that the scalar C forward stays affordable): Atari/deepqn.py:39-48, genetic_algorithm.py:125-151,223-252,
evolutionary_strategy.py:120-148 restricted to two roles (loop parity unpinned, forward pinned)."""
import numpy as np
import pytest
import torch

from coevonet_amd import lib as L
from coevonet_amd.dqn_population import DQNESTrainer, DQNGATrainer
from coevonet_amd.game_logic import initialize_env
from oracle import ref_port as rp
from tests.util import Bag, sha

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROLES = rp.DQN_ROLES


@pytest.mark.parametrize("C,n", [(4, 6), (6, 18), (3, 6), (5, 18)])
def test_forward_one_net_in_several_tasks_ragged_grid(C, n):
    """one net acting in 37 rows = tasks of 16 + 16 + 5 rows interleaved with other nets' tasks, 41 rows in all (the conv
    grid is rounded up to 48: tail workgroups), the row -> task search over interleaved tasks.
    More than 32 rows, so this is the one-workgroup-per-frame conv kernel (smaller launches take the three-launch path):
    its four instantiations - 4 and 6 planes as compile-time constants (what the bench legs time; 6 = what the reference's
    wrapper stack yields, utils/game_logic_functions.py:50-53), 3 and 5 planes at run time"""
    torch.manual_seed(77 + C)
    nets = [rp.dqn_mutate_torch(*rp.dqn_init(C, n), 0.02) for _ in range(3)]
    stride = int(L.load().coevo_dqn_slab_stride(C, n))
    flat = torch.from_numpy(np.stack(nets)).to(DEV)
    slab = torch.zeros(3, stride, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_pack", L._p(flat), L._p(slab), 3, C, n)
    layout = [(0, 16), (1, 1), (0, 16), (2, 3), (0, 5)]          # (net, rows) in ascending row order
    tasks = np.zeros(len(layout), dtype=L.DQN_TASK_DTYPE)
    net_of_row, row = [], 0
    for i, (net, r) in enumerate(layout):
        tasks[i] = (net * stride, row, r)
        net_of_row += [net] * r
        row += r
    assert row == 41 and row % 8
    g = np.random.Generator(np.random.PCG64(5))
    frames = g.integers(0, 256, size=(row, 84, 84, C), dtype=np.uint8)
    d_frames = torch.from_numpy(frames).to(DEV)
    actions = torch.full((row,), -1, dtype=torch.int32, device=DEV)
    logits = torch.zeros(row, L.DQN_LOGIT_STRIDE, dtype=torch.float32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.zeros(int(L.load().coevo_dqn_workspace_bytes(row)) // 4, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_forward_argmax", L._p(slab), L._p(L.tasks_to_device(tasks, DEV)), len(layout), 16, row, C, n,
           L._p(d_frames), L._p(actions), L._p(logits), L._p(status), L._p(ws))
    L.raise_on_status(status)
    got_l, got_a = logits[:, :n].cpu().numpy(), actions.cpu().numpy()
    for r in range(row):
        a, want = rp.dqn_forward(nets[net_of_row[r]], C, n, frames[r])
        assert np.array_equal(got_l[r].view(np.uint32), want.view(np.uint32)), r
        assert got_a[r] == a, r


def test_forward_wide_fc1_every_row_group_and_load_policy():
    """more than 16 tasks, so fc1 is the streaming kernel (one wave per task and 64 outputs; 16 tasks or fewer take the
    32-waves-per-task kernel): tasks of 1 .. 16 rows = one to four row groups, for nets with one task (non-temporal weight
    loads) and for a net cut into 16 + 16 + 7 rows (plain loads: its neighbours stream the same matrix)"""
    C, n = 4, 6
    torch.manual_seed(99)
    single = [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16]
    nets = [rp.dqn_mutate_torch(*rp.dqn_init(C, n), 0.02) for _ in range(len(single) + 2)]
    layout = [(i, r) for i, r in enumerate(single)]
    layout += [(len(single), 16), (len(single), 16), (len(single), 7), (len(single) + 1, 16), (len(single) + 1, 2)]
    assert len(layout) > 16
    lib = L.load()
    stride = int(lib.coevo_dqn_slab_stride(C, n))
    slab = torch.zeros(len(nets), stride, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_pack", L._p(torch.from_numpy(np.stack(nets)).to(DEV)), L._p(slab), len(nets), C, n)
    tasks = np.zeros(len(layout), dtype=L.DQN_TASK_DTYPE)
    net_of_row, row = [], 0
    for i, (net, r) in enumerate(layout):
        tasks[i] = (net * stride, row, r)
        net_of_row += [net] * r
        row += r
    g = np.random.Generator(np.random.PCG64(11))
    frames = g.integers(0, 256, size=(row, 84, 84, C), dtype=np.uint8)
    actions = torch.full((row,), -1, dtype=torch.int32, device=DEV)
    logits = torch.zeros(row, L.DQN_LOGIT_STRIDE, dtype=torch.float32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.zeros(int(lib.coevo_dqn_workspace_bytes(row)) // 4, dtype=torch.float32, device=DEV)
    L.call("coevo_dqn_forward_argmax", L._p(slab), L._p(L.tasks_to_device(tasks, DEV)), len(layout), 16, row, C, n,
           L._p(torch.from_numpy(frames).to(DEV)), L._p(actions), L._p(logits), L._p(status), L._p(ws))
    L.raise_on_status(status)
    got_l, got_a = logits[:, :n].cpu().numpy(), actions.cpu().numpy()
    for r in range(row):
        a, want = rp.dqn_forward(nets[net_of_row[r]], C, n, frames[r])
        assert np.array_equal(got_l[r].view(np.uint32), want.view(np.uint32)), (r, net_of_row[r])
        assert got_a[r] == a, r


def _split(tasks_np):
    """net offset -> row counts of its tasks, in task order"""
    by = {}
    for t in tasks_np:
        by.setdefault(int(t["net_off"]), []).append(int(t["n_rows"]))
    return by


def test_cfg4_full_size_step_vs_oracle():
    """BASELINE configs[3], one GPU's shard as bench.py --workload dqn-ga launches it (pop 50, HoF 10, 84x84x4, 6 actions,
    eager enqueue), horizon 6: (a) the task table of the launch, (b) the 100 deciding games + a sample across every task
    seam of the HoF nets bit for bit, (c) fitness / elite ids from those rewards, (d) generation 1 (bred children, pushed
    HoF, generation 0's evaluation games riding in its launch)."""
    pop, hof, E, C, n, T, Te = 50, 10, 2, 4, 6, 6, 4
    seed = 21
    torch.manual_seed(seed)
    args = Bag(algorithm="GA", game="pong_v3", generations=2, population=pop, hof_size=hof, elites_number=E,
               fitness_sharing=True, max_timesteps_per_episode=T, max_evaluation_steps=Te, coevo_channels=C,
               coevo_graph=False)
    env = initialize_env(args)
    tr = DQNGATrainer(env, args, collect=True)
    eng = tr.eng
    # ---- (a) what one agent-step launches
    assert eng.cohorts == 1 and eng.ro.n_games == 2 * pop * hof + 10 == 1010
    for p in range(2):
        by = _split(eng.ro.tasks_np[p])
        assert len(by) == pop + hof == 60 and sum(map(sum, by.values())) == 1010
        splits = sorted(tuple(v) for v in by.values() if len(v) > 1)
        assert splits == [(16, 16, 16, 2)] * (hof - 1) + [(16, 16, 16, 12)]       # newest HoF member: + 10 evaluation games
        assert sum(len(v) == 1 and v[0] == hof for v in by.values()) == pop
        assert len(eng.ro.tasks_np[p]) == pop + 4 * hof == 90
    tr.step()
    tr.step()
    res = tr.finish()
    # ---- the oracle's population (same torch seed, creation order of dqn_initial_population)
    torch.manual_seed(seed)
    hofs = {"second_0": [rp.dqn_init(C, n)[0] for _ in range(hof)]}
    hofs["first_0"] = [rp.dqn_init(C, n)[0] for _ in range(hof)]
    popu = {r: [] for r in ROLES}
    for _ in range(pop):
        for r in ROLES:
            popu[r].append(rp.dqn_init(C, n)[0])
    per_gen = 2 * pop * hof + 10

    def game(gen, ph, i, k, nets, hofs):
        opp = hofs[ROLES[1 - ph]][hof - 1 - k]
        a, b = (nets[ROLES[ph]][i], opp) if ph == 0 else (opp, nets[ROLES[ph]][i])
        return rp.dqn_play_game(a, b, C, n, env.seed_value, 1 + gen * per_gen + ph * pop * hof + i * hof + k, T)

    seams = [0, 15, 16, 17, 31, 32, 33, 47, 48, 49]          # rows 15/16/17 ... of a HoF net's 50 games, its 2-row task
    got0 = res.game_rewards[0].reshape(2, pop, hof, 2)
    elites, steered = {}, 0
    for ph, role in enumerate(ROLES):
        div = rp.dqn_diversity(popu[role][-1], popu[role])
        fit = []
        for i in range(pop):
            w = game(0, ph, i, hof - 1, popu, hofs)                # Q2: the last HoF game decides (rows >= 1008 among them)
            assert list(got0[ph, i, hof - 1]) == w["rewards"], (ph, i)
            fit.append(w["rewards"][ph] / hof / (1 + div))
            steered += len(set(w["actions"])) > 1
        for i in seams:
            for k in (0, 4):
                assert list(got0[ph, i, k]) == game(0, ph, i, k, popu, hofs)["rewards"], (ph, i, k)
        np.testing.assert_allclose(res.fitness[0][ph], fit, rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(res.diversity[0][ph], div, rtol=1e-5, atol=1e-6)
        ids = [int(x) for x in np.argsort(fit, kind="stable")[::-1][:E]]
        assert res.elite_ids[0][ph] == ids, ph
        elites[role] = [popu[role][i] for i in ids]
    assert steered > 0 and got0.any()                                # the frames steer the actions; some hit was credited
    # ---- (d) generation 1: hof.pop(0); hof.append(best); population = [best] + children
    for r in ROLES:
        hofs[r].append(elites[r][0])
        hofs[r].pop(0)
    sig = {"first_0": np.float32(0.05), "second_0": np.float32(0.05)}
    got1 = res.game_rewards[1].reshape(2, pop, hof, 2)
    for ph, role in enumerate(ROLES):
        child = {}
        for i in (0, 1, 2, 16, 33, 49):
            child[i] = elites[role][0] if i == 0 else rp.perturb_philox_flat(elites[role][(i - 1) % E], sig[role], 0,
                                                                             i - 1, ph)
        for i, net in child.items():
            for k in (0, hof - 1):
                opp = hofs[ROLES[1 - ph]][hof - 1 - k]
                a, b = (net, opp) if ph == 0 else (opp, net)
                w = rp.dqn_play_game(a, b, C, n, env.seed_value, 1 + per_gen + ph * pop * hof + i * hof + k, T)
                assert list(got1[ph, i, k]) == w["rewards"], (ph, i, k)
    ev = [0.0, 0.0]
    for j in range(10):                                            # generation 0's evaluation games (rows 1000 .. 1009)
        w = rp.dqn_play_game(elites["first_0"][0], elites["second_0"][0], C, n, env.seed_value, 1 + 2 * pop * hof + j, Te)
        for s in range(2):
            ev[s] += w["rewards"][s]
    assert [res.rewards[r][0] for r in ROLES] == [e / 10 for e in ev]
    for r in ROLES:
        assert sha(eng.download(r, "hof", hof - 2, 1)[0]) == sha(elites[r][0])   # (hof[-1] is generation 1's best)


def test_cfg5_full_size_generation_vs_oracle():
    """BASELINE configs[4], one GPU's shard as bench.py --workload dqn-es launches it (pop 250, 18 actions, 84x84x4, two
    game cohorts on two streams), horizon 6: the task tables, a sample of games across both cohorts and the base net's task
    seams, the chunked update over the full n = 250 bit for bit (oracle update fed with the device's rewards), the
    evaluation games of the updated base nets."""
    pop, C, n, T, Te = 250, 4, 18, 6, 4
    seed = 22
    torch.manual_seed(seed)
    args = Bag(algorithm="ES", game="boxing_v2", generations=1, population=pop, hof_size=1, learning_rate=0.1,
               fitness_sharing=False, max_timesteps_per_episode=T, max_evaluation_steps=Te, coevo_channels=C)
    env = initialize_env(args)
    tr = DQNESTrainer(env, args, collect=True)
    eng = tr.eng
    assert len(eng.ro.lanes) == 2 and [ln["n"] for ln in eng.ro.lanes] == [250, 250]
    for ln in eng.ro.lanes:
        for p in range(2):
            by = _split(ln["tasks_np"][p])
            assert len(by) == 126 and sorted(map(tuple, by.values()))[-1] == (16,) * 7 + (13,)   # the base net: 125 games
            assert sum(v == [1] for v in by.values()) == 125 and ln["max_rows"][p] == 16
    tr.step()
    res = tr.finish()
    torch.manual_seed(seed)
    base = {r: rp.dqn_init(C, n)[0] for r in ROLES}
    bn = rp.dqn_bn_segments(C, n)
    got = res.game_rewards[0]
    pert = {r: [rp.perturb_philox_flat(base[r], np.float32(0.05), 0, j, ri, bn) for j in range(pop)]
            for ri, r in enumerate(ROLES)}
    sample = sorted(set([0, 1, 7, 8, 15, 16, 17, 111, 112, 123, 124, 125, 126, 140, 141, 248, 249] + list(range(5, pop, 37))))
    for j in sample:
        for ri, r in enumerate(ROLES):
            a, b = (pert[r][j], base["second_0"]) if ri == 0 else (base["first_0"], pert[r][j])
            w = rp.dqn_play_game(a, b, C, n, env.seed_value, 1 + 2 * j + ri, T)
            assert list(got[2 * j + ri]) == w["rewards"], (j, r)
    assert got.any()
    new = {}
    for ri, r in enumerate(ROLES):
        f = np.array([got[2 * j + ri][ri] for j in range(pop)], dtype=np.float32)
        new[r] = rp.dqn_es_update_from_pert(base[r], np.stack(pert[r]), f, 0.05, 0.1, C, n)
        pert[r] = None
        assert sha(eng.download(r, "base", 0, 1)[0]) == sha(new[r]), r
    ev = [0.0, 0.0]
    for j in range(10):
        w = rp.dqn_play_game(new["first_0"], new["second_0"], C, n, env.seed_value, 1 + 2 * pop + j, Te)
        for s in range(2):
            ev[s] += w["rewards"][s]
    assert [res.rewards[r][0] for r in ROLES] == [e / 10 for e in ev]


def test_cfg4_full_size_host_frames_equal_device_frames(monkeypatch):
    """the cfg 4 shard at full size (pop 50, HoF 10: 1010 games per agent-step, 90 tasks per parity) with the env in host memory
    (three alternating cohorts, 28.5 MB of frames up per agent-step) against the device-resident rollout, horizon 6, two
    generations: all 2 x 1000 reward pairs, fitness, elite ids and evaluation means identical"""
    pop, hof, E, C, T, Te = 50, 10, 2, 4, 6, 4

    def run(frames):
        torch.manual_seed(21)
        args = Bag(algorithm="GA", game="pong_v3", generations=2, population=pop, hof_size=hof, elites_number=E,
                   fitness_sharing=True, max_timesteps_per_episode=T, max_evaluation_steps=Te, coevo_channels=C,
                   coevo_graph=False, coevo_frames=frames)
        env = initialize_env(args)
        tr = DQNGATrainer(env, args, collect=True)
        tr.step()
        tr.step()
        res = tr.finish()
        out = ([np.asarray(g).tolist() for g in res.game_rewards], res.fitness, res.elite_ids, [res.rewards[r] for r in ROLES],
               type(tr.eng.ro).__name__, len(tr.eng.ro.lanes))
        tr.close()
        return out

    for var in ("COEVO_FRAME_COHORTS", "COEVO_FRAME_THREADS"):
        monkeypatch.delenv(var, raising=False)
    dev = run("device")
    host = run("host")
    assert dev[4] == "SynthRollout" and host[4] == "HostFrameRollout" and host[5] == 3
    assert host[:4] == dev[:4]
