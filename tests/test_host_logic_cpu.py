"""Host-side mirror of the reference interface (no GPU): parameter layout, RNG-order parity of construction and
mutation with the reference's fixtures, plan building, sigma adaptation, loud failure without the HIP library."""
import os
import numpy as np
import pytest
import torch

from coevonet_amd import lib as L
from coevonet_amd.agent import MPEAgent
from coevonet_amd.fcnetwork import FCNetwork
from coevonet_amd.game_logic import create_agent, diversity_penalty, initialize_env, play_game
from coevonet_amd.genetic_algorithm import adapt_mutation_power, initial_population
from coevonet_amd.rollout import RolloutPlan, effective_steps
from tests.util import Bag, load_golden, sha


def test_fcnetwork_matches_reference_weights_and_mutation():
    """same torch seed -> byte-identical parameters as the reference's FCNetwork (fixture sha), also after
    Agent.mutate (every parameter, LayerNorm affine included)"""
    for case in load_golden("fc_forward.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        args = Bag()
        net = FCNetwork(case["D"], 5, "float32")
        if case["mutated"]:
            for p in net.parameters():
                p.data += torch.normal(0, 0.05, size=p.size())
        assert sha(net.flat()) == case["weights"]["sha256"]
        assert net.flat().size == L.fc_param_count(case["D"])


def test_weight_get_set_surface():
    torch.manual_seed(0)
    args = Bag()
    net = FCNetwork(10, 5, "float32")
    es = net.get_weights_ES()
    per = net.get_perturbable_weights()
    assert es.shape == per.shape == (138245,) and np.array_equal(es, per)  # LayerNorm excluded
    assert [n for n, _ in net.named_modules()] == ["", "fc1", "ln1", "fc2", "ln2", "output"]
    net.set_perturbable_weights(per + 1.0, args)
    assert np.allclose(net.get_perturbable_weights(), per + 1.0)
    assert np.all(net.state_dict()["ln1.weight"].numpy() == 1.0)  # untouched
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    other = FCNetwork(10, 5, "float32")
    other.load_state_dict(sd)
    assert sha(other.flat()) == sha(net.flat())
    with pytest.raises(ValueError):
        net.set_weights({"fc1.weight": torch.zeros(3, 3)}, layers=["fc1"])
    with pytest.raises(ValueError):
        FCNetwork(10, 5, "float16")


def test_agent_clone_and_create_order_consume_rng_like_reference():
    fx = load_golden("play_game.json")["cases"][0]
    torch.manual_seed(fx["torch_seed"])
    args = Bag()
    env = initialize_env(args)
    agents = [create_agent(env, args, r) for r in ("agent_0", "agent_1", "adversary_0")]
    assert [sha(a.model.flat()) for a in agents] == [w["sha256"] for w in fx["weights"]]
    c = agents[0].clone(env, args, "agent_0")
    assert isinstance(c, MPEAgent) and sha(c.model.flat()) == sha(agents[0].model.flat())
    with pytest.raises(ValueError):
        create_agent(env, Bag(game="pong_v3"), None)
    with pytest.raises(ValueError):
        play_game(env, agents[0].model, agents[1].model, None, args)


def test_initial_population_matches_ga_fixture_weights():
    """creation order of genetic_algorithm.py:63-68,110-117: generation-0 HoF members are what the fixture's saved
    HoF lists contain before the first push"""
    fx = load_golden("ga_hof2.json")
    cfg = fx["config"]
    torch.manual_seed(cfg["seed"])
    args = Bag(algorithm="GA", **cfg["args"])
    env = initialize_env(args)
    pop_flat, hof_flat = initial_population(env, args)
    saves = {s["file"]: s["agents"] for s in fx["generations"][0]["saves"]}
    # after generation 0 the HoF is [initial hof[1], best]: the survivor must be the initial member 1
    assert sha(hof_flat["agent_0"][1]) == saves["hall_of_fame_agent_0.pth"][0]["sha256"]
    assert sha(hof_flat["adversary_0"][1]) == saves["hall_of_fame_adversary.pth"][0]["sha256"]
    best = fx["generations"][0]["elite_ids"][0][0]
    assert sha(pop_flat["agent_0"][best]) == saves["hall_of_fame_agent_0.pth"][1]["sha256"]


def test_rollout_plan_groups_rows_by_weight_set():
    pop, hof = 7, 3
    games, off, D = [], {}, {}
    for i in range(pop):
        off[i], D[i] = 1000 * i, 10
    for k in range(hof):
        off[100 + k], D[100 + k] = 10 ** 6 + k, 10
        off[200 + k], D[200 + k] = 2 * 10 ** 6 + k, 8
    for i in range(pop):
        for k in range(hof):
            games.append((200 + k, i, 100 + k))
    plan = RolloutPlan(np.array(games), off, D, device=None)
    assert plan.n_rows == 3 * len(games)
    assert len(plan.light_np) == pop + 2 * hof and plan.light_max == pop and len(plan.heavy_np) == 0
    rows_seen = set()
    for t in plan.light_np:
        rows = range(int(t["row_begin"]), int(t["row_begin"]) + int(t["n_rows"]))
        for r in rows:
            g, slot = plan.row_game_np[r], plan.row_slot_np[r]
            assert off[games[g][slot]] == int(t["net_off"]) and plan.game_rows_np[g, slot] == r
            rows_seen.add(r)
    assert rows_seen == set(range(plan.n_rows))
    big = RolloutPlan(np.array([(200, i, 100) for i in range(70)]), {**{i: i for i in range(70)}, 100: 5000, 200: 9000},
                      {**{i: 10 for i in range(70)}, 100: 10, 200: 8}, device=None)
    assert [int(t["n_rows"]) for t in big.heavy_np] == [32, 32, 6, 32, 32, 6] and big.heavy_max == 32
    with pytest.raises(AssertionError):
        RolloutPlan(np.array([(0, 0, 0)]), {0: 0}, {0: 10}, device=None)  # a 10-wide net in the adversary seat


def test_effective_steps_and_sigma_adaptation():
    assert effective_steps(None, 25) == 75 and effective_steps(200, 25) == 75 and effective_steps(50, 25) == 50
    assert effective_steps(200, 70) == 200
    args = Bag(mutation_power_agent_0=0.05, mutation_power_agent_1=0.1, mutation_power_adversary=0.05,
               max_mutation_power=0.2, min_mutation_power=0.001)
    hist = {"agent_0": list(range(20, 0, -1)), "agent_1": list(range(20)), "adversary_0": [0.0] * 20}
    adapt_mutation_power(args, 19, hist)
    assert args.mutation_power_agent_0 == pytest.approx(0.1 * 1.2)  # quirk Q5: agent_1's sigma is the base
    assert args.mutation_power_agent_1 == pytest.approx(0.1 * 0.95)
    assert args.mutation_power_adversary == pytest.approx(0.05 * 0.95)
    adapt_mutation_power(args, 5, hist)  # gen <= 10: always decay
    assert args.mutation_power_agent_0 == pytest.approx(0.12 * 0.95)


def test_diversity_penalty_host_version():
    g = np.random.Generator(np.random.PCG64(0))
    w = [g.normal(size=50).astype(np.float32) for _ in range(6)]
    d = diversity_penalty(w[-1], w, Bag())
    dist = np.array([np.linalg.norm(x - w[-1]) for x in w])
    assert d == pytest.approx(np.sum(np.maximum(0, 1 - dist / dist.mean())))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """no CPU fallback: without libcoevo.so every product entry point raises"""
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "libcoevo.so"))
    with pytest.raises(L.CoevoError):
        L.load()
    with pytest.raises(L.CoevoError):
        L.fc_param_count(10)


def test_rollout_plan_cohorts_partition_games():
    """cohorts never split a per-individual net's games, cover every game once, and cut the shared-opponent rows per
    cohort; task lists are laid out cohort by cohort (coevo_rollout_desc.heavy_begin / light_begin)"""
    from coevonet_amd.rollout import RolloutPlan
    npop, nh = 30, 3
    off = {i: i * 10 for i in range(npop + 2 * nh)}
    D = {**{i: 10 for i in range(npop + nh)}, **{npop + nh + k: 8 for k in range(nh)}}
    games = [(npop + nh + k, i, npop + k) for i in range(npop) for k in range(nh)]
    for K in (1, 2, 3, 4):
        plan = RolloutPlan(np.array(games), off, D, device=None, n_cohorts=K)
        assert plan.n_cohorts == K
        co = plan.game_cohort_np
        for i in range(npop):                                   # an individual's games share a cohort
            assert len({int(co[g]) for g in range(i * nh, (i + 1) * nh)}) == 1
        assert sorted(np.bincount(co).tolist())[0] >= (npop // K) * nh
        assert plan.heavy_begin_np[0] == 0 and plan.heavy_begin_np[-1] == len(plan.heavy_np)
        assert plan.light_begin_np[0] == 0 and plan.light_begin_np[-1] == len(plan.light_np) == npop
        for k in range(K):
            for arr, begin in ((plan.heavy_np, plan.heavy_begin_np), (plan.light_np, plan.light_begin_np)):
                for t in arr[begin[k]:begin[k + 1]]:
                    rows = range(int(t["row_begin"]), int(t["row_begin"]) + int(t["n_rows"]))
                    assert all(co[plan.row_game_np[r]] == k for r in rows)
        assert sorted(plan.row_game_np.tolist()) == sorted(list(range(len(games))) * 3)
    # one net shared by every game as a per-individual net would tie all games together: a single cohort remains
    tied = RolloutPlan(np.array([(40, i, 41) for i in range(6)]), {**{i: i for i in range(6)}, 40: 100, 41: 200},
                       {**{i: 10 for i in range(6)}, 40: 8, 41: 10}, device=None, n_cohorts=3)
    assert tied.n_cohorts == 1


@pytest.mark.parametrize("n_local,K", [(9, 2), (201, 2), (200, 3), (11, 3), (10, 2), (7, 7), (5, 1)])
def test_cohort_partition_is_one_consistent_partition(n_local, K):
    """the rollout plan's game -> cohort map and the breeding / reset ranges come from the same boundaries (odd
    per-rank populations used to disagree: a data race between the cohort streams)"""
    from coevonet_amd.genetic_algorithm import cohort_partition
    bounds, of = cohort_partition(n_local, K)
    assert bounds[0] == 0 and bounds[-1] == n_local and np.all(np.diff(bounds) >= 1)
    for k in range(K):
        assert np.array_equal(np.nonzero(of == k)[0], np.arange(bounds[k], bounds[k + 1]))


def test_small_shard_rows_follow_what_is_resident_at_once(monkeypatch):
    """rows per shared-opponent task of a rank (genetic_algorithm.py:125-217 split by index): hof rows while the rank's
    workgroups - 3 n_local + 6 hof ceil(n_local / hof) - are all resident, two per CU, for the persistent whole-rollout launch;
    with COEVO_PERSISTENT=0 only while each has a CU to itself; the whole population, or hof > 8: the lean kernel's 16-row tiles"""
    from coevonet_amd.genetic_algorithm import small_shard_rows
    monkeypatch.delenv("COEVO_PERSISTENT", raising=False)
    assert small_shard_rows(25, 5, 256, 200) == 5      # a rank of 8: 75 + 150 workgroups
    assert small_shard_rows(50, 5, 256, 200) == 5      # a rank of 4: 150 + 300 <= 512
    assert small_shard_rows(100, 5, 256, 200) == 16    # a rank of 2: 300 + 600
    assert small_shard_rows(200, 5, 256, 200) == 16    # everything on this GPU
    assert small_shard_rows(25, 10, 256, 200) == 16    # tasks of more than 8 rows are not the small body's
    assert small_shard_rows(26, 5, 256, 200) == 5      # ragged last chunk: 78 + 180
    monkeypatch.setenv("COEVO_PERSISTENT", "0")
    assert small_shard_rows(25, 5, 256, 200) == 5 and small_shard_rows(50, 5, 256, 200) == 16


def test_unequal_shards_are_rejected():
    from coevonet_amd.genetic_algorithm import GAEngine
    with pytest.raises(ValueError, match="not divisible"):
        GAEngine(pop=7, hof=1, elites=1, shard=(0, 2), device="cpu")


def test_weights_only_safe_checkpoint_roundtrip(tmp_path):
    """the state_dict twin of a --save file loads with torch.load(weights_only=True) - tensors only"""
    import os
    from coevonet_amd.io_utils import agents_from_state_dicts, load_state_dicts, save_state_dicts
    args = Bag()
    env = initialize_env(args)
    agents = [create_agent(env, args, "adversary_0") for _ in range(3)]
    path = save_state_dicts(agents, os.path.join(tmp_path, "hall_of_fame_adversary.pth"), role="adversary_0")
    assert path.endswith("hall_of_fame_adversary.state_dict.pth")
    sds, role, single = load_state_dicts(path)
    assert role == "adversary_0" and not single and len(sds) == 3 and sds[0]["fc1.weight"].shape == (512, 8)
    state = torch.random.get_rng_state()
    back = agents_from_state_dicts(env, args, None, path)
    assert torch.equal(state, torch.random.get_rng_state())          # rebuilding agents leaves the RNG stream alone
    assert [sha(a.model.flat()) for a in back] == [sha(a.model.flat()) for a in agents]
    one = agents_from_state_dicts(env, args, "adversary_0",
                                  save_state_dicts(agents[1], os.path.join(tmp_path, "adversary.pth"), role="adversary_0"))
    assert sha(one.model.flat()) == sha(agents[1].model.flat())
    # the file names the offspring-noise contract it was written under; another one warns, or is refused on request
    payload = torch.load(path, weights_only=True)
    assert payload["noise"] == "philox4x32-7"
    payload["noise"] = "philox4x32-10"
    other = os.path.join(tmp_path, "other.state_dict.pth")
    torch.save(payload, other)
    with pytest.warns(RuntimeWarning, match="philox4x32-10"):
        assert len(load_state_dicts(other)[0]) == 3
    with pytest.raises(ValueError, match="offspring noise"):
        load_state_dicts(other, strict_noise=True)


def test_u8_over_255_two_term_product_is_the_ieee_quotient():
    """csrc/deepqn.hip u8_over_255: fma(x, head, x * tail) with 1/255 = head + tail equals x / 255.0f (the oracle's
    preprocess_observation divide) for every byte value; the fma is emulated exactly in float64 (24 x 24-bit product +
    a float32 addend fit its 53 bits before the single rounding to float32)"""
    x = np.arange(256, dtype=np.float32)
    head = np.float32(1.0 / 255.0)
    tail = np.float32(1.0 / 255.0 - float(head))
    low = (x * tail).astype(np.float32)
    got = (x.astype(np.float64) * np.float64(head) + low.astype(np.float64)).astype(np.float32)
    want = x / np.float32(255.0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_dqn_conv_slab_layout_is_a_permutation():
    """csrc/dqn_common.hip.h dqn_conv_slab_to_flat (restated): the lane-ordered conv weight layout visits every
    (cout, tap) of a layer exactly once"""
    for cout, taps in ((32, 256), (32, 384), (64, 512), (64, 576)):
        i = np.arange(cout * taps, dtype=np.int64)
        np_count = cout // 32
        e, lane, blk = i & 3, (i >> 2) & 63, i >> 8
        npi, qp = blk % np_count, blk // np_count
        co = 32 * npi + 16 * (e & 1) + (lane & 15)
        tap = 4 * (2 * qp + (e >> 1)) + (lane >> 4)
        flat = co * taps + tap
        assert co.max() == cout - 1 and tap.max() == taps - 1
        assert np.array_equal(np.sort(flat), i)


def test_job_struct_layouts_match_the_header():
    """the ctypes mirrors of coevo_fc_perturb_job / coevo_fc_finalize_job / coevo_reset_seg have the sizes the C side
    static_asserts (csrc/offspring.hip, select.hip, mpe_env.hip)"""
    import ctypes as C
    from coevonet_amd import lib as L
    assert (C.sizeof(L.PerturbJob), C.sizeof(L.FinalizeJob), C.sizeof(L.ResetSeg)) == (72, 40, 16)
    assert L.PerturbJob.child_first.offset == 48 and L.FinalizeJob.n_blocks.offset == 24
    assert L.ResetSeg.first_ordinal.offset == 8
    assert C.sizeof(L.GaAdaptArgs) == 88 and L.GaAdaptArgs.sig_min.offset == 56 and L.GaAdaptArgs.eval_first_game.offset == 72
    assert C.sizeof(L.RolloutDesc) == 192 and L.RolloutDesc.stamps_armed.offset == 168 and L.RolloutDesc.sync_words.offset == 176
    assert L.RolloutDesc.pack.offset == 184 and C.sizeof(L.FinalPack) == 40


def test_bench_reads_tracked_pmc_traffic():
    """bench.py's roofline.traffic comes from the tracked PMC summary (profiles/r03_pmc_hbm_traffic.json): the lookup keys it
    uses must exist there, with HBM bytes per launch close to the algorithmic bytes of the kernels it names"""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    head, src = bench.pmc_traffic("headline", "fc_cycle16_kernel<5>")
    assert head and 1.0 <= head / 171709450.0 < 1.25 and "r03_pmc_hbm_traffic.json" in src
    for wl, kern, lo, hi in (("cfg3_es", "fc_cycle_kernel", 8.0e8, 9.0e8), ("cfg4_dqn_ga", "dqn_fc1_kernel", 4.0e8, 5.0e8),
                             ("cfg4_dqn_ga", "dqn_conv_kernel", 5.5e7, 7.0e7), ("cfg5_dqn_es", "dqn_fc1_kernel", 8.0e8, 8.5e8),
                             ("cfg4_dqn_ga_c6", "dqn_conv_kernel", 7.0e7, 9.0e7)):
        b, _ = bench.pmc_traffic(wl, kern)
        assert b and lo <= b <= hi, (wl, kern, b)
    assert bench.pmc_traffic("no_such_workload", "x") == (None, None)


def test_deepqn_weight_accessors_match_reference_fixture():
    """the DeepQN mirror's get_weights / set_weights / get_weights_ES / set_weights_ES / get_perturbable_* against values
    minted from the reference's class (tests/golden/make_golden.py dqnw; Atari/deepqn.py:63-231).  No GPU call: only the
    parameter-count query of the C library."""
    import hashlib
    from coevonet_amd import deepqn as dq
    from tests.util import Bag, load_golden

    def h(a):
        return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()
    for case in load_golden("deepqn_weights.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        net = dq.DeepQN(case["C"], case["n_actions"], "float32")
        for p in net.parameters():
            p.data += torch.normal(0, case["mutate_std"], size=p.size())
        a = Bag(precision="float32")
        assert list(net.state_dict().keys()) == case["state_dict_keys"]
        assert [nm for nm, m in net.named_modules() if m in net.get_perturbable_layers()] == case["perturbable_layers"]
        assert net.get_weights_ES().size == case["all_len"] and h(net.get_weights_ES()) == case["all_sha256"]
        assert np.array_equal(net.get_weights_ES(), net.flat())            # = the canonical flat order of the C ABI
        assert net.get_perturbable_weights().size == case["perturbable_len"]
        assert h(net.get_perturbable_weights()) == case["perturbable_sha256"]
        assert h(net.get_weights_ES([net.fc1, net.vbn2])) == case["fc1_vbn2_sha256"]
        assert list(net.get_weights(["fc1", "vbn1"]).keys()) == case["get_weights_fc1_vbn1_keys"]
        v = (net.get_perturbable_weights() * np.float32(0.5) + np.float32(0.01)).astype(np.float32)
        net.set_perturbable_weights(v, a)
        assert h(net.get_weights_ES()) == case["after_set_perturbable_sha256"]
        k = int(net.fc1.weight.numel() + net.fc1.bias.numel() + 2 * 64)
        net.set_weights_ES((np.arange(k, dtype=np.float32) % np.float32(97.0)) * np.float32(1e-3), a, [net.fc1, net.vbn2])
        assert h(net.get_weights_ES()) == case["after_set_fc1_vbn2_sha256"]
        net.set_weights({key: val * 2 for key, val in net.get_weights(["output"]).items()}, layers=["output"])
        assert h(net.get_weights_ES()) == case["after_set_weights_output_sha256"]
        with pytest.raises(ValueError):
            net.set_weights({"fc1.weight": torch.zeros(3)}, layers=["fc1"])
        with pytest.raises(ValueError):
            net.set_weights({}, layers=None)


def test_load_agent_for_testing_contract(tmp_path, monkeypatch):
    """main.py --test's loader (utils/utils_pth_and_plots.py:8-74): error order and what is returned"""
    from coevonet_amd.io_utils import create_output_dir, load_agent_for_testing, save_model
    args = Bag(algorithm="GA", GA_hof_to_test_agent_0=None, GA_hof_to_test_agent_1=None, GA_hof_to_test_adversary=None)
    with pytest.raises(ValueError, match="agent_0 not specified"):
        load_agent_for_testing(args)
    files = {}
    for k in ("a0", "a1", "adv"):
        files[k] = str(tmp_path / f"{k}.pth")
    args.GA_hof_to_test_agent_0, args.GA_hof_to_test_agent_1, args.GA_hof_to_test_adversary = files["a0"], files["a1"], files["adv"]
    with pytest.raises(ValueError, match="a0.pth not found"):
        load_agent_for_testing(args)
    for k in files:
        save_model([f"{k}-old", f"{k}-new"], files[k])          # a Hall of Fame: the newest member is tested
    assert load_agent_for_testing(args) == ("a0-new", "a1-new", "adv-new")
    es = Bag(algorithm="ES", ES_model_to_test_agent_0=files["a0"], ES_model_to_test_agent_1=files["a1"],
             ES_model_to_test_adversary_0=None)
    with pytest.raises(ValueError, match="adversary_0 not specified"):
        load_agent_for_testing(es)
    es.ES_model_to_test_adversary_0 = files["adv"]
    assert load_agent_for_testing(es)[2] == ["adv-old", "adv-new"]
    monkeypatch.chdir(tmp_path)
    d = create_output_dir(Bag(algorithm="ES", generations=5, population=20, hof_size=3, max_timesteps_per_episode=400,
                              fitness_sharing=True, adaptive=True, max_mutation_power=0.5, min_mutation_power=0.001,
                              learning_rate=0.1))
    assert d == ("ES_models/gens5_pop20_hof3_gamesimple_adversary_v3_tslimit400_fitness-sharingTrue_adaptiveTrue"
                 "max_mutation0.5_min_mutation0.001_lr0.1") and os.path.isdir(d)


def test_reference_helper_functions_host_side():
    """mutate_elites (genetic_algorithm.py:32-48) and compute_weight_update (evolutionary_strategy.py:120-148) under the
    reference's names: RNG order of clone + mutate against the oracle port's torch restatement, the update against numpy"""
    from coevonet_amd import evolutionary_strategy as es
    from coevonet_amd import genetic_algorithm as ga
    from oracle import ref_port as rp
    args = Bag(population=5, elites_number=2, mutation_power_agent_0=0.05, mutation_power_agent_1=0.07,
               mutation_power_adversary=0.09, learning_rate=0.1, fitness_sharing=False)
    env = initialize_env(args)
    torch.manual_seed(7)
    elites = [create_agent(env, args, role="agent_1") for _ in range(2)]
    state = torch.random.get_rng_state()
    kids = ga.mutate_elites(env, elites, args, "agent_1")
    assert len(kids) == 4
    torch.random.set_rng_state(state)
    for i, k in enumerate(kids):
        rp.init_net(10)                                     # clone() builds a brand-new net first (MPE/mpe_agent.py:24-28)
        want = rp.mutate_torch(elites[i % 2].model.flat(), 10, args.mutation_power_agent_1)
        assert sha(k.model.flat()) == sha(want), i
    adv = ga.mutate_elites(env, [create_agent(env, args, role="adversary_0")], Bag(population=2, elites_number=1,
                           mutation_power_adversary=0.0, mutation_power_agent_0=1.0, mutation_power_agent_1=1.0), "adversary_0")
    assert len(adv) == 1                                    # sigma 0: the adversary's own attribute was taken
    g = np.random.Generator(np.random.PCG64(3))
    noises = g.normal(size=(6, 11)).astype(np.float32)
    rewards = g.normal(size=6)
    upd, div = es.compute_weight_update(list(noises), list(rewards), args, "agent_1")
    assert div is None and upd.dtype == np.float32
    np.testing.assert_array_equal(upd, ((0.1 / (6 * 0.07)) * np.dot(noises.T, rewards.astype(np.float32))).astype(np.float32))
    args.fitness_sharing = True
    popw = [g.normal(size=11).astype(np.float32) for _ in range(6)]
    upd2, div2 = es.compute_weight_update(list(noises), list(rewards), args, "agent_0", individual_weights=popw[0],
                                          population_weights=popw)
    assert div2 == diversity_penalty(popw[0], popw, args)
    np.testing.assert_allclose(upd2, (0.1 / (6 * 0.05)) * np.dot(noises.T, rewards.astype(np.float32) / (1 + div2)), rtol=1e-6)
    with pytest.raises(ValueError):
        es.get_numpy_dtype("float16")


def test_native_host_env_equals_numpy_env():
    """coevo_mpe_host_reset / _observe / _step (the env kernels' bodies compiled for the host cores) against the NumPy
    env the fixtures are stated in: resets at random ordinals, then 25 world cycles of 257 games with random actions, a random row order and ragged
    agent-step limits - observations of every live game and the final play_game triples, bit for bit
    (utils/game_logic_functions.py:138,179-190; quirk Q1 credits)"""
    from coevonet_amd import lib as L
    from coevonet_amd.mpe import simple_adversary as sa
    lib = L.load()
    n = 257
    rng = np.random.default_rng(1)
    ordinals = rng.integers(0, 4000, size=n).astype(np.int64)   # env.reset() number of each game (odd and even: Q6)
    goal, apos, lpos = sa.ResetStream(sa.ENV_SEED, skip_initial=False).take(int(ordinals.max()) + 1)
    goal, apos, lpos = goal[ordinals], apos[ordinals], lpos[ordinals]
    env = sa.VecSimpleAdversary(goal, apos, lpos)
    want0 = np.zeros((L.MPE_STATE_DOUBLES, n))
    want0[0:6] = apos.astype(np.float64).reshape(n, 6).T
    want0[12:16] = lpos.astype(np.float64).reshape(n, 4).T
    want0[16:18] = lpos.astype(np.float64)[np.arange(n), goal.astype(np.int64)].T
    want0[22] = goal
    st = np.full((L.MPE_STATE_DOUBLES, n), 7.0)
    assert lib.coevo_mpe_host_reset(st.ctypes.data, n, L.PCG64State.from_seed(sa.ENV_SEED), ordinals.ctypes.data) == 0
    assert np.array_equal(st.view(np.uint64), want0.view(np.uint64))   # PCG64 jump-ahead == numpy's Generator stream
    game_rows = rng.permutation(3 * n).astype(np.int32).reshape(n, 3).copy()
    row_game, row_slot = np.zeros(3 * n, np.int32), np.zeros(3 * n, np.int32)
    for g in range(n):
        for s in range(3):
            row_game[game_rows[g, s]], row_slot[game_rows[g, s]] = g, s
    limits = rng.integers(0, 80, size=n).astype(np.int32)
    acc, rg_prev = np.zeros((n, 3)), np.zeros(n)
    obs = np.zeros((3 * n, L.OBS_STRIDE), np.float32)
    pos_first = 1 if sa.INTEGRATE_POS_FIRST else 0
    for c in range(25):
        want = np.zeros_like(obs)
        for slot, arr in enumerate(env.observe()):
            want[game_rows[:, slot], :arr.shape[1]] = arr
        assert lib.coevo_mpe_host_observe(st.ctypes.data, n, row_game.ctypes.data, row_slot.ctypes.data, 3 * n,
                                          obs.ctypes.data) == 0
        live = 3 * c + 2 < limits   # a game past its last agent-step no longer moves here (the NumPy env moves on, unread)
        rows = np.concatenate([game_rows[live, k] for k in range(3)])
        assert np.array_equal(obs[rows].view(np.uint32), want[rows].view(np.uint32)), c
        acts_rows = rng.integers(0, 5, size=3 * n).astype(np.int32)
        t0 = 3 * c
        m0, m1, m2 = t0 < limits, t0 + 1 < limits, t0 + 2 < limits
        acc[m0, 0] += rg_prev[m0]
        acc[m1, 1] += rg_prev[m1]
        rg, ra = env.step(acts_rows[game_rows])
        acc[m2, 2] += ra[m2]
        rg_prev[m2] = rg[m2]
        assert lib.coevo_mpe_host_step(st.ctypes.data, n, game_rows.ctypes.data, acts_rows.ctypes.data, 3 * n, c,
                                       limits.ctypes.data, pos_first) == 0
    want = np.stack([acc[:, 1], acc[:, 2], acc[:, 0]], axis=1)
    assert np.array_equal(np.ascontiguousarray(st[[20, 21, 19]].T).view(np.uint64), want.view(np.uint64))
    bad = game_rows.copy()
    bad[0, 0] = 3 * n   # a row index outside the action buffer is refused, not read
    assert lib.coevo_mpe_host_step(st.ctypes.data, n, bad.ctypes.data, acts_rows.ctypes.data, 3 * n, 0, limits.ctypes.data,
                                   pos_first) == -1   # COEVO_ERR_ARG


@pytest.mark.parametrize("threads", [2, 5, 8])
def test_host_cores_pool_equals_one_thread(threads):
    """coevo_host_rollout_step (what coevo_mpe_host_rollout runs per cohort and env-cycle on its T host cores): T threads over
    slices of a cohort's game list == the single-thread coevo_mpe_host_step / _observe, bit for bit - 25 cycles, two cohorts
    of an odd game count in alternation, ragged limits, shuffled rows (utils/game_logic_functions.py:138,179-190).  No GPU."""
    from coevonet_amd import lib as L
    from coevonet_amd.mpe import simple_adversary as sa
    lib = L.load()
    n = 1013
    rng = np.random.default_rng(threads)
    ordinals = rng.integers(0, 9000, size=n).astype(np.int64)
    ref = np.zeros((L.MPE_STATE_DOUBLES, n))
    assert lib.coevo_mpe_host_reset(ref.ctypes.data, n, L.PCG64State.from_seed(sa.ENV_SEED), ordinals.ctypes.data) == 0
    st = ref.copy()
    game_rows = rng.permutation(3 * n).astype(np.int32).reshape(n, 3).copy()
    row_game, row_slot = np.zeros(3 * n, np.int32), np.zeros(3 * n, np.int32)
    for g in range(n):
        for s in range(3):
            row_game[game_rows[g, s]], row_slot[game_rows[g, s]] = g, s
    limits = rng.integers(0, 80, size=n).astype(np.int32)
    cohorts = [np.ascontiguousarray(np.nonzero(np.arange(n) % 3 != 0)[0], dtype=np.int32),
               np.ascontiguousarray(np.nonzero(np.arange(n) % 3 == 0)[0], dtype=np.int32)]
    ctx = lib.coevo_host_rollout_create(threads, 2)
    assert ctx and lib.coevo_host_rollout_threads(ctx) == threads
    obs_ref = np.zeros((3 * n, L.OBS_STRIDE), np.float32)
    obs = np.full((3 * n, L.OBS_STRIDE), 3.0, np.float32)
    pos_first = 1 if sa.INTEGRATE_POS_FIRST else 0
    acts = rng.integers(0, 5, size=3 * n).astype(np.int32)
    try:
        for c in range(26):
            for games in cohorts:   # step cycle c-1 (none at c = 0), then observe: the rollout's order per cohort
                assert lib.coevo_host_rollout_step(ctx, st.ctypes.data, n, game_rows.ctypes.data, acts.ctypes.data, 3 * n,
                                                   c - 1, limits.ctypes.data, pos_first, games.ctypes.data, len(games),
                                                   1 if c < 25 else 0, obs.ctypes.data) == 0
            if c > 0:
                assert lib.coevo_mpe_host_step(ref.ctypes.data, n, game_rows.ctypes.data, acts.ctypes.data, 3 * n, c - 1,
                                               limits.ctypes.data, pos_first) == 0
            assert np.array_equal(st.view(np.uint64), ref.view(np.uint64)), c
            if c < 25:
                assert lib.coevo_mpe_host_observe(ref.ctypes.data, n, row_game.ctypes.data, row_slot.ctypes.data, 3 * n,
                                                  obs_ref.ctypes.data) == 0
                assert np.array_equal(obs.view(np.uint32), obs_ref.view(np.uint32)), c
            acts = rng.integers(0, 5, size=3 * n).astype(np.int32)
        bad = cohorts[0].copy()
        bad[3] = n   # a game id outside the state is refused before any thread runs
        assert lib.coevo_host_rollout_step(ctx, st.ctypes.data, n, game_rows.ctypes.data, acts.ctypes.data, 3 * n, 0,
                                           limits.ctypes.data, pos_first, bad.ctypes.data, len(bad), 0, None) == -1
    finally:
        lib.coevo_host_rollout_destroy(ctx)


def test_host_reset_continuation_equals_jump_ahead():
    """coevo_mpe_host_reset / coevo_mpe_host_reset_games continue the seeded stream from game to game when ordinals are
    consecutive (two resets = 21 raw outputs) and jump ahead otherwise: runs of consecutive ordinals starting at odd and even
    positions, isolated ones, repeats and a shuffled game list all give the state of the per-game jump-ahead, which
    test_native_host_env_equals_numpy_env pins to numpy's Generator (utils/game_logic_functions.py:54,217; quirk Q6)"""
    from coevonet_amd import lib as L
    from coevonet_amd.mpe import simple_adversary as sa
    lib = L.load()
    rng = np.random.default_rng(5)
    ords = np.concatenate([np.arange(7, 60), np.arange(1000, 1041), [5, 5, 6, 99, 98, 3], np.arange(2 ** 33 + 1, 2 ** 33 + 30),
                           rng.integers(0, 10 ** 6, size=50)]).astype(np.int64)
    n = len(ords)
    goal, apos, lpos = None, None, None
    want = np.zeros((L.MPE_STATE_DOUBLES, n))
    one = np.zeros((L.MPE_STATE_DOUBLES, 1))
    seed = L.PCG64State.from_seed(sa.ENV_SEED)
    for g in range(n):   # every game alone: no predecessor, so the jump-ahead path
        o = np.array([ords[g]], dtype=np.int64)
        assert lib.coevo_mpe_host_reset(one.ctypes.data, 1, seed, o.ctypes.data) == 0
        want[:, g] = one[:, 0]
    got = np.full((L.MPE_STATE_DOUBLES, n), 9.0)
    assert lib.coevo_mpe_host_reset(got.ctypes.data, n, seed, ords.ctypes.data) == 0
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    games = rng.permutation(n).astype(np.int32)
    got2 = np.full((L.MPE_STATE_DOUBLES, n), 9.0)
    assert lib.coevo_mpe_host_reset_games(got2.ctypes.data, n, seed, ords.ctypes.data, games.ctypes.data, 0, n // 2) == 0
    assert lib.coevo_mpe_host_reset_games(got2.ctypes.data, n, seed, ords.ctypes.data, games.ctypes.data, n // 2, n) == 0
    assert np.array_equal(got2.view(np.uint64), want.view(np.uint64))
    bad = ords.copy()
    bad[3] = -1
    assert lib.coevo_mpe_host_reset_games(got2.ctypes.data, n, seed, bad.ctypes.data, games.ctypes.data, 0, n) == -1


def test_bench_scaling_arguments():
    """bench.py's reading of BASELINE.json's metric ("pop=200 ... 1/2/4/8 GPU"): `--gpus N` shards ONE population, so a rank
    count that does not divide it is refused before anything touches a GPU; the compact() helper keeps the line short"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "3"], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "not divisible" in (p.stderr + p.stdout)
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "7", "--workload", "es"], capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "not divisible" in (p.stderr + p.stdout)
    sys.path.insert(0, repo)
    import bench
    assert bench.compact({"a": 1.23456789, "b": [2.0000001, {"c": 123456.789}], "d": "x", "e": 7}) == \
        {"a": 1.235, "b": [2.0, {"c": 123500.0}], "d": "x", "e": 7}
    hc = bench.host_cores()
    assert 1 <= hc["usable"] <= hc["affinity"]


def test_bench_settling_steps_are_untimed_and_the_same_on_every_rank(monkeypatch):
    """bench.py's warm-up: W steps, then - still untimed - steps until the device has been under the workload for --settle-ms
    (the number comes from times reduced over the ranks, so every rank takes the same number: collectives stay matched);
    --settle-ms 0 = exactly W; the count is capped"""
    import sys
    import time
    import types
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a, **k: None)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.002)

    class Ctx:
        world = 1

        def max_over_ranks(self, s, dev):
            return s

    a = types.SimpleNamespace(warmup=3, settle_ms=0.0)
    assert bench._warm_up(step, a, Ctx(), None) == 0 and len(calls) == 3 and a.settle_steps == 0
    calls.clear()
    a = types.SimpleNamespace(warmup=3, settle_ms=40.0)
    n = bench._warm_up(step, a, Ctx(), None)
    assert n == a.settle_steps and 5 <= n <= 30 and len(calls) == 3 + n   # ~20 steps of 2 ms make up 40 ms
    calls.clear()
    a = types.SimpleNamespace(warmup=1, settle_ms=1e6)
    assert bench._warm_up(lambda: calls.append(1), a, Ctx(), None) == bench.MAX_SETTLE_STEPS


def test_host_rollout_entry_points_refuse_bad_descriptors():
    """argument errors of the host-side rollouts are COEVO_ERR_ARG before any GPU call (the error convention of the boundary:
    return codes, no exceptions, nothing dereferenced that was not checked)"""
    import ctypes as C
    from coevonet_amd import lib as L
    lib = L.load()
    assert not lib.coevo_host_rollout_create(0, 1) and not lib.coevo_host_rollout_create(1, 0)
    assert not lib.coevo_host_rollout_create(1, 9) and not lib.coevo_host_rollout_create(1000, 1)
    ctx = lib.coevo_host_rollout_create(2, 2)
    assert ctx and lib.coevo_host_rollout_threads(ctx) == 2 and lib.coevo_host_rollout_threads(None) == -1
    try:
        d = L.HostRolloutDesc()                       # every pointer NULL
        assert lib.coevo_mpe_host_rollout(ctx, C.byref(d), None) == -1
        assert lib.coevo_mpe_host_rollout(None, C.byref(d), None) == -1
        f = L.FramesRolloutDesc()
        assert lib.coevo_dqn_host_frames_rollout(ctx, C.byref(f), None) == -1
        buf = np.zeros(84 * 84 * 4, np.uint8)
        assert lib.coevo_synth_frame_host(None, 4, 1, 0, 0, 0xFF) == -1
        assert lib.coevo_synth_frame_host(buf.ctypes.data, 7, 1, 0, 0, 0xFF) == -1      # C > 6
        assert lib.coevo_synth_frame_host(buf.ctypes.data, 4, 1, 0, 70000, 0xFF) == -1  # t beyond the 16-bit key field
        st = np.zeros((L.MPE_STATE_DOUBLES, 4))
        gr = np.arange(12, dtype=np.int32)
        games = np.arange(4, dtype=np.int32)
        assert lib.coevo_mpe_host_step_games(st.ctypes.data, 4, gr.ctypes.data, None, 0, None, 1, games.ctypes.data, 0, 4, 0,
                                             None) == -1   # a cycle to step, but no actions
        assert lib.coevo_mpe_host_step_games(st.ctypes.data, 4, gr.ctypes.data, None, -1, None, 1, games.ctypes.data, 0, 4, 1,
                                             None) == -1   # observations asked for, no buffer
    finally:
        lib.coevo_host_rollout_destroy(ctx)


# ---- where the host cores of an env-on-the-host rollout run (csrc/host_placement.hip) ---------------------------------------
def _choose(allowed, node, l3, smt, caller, threads, ctx_index=0):
    from coevonet_amd import lib as L
    cpus = np.full(256, -1, dtype=np.int32)
    flags = np.zeros(1, dtype=np.int32)
    n = L.load().coevo_host_placement_choose(allowed.encode(), node.encode(), l3.encode(), smt.encode(), caller, threads,
                                             ctx_index, cpus.ctypes.data, flags.ctypes.data)
    return [int(c) for c in cpus[:max(n, 0)]], int(flags[0]), n


def _epyc_2s():
    """the host of a GPU box as sysfs shows it: 2 sockets x 64 cores x 2 SMT threads, 8 cores (16 CPUs) per L3 complex;
    node 0 = CPUs 0-63 + 128-191, node 1 = 64-127 + 192-255; core c's sibling is c + 128"""
    l3 = ";".join(f"{b}-{b + 7},{b + 128}-{b + 135}" for b in range(0, 128, 8))
    smt = ";".join(f"{c},{c + 128}" for c in range(128))
    return {0: "0-63,128-191", 1: "64-127,192-255"}, l3, smt


def test_host_placement_is_a_pure_function_of_mask_and_sysfs_strings():
    """coevo_host_placement_choose: the cores of a host-cores rollout (the reference steps its env on whatever core runs the
    interpreter, utils/game_logic_functions.py:138,179-190) as a function of (affinity mask, node cpulist, L3 groups, SMT
    sets, the caller's CPU, thread count, context index) - no GPU, no real topology"""
    ON, ONE_L3, UNKNOWN, SHORT = 1, 2, 4, 8
    nodes, l3, smt = _epyc_2s()
    # the caller sits on the far socket (CPU 3, node 0), the GPU hangs off node 1: both threads go to ONE complex of node 1,
    # on two distinct physical cores
    cpus, fl, n = _choose("0-255", nodes[1], l3, smt, caller=3, threads=2)
    assert n == 2 and cpus == [64, 65] and fl == ON | ONE_L3
    # the caller already runs on the GPU's node: it keeps its CPU (slot 0) and its own complex is taken
    cpus, fl, _ = _choose("0-255", nodes[1], l3, smt, caller=77, threads=2)
    assert cpus[0] == 77 and cpus[1] in range(72, 80) and cpus[1] != 77 and fl == ON | ONE_L3
    # ... on an SMT sibling of that complex: still its own CPU first, then primaries of the same complex
    cpus, fl, _ = _choose("0-255", nodes[1], l3, smt, caller=200, threads=3)
    assert fl == ON | ONE_L3 and set(cpus) <= set(range(72, 80)) | set(range(200, 208)) and len(set(cpus)) == 3
    # 16 threads (the DeepQN frame renderers): two neighbouring complexes, physical cores only
    cpus, fl, n = _choose("0-255", nodes[1], l3, smt, caller=3, threads=16)
    assert n == 16 and sorted(cpus) == list(range(64, 80)) and fl == ON
    # 20 threads on a mask of ONE complex: SMT siblings only once the cores run out
    cpus, fl, n = _choose("64-71,192-199", nodes[1], l3, smt, caller=64, threads=12)
    assert n == 12 and sorted(cpus)[:8] == list(range(64, 72)) and set(cpus[8:]) <= set(range(192, 200)) and fl == ON | ONE_L3
    cpus, fl, n = _choose("64-71,192-199", nodes[1], l3, smt, caller=64, threads=20)
    assert n == 16 and fl & SHORT and len(set(cpus)) == 16
    # context index: the next contexts / ranks move on by whole complexes and wrap inside the node
    firsts = [_choose("0-255", nodes[1], l3, smt, caller=3, threads=2, ctx_index=i)[0] for i in range(9)]
    assert [f[0] for f in firsts] == [64, 72, 80, 88, 96, 104, 112, 120, 64]
    # no core of the GPU's node is allowed (taskset to socket 0): say so (flag clear), stay inside the mask, one complex
    cpus, fl, n = _choose("8-15", nodes[1], l3, smt, caller=9, threads=2)
    assert n == 2 and cpus[0] == 9 and set(cpus) <= set(range(8, 16)) and not fl & ON and not fl & UNKNOWN and fl & ONE_L3
    # the node is unknown (a VM that hides it): the caller's own complex
    cpus, fl, n = _choose("0-255", "", l3, smt, caller=21, threads=2)
    assert n == 2 and cpus[0] == 21 and cpus[1] in range(16, 24) and fl == UNKNOWN | ONE_L3
    # no sysfs at all: consecutive CPUs of the mask, starting at the caller's
    cpus, fl, n = _choose("2-5", "", "", "", caller=4, threads=3)
    assert n == 3 and cpus[0] == 4 and set(cpus) <= {2, 3, 4, 5} and fl & UNKNOWN
    # a ragged mask across both nodes, GPU on node 0, caller on node 1
    cpus, fl, n = _choose("5,6,70-90,133", nodes[0], l3, smt, caller=80, threads=2)
    assert n == 2 and cpus == [5, 6] and fl == ON | ONE_L3
    # argument errors
    assert _choose("", nodes[1], l3, smt, 0, 2)[2] == 0           # empty mask: nothing to pin to
    assert _choose("0-255", nodes[1], l3, smt, 0, 0)[2] == -1     # COEVO_ERR_ARG
    assert _choose("garbage", "", "", "", 0, 1)[2] == 0


def test_host_placement_of_a_context_on_this_machine():
    """coevo_host_rollout_placement without a GPU: the node of the device is unknown (-1), so the context takes the caller's
    own node; its CPUs lie inside the affinity mask, the caller's own CPU stays allowed afterwards (the pin is per rollout)"""
    from coevonet_amd import lib as L
    lib = L.load()
    before = os.sched_getaffinity(0)
    ctx = lib.coevo_host_rollout_create(2, 2)
    assert ctx
    try:
        pl = L.host_placement(ctx)
        assert pl["gpu_numa_node"] == -1
        if len(before) >= 2:
            assert pl["pinned"] and len(pl["cpus"]) == 2 and set(pl["cpus"]) <= before and len(set(pl["cpus"])) == 2
        assert os.sched_getaffinity(0) == before
    finally:
        lib.coevo_host_rollout_destroy(ctx)
