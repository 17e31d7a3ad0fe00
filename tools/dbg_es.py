import sys; sys.path.insert(0, '.')
import numpy as np, torch
from coevonet_amd import evolutionary_strategy as es
from coevonet_amd.game_logic import initialize_env
from coevonet_amd.genetic_algorithm import ROLES
from oracle import ref_port as rp
from tests.util import Bag, load_golden, sha
fx = load_golden("es_small.json"); cfg = fx["config"]
torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"])
args = Bag(algorithm="ES", **cfg["args"]); env = initialize_env(args)
tr = es.ESTrainer(env, args, rng="host_reference", env_mode="device")
tr.step()
base0 = {r: tr.eng.download(r, "base", 0, 1)[0] for r in ROLES}
tr.step()
ev = tr.eng.rewards_host()[tr.eng.n_main:]
torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"])
want = rp.es_train(Bag(algorithm="ES", **cfg["args"]))
print("base after gen0 equal:", [sha(base0[r]) == sha(want[0]["base"][r]) for r in ROLES])
print("host copy equal:", [sha(tr.base_flat[r]) for r in ROLES][:1])
pop = args.population
for i in range(10):
    print(i, list(ev[i]), want[0]["games"][3*pop+i]["rewards"], want[0]["games"][3*pop+i]["ordinal"])
st = rp.Stream()
g = rp.play_game(st, base0["agent_0"], base0["agent_1"], base0["adversary_0"], 400, 25, ordinal=19)
print("oracle with product nets @19:", g["rewards"])
