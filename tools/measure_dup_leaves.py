"""Cross-game duplicate leaves per launch: in steady state, how many of the leaf boards one net
launch evaluates are the same (board, mover) as another game's leaf of the same launch?  Connect four, 1024 games,
25 x 8, staggered, shipped weights; the leaf keys of a launch are the nonzero rows of the engine's leaf_keys buffer
(cleared before every launch; the mover follows from the board).  python tools/measure_dup_leaves.py [launches]"""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
G, S, B = 1024, 25, 8
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = SelfPlayEngine(g, G, evaluators=[HipNet(net, "cuda:0")], max_batch=B, seed=0, stagger=True, searches_hint=S)
for _ in range(30):  # steady state: every phase of a game is present
    eng.search(S, B); eng.drain()
L = _lib.load()
nets = [e.h for e in eng.evaluators] + [None]
tot = uniq = 0
per = []
for i in range(n_launch):
    eng.leaf_keys.zero_()
    _lib.check(L.caro_search_staggered(eng.h, nets[0], nets[1], 1, B, C.c_void_p(eng.planes.data_ptr()),
                                       C.c_void_p(eng.leaf_keys.data_ptr()), C.c_void_p(eng._probs.data_ptr()),
                                       C.c_void_p(eng._values.data_ptr()), eng._stream()))
    k = eng.leaf_keys[:, 0]
    k = k[k != 0]
    n, u = int(k.numel()), int(torch.unique(k).numel())
    tot += n; uniq += u; per.append((n, u))
    if i % 25 == 24:
        eng.drain()
eng.close()
per = np.array(per)
out = {"launches": n_launch, "leaves": tot, "distinct_within_launch": uniq, "duplicate_fraction": 1 - uniq / tot,
       "per_launch_leaves_mean": float(per[:, 0].mean()), "per_launch_duplicate_fraction_p50_p90_max":
       [float(x) for x in np.percentile(1 - per[:, 1] / per[:, 0], [50, 90, 100])],
       "note": "connect four, 1024 games, 25 x 8, staggered, best_026_12000.dat; duplicates = leaf boards that another game "
               "put into the same launch (within a game the search de-duplicates already)"}
print(json.dumps(out))
