"""Distribution of the unique-leaf count L per minibatch index (Connect4, 1024 games, 25 x 8), steady state."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
eng = SelfPlayEngine(g, 1024, evaluators=[HipNet(net, "cuda:0")], max_batch=8, seed=0)
for _ in range(25):
    eng.search(25, 8); eng.step(); eng.drain()
L = _lib.load(); counts = (C.c_int32 * 2)()
rec = np.zeros((30, 25), np.int64)
for mv in range(30):
    for mb in range(25):
        st = eng._stream()
        _lib.check(L.caro_select(eng.h, 8, mb, None, eng.planes.data_ptr(), None, st))
        _lib.check(L.caro_leaf_counts(eng.h, counts, st))
        rec[mv, mb] = counts[0]
        eng.evaluators[0].forward_dev(eng.planes, eng._counts_dev, 0, 8192, eng._probs, eng._values, st)
        _lib.check(L.caro_expand_backup(eng.h, eng._probs.data_ptr(), eng._values.data_ptr(), st))
    eng.step(); eng.drain()
print("mb   mean    max   frac>1536")
for mb in range(25):
    print("%2d  %6.0f  %5d   %.2f" % (mb, rec[:, mb].mean(), rec[:, mb].max(), (rec[:, mb] > 1536).mean()))
print("all  %6.0f  %5d   %.2f" % (rec.mean(), rec.max(), (rec > 1536).mean()))
eng.close()
