"""Which launches run at the low clock (experiment): after three (tree, net) pairs of the staggered engine, dense net
launches follow on the same stream; the clock of the k-th of them from its workgroups' own stamps."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("tests/golden/weights/best_026_12000.dat", map_location="cpu"))
G, S, B = 1024, 25, 8
ev = HipNet(net, "cuda:0")
eng = SelfPlayEngine(g, G, evaluators=[ev], max_batch=B, seed=0, stagger=True, searches_hint=S)
for _ in range(20):
    eng.search(S, B); eng.drain()
L = _lib.load()
hn = HipNet(net, "cuda:0", mode="f32w")
rows = 1434
x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
stamps = torch.zeros(4 * 2048, dtype=torch.int64, device="cuda")
def ghz():
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64); s = s[s[:, 0] > 0]
    return np.median(s[:, 0] / (s[:, 1] * 10.0)), np.median(s[:, 1]) / 100.0
def pairs(n):
    _lib.check(L.caro_search_staggered(eng.h, ev.h, None, n, B, C.c_void_p(eng.planes.data_ptr()), C.c_void_p(eng.leaf_keys.data_ptr()),
                                       C.c_void_p(eng._probs.data_ptr()), C.c_void_p(eng._values.data_ptr()), eng._stream()))
def dense(k, stamped_last=True):
    for i in range(k - 1): hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    _lib.check(L.caro_net_forward_stamped(hn.h, x.data_ptr(), counts.data_ptr(), 0, rows, probs.data_ptr(), vals.data_ptr(), stamps.data_ptr(), None))
for npairs in (0, 1, 3, 10):
    for k in (1, 2, 5, 20):
        torch.cuda.synchronize(); stamps.zero_()
        if npairs: pairs(npairs)
        dense(k)
        torch.cuda.synchronize()
        f, w = ghz()
        print("after a sync: %2d (tree, net) pairs, then dense net launch number %2d: %.2f GHz, workgroup wall %.1f us" % (npairs, k, f, w), flush=True)
# steady state: the engine's own net launch after n continuous pairs (its workgroups stamp cycles and 100 MHz ticks)
es = torch.zeros(4 * (G * B // 3 + 8), dtype=torch.int64, device="cuda")
_lib.check(L.caro_net_debug_stamps(ev.h, C.c_void_p(es.data_ptr())))
for n in (1, 3, 25, 100, 500, 2000):
    torch.cuda.synchronize(); es.zero_()
    pairs(n)
    torch.cuda.synchronize()
    raw = es.cpu().numpy().view(np.uint64).reshape(-1, 4); raw = raw[raw[:, 0] > 0]
    cyc = raw[:, 0].astype(np.float64); dur = (raw[:, 1] & np.uint64(0xFFFFF)).astype(np.float64)
    print("engine: net launch of pair %4d after a sync: %.2f GHz (workgroup %.0f cycles, %.1f us)" % (n, np.median(cyc / (dur * 10.0)), np.median(cyc), np.median(dur) / 100.0), flush=True)
_lib.check(L.caro_net_debug_stamps(ev.h, None))
eng.close()
