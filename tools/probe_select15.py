"""Phase stamps of the tree kernels on BASELINE config 4 (15 x 15, 1024 games, 50 x 8, eviction, cap 4096): where a
block's time goes in mid-game.  python tools/probe_select15.py [warm-up moves]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.tictactoe import TicTacToe
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
g = TicTacToe(15, 5)
torch.manual_seed(0)
net = Net(g.obs_shape, g.action_space).to("cuda:0").eval()
G = 1024
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
eng = SelfPlayEngine(g, G, evaluators=[HipNet(net, "cuda:0")], max_batch=8, seed=0, searches_hint=50, evict=True, node_cap=4096)
for i in range(warm):
    eng.search(50, 8); eng.step(); eng.drain()
L = _lib.load()
_lib.check(L.caro_debug_stamps(eng.h, 1))
eng.search(50, 8)
out = np.zeros(G * 16, np.uint64)
_lib.check(L.caro_debug_read(eng.h, out.ctypes.data, out.size, None))
d = out[:G * 8].reshape(G, 8).astype(np.float64)
x = out[G * 8:].reshape(G, 8).astype(np.float64)
q = lambda v: np.percentile(v, [50, 90, 99, 100]).round(0)
print("select (wave 0 of each block), cycles since kernel start (median / p90 / p99 / max over %d blocks):" % G)
print("  noise row ready          ", q(d[:, 0]))
print("  root level done          ", q(d[:, 1]), " -> root level alone", q(d[:, 1] - d[:, 0]))
print("  all descents done        ", q(d[:, 2]), " -> levels after the root", q(d[:, 2] - d[:, 1]))
print("  end of block             ", q(d[:, 3]), " -> dedupe + records (+ planes)", q(d[:, 3] - d[:, 2]))
print("  max depth                ", q(d[:, 4] % 256))
if d[:, 7].max() > 0:
    print("  fused block: expand + backup", q(d[:, 6]), " whole block", q(d[:, 7]))
if x[:, 6].max() > 0:
    print("expand_body phases (cycles from its start): preload arrived | leaves inserted | queue flat | entries listed | rows written | backups applied")
    for k in range(1, 7):
        print("   ", k, q(x[:, k] - x[:, 0]))
    print("  queue entries", q(x[:, 7]))
c = eng.counters()
print("levels / sim", c["levels"] / c["sims"], "expansions / sim", c["expansions"] / c["sims"])
eng.close()
