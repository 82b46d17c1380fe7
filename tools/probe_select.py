"""Phase stamps of k_select (diagnostic mode): where a wave's time goes, per game, in steady state."""
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
G = 1024
eng = SelfPlayEngine(g, G, evaluators=[HipNet(net, "cuda:0")], max_batch=8, seed=0)
for _ in range(20):
    eng.search(25, 8); eng.step(); eng.drain()
L = _lib.load()
_lib.check(L.caro_debug_stamps(eng.h, 1))
eng.search(25, 8)
out = np.zeros(G * 8, np.uint64)
_lib.check(L.caro_debug_read(eng.h, out.ctypes.data, out.size, None))
d = out.reshape(G, 8).astype(np.float64)
rows, noise, loop, end, maxd = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
q = lambda x: np.percentile(x, [50, 90, 99, 100]).round(0)
print("cycles since kernel start (median / p90 / p99 / max over %d waves), last minibatch of a move:" % G)
print("  root rows + key arrived", q(rows))
print("  noise generated        ", q(noise), " -> noise gen alone", q(noise - rows))
print("  all descents done      ", q(loop), " -> descent loop after the root level", q(loop - noise))
print("  end of kernel          ", q(end), " -> dedupe + result writes", q(end - loop))
if d[:, 7].max() > 0:  # fused k_tree: expand + backup of the previous minibatch precedes the descents in the same block
    print("  k_tree, last launch with descents: expand + backup", q(d[:, 6]), " whole block", q(d[:, 7]))
    print("  k_tree, closing launch: expand + backup alone", q(d[:, 5]))
print("  max depth in the wave  ", q(maxd), " cycles per level after the root (median)", np.median((loop - noise) / np.maximum(1, maxd - 1)).round(0))
if d[:, 7].max() > 0:
    order = np.argsort(-d[:, 7])[:12]
    print("  the 12 slowest blocks: whole | expand+backup | descents: to the root level | loop | tail | max depth")
    for b in order:
        print("   %7.0f | %7.0f | %7.0f | %7.0f | %7.0f | %3.0f" % (d[b, 7], d[b, 6], noise[b], loop[b] - noise[b], end[b] - loop[b], maxd[b]))
eng.close()
