"""Phase stamps of k_tree_stag (diagnostic mode): where a block's time goes in steady state, staggered mode."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "caro_ai_amd", "data", "weights", "best_026_12000.dat"), map_location="cpu"))
G, S, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 25, 8
CAP = int(sys.argv[2]) if len(sys.argv) > 2 else None
eng = SelfPlayEngine(g, G, evaluators=[HipNet(net, "cuda:0")], max_batch=B, seed=0, stagger=True, searches_hint=S, node_cap=CAP)
for _ in range(20):
    eng.search(S, B); eng.drain()
L = _lib.load()
_lib.check(L.caro_debug_stamps(eng.h, 1))
nets = [e.h for e in eng.evaluators] + [None]
q = lambda x: np.percentile(x, [50, 90, 99, 100]).round(0)
for rep in range(3):
    _lib.check(L.caro_search_staggered(eng.h, nets[0], nets[1], 1, B, C.c_void_p(eng.planes.data_ptr()),
                                       C.c_void_p(eng.leaf_keys.data_ptr()), C.c_void_p(eng._probs.data_ptr()),
                                       C.c_void_p(eng._values.data_ptr()), eng._stream()))
    out = np.zeros(G * 16, np.uint64)
    _lib.check(L.caro_debug_read(eng.h, out.ctypes.data, out.size, None))
    d = out[:G * 8].reshape(G, 8).astype(np.float64)
    x = out[G * 8:].reshape(G, 8).astype(np.float64)  # expand_body: stamps at its phase boundaries, entry count
    ok = x[:, 6] > x[:, 0]
    ph = np.diff(x[ok, :7], axis=1)
    print("   expand_body phases (median / p90 / max): round-1 loads %s | insert %s | flatten %s | entries %s | rows %s | owners + edges %s | queue entries %s"
          % tuple([np.percentile(ph[:, i], [50, 90, 100]).round(0) for i in range(6)] + [np.percentile(x[ok, 7], [50, 90, 100])]))
    raw4 = out[:G * 8].reshape(G, 8)[:, 4]
    noise, loop, end, maxd = d[:, 0], d[:, 2], d[:, 3], (raw4 & np.uint64(0xFF)).astype(np.float64)
    hw = (raw4 >> np.uint64(8)).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    if rep == 0:
        place = {}
        for b in range(G):
            place.setdefault((b % 8, int(se[b]), int(sh[b]), int(cu[b]), int(simd[b])), []).append(b)
        per = np.bincount([len(v_) for v_ in place.values()], minlength=5)
        print("   tree waves per (XCD, SE, SH, CU, SIMD): histogram of 1, 2, 3, 4+ =", per[1], per[2], per[3], per[4:].sum(),
              "| SIMD ids used:", np.bincount(simd, minlength=4))
        print("   blocks 0..15 ->", [(int(b % 8), int(se[b]), int(cu[b]), int(simd[b])) for b in range(16)])
    step, exp, whole = (out[:G * 8].reshape(G, 8)[:, 5] & np.uint64(0xFFFFFF)).astype(np.float64), d[:, 6], d[:, 7]
    st = step > 500
    print("launch %d: whole block (median / p90 / p99 / max)" % rep, q(whole), "| blocks with a ply: %d" % st.sum())
    print("   expand + backup", q(exp), "| select part (to the end of its own stamps)", q(end))
    print("   ply (+ park / restart) where it happened", q(step[st]) if st.any() else "-", "| whole of those blocks", q(whole[st]) if st.any() else "-")
    print("   whole of the blocks WITHOUT a ply", q(whole[~st]))
    order = np.argsort(-whole)[:10]
    print("   the 10 slowest: whole | expand | ply | select | max depth")
    for b in order:
        print("   %7.0f | %7.0f | %7.0f | %7.0f | %3.0f" % (whole[b], exp[b], step[b], end[b], maxd[b]))
eng.close()
