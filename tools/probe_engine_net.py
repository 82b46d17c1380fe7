"""Per-workgroup clock of k_net_forward_w INSIDE the engine's own launches (staggered mode, slot rows, behind the tree
kernel), next to the same leaf count launched back to back on dense rows (tools/probe_small.py): where the difference
between a launch in the bench (rocprofv3 / HIP events) and a workgroup's own cycles goes.
    python tools/probe_engine_net.py [--arena]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd import _lib
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
arena = "--arena" in sys.argv
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
G, S, B = (512, 100, 8) if arena else (1024, 25, 8)
evs = [HipNet(net, "cuda:0")]
if arena:
    net2 = Net(g.obs_shape, 7); net2.load_state_dict(torch.load("caro_ai_amd/data/weights/best_025_10600.dat", map_location="cpu"))
    evs.append(HipNet(net2, "cuda:0"))
eng = SelfPlayEngine(g, G, evaluators=evs, max_batch=B, seed=0, stagger=True, searches_hint=S,
                     **({"n_stores": 2, "first_player_mode": 2, "steps_before_tau_0": 0} if arena else {}))
for _ in range(6 if arena else 20):
    eng.search(S, B); eng.drain()
L = _lib.load()
grid = G * B // 3 + 8
stamps = torch.zeros(4 * grid, dtype=torch.int64, device="cuda")
_lib.check(L.caro_net_debug_stamps(evs[0].h, C.c_void_p(stamps.data_ptr())))
nets = [e.h for e in eng.evaluators] + [None]
_lib.check(L.caro_debug_stamps(eng.h, 1))
M64 = np.uint64
for rep in range(6):
    stamps.zero_()
    torch.cuda.synchronize()
    # three launch pairs back to back (the clock ramps down across a host sync); the stamps are those of the LAST pair
    _lib.check(L.caro_search_staggered(eng.h, nets[0], nets[1], 3, B, C.c_void_p(eng.planes.data_ptr()),
                                       C.c_void_p(eng.leaf_keys.data_ptr()), C.c_void_p(eng._probs.data_ptr()),
                                       C.c_void_p(eng._values.data_ptr()), eng._stream()))
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().view(np.uint64).reshape(-1, 4)
    raw = raw[raw[:, 0] > 0]
    n_start = (raw[:, 1] >> M64(20)).astype(np.int64)            # 100 MHz ticks (44 bits)
    raw = raw[n_start > np.median(n_start) - 5000]               # rows an earlier pair with more workgroups left behind
    cyc = raw[:, 0].astype(np.float64)
    n_start = (raw[:, 1] >> M64(20)).astype(np.int64)
    n_dur = (raw[:, 1] & M64(0xFFFFF)).astype(np.int64)
    n_end = n_start + n_dur
    out = np.zeros(G * 16, np.uint64)
    _lib.check(L.caro_debug_read(eng.h, out.ctypes.data, out.size, None))
    d = out[:G * 8].reshape(G, 8)
    live = d[:, 7] > 0
    t_end = (d[live, 5] >> M64(24)).astype(np.int64) & ((1 << 40) - 1)
    whole = d[live, 7].astype(np.float64)
    ghz = np.median(cyc / (n_dur * 10.0))
    t_start = t_end - (whole / (ghz * 10.0)).astype(np.int64)
    n_start &= (1 << 40) - 1; n_end &= (1 << 40) - 1
    z = t_start.min()
    us = lambda t: (t - z) / 100.0
    print("pair %d (us from the first tree block's start): tree blocks end median %.1f p90 %.1f last %.1f | net workgroups (%d) start first %.1f median %.1f last %.1f | end first %.1f median %.1f last %.1f"
          % (rep, us(np.median(t_end)), us(np.percentile(t_end, 90)), us(t_end.max()), raw.shape[0], us(n_start.min()),
             us(np.median(n_start)), us(n_start.max()), us(n_end.min()), us(np.median(n_end)), us(n_end.max())))
    print("        workgroup cycles median %.0f max %.0f (%.1f us at %.2f GHz) | conv_in %.0f trunk %.0f heads %.0f"
          % (np.median(cyc), cyc.max(), np.median(cyc) / ghz / 1e3, ghz, np.median(raw[:, 2].astype(np.float64)),
             np.median((raw[:, 3] - raw[:, 2]).astype(np.float64)), np.median((raw[:, 0] - raw[:, 3]).astype(np.float64))), flush=True)
_lib.check(L.caro_net_debug_stamps(evs[0].h, None))
eng.close()
