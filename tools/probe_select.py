"""Where does k_select's time go?  Variants: generated vs explicit noise, node_cap, G."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet

g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load("tests/golden/weights/best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0")

def run(G, cap, explicit, moves=12):
    eng = SelfPlayEngine(g, G, evaluators=[hn], max_batch=8, node_cap=cap, seed=0)
    nz = torch.full((G, 8, 7), 1.0 / 7, dtype=torch.float64, device="cuda:0") if explicit else None
    for _ in range(6):
        eng.search(25, 8, None if nz is None else [nz] * 25); eng.step(); eng.drain()
    eng.profile(True); eng.profile_read()
    c0 = eng.counters(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(moves):
        eng.search(25, 8, None if nz is None else [nz] * 25); eng.step(); eng.drain()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c1 = eng.counters(); pr = eng.profile_read(); eng.profile(False)
    us = {k: round(v[0] * 1e3 / max(1, v[1]), 1) for k, v in pr.items()}
    print("G=%5d cap=%6d noise=%-8s  exp/s %.2fM  depth %.2f  us/launch %s" % (
        G, cap, "explicit" if explicit else "device", (c1["expansions"] - c0["expansions"]) / dt / 1e6,
        (c1["levels"] - c0["levels"]) / (c1["sims"] - c0["sims"]), us), flush=True)
    eng.close()

run(1024, 8464, False)
run(1024, 8464, True)
run(1024, 3072, False)
run(256, 8464, False)
run(2048, 8464, False)
run(4096, 8464, False)
