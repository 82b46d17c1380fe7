"""Experiment: does replaying a move's 25 launch pairs as a HIP graph beat enqueuing them one by one?
(bench's idle_frac says the kernels fill 97 % of the step; a graph can only recover part of the rest.)
Captures caro_search_staggered's launches of one move with torch.cuda.graph -- twice, because the leaf-count ping-pong
flips 25 times per move: graph A starts on parity 0, graph B on parity 1 -- and replays A, B, A, B ... with the normal
drain between moves.  python tools/exp/graph_probe.py [moves]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from caro_ai_amd.engine import SelfPlayEngine
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
g = ConnectFour()
net = Net(g.obs_shape, 7); net.load_state_dict(torch.load(os.path.join(ROOT, "tests/golden/weights/best_026_12000.dat"), map_location="cpu"))
G, S, B = 1024, 25, 8
MOVES = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def run(graphs):
    eng = SelfPlayEngine(g, G, evaluators=[HipNet(net, "cuda:0")], max_batch=B, seed=0, stagger=True, searches_hint=S)
    for _ in range(20):
        eng.move(S, B)
    eng.flush()
    torch.cuda.synchronize()
    gs = None
    if graphs:
        side = torch.cuda.Stream()
        gs = []
        for _ in range(2):  # parity 0 start, parity 1 start
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(gr, stream=side):
                    eng.search(S, B)
            gs.append(gr)
        torch.cuda.synchronize()
    c0 = eng.counters()
    t0 = time.perf_counter()
    for i in range(MOVES):
        if gs:
            gs[i & 1].replay()
            d = eng.drain()
        else:
            eng.search(S, B)
            d = eng.drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    eng.close()
    return (c1["expansions"] - c0["expansions"]) / dt, dt * 1e3 / MOVES, c1["overflows"], c1["sims"] - c0["sims"]


for mode in (False, True, False, True):
    try:
        v, ms, ov, sims = run(mode)
        print("graphs %-5s %.3f M node-expansions/s  %.3f ms per move  overflows %d  sims %d" % (mode, v / 1e6, ms, ov, sims), flush=True)
    except Exception as e:
        print("graphs %s failed: %r" % (mode, e), flush=True)
