#!/usr/bin/env python3
"""Build container only (needs /root/reference): how fast is the C port of the path (oracle/caro_oracle.c,
the thing bench.py times as `cpu_baseline` on the GPU box) relative to the reference's own Python, on ONE core
of the SAME host, same workload: Connect4 self-play, 25 x 8 sims/move, shipped best_026_12000.dat, eval-mode
batch-norm, torch CPU float32 forward on one thread, tau = 1 for 10 plies.  Then the same pair as N independent
single-thread processes, one per vCPU ("8 independent single-thread processes ... aggregate
node-expansions/s, stating 8 cores").  Writes profiles/cpu_ratio_r03.json, which bench.py attaches to its
cpu_baseline record.

    python tools/measure_cpu_ratio.py [seconds per side, default 60] [processes of the aggregate leg, default 8]
"""
import collections
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
torch.set_num_threads(1)


def time_reference(seconds, seed=0):
    sys.path.insert(0, REF)
    from lib import mcts as ref_mcts, utils as ref_utils, model as ref_model
    from lib.game.connect_four.connect_four import ConnectFour
    game = ConnectFour()
    net = ref_model.Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(os.path.join(REF, "saves/trained_connect4/best_026_12000.dat"), map_location="cpu"))
    net.eval()
    np.random.seed(seed)
    nodes = games = plies = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t0 < seconds:
            store = ref_mcts.MCTS(game)  # a fresh tree per game (what the engine does)
            rb = collections.deque()
            ref_utils.play_game(game, store, rb, net, net, 10, 25, 8, net1_plays_first=bool(games & 1))
            nodes += len(store)
            plies += len(rb)
            games += 1
    dt = time.perf_counter() - t0
    sys.path.remove(REF)
    for m in [m for m in sys.modules if m == "lib" or m.startswith("lib.") or m == "config"]:
        del sys.modules[m]
    return {"node_expansions_per_s": nodes / dt, "games": games, "plies": plies, "seconds": dt}


def time_port(seconds, base=0):
    from caro_ai_amd.lib.model import Net
    from oracle.oracle import Oracle
    o = Oracle(Oracle.C4)
    net = Net((2, 6, 7), 7)
    net.load_state_dict(torch.load(os.path.join(ROOT, "caro_ai_amd/data/weights/best_026_12000.dat"), map_location="cpu"))
    net.eval()

    def fn(planes, states, players):
        with torch.no_grad():
            lg, vl = net(torch.from_numpy(np.ascontiguousarray(planes)))
            return torch.softmax(lg, dim=1).numpy(), vl.numpy()[:, 0]

    o.set_net(0, fn)
    o.set_net(1, fn)
    games = plies = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        o.set_stream(0, base + games)
        r = o.play_game(10, 25, 8, games & 1)
        plies += r["plies"]
        games += 1
    dt = time.perf_counter() - t0
    return {"node_expansions_per_s": o.counters()["expansions"] / dt, "games": games, "plies": plies, "seconds": dt}


def _worker(job):
    kind, seconds, wid = job
    torch.set_num_threads(1)
    return time_reference(seconds, seed=wid) if kind == "reference" else time_port(seconds, base=wid * 100000)


def aggregate(kind, seconds, procs):
    """`procs` independent single-thread processes at once; the sum of their own rates"""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_worker, [(kind, seconds, w) for w in range(procs)])
    return {"node_expansions_per_s": sum(r["node_expansions_per_s"] for r in res), "processes": procs,
            "per_process": [r["node_expansions_per_s"] for r in res], "games": sum(r["games"] for r in res),
            "seconds": max(r["seconds"] for r in res)}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ref = time_reference(seconds)
    port = time_port(seconds)
    ref_n = aggregate("reference", seconds, procs)
    port_n = aggregate("port", seconds, procs)
    cpu = "?"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    out = {"port_per_core": port["node_expansions_per_s"], "reference_per_core": ref["node_expansions_per_s"],
           "ratio": port["node_expansions_per_s"] / ref["node_expansions_per_s"],
           "where": "build container, one core of %s, torch %s, numpy %s" % (cpu, torch.__version__, np.__version__),
           "workload": "Connect4 self-play, 25x8 sims/move, best_026_12000.dat, eval-mode BN, 1 torch thread",
           "port": port, "reference": ref,
           "reference_aggregate": ref_n, "port_aggregate": port_n,
           "aggregate_cores": procs, "aggregate_ratio": port_n["node_expansions_per_s"] / ref_n["node_expansions_per_s"],
           "note": "reference = /root/reference lib/utils.play_game + lib/mcts.MCTS timed in place; port = "
                   "oracle/caro_oracle.c driving the same torch CPU forward.  Multiply a GPU-box cpu_baseline.per_core "
                   "by 1/ratio to estimate what the reference's Python would do on that host."}
    with open(os.path.join(ROOT, "profiles", "cpu_ratio_r03.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
