#!/usr/bin/env python3
"""Round-2 additions to the golden vectors, again by RUNNING THE REFERENCE (build container only; the
outputs are committed, the reference is not).  Same harness as make_golden.py (imported from it), larger
real-weight samples so that the GPU-net tolerance can be stated on hundreds of plies instead of 33:

  real_c4_x32.json.gz     32 self-play games, shipped best_026_12000.dat, 25 x 8 sims/move (BASELINE config 2's
                          per-game settings), tau = 1 for 10 plies, first player = uid & 1
  arena_c4_800.json.gz    8 arena games best_026 vs best_025 at config 5's 100 x 8 sims/move, tau = 0, one
                          store per player

Usage:  python tests/golden/make_golden_r2.py
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts /root/reference on sys.path and imports its lib)


def slim(g):
    """keep what the GPU-side comparison reads; W/Q traces of 32 games are not needed a second time"""
    g = mg.strip(g, False)
    g["trace"] = [{"N": t["N"], "nodes": t["nodes"]} for t in g["trace"]]
    return g


def main():
    t0 = time.time()
    c4 = mg.ConnectFour()
    w26 = os.path.join(mg.REF, "saves/trained_connect4/best_026_12000.dat")
    w25 = os.path.join(mg.REF, "saves/trained_connect4/best_025_10600.dat")
    n26, n25 = mg.load_net(c4, w26), mg.load_net(c4, w25)
    games = []
    for i in range(32):
        games.append(slim(mg.play_reference(c4, n26, n26, 1, 10, 25, 8, i & 1, 31, 2000 + i, False)))
        print("self-play game %d: %d plies, %.0f s" % (i, games[-1]["plies"], time.time() - t0), flush=True)
    mg.dump("real_c4_x32.json.gz", {"kind": "c4", "weights": "best_026_12000.dat", "games": games})
    games = []
    for i in range(8):
        games.append(slim(mg.play_reference(c4, n26, n25, 2, 0, 100, 8, i & 1, 37, 3000 + i, False)))
        print("arena game %d: %d plies, result %d, %.0f s" % (i, games[-1]["plies"], games[-1]["result"],
                                                              time.time() - t0), flush=True)
    mg.dump("arena_c4_800.json.gz", {"kind": "c4", "weights": ["best_026_12000.dat", "best_025_10600.dat"],
                                     "games": games})
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
