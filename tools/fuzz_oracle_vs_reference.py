#!/usr/bin/env python3
"""Build-container check (needs /root/reference): a differential fuzz of the ORACLE against the REFERENCE beyond the
committed fixtures -- whole games with the table net on randomly drawn configurations (connect four, and m,n,k boards of
3x3 .. 7x7 with k from 3 to n; searches 2 .. 30 (with ONE search from an unexpanded root no edge is visited and the
reference's get_policy_value divides by zero, lib/mcts.py:311), batch 1 .. 16, one store or one per player, tau switch after 0 .. 12 plies,
either opener), every game played by the reference's `play_game` under the harness of tests/golden/make_golden.py and
replayed by oracle/caro_oracle.c with `tests/test_oracle_golden._check_game` (result, steps, boards, players, z, root N and
node count after every search, float64 pi: all exact).

    python tools/fuzz_oracle_vs_reference.py [games] [seed]     -> summary; profiles/r05_oracle_fuzz.txt holds a run
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402  (reference first on sys.path)
from oracle.oracle import Oracle  # noqa: E402
from tests.test_oracle_golden import _check_game  # noqa: E402


def main():
    n_games = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = {"games": 0, "plies": 0, "draws": 0, "two_stores": 0, "c4": 0, "shapes": set()}
    for i in range(n_games):
        if rng.random() < 0.3:
            game, kind, n, k = mg.ConnectFour(), "c4", 0, 0
        else:
            n = int(rng.integers(3, 8))
            k = int(rng.integers(3, n + 1))
            game, kind = mg.TicTacToe(n, k), "mnk"
        S, B = int(rng.integers(2, 31)), int(rng.choice([1, 2, 3, 4, 8, 16]))
        ns, sbt0, fp = int(rng.integers(1, 3)), int(rng.integers(0, 13)), int(rng.integers(0, 2))
        g = mg.strip(mg.play_reference(game, mg.SynthNet(game), mg.SynthNet(game), ns, sbt0, S, B, fp, seed, 100000 + i, True), False)
        o = Oracle(Oracle.C4, n_stores=ns) if kind == "c4" else Oracle(Oracle.MNK, n, k, n_stores=ns)
        _check_game(o, g, lambda oo: oo.use_synth_net())
        stats["games"] += 1
        stats["plies"] += g["plies"]
        stats["draws"] += int(g["result"] == 0)
        stats["two_stores"] += int(ns == 2)
        stats["c4"] += int(kind == "c4")
        stats["shapes"].add((kind, n, k))
        if (i + 1) % 50 == 0:
            print("%d games equal so far (%.0f s)" % (i + 1, time.time() - t0), flush=True)
    stats["shapes"] = sorted(stats["shapes"])
    print("oracle == reference on %(games)d random whole games (%(plies)d plies; %(draws)d draws; %(two_stores)d with one store per "
          "player; %(c4)d connect four; board shapes %(shapes)s)" % stats)


if __name__ == "__main__":
    main()
