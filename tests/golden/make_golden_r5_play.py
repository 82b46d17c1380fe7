#!/usr/bin/env python3
"""Round-5 golden vector of the reference's ARENA SCRIPT: `play.py` itself (ref play.py:15-76), run as `__main__` in the
build container through runpy with numpy's global generator seeded -- the script draws the opener
(`np.random.choice(2)`), one Dirichlet row per descent and one `choice` per ply from it, so a seed makes its run
reproducible -- on the shipped checkpoints, two rounds per ordered pair, its own 40 x 8 sims per move
(ref config.py:18-19), fresh stores per game (`mcts_stores=None`).  What it printed and the (result, steps) of every
game are recorded.

Harness rule as for every other fixture (declared deviation): the nets run in eval mode without autograd --
the script itself leaves them in train mode; `lib.utils.play_game` is wrapped for the duration of the run to switch the
two nets it is handed to `.eval()` and to call the reference's function under `torch.no_grad()`.  Nothing else is
touched; the wrapper also logs each game's return value.

tests/test_gpu_shim.py::test_reference_play_script_loop_on_this_packages_play_game runs the same loop on THIS package's
`lib.utils.play_game` (INTEGRATION level 1: swap the imports), same seed, nets on the CPU so that the net arithmetic is
the reference's: same lines.

Usage:  python tests/golden/make_golden_r5_play.py
"""
import contextlib
import io
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

SEED, ROUNDS = 7, 2
MODELS = ["best_026_12000.dat", "best_025_10600.dat"]


def main():
    torch.set_num_threads(1)
    games = []
    real = mg.ref_utils.play_game

    def play_game(*a, **k):
        for key in ("net1", "net2"):
            k[key].eval()
        with torch.no_grad():
            r, steps = real(*a, **k)
        games.append([int(r), int(steps)])
        return r, steps

    cwd, argv = os.getcwd(), sys.argv
    buf, err = io.StringIO(), io.StringIO()
    mg.ref_utils.play_game = play_game
    try:
        os.chdir(os.path.join(mg.REF, "saves", "trained_connect4"))
        sys.argv = ["play.py"] + MODELS + ["-r", str(ROUNDS), "-g", "0"]
        np.random.seed(SEED)
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):
            runpy.run_path(os.path.join(mg.REF, "play.py"), run_name="__main__")
    except SystemExit:
        sys.stderr.write(err.getvalue())
        raise
    finally:
        mg.ref_utils.play_game = real
        os.chdir(cwd)
        sys.argv = argv
    lines = buf.getvalue().splitlines()
    print("\n".join(lines))
    print(games)
    assert len(games) == 2 * ROUNDS and lines[-3] == "Leaderboard:"
    mg.dump("play_script_c4.json.gz", {"kind": "c4", "seed": SEED, "rounds": ROUNDS, "models": MODELS,
                                       "searches": 40, "batch": 8, "stdout": lines, "games": games})


if __name__ == "__main__":
    main()
