#!/usr/bin/env python3
"""Round-5 golden vector of the reference's line helpers (ref lib/game/tictactoe/tictactoe_helpers.py:7-179): the six
functions run on random boards of 3x3 .. 15x15 and random cells, a SHA-256 per board size over everything they return
(get_row | get_col | get_diag | get_antidiag of the cell, k_in_a_row of each line and check_win for both tokens and
k = 3, 4, 5); and of the codec helpers of the two game classes (list / matrix views of random positions:
tests/rules_digest.py::codec_digest).  tests/test_cpu_product.py recomputes the digests with this package's modules.

Usage:  python tests/golden/make_golden_r5_helpers.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from lib.game.tictactoe import tictactoe_helpers as ref_helpers  # noqa: E402
from tests.rules_digest import codec_digest, helpers_digest  # noqa: E402

SEED, CASES = 20261005, 300


def main():
    out = {"seed": SEED, "cases": CASES, "digests": {str(n): helpers_digest(ref_helpers, n, SEED, CASES) for n in (3, 4, 5, 8, 15)}}
    out["codec"] = {"c4": codec_digest(mg.ConnectFour(), SEED, CASES)}
    for n, k in ((3, 3), (5, 4), (15, 5)):
        out["codec"]["mnk%d" % n] = codec_digest(mg.TicTacToe(n, k), SEED, CASES)
    print(out)
    mg.dump("helpers_digest.json.gz", out)


if __name__ == "__main__":
    main()
