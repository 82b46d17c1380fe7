#!/usr/bin/env python3
"""Round-5 golden vector of the reference's interactive session: `lib/play_session.Session`
(ref lib/play_session.py:7-49) itself, driven in the build container through whole games of connect four and
TicTacToe -- the "human" plays a scripted rule (the legal move at index (3 * turn + 1) mod the number of legal moves),
the bot answers with `move_bot` (BOT_MCTS_SEARCHES x BOT_MCTS_BATCH_SIZE = 40 x 8 sims on the session's persistent
store, tau = 0, move drawn with numpy from the one-hot policy) -- numpy's global generator seeded.  Harness rule as
everywhere (Q9): the session's net in eval mode, no autograd.  Recorded per bot turn: the move, the position value the
session reports, the state after it, `render()`; per game who won.

Usage:  python tests/golden/make_golden_r5_session.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from lib import play_session as ref_session  # noqa: E402  (the reference's, /root/reference is first on sys.path)


def human_move(game, state, turn):
    legal = game.possible_moves(state)
    return int(legal[(3 * turn + 1) % len(legal)])


def drive(game, model_file, player_moves_first, seed):
    np.random.seed(seed)
    s = ref_session.Session(game, model_file, player_moves_first)
    s.model.eval()
    turns, outcome, turn = [], None, 0
    with torch.no_grad():
        while outcome is None:
            if player_moves_first or turns:  # the human opens, or answers the bot's last move
                mv = human_move(game, s.state, turn)
                assert s.is_valid_move(mv)
                if s.move_player(mv):
                    outcome = "human"
                    break
                if s.is_draw():
                    outcome = "draw"
                    break
            won = s.move_bot()
            turns.append({"move": int(s.moves[-1]), "value": float(s.value), "state": str(s.state), "render": s.render()})
            if won:
                outcome = "bot"
            elif s.is_draw():
                outcome = "draw"
            turn += 1
    return {"seed": seed, "player_moves_first": player_moves_first, "moves": [int(m) for m in s.moves],
            "turns": turns, "outcome": outcome, "store_len": len(s.mcts_store)}


def main():
    torch.set_num_threads(1)
    c4, ttt = mg.ConnectFour(), mg.TicTacToe()
    w26 = os.path.join(mg.REF, "saves/trained_connect4/best_026_12000.dat")
    wt5 = os.path.join(mg.REF, "saves/trained_tictactoe/best_005_00900.dat")
    out = {"searches": ref_session.cfg.BOT_MCTS_SEARCHES, "batch": ref_session.cfg.BOT_MCTS_BATCH_SIZE,
           "c4": {"weights": "best_026_12000.dat", "games": [drive(c4, w26, True, 21), drive(c4, w26, False, 22)]},
           "ttt3": {"weights": "best_005_00900.dat", "games": [drive(ttt, wt5, True, 23), drive(ttt, wt5, False, 24)]}}
    for k in ("c4", "ttt3"):
        for g in out[k]["games"]:
            print(k, g["player_moves_first"], g["outcome"], g["moves"], g["store_len"])
    # `render` of other boards (the session only shows 6x7 and 3x3): a few random positions each
    rng = np.random.default_rng(5)
    renders = []
    for kind, n, k, game in (("c4", 0, 0, c4), ("mnk", 3, 3, ttt), ("mnk", 5, 4, mg.TicTacToe(5, 4)),
                             ("mnk", 15, 5, mg.TicTacToe(15, 5))):
        for _ in range(3):
            st, pl = game.initial_state, 0
            for _ply in range(int(rng.integers(1, 12))):
                legal = game.possible_moves(st)
                st, won = game.move(st, int(legal[int(rng.integers(len(legal)))]), pl)
                pl = 1 - pl
                if won:
                    break
            renders.append({"kind": kind, "n": n, "k": k, "state": str(st), "render": game.render(st)})
    out["renders"] = renders
    mg.dump("session.json.gz", out)


if __name__ == "__main__":
    main()
