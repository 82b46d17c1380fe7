#!/usr/bin/env python3
"""Round-5 golden vectors, again by RUNNING THE REFERENCE (build container only; outputs committed, reference not).
They close the pinning gaps the round-4 review named:

  draws_ttt3.json.gz        whole TicTacToe(3,3) games that end in a DRAW (ref lib/utils.py:86-96: result 0, every z 0;
                            lib/mcts.py:144-146: a full board inside the tree backs up 0.0), table net, mixed
                            searches x batch, one and two stores -- every earlier recorded game ended +-1
  rules_digest.json.gz      SURVEY 8(c) G1 at its stated size: 10^5 random connect-four plies, 10^4 each for 3x3 and
                            15x15 k=5 (+ 5 000 each for 5x5 k=4, 8x8 k=5, 10x10 k=5: the other lane geometries of the
                            rule kernels), as one SHA-256 per 1000-ply block (tests/rules_digest.py says what a block
                            is and what its digest absorbs: next state, won, legal mask, planes)
  arena_c4_320_x16.json.gz  SURVEY 8(c) G5: 32 seeded tau=0 arena games best_026 vs best_025, one store per player:
  arena_c4_800_x16.json.gz  16 at 40 x 8 sims/move (ref play.py:47-52 with config.py:18-19) and 16 at 100 x 8
                            (BASELINE config 5); per ply the root N vector (= the argmax move), pi, z; W / L / D

Harness as make_golden.py (table-driven np.random.dirichlet / np.random.choice keyed (seed, uid, ply, sim),
net.eval() + no_grad, fresh stores per game).  Also synth_mid.json.gz: whole table-net games on 6x6, 8x8 and 10x10.
Usage:  python tests/golden/make_golden_r5.py [draws|rules|arena|mid ...]
"""
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (puts /root/reference on sys.path and imports its lib)
from make_golden_r2 import slim  # noqa: E402
from tests import rules_digest as rd  # noqa: E402

RULES_SEED = 20261005
BLOCK = 1000


def draws():
    ttt = mg.TicTacToe()
    shapes = [(25, 1, 1, 10), (10, 8, 1, 2), (25, 4, 1, 0), (10, 8, 2, 0), (20, 16, 2, 0)]  # S, B, stores, tau-1 plies
    kept, per_shape, played = [], [0] * len(shapes), 0
    uid = 9000
    while min(per_shape) < 1 or len(kept) < 6:
        i = (uid - 9000) % len(shapes)
        S, B, ns, sbt0 = shapes[i]
        g = mg.play_reference(ttt, mg.SynthNet(ttt), mg.SynthNet(ttt), ns, sbt0, S, B, uid & 1, 11, uid, True)
        played += 1
        if g["result"] == 0 and per_shape[i] < 2:
            assert g["plies"] == 9 and all(z == 0 for z in g["z"])
            kept.append(mg.strip(g, keep_tables=False))
            per_shape[i] += 1
        uid += 1
        assert uid < 9400
    print("draws: kept %d of %d games played (per shape %s)" % (len(kept), played, per_shape))
    mg.dump("draws_ttt3.json.gz", {"kind": "mnk", "n": 3, "k": 3, "games": kept})


MID = {"mnk5": (5, 4), "mnk8": (8, 5), "mnk10": (10, 5)}  # the other lane geometries of the rule kernels: 5 blocks each


def _rules_block(job):
    name, b = job
    makers = {"c4": mg.ConnectFour, "ttt3": mg.TicTacToe, "mnk15": lambda: mg.TicTacToe(15, 5)}
    makers.update({k: (lambda nk=nk: mg.TicTacToe(*nk)) for k, nk in MID.items()})
    game = makers[name]()
    return name, b, rd.block_digest(game, RULES_SEED, b, BLOCK, game.action_space)


def rules():
    jobs = [("c4", b) for b in range(100)] + [("ttt3", b) for b in range(10)] + [("mnk15", b) for b in range(10)]
    out = {"c4": {"kind": "c4", "blocks": [None] * 100}, "ttt3": {"kind": "mnk", "n": 3, "k": 3, "blocks": [None] * 10},
           "mnk15": {"kind": "mnk", "n": 15, "k": 5, "blocks": [None] * 10}}
    for name, (n, k) in MID.items():
        jobs += [(name, b) for b in range(5)]
        out[name] = {"kind": "mnk", "n": n, "k": k, "blocks": [None] * 5}
    with mp.Pool(8) as pool:
        for name, b, dg in pool.imap_unordered(_rules_block, jobs):
            out[name]["blocks"][b] = dg
    for name, d in out.items():
        print(name, "plies", BLOCK * len(d["blocks"]), "wins", sum(x["wins"] for x in d["blocks"]),
              "draws", sum(x["draws"] for x in d["blocks"]))
    mg.dump("rules_digest.json.gz", {"seed": RULES_SEED, "block": BLOCK, "sets": out})


def mid():
    """whole table-net games on the board sizes between 5x5 and 15x15 (lane geometries 64 x 1 with one and two actions per
    lane), recorded like make_golden.py's: synth_mid.json.gz"""
    games = []
    for i, (n, k, S, B, ns, sbt0) in enumerate([(6, 4, 8, 8, 1, 3), (8, 5, 10, 1, 1, 4), (8, 5, 6, 8, 2, 0),
                                                (10, 5, 6, 8, 1, 5), (10, 5, 16, 1, 1, 2)]):
        g = mg.TicTacToe(n, k)
        rec = mg.strip(mg.play_reference(g, mg.SynthNet(g), mg.SynthNet(g), ns, sbt0, S, B, i & 1, 19, 9500 + i, True), False)
        rec["n"], rec["k"] = n, k
        games.append(rec)
        print("mid game %d: %dx%d k=%d, %dx%d sims, %d stores: %d plies, result %d" % (i, n, n, k, S, B, ns, rec["plies"],
                                                                                   rec["result"]), flush=True)
    mg.dump("synth_mid.json.gz", {"kind": "mnk-mixed", "games": games})


_nets = None


def _arena_game(job):
    global _nets
    S, B, seed, uid = job
    c4 = mg.ConnectFour()
    if _nets is None:
        _nets = (mg.load_net(c4, os.path.join(mg.REF, "saves/trained_connect4/best_026_12000.dat")),
                 mg.load_net(c4, os.path.join(mg.REF, "saves/trained_connect4/best_025_10600.dat")))
    return slim(mg.play_reference(c4, _nets[0], _nets[1], 2, 0, S, B, uid & 1, seed, uid, False))


def arena():
    for S, seed, uid0, name in [(40, 53, 7000, "arena_c4_320_x16.json.gz"), (100, 59, 7100, "arena_c4_800_x16.json.gz")]:
        with mp.Pool(8) as pool:
            games = pool.map(_arena_game, [(S, 8, seed, uid0 + i) for i in range(16)], chunksize=1)
        res = [g["result"] for g in games]
        tally = {"wins": res.count(1), "losses": res.count(-1), "draws": res.count(0)}
        print(name, tally, "plies", sum(g["plies"] for g in games))
        mg.dump(name, {"kind": "c4", "weights": ["best_026_12000.dat", "best_025_10600.dat"], "games": games,
                       "tally": tally})


def main():
    t0 = time.time()
    what = sys.argv[1:] or ["draws", "rules", "arena", "mid"]
    for w in what:
        {"draws": draws, "rules": rules, "arena": arena, "mid": mid}[w]()
        print("%s done, %.0f s" % (w, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
