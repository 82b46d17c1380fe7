#!/usr/bin/env python3
"""Per-phase breakdown of train.fit iterations at 1 024 self-play games (VERDICT r5 task 1): where an iteration of the
reference's loop (train.py:165-217: self-play with the best net -> replay buffer -> TRAIN_ROUNDS SGD steps -> every
EVALUATE_EVERY_STEP iterations the arena gate) spends its wall time on the MI355X, for the exact and the stream form of
self-play.  Writes one JSON object (default profiles/r06_fit_breakdown.json).

    python tools/fit_breakdown.py [--games 1024] [--iterations 6] [--out profiles/r06_fit_breakdown.json]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--concurrent", type=int, default=1024)
    ap.add_argument("--iterations", type=int, default=6)
    ap.add_argument("--evaluate-every", type=int, default=3, help="(config.EVALUATE_EVERY_STEP is 100: lowered here so that "
                                                                  "the gate shows up inside a few iterations)")
    ap.add_argument("--reference-evaluate", type=int, default=1, help="1: the gate with the reference's evaluate semantics "
                                                                       "(fit's default in a single process); 0: sharded rounds")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_fit_breakdown.json"))
    args = ap.parse_args()
    from caro_ai_amd import config as cfg
    from caro_ai_amd import train
    from caro_ai_amd.data import weights_path
    from caro_ai_amd.lib.game.connect_four import ConnectFour
    from caro_ai_amd.lib.model import Net
    cfg.EVALUATE_EVERY_STEP = args.evaluate_every
    dev = "cuda:0"
    game = ConnectFour()
    out = {"config": {"game": "Connect4", "games_per_iteration": args.games, "concurrent": args.concurrent,
                      "searches": cfg.MCTS_SEARCHES, "batch": cfg.MCTS_BATCH_SIZE, "train_rounds": cfg.TRAIN_ROUNDS,
                      "batch_size": cfg.BATCH_SIZE, "evaluate_every": args.evaluate_every,
                      "evaluation_rounds": cfg.EVALUATION_ROUNDS, "weights": "best_026_12000.dat (start)"},
           "note": "seconds per phase of each train.fit iteration, host wall clock with a device synchronisation at "
                   "every phase boundary; self_play_setup / _play / _gather split the self-play phase (setup = HipNet for "
                   "the weights + engine construction (first iteration) or in-place restart; play = the move loop; "
                   "gather = tuple exchange + append to the device replay buffer); iteration 1 carries the first-use "
                   "costs (code objects, torch kernel selection for the SGD step)"}
    for form in ("exact", "stream"):
        train.release_engines()
        torch.manual_seed(0)
        net = Net(game.obs_shape, game.action_space)
        net.load_state_dict(torch.load(weights_path("best_026_12000.dat"), map_location="cpu"))
        net = net.to(dev)
        t0 = time.time()
        args.reference_evaluate = bool(args.reference_evaluate)
        hist = train.fit(game, net, dev, args.games, iterations=args.iterations, sample_seed=1, log=None,
                         concurrent=args.concurrent, stream=(form == "stream"), reference_evaluate=args.reference_evaluate)
        torch.cuda.synchronize()
        total = time.time() - t0
        ph = hist["phases"]
        steady = ph[1:] or ph
        keys = ("self_play", "self_play_setup", "self_play_play", "self_play_gather", "train", "broadcast", "evaluate")
        mean = {k: sum(p[k] for p in steady) / len(steady) for k in keys}
        nodes = sum(p["nodes"] for p in steady)
        out[form] = {"iterations": ph, "seconds_total": total,
                     "mean_after_first": mean,
                     "speed_nodes_self_play_phase": nodes / max(sum(p["self_play"] for p in steady), 1e-9),
                     "speed_nodes_whole_loop": nodes / max(sum(sum(p[k] for k in ("self_play", "train", "broadcast", "evaluate"))
                                                               for p in steady), 1e-9),
                     "evaluations": hist["evaluations"], "promotions": hist["promotions"],
                     "loss_total": hist["loss_total"]}
        print("[fit_breakdown] %s: %s" % (form, json.dumps(out[form]["mean_after_first"])), file=sys.stderr, flush=True)
    train.release_engines()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k]["mean_after_first"] for k in ("exact", "stream")}))


if __name__ == "__main__":
    main()
