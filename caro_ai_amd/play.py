#!/usr/bin/env python3
"""Round-robin arena between checkpoints, the drop-in for the reference's play.py:15-76.

Same CLI (`models...`, `-r/--rounds`, `--cuda`, `-g/--game`), same pairing (every ordered pair plays `rounds`
games), same settings (tau = 0 from move 0, PLAY_MCTS_SEARCHES x PLAY_MCTS_BATCH_SIZE, a fresh pair of trees
per game, first player random) and the same output lines; each pair's games run concurrently on the HIP engine.

    python -m caro_ai_amd.play -g 0 --cuda a.dat b.dat -r 64
"""
import argparse
import sys
import time

import torch

from caro_ai_amd import config as cfg
from caro_ai_amd.lib import model, utils
from caro_ai_amd.lib.game import game_provider


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("models", nargs="+", help="The list of models (at least 2) to play against each other")
    parser.add_argument("-r", "--rounds", type=int, default=2, help="Count of rounds to perform for every pair")
    parser.add_argument("--cuda", default=False, action="store_true", help="Enable CUDA")
    parser.add_argument("--seed", type=int, default=0)
    game_provider.add_game_argument(parser)
    args = parser.parse_args(argv)
    device = "cuda:0"  # the engine is GPU only; --cuda is accepted for CLI compatibility

    game = game_provider.get_game(args)
    nets = []
    for fname in args.models:
        net = model.Net(game.obs_shape, game.action_space)
        net.load_state_dict(torch.load(fname, map_location=lambda storage, loc: storage))
        nets.append((fname, net.to(device)))

    total_agent, total_pairs = {}, {}
    uid = 0
    for idx1, n1 in enumerate(nets):
        for idx2, n2 in enumerate(nets):
            if idx1 == idx2:
                continue
            ts = time.time()
            res = utils.play_games(game, args.rounds, None, n1[1], n2[1], steps_before_tau_0=0,
                                   mcts_searches=cfg.PLAY_MCTS_SEARCHES, mcts_batch_size=cfg.PLAY_MCTS_BATCH_SIZE,
                                   concurrent=min(args.rounds, 1024), seed=args.seed, uid_base=uid, device=device,
                                   first_player_mode=2)
            uid += args.rounds
            wins, losses, draws = res.count(1), res.count(-1), res.count(0)
            speed_games = args.rounds / (time.time() - ts)
            print("%s vs %s -> w=%d, l=%d, d=%d" % (n1[0], n2[0], wins, losses, draws))
            sys.stderr.write("Speed %.2f games/s\n" % speed_games)
            sys.stdout.flush()
            utils.update_counts(total_agent, n1[0], (wins, losses, draws))
            utils.update_counts(total_agent, n2[0], (losses, wins, draws))
            utils.update_counts(total_pairs, (n1[0], n2[0]), (wins, losses, draws))

    leaders = sorted(total_agent.items(), reverse=True, key=lambda p: p[1][0])
    print("Leaderboard:")
    for name, (wins, losses, draws) in leaders:
        print("%s: \t w=%d, l=%d, d=%d" % (name, wins, losses, draws))
    return total_agent, total_pairs


if __name__ == "__main__":
    main()
