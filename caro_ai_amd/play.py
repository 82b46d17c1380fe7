#!/usr/bin/env python3
"""Round-robin arena between checkpoints: the command-line surface of the reference's play.py:15-76 (same
arguments, same pairing rule, same printed lines) on the batched HIP engine.

Every ordered pair (first, second) of the given checkpoints meets `--rounds` times: tau = 0 from the first move,
PLAY_MCTS_SEARCHES x PLAY_MCTS_BATCH_SIZE simulations per move, a fresh pair of trees per game, the opening side
alternating with the game id (the reference draws it at random, play.py:47-52 / utils.py:65-66).  All rounds of
a pairing advance together on the GPU; with several ranks (torchrun) each rank plays a contiguous share of the
rounds and the win / loss / draw counts are all-reduced.

    python -m caro_ai_amd.play -g 0 --cuda a.dat b.dat -r 64
"""
import argparse
import itertools
import sys
import time
from dataclasses import dataclass

import torch

from caro_ai_amd import config as cfg
from caro_ai_amd import parallel
from caro_ai_amd.lib import model, utils
from caro_ai_amd.lib.game import game_provider


@dataclass(frozen=True)
class Tally:
    wins: int = 0
    losses: int = 0
    draws: int = 0

    @classmethod
    def of(cls, net1_results):
        return cls(sum(r > 0 for r in net1_results), sum(r < 0 for r in net1_results),
                   sum(r == 0 for r in net1_results))

    def as_tuple(self):
        return (self.wins, self.losses, self.draws)

    def mirrored(self):
        """the same games seen from the other side of the board"""
        return (self.losses, self.wins, self.draws)

    def __str__(self):
        return "w=%d, l=%d, d=%d" % self.as_tuple()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="+", help="The list of models (at least 2) to play against each other")
    ap.add_argument("-r", "--rounds", type=int, default=2, help="Count of rounds to perform for every pair")
    ap.add_argument("--cuda", default=False, action="store_true", help="Enable CUDA")
    ap.add_argument("--seed", type=int, default=0, help="key of the generated noise / move uniforms")
    ap.add_argument("--node-cap", type=int, default=0,
                    help="nodes per tree (default: searches x batch x cells, which cannot overflow; an overflow ends the run)")
    game_provider.add_game_argument(ap)
    return ap.parse_args(argv)


def load_checkpoint(game, path, device):
    """a `.dat` file is torch.save(net.state_dict()) (train.py:214-216)"""
    net = model.Net(game.obs_shape, game.action_space)
    net.load_state_dict(torch.load(path, map_location=lambda storage, loc: storage))
    return net.to(device).eval()


def meet(game, first, second, rounds, seed, uid_base, device, node_cap=None):
    """`rounds` games `first` (player 0) vs `second`; this rank plays its share, every rank gets the total"""
    rank, _, world = parallel.env_rank() if parallel.is_dist() else (0, 0, 1)
    lo, n = parallel.shard_rounds(rounds, rank, world)
    mine = []
    if n:
        mine = utils.play_games(game, n, None, first, second, steps_before_tau_0=0,
                                mcts_searches=cfg.PLAY_MCTS_SEARCHES, mcts_batch_size=cfg.PLAY_MCTS_BATCH_SIZE,
                                concurrent=min(n, 1024), seed=seed, uid_base=uid_base + lo, device=device,
                                first_player_mode=2, node_cap=node_cap)
    return Tally(*parallel.allreduce_counts(Tally.of(mine).as_tuple(), device))


def main(argv=None):
    args = parse_args(argv)
    rank, local_rank, world = parallel.init()
    device = parallel.local_device(local_rank) if world > 1 else "cuda:0"  # the engine is GPU only; --cuda is accepted as is
    game = game_provider.get_game(args)
    agents = [(path, load_checkpoint(game, path, device)) for path in args.models]

    per_agent, per_pair = {}, {}
    say = print if rank == 0 else (lambda *a, **k: None)
    pairings = list(itertools.permutations(range(len(agents)), 2))  # every ordered pair, first index outermost
    for k, (i, j) in enumerate(pairings):
        (name_i, net_i), (name_j, net_j) = agents[i], agents[j]
        started = time.time()
        tally = meet(game, net_i, net_j, args.rounds, args.seed, k * args.rounds, device, args.node_cap or None)
        say("%s vs %s -> %s" % (name_i, name_j, tally))
        if rank == 0:
            sys.stderr.write("Speed %.2f games/s\n" % (args.rounds / (time.time() - started)))
        sys.stdout.flush()
        utils.update_counts(per_agent, name_i, tally.as_tuple())
        utils.update_counts(per_agent, name_j, tally.mirrored())
        utils.update_counts(per_pair, (name_i, name_j), tally.as_tuple())

    say("Leaderboard:")
    for name, (w, l, d) in sorted(per_agent.items(), key=lambda item: item[1][0], reverse=True):  # by total wins
        say("%s: \t w=%d, l=%d, d=%d" % (name, w, l, d))
    return per_agent, per_pair


if __name__ == "__main__":
    main()
