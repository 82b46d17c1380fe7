"""One interactive game of a human against a checkpoint: the object the reference's chat front end drives
(lib/play_session.py:6-49 -- same constructor arguments, attributes and method names).

The bot side is `MCTS.search_batch(BOT_MCTS_SEARCHES, BOT_MCTS_BATCH_SIZE)` on a tree that lives on the GPU and
persists for the whole game, followed by the tau = 0 policy; the move is drawn with numpy from that one-hot
policy, which is what keeps a seeded numpy stream in step with the reference."""
import numpy as np
import torch

from caro_ai_amd import config as cfg
from caro_ai_amd.lib import mcts, model


class Session:
    def __init__(self, game, model_file, player_moves_first, device="cuda:0"):
        self.game = game
        self.BOT_PLAYER, self.USER_PLAYER = game.player_black, game.player_white
        self.model_file = model_file
        self.player_moves_first = player_moves_first
        self.device = device
        weights = torch.load(model_file, map_location=lambda storage, loc: storage)
        self.model = model.Net(input_shape=game.obs_shape, actions_n=game.action_space)
        self.model.load_state_dict(weights)
        self.model.to(device).eval()
        self.mcts_store = mcts.MCTS(game, tree_device=device)
        self.state = game.initial_state
        self.moves = []     # every move of the game, both sides, in order
        self.value = None   # the bot's own estimate of its last move

    def _play(self, move, who) -> bool:
        """put `who`'s token, remember the move; True when it wins the game"""
        self.moves.append(move)
        self.state, won = self.game.move(self.state, move, who)
        return won

    def move_player(self, move: int) -> bool:
        return self._play(move, self.USER_PLAYER)

    def move_bot(self) -> bool:
        tree = self.mcts_store
        tree.search_batch(cfg.BOT_MCTS_SEARCHES, cfg.BOT_MCTS_BATCH_SIZE, self.state, self.BOT_PLAYER, self.model,
                          device=self.device)
        policy, q_values = tree.get_policy_value(self.state, tau=0)
        move = int(np.random.choice(self.game.action_space, p=policy))
        self.value = q_values[move]
        return self._play(move, self.BOT_PLAYER)

    def is_valid_move(self, move: int) -> bool:
        return move in self.game.possible_moves(self.state)

    def is_draw(self) -> bool:
        return not self.game.possible_moves(self.state)

    def render(self) -> str:
        head = "" if self.value is None else "Position evaluation: %.2f\n" % float(self.value)
        return "%s<pre>%s</pre>" % (head, self.game.render(self.state))
