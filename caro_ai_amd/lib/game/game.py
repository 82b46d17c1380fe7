"""Game plugin interface of the self-play path.

Same eight members, names and argument meaning as the reference's `BaseGame`
(lib/game/game.py:9-120), so games written against the reference plug in
unchanged.  The two shipped games add `kind`, `key_words`, `to_key` and
`from_key`: the packed board form the HIP engine works on (include/caro_hip.h).
"""
from abc import ABC, abstractmethod
from typing import List, Tuple

import numpy as np


class BaseGame(ABC):
    @property
    @abstractmethod
    def initial_state(self) -> int:
        """State of the empty board, in MCTS (int) form."""

    @property
    @abstractmethod
    def obs_shape(self) -> Tuple[int, ...]:
        """Shape of one network input, (2, H, W)."""

    @property
    @abstractmethod
    def action_space(self) -> int:
        """Number of actions, legal or not."""

    @abstractmethod
    def possible_moves(self, mcts_state: int) -> List:
        """Legal actions in ascending order."""

    @abstractmethod
    def invalid_moves(self, mcts_state: int) -> List:
        """Illegal actions."""

    @abstractmethod
    def states_to_training_batch(self, state_lists: List, who_moves_lists: List[int]) -> np.ndarray:
        """float32[L, 2, H, W]: plane 0 = tokens of the player to move, plane 1 = the opponent's."""

    @abstractmethod
    def move(self, mcts_state: int, move: int, player: int) -> Tuple[int, bool]:
        """(new state, whether `player` just won)."""

    @abstractmethod
    def render(self, mcts_state: int) -> str:
        """Human-readable board."""
