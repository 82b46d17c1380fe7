"""Connect four, 6 x 7, behind the reference's `ConnectFour` interface
(lib/game/connect_four/connect_four.py:9-281).

The MCTS state int is the reference's own 63-bit word (42 cell bits column by
column from the bottom, then seven 3-bit free-slot counts, :36-56); the engine
computes directly on it, so `to_key` is the identity."""
import numpy as np

from caro_ai_amd import _lib
from caro_ai_amd.lib.game._packed import PackedGame


class ConnectFour(PackedGame):
    kind = _lib.GAME_CONNECT4

    def __init__(self):
        super().__init__()
        self.game_rows = 6
        self.game_cols = 7
        self.bits_in_len = 3
        self.player_black = 1
        self.player_white = 0
        self.count_to_win = 4
        self._setup()

    @property
    def obs_shape(self):
        return (2, self.game_rows, self.game_cols)

    def to_key(self, state_int):
        assert isinstance(state_int, (int, np.integer))
        return np.array([int(state_int)], dtype=np.uint64)

    def from_key(self, key):
        return int(np.asarray(key, dtype=np.uint64).reshape(-1)[0])

    def to_keys(self, states):
        return np.array([int(s) for s in states], dtype=np.uint64).reshape(-1, 1)

    def from_keys(self, keys):
        return np.asarray(keys, dtype=np.uint64).reshape(-1).tolist()

    # list form <-> int, for callers that used the reference's codec (:94-155)
    @staticmethod
    def bits_to_int(bits):
        """most significant bit first (:95-100)"""
        value = 0
        for bit in bits:
            value = (value << 1) | int(bit)
        return value

    @staticmethod
    def int_to_bits(num, bits):
        """the low `bits` bits of num, most significant first (:103-108)"""
        return [(int(num) >> shift) & 1 for shift in range(bits - 1, -1, -1)]

    def convert_mcts_state_to_nn_state(self, mcts_state):
        """the list view of a state (:149-155)"""
        return self.decode_binary(mcts_state)

    def decode_binary(self, state_int):
        assert isinstance(state_int, int)
        cols = []
        for c in range(self.game_cols):
            free = (state_int >> (3 * (6 - c))) & 7
            cols.append([(state_int >> (62 - (6 * c + r))) & 1 for r in range(self.game_rows - free)])
        return cols

    def encode_lists(self, field_lists):
        assert isinstance(field_lists, list)
        assert len(field_lists) == self.game_cols
        s = 0
        for c, col in enumerate(field_lists):
            for r, tok in enumerate(col):
                s |= int(tok) << (62 - (6 * c + r))
            s |= (self.game_rows - len(col)) << (3 * (6 - c))
        return s

    def move(self, state_int, col, player):
        assert isinstance(state_int, int)
        assert isinstance(col, int)  # (the reference refuses numpy integers too, connect_four.py:250-255)
        assert 0 <= col < self.game_cols
        assert player == self.player_black or player == self.player_white
        return self._move_key(state_int, col, player)

    def possible_moves(self, state_int):
        assert isinstance(state_int, int)
        return super().possible_moves(state_int)

    def render(self, state_int):
        rows = [[" "] * self.game_cols for _ in range(self.game_rows)]
        for c, col in enumerate(self.decode_binary(state_int)):
            for r, tok in enumerate(col):
                rows[self.game_rows - 1 - r][c] = "X" if tok else "O"
        body = "\n".join("".join(r) for r in rows)
        return "0123456\n-------\n" + body + "\n-------\n0123456"
