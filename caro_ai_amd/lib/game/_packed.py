"""Shared implementation of the two shipped games on top of the C-ABI's
host-side single-state helpers (caro_host_*, compiled from the same
caro_rules.h as the HIP kernels)."""
import ctypes as C

import numpy as np

from caro_ai_amd import _lib
from caro_ai_amd.lib.game.game import BaseGame


class PackedGame(BaseGame):
    kind = None
    n = 0
    k = 0

    def _setup(self):
        L = _lib.load()
        self._L = L
        self.key_words = L.caro_key_words(self.kind, self.n)
        self._A = L.caro_action_space(self.kind, self.n)
        self._cells = L.caro_obs_cells(self.kind, self.n)

    # --- packed form ---
    def to_key(self, state_int) -> np.ndarray:
        raise NotImplementedError

    def from_key(self, key) -> int:
        raise NotImplementedError

    def to_keys(self, states) -> np.ndarray:
        out = np.empty((len(states), self.key_words), dtype=np.uint64)
        for i, s in enumerate(states):
            out[i] = self.to_key(s)
        return out

    def from_keys(self, keys):
        keys = np.asarray(keys, dtype=np.uint64).reshape(-1, self.key_words)
        return [self.from_key(k) for k in keys]

    # --- BaseGame ---
    @property
    def initial_state(self) -> int:
        key = np.zeros(self.key_words, dtype=np.uint64)
        _lib.check(self._L.caro_host_initial(self.kind, self.n, self.k, key.ctypes.data))
        return self.from_key(key)

    @property
    def action_space(self) -> int:
        return self._A

    def _legal_mask(self, mcts_state):
        key = np.ascontiguousarray(self.to_key(mcts_state))
        legal = np.zeros(self._A, dtype=np.uint8)
        _lib.check(self._L.caro_host_legal(self.kind, self.n, self.k, key.ctypes.data, legal.ctypes.data))
        return legal

    def possible_moves(self, mcts_state):
        return np.flatnonzero(self._legal_mask(mcts_state)).tolist()

    def invalid_moves(self, mcts_state):
        return np.flatnonzero(self._legal_mask(mcts_state) == 0).tolist()

    def states_to_training_batch(self, state_ints, who_moves_lists):
        out = np.zeros((len(state_ints),) + tuple(self.obs_shape), dtype=np.float32)
        for i, (s, w) in enumerate(zip(state_ints, who_moves_lists)):
            key = np.ascontiguousarray(self.to_key(s))
            _lib.check(self._L.caro_host_encode(self.kind, self.n, self.k, key.ctypes.data, int(w),
                                                out[i].ctypes.data))
        return out

    def _move_key(self, mcts_state, move, player):
        key = np.ascontiguousarray(self.to_key(mcts_state))
        won = C.c_int(0)
        rc = self._L.caro_host_move(self.kind, self.n, self.k, key.ctypes.data, int(move), int(player),
                                    C.addressof(won))
        assert rc == 0, self._L.caro_last_error().decode()  # the reference asserts on bad moves too
        return self.from_key(key), bool(won.value)
