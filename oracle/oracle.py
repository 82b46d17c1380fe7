"""ctypes wrapper of oracle/caro_oracle.c -- TEST INFRASTRUCTURE ONLY.

May be imported from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, and from nowhere else (nothing under caro_ai_amd/ does).

States cross this wrapper in the reference's own form (Python ints:
connect_four.py:36-56 bit layout, tictactoe.py:14-24 digit string); inside the
C file they are cell arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcaro_oracle.so")

NET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                     C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float))


def build(force=False):
    src = os.path.join(_HERE, "caro_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "caro_noise.h")
    if (force or not os.path.exists(_SO)
            or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        # on the GPU box the prebuilt .so travels with the snapshot; rebuild only if absent/stale
        build()
        L = C.CDLL(_SO)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_clear.argtypes = [C.c_void_p]
        L.oracle_set_net.argtypes = [C.c_void_p, C.c_int, NET_FN, C.c_void_p]
        L.oracle_use_synth_net.argtypes = [C.c_void_p]
        L.oracle_use_synth_nets.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.oracle_set_noise_table.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.oracle_set_uniform_table.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        L.oracle_set_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.oracle_action_space.argtypes = [C.c_void_p]
        L.oracle_store_len.argtypes = [C.c_void_p, C.c_int]
        L.oracle_noise_pos.argtypes = [C.c_void_p]
        L.oracle_noise_pos.restype = C.c_long
        L.oracle_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_initial_state.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_move.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_possible_moves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_encode_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_c4_cells_to_int.argtypes = [C.c_void_p]
        L.oracle_c4_cells_to_int.restype = C.c_uint64
        L.oracle_c4_int_to_cells.argtypes = [C.c_uint64, C.c_void_p]
        L.oracle_search_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_int, C.c_uint32]
        L.oracle_get_node.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.oracle_get_policy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_poke_node.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.oracle_backup.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_play_game.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9
        L.oracle_noise_row.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_void_p]
        L.oracle_move_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.oracle_move_uniform.restype = C.c_double
        L.oracle_sample_index.argtypes = [C.c_void_p, C.c_int, C.c_double]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def noise_row(seed, uid, ply, sim, A, alpha=0.3):
    out = np.empty(A, dtype=np.float64)
    lib().oracle_noise_row(seed, uid, ply, sim, A, alpha, _ptr(out))
    return out


def move_uniform(seed, uid, ply):
    return lib().oracle_move_uniform(seed, uid, ply)


def sample_index(pi, u):
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    return lib().oracle_sample_index(_ptr(pi), len(pi), float(u))


class Oracle:
    """One game kind + up to two MCTS stores + play_game, reference semantics."""

    C4, MNK = 0, 1

    def __init__(self, kind, n=3, k=3, n_stores=1, c_puct=1.0, alpha=0.30, explore=0.25):
        self.L = lib()
        self.kind, self.n, self.k = kind, n, k
        self.h = self.L.oracle_create(kind, n, k, n_stores, c_puct, alpha, explore)
        self.A = self.L.oracle_action_space(self.h)
        self.rows, self.cols = (6, 7) if kind == 0 else (n, n)
        self.hw = self.rows * self.cols
        self._keep = []  # keep callbacks / tables alive

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    # ---- state conversion (reference int <-> cells) ----
    def to_cells(self, state_int):
        cells = np.empty(self.hw, dtype=np.uint8)
        if self.kind == 0:
            self.L.oracle_c4_int_to_cells(int(state_int), _ptr(cells))
        else:
            s = str(int(state_int)).rjust(self.hw, "0")  # tictactoe.py:88-100
            cells[:] = np.frombuffer(s.encode(), dtype=np.uint8) - ord("0")
        return cells

    def to_int(self, cells):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        if self.kind == 0:
            return int(self.L.oracle_c4_cells_to_int(_ptr(cells)))
        return int("".join(str(int(c)) for c in cells))  # tictactoe.py:102-115

    # ---- rules ----
    @property
    def initial_state(self):
        cells = np.empty(self.hw, dtype=np.uint8)
        self.L.oracle_initial_state(self.h, _ptr(cells))
        return self.to_int(cells)

    def move(self, state_int, move, player):
        cells = self.to_cells(state_int)
        won = self.L.oracle_move(self.h, _ptr(cells), int(move), int(player))
        if won < 0:
            raise AssertionError("illegal move")
        return self.to_int(cells), bool(won)

    def possible_moves(self, state_int):
        cells = self.to_cells(state_int)
        out = np.empty(self.A, dtype=np.int32)
        n = self.L.oracle_possible_moves(self.h, _ptr(cells), _ptr(out))
        return out[:n].tolist()

    def states_to_training_batch(self, states, who_moves):
        out = np.zeros((len(states), 2, self.rows, self.cols), dtype=np.float32)
        for i, (s, w) in enumerate(zip(states, who_moves)):
            cells = self.to_cells(s)
            self.L.oracle_encode_planes(self.h, _ptr(cells), int(w), _ptr(out[i]))
        return out

    # ---- nets ----
    def use_synth_net(self, salt0=0, salt1=None):
        """the table net (salt 0: the one the reference-recorded vectors use); two salts = two different nets"""
        if salt0 == 0 and salt1 in (None, 0):
            self.L.oracle_use_synth_net(self.h)
        else:
            self.L.oracle_use_synth_nets(self.h, salt0, salt0 if salt1 is None else salt1)

    def set_net(self, which, fn):
        """fn(planes float32[L,2,H,W], states list[int], players int32[L]) -> (P float32[L,A], v float32[L])"""
        A, hw = self.A, self.hw

        def _cb(ctx, L, planes, cells, players, P, v):
            pl = np.ctypeslib.as_array(planes, shape=(L, 2, self.rows, self.cols))
            ce = np.ctypeslib.as_array(cells, shape=(L, hw))
            py = np.ctypeslib.as_array(players, shape=(L,))
            states = [self.to_int(ce[i]) for i in range(L)] if getattr(fn, "wants_states", False) else None
            p_out, v_out = fn(pl, states, py)
            np.ctypeslib.as_array(P, shape=(L, A))[:] = np.asarray(p_out, dtype=np.float32)
            np.ctypeslib.as_array(v, shape=(L,))[:] = np.asarray(v_out, dtype=np.float32).reshape(L)

        cb = NET_FN(_cb)
        self._keep.append(cb)
        self.L.oracle_set_net(self.h, which, cb, None)

    # ---- random inputs ----
    def set_noise_table(self, table):
        if table is None:
            self.L.oracle_set_noise_table(self.h, None, 0)
            return
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1, self.A)
        self._keep.append(t)
        self.L.oracle_set_noise_table(self.h, _ptr(t), t.shape[0])

    def set_uniform_table(self, table):
        if table is None:
            self.L.oracle_set_uniform_table(self.h, None, 0)
            return
        t = np.ascontiguousarray(table, dtype=np.float64).reshape(-1)
        self._keep.append(t)
        self.L.oracle_set_uniform_table(self.h, _ptr(t), t.shape[0])

    def set_stream(self, seed, game_uid):
        self.L.oracle_set_stream(self.h, seed, game_uid)

    # ---- search ----
    def clear(self):
        self.L.oracle_clear(self.h)

    def store_len(self, store=0):
        return self.L.oracle_store_len(self.h, store)

    def noise_pos(self):
        return self.L.oracle_noise_pos(self.h)

    def counters(self):
        out = np.zeros(7, dtype=np.int64)
        self.L.oracle_counters(self.h, _ptr(out))
        return dict(zip(["sims", "levels", "expansions", "terminals", "dropped", "net_calls", "net_rows"],
                        out.tolist()))

    def search_batch(self, count, batch_size, state_int, player, store=0, which_net=0, ply=0):
        cells = self.to_cells(state_int)
        self.L.oracle_search_batch(self.h, store, which_net, count, batch_size, _ptr(cells), int(player), ply)

    def get_node(self, state_int, store=0):
        cells = self.to_cells(state_int)
        N = np.zeros(self.A, np.int32); W = np.zeros(self.A); Wf = np.zeros(self.A, np.int32)
        Q = np.zeros(self.A); P = np.zeros(self.A, np.float32)
        ok = self.L.oracle_get_node(self.h, store, _ptr(cells), _ptr(N), _ptr(W), _ptr(Wf), _ptr(Q), _ptr(P))
        if not ok:
            return None
        return {"N": N, "W": W, "W_is_f32": Wf, "Q": Q, "P": P}

    def get_policy(self, state_int, tau=1, store=0):
        cells = self.to_cells(state_int)
        out = np.zeros(self.A)
        self.L.oracle_get_policy(self.h, store, _ptr(cells), int(tau), _ptr(out))
        return out

    def poke_node(self, state_cells, N, W, Q, P, store=0):
        cells = np.ascontiguousarray(state_cells, dtype=np.uint8)
        self.L.oracle_poke_node(self.h, store, _ptr(cells), _ptr(np.asarray(N, np.int32)),
                                _ptr(np.asarray(W, np.float64)), _ptr(np.asarray(Q, np.float64)),
                                _ptr(np.asarray(P, np.float32)))

    def backup(self, value, path_cells, actions, store=0, value_is_f32=False):
        pc = np.ascontiguousarray(path_cells, dtype=np.uint8).reshape(len(actions), self.hw)
        ac = np.asarray(actions, np.int32)
        self.L.oracle_backup(self.h, store, float(value), int(value_is_f32), len(actions), _ptr(pc), _ptr(ac))

    def get_node_cells(self, cells, store=0):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        N = np.zeros(self.A, np.int32); W = np.zeros(self.A); Wf = np.zeros(self.A, np.int32)
        Q = np.zeros(self.A); P = np.zeros(self.A, np.float32)
        ok = self.L.oracle_get_node(self.h, store, _ptr(cells), _ptr(N), _ptr(W), _ptr(Wf), _ptr(Q), _ptr(P))
        return {"N": N, "W": W, "W_is_f32": Wf, "Q": Q, "P": P} if ok else None

    def play_game(self, steps_before_tau_0, searches, batch_size, first_player, max_plies=None):
        """Returns dict(result, steps, plies, states, players, pi, actions, rootN, nodes, z)."""
        mp = max_plies or (self.hw + 1)
        cells = np.zeros((mp, self.hw), np.uint8); player = np.zeros(mp, np.int32)
        pi = np.zeros((mp, self.A)); action = np.zeros(mp, np.int32)
        rootN = np.zeros((mp, self.A), np.int32); nodes = np.zeros(mp, np.int32); z = np.zeros(mp, np.int32)
        steps = C.c_int32(0); plies = C.c_int32(0)
        r = self.L.oracle_play_game(self.h, steps_before_tau_0, searches, batch_size, int(first_player), mp,
                                    _ptr(cells), _ptr(player), _ptr(pi), _ptr(action), _ptr(rootN), _ptr(nodes),
                                    _ptr(z), C.byref(steps), C.byref(plies))
        n = plies.value
        return {"result": r, "steps": steps.value, "plies": n,
                "states": [self.to_int(cells[i]) for i in range(n)], "cells": cells[:n],
                "players": player[:n], "pi": pi[:n], "actions": action[:n], "rootN": rootN[:n],
                "nodes": nodes[:n], "z": z[:n]}
