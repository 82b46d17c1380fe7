"""Random playouts in blocks of plies, and the SHA-256 that stands for a block (SURVEY 8(c) G1 at its stated size:
10^5 connect-four plies, 10^4 each for 3x3 and 15x15 k=5, kept as one digest per 1000-ply block instead of 45 MB of
records).  Shared by tests/golden/make_golden_r5.py (which runs it on the REFERENCE's game classes) and by the tests
(oracle on the CPU, batched rule kernels on the GPU): test infrastructure, no product code.

A block is self-contained: block b of a game kind draws from `np.random.default_rng([seed, b])`, starts at the initial
state and restarts a game whenever one ends (win, or no legal move left); the last game of a block is cut off at the
block's length.  Per ply the digest absorbs, in this order:
    str(next_state)  |  b"1" / b"0" (won)  |  packbits(legal-move mask of the state BEFORE the move, A bits)
    |  packbits(planes of next_state seen by the player to move next, 2*H*W bits)
"""
import hashlib

import numpy as np


def absorb(h, s2, won, legal, A, planes):
    h.update(str(int(s2)).encode())
    h.update(b"1" if won else b"0")
    mask = np.zeros(A, np.uint8)
    mask[np.asarray(legal, dtype=np.int64)] = 1
    h.update(np.packbits(mask).tobytes())
    h.update(np.packbits(np.asarray(planes).astype(np.uint8).reshape(-1)).tobytes())


def playout_block(game, seed, block, n_plies, A, sink):
    """game: anything with initial_state / possible_moves(s) / move(s, m, p) -> (s2, won) (the reference's BaseGame,
    this repo's shim, or the oracle).  Calls sink(s, legal, m, p, s2, won) once per ply; returns (wins, draws)."""
    rng = np.random.default_rng([seed, block])
    wins = draws = 0
    s, p = game.initial_state, int(rng.integers(2))
    done = 0
    while done < n_plies:
        legal = [int(x) for x in game.possible_moves(s)]
        if not legal:
            draws += 1
            s, p = game.initial_state, int(rng.integers(2))
            continue
        m = legal[int(rng.integers(len(legal)))]
        s2, won = game.move(s, m, p)
        sink(s, legal, m, p, s2, bool(won))
        done += 1
        if won:
            wins += 1
            s, p = game.initial_state, int(rng.integers(2))
        else:
            s, p = s2, 1 - p
    return wins, draws


def block_digest(game, seed, block, n_plies, A):
    """digest of a block with every quantity taken from `game` itself (states_to_training_batch for the planes)"""
    h = hashlib.sha256()

    def sink(s, legal, m, p, s2, won):
        absorb(h, s2, won, legal, A, game.states_to_training_batch([s2], [1 - p])[0])

    wins, draws = playout_block(game, seed, block, n_plies, A, sink)
    return {"sha256": h.hexdigest(), "wins": wins, "draws": draws}


def helpers_digest(helpers, n, seed, cases):
    """SHA-256 over what the six line helpers of the m,n,k game return on `cases` random n x n boards and cells
    (tests/golden/make_golden_r5_helpers.py runs it on the reference's module, the test on this package's)"""
    rng = np.random.default_rng([seed, n])
    h = hashlib.sha256()
    for _ in range(cases):
        m = rng.integers(0, 3, size=(n, n)).tolist()  # tokens 0 / 1, 2 = empty
        c = (int(rng.integers(n)), int(rng.integers(n)))
        lines = [f(m, c) for f in (helpers.get_row, helpers.get_col, helpers.get_diag, helpers.get_antidiag)]
        h.update(repr(lines).encode())
        for k in (3, 4, 5):
            for tok in (0, 1):
                h.update(bytes([int(helpers.k_in_a_row(ln, k, tok)) for ln in lines]))
                h.update(bytes([int(helpers.check_win(m, c, k, tok))]))
    return h.hexdigest()


def codec_digest(game, seed, cases):
    """SHA-256 over the list / matrix views of `cases` random positions of `game` through the codec helpers a caller of
    the reference can reach: connect four -- decode_binary, convert_mcts_state_to_nn_state, int_to_bits, bits_to_int,
    encode_lists; m,n,k -- convert_mcts_state_to_list_state, encode_game_state, flatten_nested_list, _pad_mcts_state"""
    rng = np.random.default_rng([seed, game.action_space])
    h = hashlib.sha256()
    c4 = hasattr(game, "decode_binary")
    for _ in range(cases):
        s, p = game.initial_state, 0
        for _ply in range(int(rng.integers(0, min(30, game.action_space * 3)))):
            legal = game.possible_moves(s)
            if not legal:
                break
            s, won = game.move(s, int(legal[int(rng.integers(len(legal)))]), p)
            p = 1 - p
            if won:
                break
        if c4:
            cols = game.decode_binary(s)
            bits = game.int_to_bits(s, 63)
            assert game.encode_lists(cols) == s and game.bits_to_int(bits) == s
            h.update(repr((cols, game.convert_mcts_state_to_nn_state(s), bits, game.int_to_bits(s, 3))).encode())
        else:
            m = game.convert_mcts_state_to_list_state(s)
            assert game.encode_game_state(m) == s
            h.update(repr((m, game.flatten_nested_list(m), game._pad_mcts_state(str(s)))).encode())
    return h.hexdigest()
