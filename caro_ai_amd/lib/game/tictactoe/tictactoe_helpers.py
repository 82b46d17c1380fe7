"""Line helpers of the m,n,k game on the matrix form of a board, with the names, arguments and results of the
reference's lib/game/tictactoe/tictactoe_helpers.py:7-179 (a caller of the reference may import them from here).

The engine does not use them: on the GPU the four lines through a move are gathered with one ballot and a run of k is
found with k - 1 shift-ANDs (caro_ai_amd/csrc/caro_rules.h, `MnkRules::move_group`).  These are the host-side
functions for code that works on lists of lists: every line is described by its starting cell and a step, and walked
with one generator.
"""
from typing import Iterator, List, Sequence, Tuple

Matrix = List[List[int]]
Coord = Tuple[int, int]


def _walk(matrix: Matrix, row: int, col: int, d_row: int, d_col: int) -> Iterator[int]:
    """cells from (row, col) in steps of (d_row, d_col) until the board ends"""
    n_rows, n_cols = len(matrix), len(matrix[0])
    while 0 <= row < n_rows and 0 <= col < n_cols:
        yield matrix[row][col]
        row, col = row + d_row, col + d_col


def get_row(matrix: Matrix, coord: Sequence[int]) -> List[int]:
    """the whole row through `coord` = (row, column), left to right (:59-69)"""
    return list(_walk(matrix, coord[0], 0, 0, 1))


def get_col(matrix: Matrix, coord: Sequence[int]) -> List[int]:
    """the whole column through `coord`, top to bottom (:72-80)"""
    return list(_walk(matrix, 0, coord[1], 1, 0))


def get_diag(matrix: Matrix, coord: Sequence[int]) -> List[int]:
    """the diagonal through `coord` from its top-left end to its bottom-right end (:83-130)"""
    back = min(coord[0], coord[1])  # steps from the cell up-left to the board's edge
    return list(_walk(matrix, coord[0] - back, coord[1] - back, 1, 1))


def get_antidiag(matrix: Matrix, coord: Sequence[int]) -> List[int]:
    """the anti-diagonal through `coord` from its bottom-left end to its top-right end; square boards only (:133-179)"""
    assert len(matrix) == len(matrix[0]), "we only handle squares"
    back = min(len(matrix) - 1 - coord[0], coord[1])  # steps from the cell down-left to the board's edge
    return list(_walk(matrix, coord[0] + back, coord[1] - back, -1, 1))


def k_in_a_row(arr: List[int], k: int, token: int) -> bool:
    """True iff `arr` holds a run of at least k cells equal to `token` (:26-56)"""
    assert k > 1, "We do not handle trivial cases where k <= 1"
    run = 0
    for cell in arr:
        run = run + 1 if cell == token else 0
        if run >= k:
            return True
    return False


def check_win(matrix: Matrix, move: Sequence[int], k: int, token: int) -> bool:
    """True iff one of the four lines through `move` = (row, column) holds k of `token` in a row (:7-23)"""
    return any(k_in_a_row(line(matrix, move), k, token) for line in (get_row, get_col, get_diag, get_antidiag))
