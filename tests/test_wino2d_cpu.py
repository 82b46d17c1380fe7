"""2-D Winograd F(2x2,3x3) (caro_ai_amd/net_hip.py pack_net_w2, csrc/caro_net.hip trunk_w2d): the arithmetic of the
kernel restated in numpy ON THE PACKED WEIGHT IMAGE -- the chunk / row / slot a lane of the kernel reads, the column
combination and row transform of its operand stream, the two-phase fold and the partner exchange -- against
torch's conv2d in float64.  CPU only: pins the packer and the algebra; the kernel itself is tests/test_gpu_net.py."""
import numpy as np
import pytest
import torch

from caro_ai_amd.lib.model import Net, _fold
from caro_ai_amd.net_hip import W2_PHASE_B, pack_net_w2, wino2d_weights


def _image_tap(img, layer, phase, bh, a, co, ci):
    """weight U[a][b(phase, bh)][co][ci] the way trunk_w2d finds it in the packed image"""
    h, rem = divmod(ci, 32)
    G, j = divmod(rem, 4)
    cq, gi = divmod(G, 2)
    slot = gi ^ ((co >> 3) & 1)
    return img[layer, phase * 4 + cq, bh, a, h, co, slot * 4 + j]


@pytest.mark.parametrize("n", [15, 13])
def test_packed_image_and_two_phase_fold_reproduce_the_convolution(n):
    torch.manual_seed(n)
    net = Net((2, n, n), n * n).eval()
    img = pack_net_w2(net).reshape(5, 8, 2, 4, 2, 64, 8).astype(np.float64)
    layer = 2
    w, _ = _fold(list(net.residual_blocks())[layer])
    w = w.double()
    # the image holds U = G w G^T for the (phase, bh) -> b map of the kernel
    u = wino2d_weights(w.numpy())
    rng = np.random.default_rng(n)
    for _ in range(200):
        phase, bh, a, co, ci = (int(rng.integers(k)) for k in (2, 2, 4, 64, 64))
        want = np.float32(u[a, W2_PHASE_B[phase][bh], co, ci])
        assert _image_tap(img, layer, phase, bh, a, co, ci) == want
    # one layer on a random activation, the kernel's way
    x = torch.randn(1, 64, n, n, dtype=torch.float64)
    ref = torch.nn.functional.conv2d(x, w, padding=1)[0].numpy()          # [co, y, x]
    xp = np.zeros((64, n + 3, n + 3))
    xp[:, 1:n + 1, 1:n + 1] = x[0].numpy()                                 # xp[:, y + 1, x + 1] = cell (y, x)
    T = (n + 1) // 2
    out = np.zeros((64, 2 * T, 2 * T))
    cols = {(0, 0): (1, 2, 1.0), (0, 1): (0, 2, -1.0), (1, 0): (2, 1, -1.0), (1, 1): (1, 3, -1.0)}  # (bh, phase) -> jA, jB, sg
    U = np.zeros((2, 2, 4, 64, 64))  # [phase][bh][a][co][ci] read back from the image
    for phase in range(2):
        for bh in range(2):
            for a in range(4):
                for ci in range(64):
                    h, rem = divmod(ci, 32)
                    G, j = divmod(rem, 4)
                    cq, gi = divmod(G, 2)
                    for sl in range(2):
                        m = (gi ^ ((np.arange(64) >> 3) & 1)) == sl
                        U[phase, bh, a, m, ci] = img[layer, phase * 4 + cq, bh, a, h, m, sl * 4 + j]
    for ty in range(T):
        for tx in range(T):
            d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]               # d[ci][r][j] = cell (2ty-1+r, 2tx-1+j)
            Z = {}
            for bh in range(2):
                for phase in range(2):
                    jA, jB, sg = cols[(bh, phase)]
                    c = d[:, :, jA] + sg * d[:, :, jB]                      # [ci][r]
                    V = np.stack([c[:, 0] - c[:, 2], c[:, 1] + c[:, 2], c[:, 2] - c[:, 1], c[:, 1] - c[:, 3]])  # [a][ci]
                    M = np.einsum("aoi,ai->ao", U[phase, bh], V)            # [a][co]
                    Z[(bh, phase)] = ((M[0] + M[1]) + M[2], (M[1] - M[2]) - M[3])
            for u_ in range(2):
                # bh 0 finishes column v = 0 with the partner's phase-0 fold, bh 1 column v = 1
                out[:, 2 * ty + u_, 2 * tx] = (Z[(0, 1)][u_] + Z[(0, 0)][u_]) + Z[(1, 0)][u_]
                out[:, 2 * ty + u_, 2 * tx + 1] = (Z[(0, 0)][u_] - Z[(1, 0)][u_]) - Z[(1, 1)][u_]
    err = np.abs(out[:, :n, :n] - ref).max()
    assert err < 5e-6 * max(1.0, np.abs(ref).max()), err  # float32-rounded U against float64 weights


def test_lane_to_tile_map_covers_the_board_once_and_read_groups_are_conflict_free():
    """the lane -> tile map of trunk_w2d and the swizzle key akey<true>: every ds_read_b128 lane group
    ({0-3,12-15,20-27} / {4-11,16-19,28-31} per half, MI355X_MICROARCH.md) reads 16 different 16-byte slots for any
    of the 16 cell offsets of a tile, and the 64 tiles of a board are covered exactly once"""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
              [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    seen = set()
    for rt in range(2):
        tiles = {}
        for i in range(32):
            quad = i >> 2
            grp = (0x96 >> quad) & 1
            qq = ((quad >> 1) << 2) | (i & 3)
            tiles[i] = (rt * 4 + grp * 2 + (qq >> 3), qq & 7)
            seen.add(tiles[i])
        for g in groups:
            for r in range(4):
                for j in range(4):
                    keys = {((tiles[i][1] + (j >> 1)) & 7) | (((tiles[i][0] + (r >> 1)) & 1) << 3) for i in g}
                    assert len(keys) == 16, (rt, r, j)
    assert seen == {(ty, tx) for ty in range(8) for tx in range(8)}
    # the stored key of a real cell equals the key the tile-relative formula uses
    for ty in range(8):
        for tx in range(8):
            for r in range(4):
                for j in range(4):
                    y, x = 2 * ty - 1 + r, 2 * tx - 1 + j
                    if 0 <= y < 15 and 0 <= x < 15:
                        stored = (((x + 1) >> 1) & 7) | ((((y + 1) >> 1) & 1) << 3)
                        assert stored == ((tx + (j >> 1)) & 7) | (((ty + (r >> 1)) & 1) << 3)
    # weight rows: [co][2 slots of 16 B], slot = gi ^ (co >> 3 & 1): a 256-byte bank row holds 8 co x 2 slots
    for g in groups:
        for gi in range(2):
            slots = {((co % 8) * 2 + (gi ^ ((co >> 3) & 1))) for co in g}
            assert len(slots) == 16
