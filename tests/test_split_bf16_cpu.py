"""The bf16x3 split-operand form (caro_ai_amd/net_hip.py split_bf16x3 / pack_net_x3, csrc/caro_net.hip k_net_forward_x3)
on the CPU: the decomposition float32 = hi + mid + lo in bfloat16 parts, the packed weight image where a lane of the
kernel looks for it, and the arithmetic of the six part products restated in numpy against the float32 convolution sum.
Pins the packer and the algebra; the kernel itself is tests/test_gpu_net.py."""
import numpy as np
import torch

from caro_ai_amd.lib.model import Net, _fold
from caro_ai_amd.net_hip import bf16_round, bf16_value, pack_net_x3, split_bf16x3


def test_bf16_rounding_is_to_nearest_even():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -8, -1.0 - 2.0 ** -8, 0.0, 3.0e-30, 65504.0],
                 np.float32)
    b = bf16_value(bf16_round(x))
    # ties go to the even mantissa: 1 + 2^-8 -> 1, 1 + 3 * 2^-8 -> 1 + 2^-6
    assert b[0] == 1.0 and b[1] == 1.0 and b[2] == np.float32(1.0 + 2.0 ** -7) and b[3] == np.float32(1.0 + 2.0 ** -6)
    assert b[4] == -1.0 and b[5] == 0.0
    assert np.all(np.abs(b[6:] - x[6:]) <= np.abs(x[6:]) * 2.0 ** -8)
    # same bits as torch's own conversion on a million random values
    g = torch.Generator().manual_seed(1)
    r = (torch.randn(1 << 20, generator=g) * torch.exp(4 * torch.randn(1 << 20, generator=g))).float()
    assert np.array_equal(bf16_value(bf16_round(r.numpy())), r.bfloat16().float().numpy())


def test_three_parts_are_an_exact_decomposition():
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(1 << 18, generator=g) * torch.exp(3 * torch.randn(1 << 18, generator=g))).float().numpy()
    hi, mid, lo = split_bf16x3(x)
    back = (bf16_value(hi).astype(np.float64) + bf16_value(mid)) + bf16_value(lo)
    # 8 + 8 + 8 significant bits with signs: what is left is below 2^-24 of |x| (mostly nothing at all)
    assert np.all(np.abs(back - x) <= np.abs(x) * 2.0 ** -24)
    assert (back.astype(np.float32) == x).mean() > 0.99
    assert np.all(np.abs(bf16_value(mid)) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(bf16_value(lo)) <= np.abs(x) * 2.0 ** -16)


def _image_weight(img, layer, tap, part, co, ci):
    """part `part` of w[co][ci] of (layer, tap) where a lane of k_net_forward_x3 finds it: half c = ci / 32, then
    [part][kg = (ci % 32) / 8][co][ci % 8]"""
    c, rem = divmod(ci, 32)
    kg, j = divmod(rem, 8)
    return img[layer, tap, c, part, kg, co, j]


def test_packed_image_holds_the_folded_weights_and_six_products_reproduce_the_sum():
    torch.manual_seed(5)
    net = Net((2, 6, 7), 7).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.5, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    img = pack_net_x3(net)
    assert img.dtype == np.uint16 and img.size == 45 * 2 * 3 * 4 * 64 * 8
    img = img.reshape(5, 9, 2, 3, 4, 64, 8)
    rng = np.random.default_rng(0)
    for layer in (0, 4):
        w, _ = _fold(list(net.residual_blocks())[layer])
        w = w.detach().numpy().astype(np.float32)  # [co, ci, ky, kx]
        for tap in (0, 4, 8):
            for co, ci in ((0, 0), (63, 63), (17, 40), (32, 31), (5, 32)):
                parts = [bf16_value(np.array([_image_weight(img, layer, tap, p_, co, ci)], np.uint16))[0] for p_ in range(3)]
                assert np.float32((np.float64(parts[0]) + parts[1]) + parts[2]) == w[co, ci, tap // 3, tap % 3]
        # one output of the layer: the sum over 9 taps x 64 channels of a * w from the six part products of weight
        # >= 2^-16, accumulated in float64 here (the kernel: float32 MFMA accumulation) -- against the exact sum
        a = (rng.standard_normal((9, 64)) * np.exp(rng.standard_normal((9, 64)))).astype(np.float32)
        ah, am, al = (bf16_value(p_).astype(np.float64) for p_ in split_bf16x3(a))
        co = 11
        wt = np.stack([w[co, :, t // 3, t % 3] for t in range(9)])  # [tap, ci]
        wh, wm, wl = (bf16_value(p_).astype(np.float64) for p_ in split_bf16x3(wt))
        six = (am * wm + ah * wl + al * wh + ah * wm + am * wh + ah * wh).sum()
        exact = (a.astype(np.float64) * wt.astype(np.float64)).sum()
        scale = np.abs(a.astype(np.float64) * wt.astype(np.float64)).sum()
        assert abs(six - exact) <= scale * 3 * 2.0 ** -24, (six, exact)
