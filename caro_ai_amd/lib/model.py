"""Policy/value network of the self-play path.

Mirrors the interface of the reference's lib/model.py:10-107 (`Net`,
`NetWrapper`): same constructor arguments, same forward contract
(`x[B,2,H,W] -> (policy logits [B,A], value [B,1])`) and -- the part that makes
`.dat` checkpoints interchange both ways -- the same `state_dict` key names and
tensor shapes (62 entries: conv_in / conv_1..conv_5 / conv_val / conv_policy as
`<name>.0.*` conv + `<name>.1.*` batch-norm, `value.0`, `value.2`, `policy.0`).

The arithmetic itself is PyTorch-ROCm's (MIOpen / rocBLAS); this module only
describes the architecture.  `FoldedNet` is the inference form used by the
self-play engine: eval-mode batch-norm folded into the preceding convolution,
so a leaf batch costs 8 conv + 3 GEMM launches instead of 8 conv + 8 BN.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_FILTERS = 64  # reference lib/model.py:7 (config.NUM_FILTERS is unused there too)


def _conv_block(c_in, c_out, kernel, padding):
    # index 0 = conv, 1 = batch norm, 2 = activation: the indices are part of the checkpoint format
    return nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=kernel, padding=padding),
                         nn.BatchNorm2d(c_out), nn.LeakyReLU())


class Net(nn.Module):
    N_RESIDUAL = 5

    def __init__(self, input_shape, actions_n):
        super().__init__()
        planes, height, width = input_shape
        self.input_shape = tuple(input_shape)
        self.actions_n = actions_n
        self.conv_in = _conv_block(planes, NUM_FILTERS, 3, 1)
        for i in range(1, self.N_RESIDUAL + 1):
            setattr(self, "conv_%d" % i, _conv_block(NUM_FILTERS, NUM_FILTERS, 3, 1))
        cells = height * width
        self.conv_val = _conv_block(NUM_FILTERS, 1, 1, 0)
        self.value = nn.Sequential(nn.Linear(cells, 20), nn.LeakyReLU(), nn.Linear(20, 1), nn.Tanh())
        self.conv_policy = _conv_block(NUM_FILTERS, 2, 1, 0)
        self.policy = nn.Sequential(nn.Linear(2 * cells, actions_n))

    # the reference's shape probes (lib/model.py:74-80): how many features the 1x1 heads hand to their linear layers
    # for a trunk output of `shape` = (filters, H, W).  The constructor above knows the answer (1 or 2 planes of H*W
    # cells); the probes stay for callers that ask the net itself.
    def _get_conv_val_size(self, shape):
        return int(self.conv_val[0].out_channels * shape[1] * shape[2])

    def _get_conv_policy_size(self, shape):
        return int(self.conv_policy[0].out_channels * shape[1] * shape[2])

    def residual_blocks(self):
        return [getattr(self, "conv_%d" % i) for i in range(1, self.N_RESIDUAL + 1)]

    def forward(self, x):
        n = x.shape[0]
        h = self.conv_in(x)
        for block in self.residual_blocks():
            h = h + block(h)
        val = self.value(self.conv_val(h).reshape(n, -1))
        pol = self.policy(self.conv_policy(h).reshape(n, -1))
        return pol, val


class NetWrapper:
    """`model` is trained, `target_model` is the frozen best copy (reference lib/model.py:97-107)."""

    def __init__(self, model):
        self.model = model
        self.target_model = copy.deepcopy(model)

    def sync(self):
        self.target_model.load_state_dict(self.model.state_dict())


def _fold(block):
    conv, bn = block[0], block[1]
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    w = conv.weight * scale.reshape(-1, 1, 1, 1)
    b = (conv.bias - bn.running_mean) * scale + bn.bias
    return w.detach().clone(), b.detach().clone()


class FoldedNet(nn.Module):
    """Inference-only form of a `Net` in eval mode (batch-norm folded).

    Numerically this differs from `Net.eval()` only by the re-association of
    the BN affine into the conv weights (float32 rounding, ~1e-6 relative);
    tests/test_model.py states the tolerance.
    """

    def __init__(self, net: Net, negative_slope=0.01):
        super().__init__()
        self.slope = negative_slope
        self.actions_n = net.actions_n
        blocks = [net.conv_in] + net.residual_blocks()
        self.ws = nn.ParameterList()
        self.bs = nn.ParameterList()
        for blk in blocks + [net.conv_val, net.conv_policy]:
            w, b = _fold(blk)
            self.ws.append(nn.Parameter(w, requires_grad=False))
            self.bs.append(nn.Parameter(b, requires_grad=False))
        self.value = copy.deepcopy(net.value)
        self.policy = copy.deepcopy(net.policy)
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        n = x.shape[0]
        nb = len(self.ws) - 2
        h = F.leaky_relu(F.conv2d(x, self.ws[0], self.bs[0], padding=1), self.slope)
        for i in range(1, nb):
            h = h + F.leaky_relu(F.conv2d(h, self.ws[i], self.bs[i], padding=1), self.slope)
        val = F.leaky_relu(F.conv2d(h, self.ws[nb], self.bs[nb]), self.slope)
        pol = F.leaky_relu(F.conv2d(h, self.ws[nb + 1], self.bs[nb + 1]), self.slope)
        return self.policy(pol.reshape(n, -1)), self.value(val.reshape(n, -1))


class GemmNet(nn.Module):
    """Inference form used on the leaf batch: every convolution is ONE gather +
    ONE GEMM over channels-last activations X[L*HW, C],

        patches[L*HW, 9*C] = X_padded[neighbour index]      (3x3, padding 1)
        Y[L*HW, 64]        = patches @ Wmat[9*C, 64] + b    (rocBLAS / hipBLASLt, MFMA)

    so the leaf-batch size L can change every minibatch without any kernel
    being compiled or searched for (MIOpen has no tuned/compiled database for
    gfx950 in this image and would JIT per shape).  Eval-mode batch-norm is
    folded into Wmat / b.  Same function as `Net.eval()` up to float32
    re-association (tests/test_model.py states the tolerance).
    """

    def __init__(self, net: Net, negative_slope=0.01):
        super().__init__()
        self.slope = negative_slope
        _, H, W = net.input_shape
        self.H, self.W, self.HW = H, W, H * W
        self.actions_n = net.actions_n
        idx = torch.full((H * W, 9), H * W, dtype=torch.long)  # H*W = the zero padding row
        for y in range(H):
            for x in range(W):
                for ky in range(3):
                    for kx in range(3):
                        yy, xx = y + ky - 1, x + kx - 1
                        if 0 <= yy < H and 0 <= xx < W:
                            idx[y * W + x, ky * 3 + kx] = yy * W + xx
        self.register_buffer("idx", idx.reshape(-1))
        self.wm = nn.ParameterList()
        self.bs = nn.ParameterList()
        for blk in [net.conv_in] + net.residual_blocks():
            w, b = _fold(blk)  # [Cout, Cin, 3, 3]
            wm = w.permute(2, 3, 1, 0).reshape(-1, w.shape[0]).contiguous()  # [(ky,kx,c), Cout]
            self.wm.append(nn.Parameter(wm, requires_grad=False))
            self.bs.append(nn.Parameter(b, requires_grad=False))
        wv, bv = _fold(net.conv_val)      # [1, 64, 1, 1]
        wp, bp = _fold(net.conv_policy)   # [2, 64, 1, 1]
        self.wh = nn.Parameter(torch.cat([wv, wp]).reshape(3, -1).t().contiguous(), requires_grad=False)  # [64, 3]
        self.bh = nn.Parameter(torch.cat([bv, bp]), requires_grad=False)
        self.value = copy.deepcopy(net.value)
        self.policy = copy.deepcopy(net.policy)
        for p in self.parameters():
            p.requires_grad_(False)

    def _conv3(self, x, i):
        # x: [L, HW, C]
        L, HW, Cc = x.shape
        xp = F.pad(x, (0, 0, 0, 1))                      # zero row at index HW
        patches = xp.index_select(1, self.idx).reshape(L * HW, 9 * Cc)
        y = torch.addmm(self.bs[i], patches, self.wm[i])
        return F.leaky_relu(y, self.slope).reshape(L, HW, -1)

    def forward(self, x):
        L = x.shape[0]
        h = x.reshape(L, x.shape[1], self.HW).transpose(1, 2)  # [L, HW, 2] channels-last view
        h = self._conv3(h.contiguous(), 0)
        for i in range(1, len(self.wm)):
            h = h + self._conv3(h, i)
        heads = F.leaky_relu(torch.addmm(self.bh, h.reshape(L * self.HW, -1), self.wh), self.slope)
        heads = heads.reshape(L, self.HW, 3)
        val = self.value(heads[:, :, 0])
        pol = self.policy(heads[:, :, 1:3].transpose(1, 2).reshape(L, 2 * self.HW))
        return pol, val
