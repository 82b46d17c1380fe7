"""`HipNet`: the fused float32 HIP inference kernel (csrc/caro_net.hip) for a
`lib.model.Net` in eval mode -- same function as `Net.eval()` followed by the
`F.softmax` of lib/mcts.py:216, batch-norm folded into the convolutions.

The kernel reads the leaf count from device memory, so the engine can enqueue
select -> net -> expand+backup for a whole move without a host round trip.
"""
import ctypes as C

import numpy as np
import torch

from caro_ai_amd import _lib
from caro_ai_amd.lib.model import Net, _fold


def pack_net(net: Net) -> np.ndarray:
    """Flat float32 buffer in the order include/caro_hip.h documents."""
    net = net.eval()
    H, W = net.input_shape[1], net.input_shape[2]
    parts = []
    w0, b0 = _fold(net.conv_in)                                   # [64, 2, 3, 3]
    parts += [w0.permute(2, 3, 1, 0).reshape(9, 2, 64), b0]       # [tap][ci][co]
    co = np.arange(64)[None, :, None]
    h = np.arange(2)[:, None, None]
    j = np.arange(32)[None, None, :]
    idx = ((h * 64 + co) * 32 + ((((j >> 2) ^ ((co >> 1) & 7)) << 2) | (j & 3))).reshape(-1)  # LDS image index
    ci = np.broadcast_to(h * 32 + j, (2, 64, 32)).reshape(-1)
    cc = np.broadcast_to(co, (2, 64, 32)).reshape(-1)
    res_w, res_b = [], []
    for blk in net.residual_blocks():
        w, b = _fold(blk)                                         # [co, ci, ky, kx]
        w = w.cpu().numpy()
        chunks = np.zeros((9, 4096), np.float32)
        for tap in range(9):
            chunks[tap, idx] = w[cc, ci, tap // 3, tap % 3]
        res_w.append(chunks)
        res_b.append(b.cpu().numpy())
    parts += [np.stack(res_w), np.stack(res_b)]
    wv, bv = _fold(net.conv_val)
    wp, bp = _fold(net.conv_policy)
    parts += [torch.cat([wv, wp]).reshape(3, 64), torch.cat([bv, bp])]
    parts += [net.value[0].weight, net.value[0].bias, net.value[2].weight.reshape(-1), net.value[2].bias]
    parts += [net.policy[0].weight, net.policy[0].bias]
    flat = [np.ascontiguousarray(p.detach().cpu().numpy() if torch.is_tensor(p) else p, dtype=np.float32).reshape(-1)
            for p in parts]
    out = np.concatenate(flat)
    assert out.size == _lib.load().caro_net_packed_size(H, W, net.actions_n), out.size
    return out


class HipNet:
    """Device-resident packed weights + the forward launch."""

    device_counts = True  # the engine may call forward_dev without knowing L on the host

    def __init__(self, net: Net, device="cuda:0", negative_slope=0.01):
        self.L = _lib.load()
        self.device = torch.device(device)
        self.H, self.W = net.input_shape[1], net.input_shape[2]
        self.A = net.actions_n
        packed = pack_net(net)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(self.L.caro_net_create(self.H, self.W, self.A, negative_slope, packed.ctypes.data, packed.size,
                                          self.device.index or 0, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.caro_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward_dev(self, planes, counts_dev_ptr, which, max_rows, probs, values, stream):
        _lib.check(self.L.caro_net_forward(self.h, planes.data_ptr(), counts_dev_ptr, which, max_rows,
                                           probs.data_ptr(), values.data_ptr(), stream))

    def __call__(self, planes):
        """evaluator form (L known on the host): planes[L,2,H,W] -> (P[L,A], v[L])"""
        L = planes.shape[0]
        counts = torch.tensor([L, 0], dtype=torch.int32, device=planes.device)
        probs = torch.empty((L, self.A), dtype=torch.float32, device=planes.device)
        values = torch.empty(L, dtype=torch.float32, device=planes.device)
        st = C.c_void_p(torch.cuda.current_stream(planes.device).cuda_stream)
        self.forward_dev(planes.contiguous(), counts.data_ptr(), 0, L, probs, values, st)
        return probs, values
