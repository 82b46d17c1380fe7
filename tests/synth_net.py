"""Torch twin of oracle_synth_net (oracle/caro_oracle.c): a deterministic "net"
whose priors and value are exact dyadic float32 functions of an integer hash of
the input planes.  All arithmetic is int64 with two's-complement wrap-around
(== uint64 arithmetic), so CPU, GPU and the C oracle agree bit for bit.
"""
import numpy as np
import torch

MASK = (1 << 64) - 1


def _s64(x):  # uint64 constant as the int64 with the same bits
    x &= MASK
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix64_py(z):
    z &= MASK
    z ^= z >> 30
    z = (z * 0xbf58476d1ce4e5b9) & MASK
    z ^= z >> 27
    z = (z * 0x94d049bb133111eb) & MASK
    z ^= z >> 31
    return z


def _lsr(x, s):  # logical shift right on int64 tensors
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix64(z):
    z = z ^ _lsr(z, 30)
    z = z * _s64(0xbf58476d1ce4e5b9)
    z = z ^ _lsr(z, 27)
    z = z * _s64(0x94d049bb133111eb)
    z = z ^ _lsr(z, 31)
    return z


class SynthNet:
    def __init__(self, n_inputs, A, device, salt=0):
        self.A = A
        self.salt = _s64(salt)
        coef = [_s64(_mix64_py(0x5851f42d4c957f2d + j) | 1) for j in range(n_inputs)]
        self.coef = torch.tensor(coef, dtype=torch.int64, device=device)
        self.astep = torch.tensor([_s64(0x9E3779B97F4A7C15 * (a + 1)) for a in range(A)], dtype=torch.int64,
                                 device=device)

    @torch.no_grad()
    def __call__(self, planes):
        L = planes.shape[0]
        mask = (planes.reshape(L, -1) != 0).to(torch.int64)
        h = (mask * self.coef).sum(dim=1) + self.salt  # wraps mod 2^64
        ha = _mix64(h[:, None] + self.astep[None, :])
        P = ((_lsr(ha, 20) & 1023) + 1).to(torch.float32) / 8192.0
        hv = _mix64(h ^ _s64(0xA5A5A5A5A5A5A5A5))
        v = ((_lsr(hv, 20) % 2001) - 1000).to(torch.float32) / 1024.0
        return P.contiguous(), v.contiguous()


def synth_numpy(planes, A, salt=0):
    """numpy/CPU form through torch CPU (used by CPU-side checks)"""
    net = SynthNet(int(np.prod(planes.shape[1:])), A, "cpu", salt)
    P, v = net(torch.from_numpy(np.ascontiguousarray(planes)))
    return P.numpy(), v.numpy()
