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


def _lds_image_index():
    """index of weight (ci = 32h + j, co) inside one 4096-float tap chunk, and the matching (co, ci) lists"""
    co = np.arange(64)[None, :, None]
    h = np.arange(2)[:, None, None]
    j = np.arange(32)[None, None, :]
    idx = ((h * 64 + co) * 32 + ((((j >> 2) ^ ((co >> 1) & 7)) << 2) | (j & 3))).reshape(-1)
    ci = np.broadcast_to(h * 32 + j, (2, 64, 32)).reshape(-1)
    cc = np.broadcast_to(co, (2, 64, 32)).reshape(-1)
    return idx, cc, ci


# F(2,3) weight transform: rows p of G applied along ky
_WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def pack_net_w(net: Net) -> np.ndarray:
    """float32[5 * 4 * 3 * 4096]: per residual layer the transformed taps U[p][dx] = sum_ky G[p][ky] w[:, :, ky, dx]
    (float64 sum, rounded once), each in the LDS image order of the plain tap chunks (k_net_forward_w)."""
    net = net.eval()
    idx, cc, ci = _lds_image_index()
    out = np.zeros((5, 4, 3, 4096), np.float32)
    for li, blk in enumerate(net.residual_blocks()):
        w, _ = _fold(blk)
        w = w.cpu().numpy().astype(np.float64)                # [co, ci, ky, kx]
        u = np.einsum("pk,oikx->pxoi", _WINO_G, w)            # [p, dx, co, ci]
        for pp in range(4):
            for dx in range(3):
                out[li, pp, dx, idx] = u[pp, dx][cc, ci].astype(np.float32)
    return out.reshape(-1)


# 2-D Winograd F(2x2,3x3): which transformed column b a (phase, b-half) pair of the kernel processes (trunk_w2d)
W2_PHASE_B = ((1, 2), (0, 3))  # [phase][bh]


def wino2d_weights(w: np.ndarray) -> np.ndarray:
    """U[a][b][co][ci] = sum_{ky,kx} G[a][ky] G[b][kx] w[co][ci][ky][kx], float64"""
    return np.einsum("ak,bl,oikl->aboi", _WINO_G, _WINO_G, w.astype(np.float64))


def pack_net_w2(net: Net) -> np.ndarray:
    """float32[5][8 chunks][2 bh][4 a][2 h][64 co][8]: the transformed taps of the five residual layers in the LDS image
    order of k_net_forward_w2 / trunk_w2d.  Chunk c = phase * 4 + cq holds, for the two b of the phase (W2_PHASE_B) and
    all four a, the channel granules G = 2 cq and 2 cq + 1 of both lane halves: channels 32 h + 4 G + 0..3 at 16-byte
    slot (G & 1) ^ ((co >> 3) & 1) of row (bh, a, h, co).  float64 transform, rounded once."""
    net = net.eval()
    out = np.zeros((5, 8, 2, 4, 2, 64, 8), np.float32)
    co = np.arange(64)
    for li, blk in enumerate(net.residual_blocks()):
        w, _ = _fold(blk)
        u = wino2d_weights(w.cpu().numpy())                    # [a, b, co, ci]
        for phase in range(2):
            for bh in range(2):
                b = W2_PHASE_B[phase][bh]
                for cq in range(4):
                    for gi in range(2):
                        G = 2 * cq + gi
                        for h in range(2):
                            ch = 32 * h + 4 * G
                            blk4 = u[:, b, :, ch:ch + 4].astype(np.float32)  # [a, co, 4]
                            slot = gi ^ ((co >> 3) & 1)                       # [co]
                            for a in range(4):
                                for sl in range(2):
                                    m = slot == sl
                                    out[li, phase * 4 + cq, bh, a, h, m, sl * 4:sl * 4 + 4] = blk4[a, m]
    return out.reshape(-1)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> the nearest bfloat16 (ties to even), as uint16 bit patterns; finite inputs"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_value(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def split_bf16x3(x: np.ndarray):
    """x (float32) -> three bfloat16 bit arrays hi, mid, lo with hi + mid + lo == x (each residual is exact)"""
    x = np.ascontiguousarray(x, np.float32)
    hi = bf16_round(x)
    r1 = x - bf16_value(hi)
    mid = bf16_round(r1)
    r2 = r1 - bf16_value(mid)
    lo = bf16_round(r2)
    return hi, mid, lo


def pack_net_x3(net: Net) -> np.ndarray:
    """uint16[45 (layer, tap)][2 c][3 parts][4 kg][64 co][8 ci]: the folded residual weights split into three bfloat16
    parts, in the LDS image order of k_net_forward_x3 (ci = 32 c + 8 kg + 0..7)."""
    net = net.eval()
    out = np.zeros((5, 9, 2, 3, 4, 64, 8), np.uint16)
    for li, blk in enumerate(net.residual_blocks()):
        w, _ = _fold(blk)
        w = w.detach().cpu().numpy().astype(np.float32)           # [co, ci, ky, kx]
        for tap in range(9):
            wt = w[:, :, tap // 3, tap % 3].reshape(64, 2, 4, 8).transpose(1, 2, 0, 3)  # [c, kg, co, 8]
            for part, bits in enumerate(split_bf16x3(wt)):
                out[li, tap, :, part] = bits
    return out.reshape(-1)


def wino2d_pays(H: int, W: int) -> bool:
    """the 2-D form executes 64 tiles x 16 taps per board, the row form ceil(tiles / 32) * 32 x 12 with
    ceil(H / 2) * W tiles: take the 2-D form where it is supported and at least 1/4 cheaper (13x13 up)"""
    L = _lib.load()
    if not L.caro_net_winograd2d_supported(H, W):
        return False
    rows = -(-(((H + 1) // 2) * W) // 32) * 32
    return 64 * 16 <= 0.75 * rows * 12


class HipNet:
    """Device-resident packed weights + the forward launch.
    mode "f32w" (default): float32 on v_mfma_f32_32x32x2_f32, the 3x3 convolutions in Winograd form -- the row form
                 F(2,3) ("f32w1"), or on large boards (13x13 up, one board per workgroup) the 2-D form F(2x2,3x3) ("f32w2").
    mode "f32w1" / "f32w2": that form, forced.  split_tiles=False (row form): full tiles only, see __init__.
    mode "f32": the same with direct 3x3 convolutions (a plain fma chain in k order): the A/B baseline.
    mode "bf16x3": an EXTRA mode, never a default: the direct form with every float32 trunk operand split into three
                 bfloat16 parts, six part products per multiply on v_mfma_f32_16x16x32_bf16, float32 accumulation (k_net_forward_x3);
                 not bit-identical to the float32 modes, within the tolerance tests/test_gpu_net.py states."""

    device_counts = True  # the engine may call forward_dev without knowing L on the host

    def __init__(self, net: Net, device="cuda:0", negative_slope=0.01, mode="f32w", split_tiles=True):
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
        if mode == "f32w":
            mode = "f32w2" if wino2d_pays(self.H, self.W) else "f32w1"
        self.mode = mode
        if mode == "f32w2":
            w2 = pack_net_w2(net)
            assert w2.size == self.L.caro_net_winograd2d_size()
            _lib.check(self.L.caro_net_enable_winograd2d(self.h, w2.ctypes.data, w2.size))
        elif mode == "f32w1":
            ww = pack_net_w(net)
            # split_tiles=False: full tiles only -- no 2- / 4-way K-split tiles for small launches (the library reads
            # CARO_NO_SPLIT_TILES when the mode is enabled).  A half-size launch then occupies half the compute units
            # instead of all of them: what two engines on two streams want (bench.py `two_streams`).
            import os
            old = os.environ.get("CARO_NO_SPLIT_TILES")
            if not split_tiles:
                os.environ["CARO_NO_SPLIT_TILES"] = "1"
            try:
                _lib.check(self.L.caro_net_enable_winograd(self.h, ww.ctypes.data, ww.size))
            finally:
                if not split_tiles:
                    if old is None:
                        del os.environ["CARO_NO_SPLIT_TILES"]
                    else:
                        os.environ["CARO_NO_SPLIT_TILES"] = old
        elif mode == "bf16x3":
            wx = pack_net_x3(net)
            assert wx.size == self.L.caro_net_split_bf16_size()
            _lib.check(self.L.caro_net_enable_split_bf16(self.h, wx.ctypes.data, wx.size))
        else:
            assert mode == "f32", mode

    def close(self):
        if getattr(self, "h", None):
            self.L.caro_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workgroup_mfma_flops(self):
        """flops ONE workgroup of the mode's trunk executes on the matrix pipe (v_mfma_f32_32x32x2_f32 = 4096 flop),
        padding rows included: what bench.py's `roofline.achieved` counts.  f32w1: 5 layers x 12 transformed taps
        (4 p x 3 dx), each 32 k-steps on 8 waves; f32 (direct): 5 x 9 taps; f32w2: 5 x 16 taps on 64 tiles."""
        if self.mode == "f32w2":  # 5 layers x 16 taps x (2 row tiles x 2 column tiles) blocks of 32 k-steps
            return 5 * 16 * 4 * 32 * 4096.0
        if self.mode == "bf16x3":  # 45 taps x 96 v_mfma_f32_16x16x32_bf16 (16384 flop: six part products per multiply) x 8 waves
            return 45 * 96 * 8 * 16384.0
        taps = {"f32w1": 60, "f32": 45}.get(self.mode)
        return None if taps is None else taps * 32 * 8 * 4096.0

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


def weights_version(net):
    """changes whenever a parameter or batch-norm buffer of `net` is written in place (optimizer step,
    load_state_dict, NetWrapper.sync) or replaced (`.to()`): tensor version counters + storage addresses"""
    return tuple((t._version, t.data_ptr()) for t in net.state_dict(keep_vars=True).values())


_HIPNETS = {}  # (id(net), device, mode) -> (weights version, HipNet), most recently used last
HIPNET_CACHE = 4


def hipnet_for(net: Net, device="cuda:0", mode="f32w", split_tiles=True) -> HipNet:
    """The `HipNet` of `net` as its weights are NOW, built once per weight version: callers that run the same net
    again and again (train.self_play with the best net between two promotions, train.py:185-217) do not pack and
    upload it again.  The cache only drops its reference when an entry goes stale or falls out; an engine that
    still launches on a HipNet keeps it alive."""
    key = (id(net), str(torch.device(device)), mode, bool(split_tiles))
    ver = weights_version(net)
    hit = _HIPNETS.pop(key, None)
    if hit is None or hit[0] != ver:
        hit = (ver, HipNet(net, device, mode=mode, split_tiles=split_tiles))
    _HIPNETS[key] = hit
    while len(_HIPNETS) > HIPNET_CACHE:
        _HIPNETS.pop(next(iter(_HIPNETS)))
    return hit[1]


def release_hipnets():
    _HIPNETS.clear()


class HashNet:
    """The table evaluator of include/caro_hip.h (`caro_net_create_hash`): priors and value are exact integer-hash
    functions of the leaf planes.  Same device-side interface as `HipNet` (leaf counts read on the device), so an
    engine built on it runs the very launches the conv net runs -- the form in which the search is compared bit for
    bit with the oracle.  Not a model: it has no weights."""

    device_counts = True

    def __init__(self, game_or_shape, actions_n=None, device="cuda:0", salt=0):
        shape = getattr(game_or_shape, "obs_shape", game_or_shape)
        self.H, self.W = int(shape[1]), int(shape[2])
        self.A = int(actions_n if actions_n is not None else game_or_shape.action_space)
        self.L = _lib.load()
        self.device = torch.device(device)
        self.salt = int(salt)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(self.L.caro_net_create_hash(self.H, self.W, self.A, self.salt, self.device.index or 0, C.byref(h)))
        self.h = h

    close = HipNet.close
    __del__ = HipNet.__del__
    forward_dev = HipNet.forward_dev
    __call__ = HipNet.__call__
