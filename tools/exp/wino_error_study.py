"""Larger Winograd tiles for the connect-four net, priced before building anything (CPU, torch): the error of
F(mh x mw, 3x3) trunks in float32 against the float64 forward, next to torch float32 (the tolerance of
tests/test_gpu_net.py: |dP| < 1e-4 and no further from float64 than 4x torch).  python tools/exp/wino_error_study.py"""
import sys, torch, numpy as np
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from caro_ai_amd.lib.model import Net
import torch.nn.functional as F
torch.manual_seed(0)
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "..", "tests/golden/weights/best_026_12000.dat"), map_location="cpu")); net.eval()
L = 300
g = torch.Generator().manual_seed(L)
x = (torch.rand((L, 2, 6, 7), generator=g) < 0.3).float(); x[:, 1] *= (1 - x[:, 0])

def mats(m, dtype):
    if m == 4:
        BT = torch.tensor([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], dtype=dtype)
        G = torch.tensor([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], dtype=torch.float64).to(dtype)
        AT = torch.tensor([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], dtype=dtype)
    else:
        BT = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=dtype)
        G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=dtype)
        AT = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=dtype)
    return BT, G, AT

def wino_conv(x, w, b, mh, mw):
    """3x3 same conv, F(mh x mw, 3x3), everything in x.dtype (mh/mw in {1 (direct along that axis), 2, 4})"""
    dt = x.dtype
    N, C, H, W = x.shape
    def ax(m):
        if m == 1: return None
        return mats(m, dt)
    Mh, Mw = ax(mh), ax(mw)
    th, tw = (mh + 2), (mw + 2)
    nh, nw = -(-H // mh), -(-W // mw)
    xp = F.pad(x, (1, 1 + nw * mw - W, 1, 1 + nh * mh - H))
    tiles = xp.unfold(2, th, mh).unfold(3, tw, mw)       # N C nh nw th tw
    BTh, Gh, ATh = Mh; BTw, Gw, ATw = Mw
    V = torch.einsum("ij,nchwjk,lk->nchwil", BTh, tiles, BTw)
    U = torch.einsum("ij,ocjk,lk->ocil", Gh.to(dt), w.to(dt), Gw.to(dt))
    Mm = torch.einsum("nchwil,ocil->nohwil", V, U)
    Y = torch.einsum("ij,nohwjk,lk->nohwil", ATh, Mm, ATw)   # N O nh nw mh mw
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], nh * mh, nw * mw)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1).to(dt)

orig = F.conv2d
def run(mode, dtype):
    def patched(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if mode and weight.shape[1] == 64 and weight.shape[2] == 3:
            return wino_conv(inp, weight, bias if bias is not None else torch.zeros(weight.shape[0], dtype=inp.dtype), *mode)
        return orig(inp, weight, bias, stride, padding, dilation, groups)
    F.conv2d = patched; torch.nn.functional.conv2d = patched
    import torch.nn as nn
    old = nn.Conv2d._conv_forward
    nn.Conv2d._conv_forward = lambda self, i, w, b: patched(i, w, b, self.stride, self.padding, self.dilation, self.groups)
    n = net.double() if dtype == torch.float64 else net.float()
    with torch.no_grad():
        lg, v = n(x.to(dtype))
    nn.Conv2d._conv_forward = old; F.conv2d = orig
    return torch.softmax(lg, 1).double(), v.double()
p64, v64 = run(None, torch.float64)
pw64, _ = run((4, 4), torch.float64)
print("F(4x4) in float64 vs direct float64 (algorithm check):", (pw64 - p64).abs().max().item())
p32, v32 = run(None, torch.float32)
e_ref = (p32 - p64).abs().max().item()
print("torch float32 direct: max |dP| vs float64 %.3e  |dv| %.3e" % (e_ref, (v32 - v64).abs().max().item()))
for mode in [(2, 1), (2, 2), (2, 4), (4, 2), (4, 4)]:
    try:
        p, v = run(mode, torch.float32)
        print("F(%dx%d): max |dP| vs float64 %.3e (x%.1f of torch's), vs torch float32 %.3e, |dv| %.3e" % (mode + ((p - p64).abs().max().item(), (p - p64).abs().max().item() / e_ref, (p - p32).abs().max().item(), (v - v64).abs().max().item())))
    except Exception as e:
        print(mode, "failed", repr(e)[:200])
