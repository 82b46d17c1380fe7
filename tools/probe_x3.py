"""bf16x3 (split-operand) net kernel against the float32 modes: distance from a float64 forward, and launch time at the
bench's leaf count.  `python tools/probe_x3.py [rows]`"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1434
W = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "caro_ai_amd/data/weights/")


def rand_net(shape, A, seed):
    torch.manual_seed(seed)
    net = Net(shape, A)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.5, 0.5); m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.3, 0.3)
    return net.eval()


cases = [("c4 trained", (2, 6, 7), 7, "best_026_12000.dat"), ("c4 random", (2, 6, 7), 7, None),
         ("ttt trained", (2, 3, 3), 9, "best_005_00900.dat"), ("10x10 random", (2, 10, 10), 100, None),
         ("15x15 random", (2, 15, 15), 225, None)]
for name, shape, A, wf in cases:
    if wf:
        net = Net(shape, A); net.load_state_dict(torch.load(W + wf, map_location="cpu")); net.eval()
    else:
        net = rand_net(shape, A, 1)
    g = torch.Generator().manual_seed(7)
    x = (torch.rand((600,) + shape, generator=g) < 0.3).float(); x[:, 1] *= (1 - x[:, 0])
    with torch.no_grad():
        lg, vl = net(x); p32 = torch.softmax(lg, 1)
        lg64, vl64 = net.double()(x.double()); p64 = torch.softmax(lg64, 1)
    net.float()
    line = [f"{name}: torch32 dP {((p32.double() - p64).abs().max().item()):.2e} dv {((vl.double() - vl64).abs().max().item()):.2e}"]
    for mode in ("f32", "f32w", "bf16x3"):
        hn = HipNet(net, "cuda:0", mode=mode)
        p, v = hn(x.cuda()); torch.cuda.synchronize()
        line.append(f"{mode} dP {((p.cpu().double() - p64).abs().max().item()):.2e} dv {((v.cpu().double() - vl64[:, 0]).abs().max().item()):.2e}"
                    f" |P-torch32| {((p.cpu() - p32).abs().max().item()):.2e}")
        hn.close()
    print(" | ".join(line), flush=True)

net = Net((2, 6, 7), 7); net.load_state_dict(torch.load(W + "best_026_12000.dat", map_location="cpu"))
x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
for mode in ("f32", "f32w", "bf16x3", "f32w", "bf16x3"):
    hn = HipNet(net, "cuda:0", mode=mode)
    for _ in range(50):
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400):
        hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
    e1.record(); torch.cuda.synchronize()
    print(f"{mode}: {e0.elapsed_time(e1) / 400 * 1000:.1f} us per launch at {rows} rows", flush=True)
    hn.close()
