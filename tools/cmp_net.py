"""Outputs of the net kernel for fixed inputs -> npz (argv[1]); with two more arguments: compare two such files bit for bit.
Used to check a new build of libcaro_hip against a kept one (CARO_HIP_LIB) at every tile class."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    bad = 0
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        bad += not same
        print("%-28s %s  max |d| %.3e" % (k, "identical" if same else "DIFFERENT", float(np.abs(a[k] - b[k]).max())))
    sys.exit(1 if bad else 0)
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
out = {}
torch.manual_seed(7)
for name, shape, A, sizes in (("c4", (2, 6, 7), 7, (1, 5, 6, 7, 100, 256, 600, 1434, 1700, 2300, 3100)),
                              ("ttt3", (2, 3, 3), 9, (1, 40, 3000)),
                              ("g15", (2, 15, 15), 225, (1, 3, 300))):
    net = Net(shape, A)
    if name == "c4":
        net.load_state_dict(torch.load("caro_ai_amd/data/weights/best_026_12000.dat", map_location="cpu"))
    else:
        with torch.no_grad():
            for prm in net.parameters(): prm.mul_(1.5)
    net.eval()
    hn = HipNet(net, "cuda:0")
    for rows in sizes:
        g = torch.Generator().manual_seed(rows)
        x = torch.zeros((rows,) + shape)
        r = torch.rand((rows,) + shape[1:], generator=g)
        x[:, 0] = (r < 0.3).float(); x[:, 1] = ((r >= 0.3) & (r < 0.55)).float()
        pr, v = hn(x.cuda())
        torch.cuda.synchronize()
        out["%s_%d_p" % (name, rows)] = pr.cpu().numpy(); out["%s_%d_v" % (name, rows)] = v.cpu().numpy()
        with torch.no_grad():
            lg, vv = net(x)
        d = (torch.softmax(lg, 1) - pr.cpu()).abs().max().item()
        print("%s rows %d: max |P - torch cpu| %.2e, max |v - torch cpu| %.2e" % (name, rows, d, (vv.reshape(-1) - v.cpu()).abs().max().item()))
np.savez(sys.argv[1], **out)
