"""A few hundred back-to-back launches of the fused net kernel at the bench's leaf count (for rocprofv3 --pmc)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caro_ai_amd.lib.model import Net
from caro_ai_amd.net_hip import HipNet
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
net = Net((2, 6, 7), 7); net.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "caro_ai_amd/data/weights/") + "best_026_12000.dat", map_location="cpu"))
hn = HipNet(net, "cuda:0", mode=mode)
rows = 1434
x = (torch.rand((rows, 2, 6, 7), device="cuda") < 0.3).float()
counts = torch.tensor([rows, 0], dtype=torch.int32, device="cuda")
probs = torch.empty((rows, 7), device="cuda"); vals = torch.empty(rows, device="cuda")
for _ in range(n):
    hn.forward_dev(x, counts.data_ptr(), 0, rows, probs, vals, None)
torch.cuda.synchronize()
print("done", mode, n)
