"""Digests of the float32 net kernel's outputs (k_net_forward_w, the default form) for fixed inputs and the shipped
weights, at launch sizes that reach every tile class -> tests/golden/net_hip_digest.json.  Run on an MI355X:

    python tests/golden/make_net_digest.py

The kernel is deterministic; the digests pin its bits, so that a re-ordering of the kernel that is meant to keep the
arithmetic (as round 2's re-ordered trunk was: every digest equal to the round-1 kernel's) can be checked against the
committed file, and one that is meant to change it shows up as exactly that (tests/test_gpu_net.py)."""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = (("c4", (2, 6, 7), 7, "best_026_12000.dat", (1, 5, 6, 7, 100, 256, 257, 600, 768, 769, 1434, 1537, 1700, 2300, 3100)),
         ("ttt3", (2, 3, 3), 9, "best_005_00900.dat", (1, 40, 3000)))


def boards(rows, shape):
    g = torch.Generator().manual_seed(rows)
    r = torch.rand((rows,) + shape[1:], generator=g)
    x = torch.zeros((rows,) + shape)
    x[:, 0] = (r < 0.3).float()
    x[:, 1] = ((r >= 0.3) & (r < 0.55)).float()
    return x


def digests():
    from caro_ai_amd.lib.model import Net
    from caro_ai_amd.net_hip import HipNet
    out = {}
    for name, shape, A, weights, sizes in CASES:
        net = Net(shape, A)
        net.load_state_dict(torch.load(os.path.join(ROOT, "tests", "golden", "weights", weights), map_location="cpu"))
        hn = HipNet(net.eval(), "cuda:0")
        for rows in sizes:
            p, v = hn(boards(rows, shape).cuda())
            torch.cuda.synchronize()
            h = hashlib.sha256(p.cpu().numpy().tobytes() + v.cpu().numpy().tobytes()).hexdigest()
            out["%s_%d" % (name, rows)] = h
        hn.close()
    return out


if __name__ == "__main__":
    d = digests()
    with open(os.path.join(ROOT, "tests", "golden", "net_hip_digest.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print(json.dumps(d, indent=1, sort_keys=True))
