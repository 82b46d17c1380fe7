"""SHA-256 of the kernel sources (caro_ai_amd/csrc + include/): what a PMC pass was taken on.  tools/profile_r06.sh writes it beside
the counters; bench.py quotes `mfma_busy_pmc` / `traffic` from the committed PMC file only while the sources still hash
to it (a kernel change without a new PMC pass must not carry stale counters)."""
import hashlib
import json
import os


def source_sha256(root=None):
    root = root or os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    out = {}
    for d in (os.path.join(root, "caro_ai_amd", "csrc"), os.path.join(root, "include")):  # (the kernels include both)
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".inc")):
                out[f] = hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()
    return out


if __name__ == "__main__":
    print(json.dumps(source_sha256()))
