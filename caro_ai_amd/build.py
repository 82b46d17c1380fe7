"""Builds libcaro_hip.so (HIP kernels + C-ABI) for gfx950, in-tree.

    python -m caro_ai_amd.build [--force]

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with
the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcaro_hip.so")
SOURCES = [os.path.join(CSRC, "caro_engine.hip"), os.path.join(CSRC, "caro_net.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "caro_rules.h"), os.path.join(CSRC, "caro_variants.h"),
                  os.path.join(CSRC, "caro_host.inc"),
                  os.path.join(HERE, "..", "include", "caro_hip.h"),
                  os.path.join(HERE, "..", "include", "caro_noise.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# caro_engine.hip: PUCT / noise arithmetic must not be fused (bit parity with the oracle)
PER_FILE = {"caro_engine.hip": ["-ffp-contract=off", "-mllvm", "-disable-promote-alloca-to-lds"],
            # keep small per-thread arrays in registers: LDS is budgeted by hand in the net kernel
            "caro_net.hip": ["-mllvm", "-disable-promote-alloca-to-lds"]}


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + PER_FILE[os.path.basename(src)] + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", OUT)
