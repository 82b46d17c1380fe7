#!/usr/bin/env python3
"""Experiment builds of the net kernel (timing experiments only; the removal builds compute WRONG results).

    python tools/exp/build_exp.py 21 22 24      ->  tools/exp/_build/libcaro_exp<N>.so

The product source (caro_ai_amd/csrc/caro_net.hip) contains no experiment code: its probe points are comments of
the form /*@NAME(args)*/.  This script makes a COPY of the file in which those comments become the macros of
tools/exp/caro_net_exp.h (CARO_NAME(args)), compiles the copy with -DCARO_EXP=N and links it with the product's
engine object.  Load the result with CARO_HIP_LIB=tools/exp/_build/libcaro_exp<N>.so (tools/probe_*.py).
tools/exp/_build/ is listed in .gitignore and .gpurunignore: build on the box that runs the experiment.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "caro_ai_amd", "csrc")
OUT = os.path.join(HERE, "_build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wno-unused-function", "-mllvm",
         "-disable-promote-alloca-to-lds"]


def instrumented_source():
    s = open(os.path.join(CSRC, "caro_net.hip")).read()
    s = s.replace("/*@CHUNK_BARRIER*/ __syncthreads();", "CARO_CHUNK_BARRIER")
    s = s.replace("/*@FETCH_ON*/", "CARO_FETCH_ON &&")
    s, n = re.subn(r"/\*@([A-Z_]+\([^*]*\))\*/", r"CARO_\1", s)
    assert n > 20, "probe points not found"
    anchor = '#include "../../include/caro_noise.h"\n'
    assert anchor in s
    s = s.replace(anchor, anchor + '#include "caro_net_exp.h"\n')
    s = s.replace('"../../include/', '"%s/include/' % ROOT)
    return s


def build(n):
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(OUT, "caro_net_exp%d.hip" % n)
    open(src, "w").write(instrumented_source())
    obj = os.path.join(OUT, "caro_net_exp%d.o" % n)
    subprocess.check_call([HIPCC] + FLAGS + ["-DCARO_EXP=%d" % n, "-I", HERE, "-I", CSRC, "-c", src, "-o", obj])
    eng = os.path.join(CSRC, "caro_engine.hip.o")
    if not os.path.exists(eng):  # the object does not travel to the GPU box (.gpurunignore): compile it here, as build.py does
        eng = os.path.join(OUT, "caro_engine.hip.o")
        if not os.path.exists(eng):
            subprocess.check_call([HIPCC] + FLAGS + ["-ffp-contract=off", "-c", os.path.join(CSRC, "caro_engine.hip"), "-o", eng])
    so = os.path.join(OUT, "libcaro_exp%d.so" % n)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", eng, obj, "-o", so])
    return so


if __name__ == "__main__":
    for a in sys.argv[1:]:
        print("built", build(int(a)))
