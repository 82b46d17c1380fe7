#!/bin/bash
# builds experiment variants of the library: caro_ai_amd/libcaro_exp<N>.so with -DCARO_EXP=N (timing experiments only)
set -e
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -Wno-unused-function -mllvm -disable-promote-alloca-to-lds -DCARO_EXP=$n -c caro_ai_amd/csrc/caro_net.hip -o /tmp/caro_net_exp$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC caro_ai_amd/csrc/caro_engine.hip.o /tmp/caro_net_exp$n.o -o caro_ai_amd/libcaro_exp$n.so
  echo built exp$n
done
