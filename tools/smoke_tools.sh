#!/bin/bash
# Runs every diagnostic under tools/ once on the GPU box (bounded), and says which still work against the current
# C-ABI: gpurun_out/<dir>/tools_smoke.txt.  usage: tools/smoke_tools.sh <dir>
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-tools_smoke}
mkdir -p $OUT
cd $ROOT
: > $OUT/tools_smoke.txt
for t in probe_clock.py probe_clock15.py probe_engine_net.py probe_grid.py \
         probe_leaves.py probe_netloop.py probe_select.py probe_select15.py probe_small.py probe_stag.py \
         measure_dup_leaves.py measure_move_latency.py measure_train_step.py cmp_net.py bench_noise.py probe_x3.py probe_gate.py; do
  a=""; if [ $t = cmp_net.py ]; then a=$OUT/cmp_net.npz; fi
  timeout -k 5 150 python tools/$t $a > $OUT/smoke_$t.log 2>&1
  echo "$t rc=$?" >> $OUT/tools_smoke.txt
done
cat $OUT/tools_smoke.txt
