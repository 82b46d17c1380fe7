#!/bin/bash
# Round-5 experiment (VERDICT r4 task 1): does the 2 k-cycle tail of a tree level's load at 1024 resident games come
# from address aliasing?  Every tree is a 4 MiB-strided region and the slot hash has no per-tree term, so the hot
# nodes of all trees (the same opening boards) sit at the same offset mod 4 MiB.  Four settings, each measured with
# tools/probe_stag.py (block / level clocks) and rocprofv3 --kernel-trace --stats of the bench command:
#   base   CARO_SLOT_ROT=0 CARO_TREE_SKEW=0
#   rot    per-tree slot rotation   home = (hash + t * ROT) & mask
#   skew   tree tables hcap + SKEW slots apart
#   both
# Output: gpurun_out/<dir>/<setting>.{probe,bench,stats}
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-slot_alias}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-configs --sustained-moves 0 --no-profile --steps 40 --warmup 20"
one() {  # name rot skew
  export CARO_SLOT_ROT=$2 CARO_TREE_SKEW=$3
  python3 $ROOT/tools/probe_stag.py 1024 > $OUT/$1.probe 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_stats -- $B > $OUT/$1.bench 2> $OUT/$1.err
  python3 - "$OUT/$1_stats" <<'EOF' > $OUT/$1.stats
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("void caro::k_tree_stag", "k_net_forward_w", "void caro::k_net_forward_w")) or "k_tree_stag" in r["Name"] or "k_net_forward_w" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
EOF
  find $OUT/$1_stats -name "*_agent_info.csv" -delete
  find $OUT/$1_stats -name "*kernel_trace.csv" -delete
  echo "$1 done"; cat $OUT/$1.stats; grep -o '"value": [0-9.]*' $OUT/$1.bench | head -1
}
one base 0 0
one rot 0x9E3779B1 0
one skew 0 37
one both 0x9E3779B1 37
