#!/bin/bash
# What the memory path of k_tree_stag looks like at 256 and at 1024 resident games (NOTES round 5): per launch, from
# rocprofv3 --pmc, one pass per counter group and size:
#   A  TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum   -> mean L1->L2 read round trip (cycles) = LATENCY / REQ
#   B  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum           -> mean L2->fabric read round trip = LEVEL / RDREQ
#   C  TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum -> L2 hit rate, read / write mix
# Output: gpurun_out/<dir>/summary.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-tcc_hits}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/summary.txt
for G in 256 1024; do
  for P in "A TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "B TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "C TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum"; do
    set -- $P; tag=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/g${G}_$tag -- python3 $ROOT/bench.py --games $G --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --sustained-moves 0 --no-profile > $OUT/g${G}_$tag.json 2> $OUT/g${G}_$tag.err || { echo "pass $tag at G=$G failed" >> $OUT/summary.txt; tail -3 $OUT/g${G}_$tag.err >> $OUT/summary.txt; continue; }
    python3 - "$OUT/g${G}_$tag" "$G" "$tag" <<'PYEOF' >> $OUT/summary.txt
import csv, glob, sys
from collections import defaultdict
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
name = lambda r: "k_tree_stag" if "k_tree_stag" in r["Kernel_Name"] else "k_net_forward_w" if "k_net_forward_w" in r["Kernel_Name"] else None
big = defaultdict(int)
for r in rows:
    if name(r):
        big[name(r)] = max(big[name(r)], int(r["Grid_Size"]))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = name(r)
    if k and int(r["Grid_Size"]) == big[k]:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    extra = ""
    if "TCP_TCC_READ_REQ_LATENCY_sum" in m:
        extra = "mean L1->L2 read round trip %.0f cycles" % (m["TCP_TCC_READ_REQ_LATENCY_sum"] / max(1.0, m["TCP_TCC_READ_REQ_sum"]))
    if "TCC_EA0_RDREQ_LEVEL_sum" in m:
        extra = "mean L2->fabric read round trip %.0f cycles" % (m["TCC_EA0_RDREQ_LEVEL_sum"] / max(1.0, m["TCC_EA0_RDREQ_sum"]))
    if "TCC_HIT_sum" in m:
        extra = "L2 hit rate %.3f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]))
    print("G=%-5s pass %s %-16s %d launches | " % (sys.argv[2], sys.argv[3], k, len(next(iter(c.values())))) + "  ".join("%s %.0f" % (n, v) for n, v in sorted(m.items())) + " | " + extra)
PYEOF
    rm -rf $OUT/g${G}_$tag
  done
done
cat $OUT/summary.txt
