#!/bin/bash
# One gpurun call: the whole GPU suite, then the driver's bench command, the 400-move soak and the lock-step engine.
# Output under gpurun_out/<dir>/ ; usage: tools/endrun.sh <dir>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-endrun}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -2 $OUT/tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench.err && \
timeout -k 10 200 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extra-configs --sustained-moves 0 > $OUT/soak_400.json 2>> $OUT/bench.err && \
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --stagger 0 --no-cpu-baseline > $OUT/lockstep.json 2>> $OUT/bench.err
python - <<PY
import json
for n in ("bench_20_5", "soak_400", "lockstep"):
    try:
        d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "unreadable", e); continue
    r = d.get("roofline", {})
    print("%s: %.0f ms %.3f sustained %s extras_rc %s | net %.1f us frac %.3f issued %s | tree %s" % (
        n, d["value"], d["ms_per_step"], (d.get("sustained") or {}).get("value"), d.get("extras_rc"),
        r.get("avg_launch_us", 0), r.get("frac", 0), (r.get("mfma_issued") or {}).get("frac"), (d.get("roofline_tree") or {}).get("avg_launch_us")))
    for k in ("config5", "config4"):
        c = d.get(k)
        if c:
            cr = c.get("roofline", {})
            print("   %s %.0f ms %.2f games/s %s finished %s ovf %s issued %s net %.1f us" % (
                k, c["value"], c["ms_per_step"], c.get("games_per_s"), c.get("games_finished"), c.get("overflows"),
                (cr.get("mfma_issued") or {}).get("frac"), cr.get("avg_launch_us", 0)))
PY
