#!/bin/bash
# Profiles on the GPU box (one gpurun call): kernel-trace stats of the driver's bench command, and the PMC passes (HBM
# bytes, MFMA busy), each in its own rocprofv3 run as the pool requires -- at the driver's --steps 20 --warmup 5 for the
# headline, and the same three passes on BASELINE configs 5 and 4 (+ a kernel-trace pass of config 4: its tree kernel).
# Output: gpurun_out/<dir>/ ; summarise with  python tools/prof_summary.py gpurun_out/<dir> r06_a pmc_r06.json
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-prof_r06}
WHAT=${2:-all}   # "stats": the kernel-trace pass of the headline only; "x3stats": that of the --net hipx3 leg only
mkdir -p $OUT
# what is profiled: bench.py quotes the counters of pmc_r06.json only while the kernel sources still hash to this
python3 $ROOT/tools/source_sha.py > $OUT/source_sha256.json
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-configs --sustained-moves 0 --train-loop-games 0 --no-profile"
H="--steps 20 --warmup 5"
C5="--arena --games 512 --searches 100 --steps 3 --warmup 2"
C4="--game gomoku15 --searches 50 --steps 2 --warmup 1"
x3_stats() {  # kernel-trace pass of the labelled extra leg's configuration (--net hipx3)
  echo "rocprofv3 --kernel-trace --stats -- $B $H --net hipx3" | sed "s#$ROOT/##g" > $OUT/stats_x3.cmd
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x3 -- $B $H --net hipx3 > $OUT/stats_x3.json 2> $OUT/stats_x3.err
  echo x3 stats done
}
if [ "$WHAT" = "x3stats" ]; then x3_stats; exit 0; fi
echo "rocprofv3 --kernel-trace --stats -- $B $H" | sed "s#$ROOT/##g" > $OUT/stats.cmd  # (what the summary's header quotes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B $H > $OUT/stats.json 2> $OUT/stats.err
echo stats done
if [ "$WHAT" = "stats" ]; then exit 0; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lockstep -- $B $H --stagger 0 > $OUT/stats_lockstep.json 2> $OUT/stats_lockstep.err
echo lock-step stats done
run_pmc() {  # $1 prefix, $2.. bench arguments
  local pre=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${pre}_fetch -- $B "$@" > $OUT/${pre}_fetch.json 2> $OUT/${pre}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${pre}_write -- $B "$@" > $OUT/${pre}_write.json 2> $OUT/${pre}_write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${pre}_mfma -- $B "$@" > $OUT/${pre}_mfma.json 2> $OUT/${pre}_mfma.err
  echo $pre done
}
run_pmc pmc $H
run_pmc config5 $C5
run_pmc config4 $C4
run_pmc x3 $H --net hipx3   # the labelled extra leg: bf16x3 split operands (k_net_forward_x3)
echo "rocprofv3 --kernel-trace --stats -- $B $C4" | sed "s#$ROOT/##g" > $OUT/stats_config4.cmd
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_config4 -- $B $C4 > $OUT/stats_config4.json 2> $OUT/stats_config4.err
echo config4 stats done
x3_stats
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
