#!/bin/bash
# Round-2 profiles on the GPU box (one gpurun call): kernel-trace stats of the driver's bench command, and the
# PMC passes (HBM bytes, MFMA busy), each in its own rocprofv3 run as the pool requires.  Output: gpurun_out/prof_r02/
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/${1:-prof_r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-extra-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 20 --warmup 5 --no-profile > $OUT/stats.json 2> $OUT/stats.err
echo stats done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B --steps 3 --warmup 2 --no-profile > $OUT/fetch.json 2> $OUT/fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B --steps 3 --warmup 2 --no-profile > $OUT/write.json 2> $OUT/write.err
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $B --steps 3 --warmup 2 --no-profile > $OUT/mfma.json 2> $OUT/mfma.err
echo mfma done
# keep the merge-back small: per-kernel stats and the counter tables only
find $OUT -name "*_agent_info.csv" -delete
# (the kernel trace of the stats run is what tools/prof_summary.py reads)
ls -R $OUT | head -40
du -sh $OUT
