#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + the two PMC passes of the default
# bench, summaries into gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep "^{\"metric\"" $OUT/bench_stats.log | tail -1 > $OUT/bench_under_rocprof.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats
cat $OUT/traffic.json | head -40
