#!/bin/bash
# Round evidence (run on the GPU box through gpurun): bench.py default run, rocprofv3 kernel-trace
# stats of the same command, and two PMC passes (FETCH_SIZE; WRITE_SIZE) for HBM traffic.
# usage: bash tools/bench_prof.sh <tag> [extra bench args]
set -e
TAG=$1; shift || true
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $R
python3 bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-sample 0 "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --cpu-sample 0 --steps 5 --warmup 1 "$@" > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --cpu-sample 0 --steps 5 --warmup 1 "$@" > $OUT/pmc_write.json 2> $OUT/pmc_write.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/traffic_summary.py $OUT > $OUT/traffic.json
cat $OUT/bench.json; head -6 $OUT/kernel_stats.csv; cat $OUT/traffic.json
