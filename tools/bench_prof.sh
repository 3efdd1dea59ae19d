#!/bin/bash
# Round-end evidence: bench.py default run + rocprofv3 kernel-trace stats of the same command.
# usage (GPU box): bash tools/bench_prof.sh <tag>
set -e
TAG=$1
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cat $OUT/bench.json; cat $OUT/kernel_stats.csv | head -6
