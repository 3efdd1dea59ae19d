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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --cpu-sample 0 --no-e2e --no-traffic "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --cpu-sample 0 --no-e2e --no-traffic --steps 5 --warmup 1 "$@" > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --cpu-sample 0 --no-e2e --no-traffic --steps 5 --warmup 1 "$@" > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# instruction mix and pipe activity of the same kernels (two passes: the counters do not fit one)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_insts -- python3 $R/bench.py --cpu-sample 0 --no-e2e --no-traffic --steps 5 --warmup 1 "$@" > $OUT/pmc_insts.json 2> $OUT/pmc_insts.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_pipes -- python3 $R/bench.py --cpu-sample 0 --no-e2e --no-traffic --steps 5 --warmup 1 "$@" > $OUT/pmc_pipes.json 2> $OUT/pmc_pipes.err
python3 $R/tools/prof_summary.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/traffic_summary.py $OUT > $OUT/traffic.json
cat $OUT/bench.json; head -6 $OUT/kernel_stats.csv; cat $OUT/traffic.json
