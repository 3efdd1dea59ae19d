#!/bin/bash
# Round evidence for the loader chain (run on the GPU box through gpurun): tools/ingest_bench.py, then rocprofv3 kernel-trace
# stats and PMC passes (FETCH_SIZE; WRITE_SIZE; instruction mix; pipe activity) of the same command.
# usage: bash tools/ingest_prof.sh <tag> [workload] [sorted 0|1]
set -e
TAG=$1; WL=${2:-c3}; SORTED=${3:-0}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ingest_$TAG
mkdir -p $OUT
cd $R
python3 tools/ingest_bench.py $WL 3 $SORTED > $OUT/ingest.json 2> $OUT/ingest.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/ingest_bench.py $WL 2 $SORTED > /dev/null 2> $OUT/rocprof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/ingest_bench.py $WL 1 $SORTED > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/tools/ingest_bench.py $WL 1 $SORTED > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_insts -- python3 $R/tools/ingest_bench.py $WL 1 $SORTED > /dev/null 2> $OUT/pmc_insts.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_pipes -- python3 $R/tools/ingest_bench.py $WL 1 $SORTED > /dev/null 2> $OUT/pmc_pipes.err
python3 $R/tools/prof_summary.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv | cut -c1-200; grep -v "count_fast\|em_\|pack_\|cleanup\|recount" $OUT/pmc_summary.txt | head -150
