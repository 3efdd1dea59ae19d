#!/bin/bash
# Developer profiling recipe (run on the GPU box through gpurun): kernel trace + two PMC passes
# of the count kernel on one workload.  usage: tools/prof.sh <tag> <workload> [budget] [mult]
set -e
TAG=$1; WL=${2:-c3}; BUD=${3:-4096}; MULT=${4:-1}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/kbench.py $WL $BUD $MULT > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $R/tools/kbench.py $WL $BUD $MULT > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 $R/tools/kbench.py $WL $BUD $MULT > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $R/tools/kbench.py $WL $BUD $MULT > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -- python3 $R/tools/kbench.py $WL $BUD $MULT > $OUT/pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
python3 $R/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
