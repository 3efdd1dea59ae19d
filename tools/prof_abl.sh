#!/bin/bash
# Developer recipe (GPU box): instruction counts of the count kernel under the developer build's ablation switches.
# usage: tools/prof_abl.sh <tag> <workload> <mult> <abl> [<abl> ...]
set -e
TAG=$1; WL=$2; MULT=$3; shift 3
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/abl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for abl in "$@"; do
	export KB_ABLATE=$abl
	rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a$abl/pmc1 -- python3 $R/tools/kbench.py $WL 8192 $MULT > $OUT/a$abl.log 2>&1
	python3 $R/tools/prof_summary.py $OUT/a$abl 2>/dev/null | grep "count_fast_kernel" | sed "s/^/abl=$abl /" >> $OUT/summary.txt
done
cat $OUT/summary.txt
