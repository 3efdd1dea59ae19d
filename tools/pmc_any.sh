#!/bin/bash
# Developer recipe (GPU box): counter passes over any command; mean per dispatch of kernels whose name contains a pattern.
# usage: tools/pmc_any.sh <kernel-name pattern> "<command>" "<counters of pass 1>" ["<counters of pass 2>" ...]
R=$GRAFT_REPO_ROOT
PAT=$1; CMD=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcany/p$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcany/p$i -- $CMD > /dev/null 2>&1
  python3 - <<PY
import csv,collections,glob
fs=glob.glob('$R/gpurun_out/pmcany/p$i/*/*_counter_collection.csv')
d=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if '$PAT' in r['Kernel_Name']: d[(r['Kernel_Name'].split('(')[0].split('::')[-1][:28], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print('%-28s %-28s %16.1f  (n=%d)'%(k[0],k[1],sum(v)/len(v),len(v)))
PY
done
rm -rf $R/gpurun_out/pmcany
