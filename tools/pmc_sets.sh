#!/bin/bash
# Developer recipe (GPU box): several counter passes over tools/kbench.py c3 for the release library; mean per dispatch of the
# count kernel.  usage: tools/pmc_sets.sh "<counters of pass 1>" "<counters of pass 2>" ...
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export LSQ_LIB=${LSQ_LIB:-$R/lesseq_amd/_build/liblesseq_hip.so}
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcsets/p$i
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcsets/p$i -- python3 $R/tools/kbench.py c3 8192 3 > /dev/null 2>&1
  python3 - <<PY
import csv,collections,glob
fs=glob.glob('$R/gpurun_out/pmcsets/p$i/*/*_counter_collection.csv')
d=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'count_fast' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(d.items()): print('%-34s %14.1f  (n=%d)'%(k,sum(v)/len(v),len(v)))
PY
done
