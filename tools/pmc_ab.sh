#!/bin/bash
# usage: pmc_ab.sh lib1 lib2 ... (names under lesseq_amd/_build: hip, hip_base2, ...)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export LSQ_LIB=$R/lesseq_amd/_build/liblesseq_$lib.so
  rm -rf $R/gpurun_out/pmcab/$lib
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmcab/$lib -- python3 $R/tools/kbench.py c3 8192 3 > /dev/null 2>&1
  python3 - <<PY
import csv,collections,glob
f=glob.glob('$R/gpurun_out/pmcab/$lib/*/*_counter_collection.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'count_fast' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
print('$lib', ' '.join('%s=%.2fM'%(k.replace('SQ_',''),sum(v)/len(v)/1e6) for k,v in sorted(d.items())))
PY
done
