#!/bin/bash
# Developer helper: builds the release library of another commit as lesseq_amd/_build/liblesseq_hip_<name>.so, for same-box
# A/B runs against the working tree (LSQ_LIB=... python tools/step_ab.py).  usage: tools/build_ref.sh <git ref> <name>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
ref=$1; name=$2
T=$(mktemp -d)
git -C $R archive $ref lesseq_amd/csrc include examples | tar -x -C $T
make -C $T/lesseq_amd/csrc -j8 ../_build/liblesseq_hip.so > $T/make.log 2>&1 || { tail -20 $T/make.log; exit 1; }
cp $T/lesseq_amd/_build/liblesseq_hip.so $R/lesseq_amd/_build/liblesseq_hip_$name.so
rm -rf $T
echo built lesseq_amd/_build/liblesseq_hip_$name.so from $ref
