#!/bin/bash
# Developer helper: builds lesseq_amd/_build/liblesseq_hip_<name>.so with extra -D flags on the
# device files, for same-box A/B runs (LSQ_LIB=... python tools/kbench.py ...).
# usage: tools/build_variant.sh <name> [-DLSQ_FAST_WAVES=6 ...]
set -e
cd "$(dirname "$0")/../lesseq_amd/csrc"
name=$1; shift
B=../_build
OBJS=""
for f in lsq_device lsq_count lsq_em lsq_ingest lsq_replay; do
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DLSQ_DEV "$@" -c -o $B/${f}_$name.o $f.hip &
	OBJS="$OBJS $B/${f}_$name.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $B/liblesseq_hip_$name.so $B/lsq_annot.o $B/lsq_mrf.o $B/lsq_cli.o $B/lsq_synth.o $OBJS -pthread
echo built $B/liblesseq_hip_$name.so
