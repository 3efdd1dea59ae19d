#!/bin/bash
# Host-side AddressSanitizer run (CPU only; GPU sanitizers are not available on the pool): the host translation units
# compiled with -fsanitize=address, linked with the regular device objects into /tmp/lsq_asan/liblesseq_hip.so, and the
# CPU test suite run against it (annotation loaders, event compiler and device plan, host MRF parser, classify, formatters,
# shard bounds, gathered-block unpacking).  usage: tools/asan_cpu_tests.sh   (after `make -C lesseq_amd/csrc`)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/lsq_asan; mkdir -p $O
cd $R/lesseq_amd/csrc
for f in lsq_annot lsq_mrf lsq_cli lsq_synth; do
	g++ -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-omit-frame-pointer -ffp-contract=off -pthread -c -o $O/$f.o $f.cpp &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fsanitize=address -o $O/liblesseq_hip.so $O/lsq_annot.o $O/lsq_mrf.o $O/lsq_cli.o $O/lsq_synth.o \
	../_build/lsq_device.o ../_build/lsq_count.o ../_build/lsq_em.o ../_build/lsq_ingest.o ../_build/lsq_replay.o -pthread
cp ../_build/liblesseq_rccl.so $O/
cd $R
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) LSQ_LIB=$O/liblesseq_hip.so \
	python -m pytest tests -x -q -m "not gpu" -k "not exports and not crosses_the_boundary"      # (the boundary test caps the address space, which the sanitizer's shadow memory does not survive)
