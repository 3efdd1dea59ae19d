#!/bin/bash
# Developer aid (GPU box): wall-clock and phase times of the solve executable on C3 from MRF text, three runs with the
# early exit and three with the orderly teardown.
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import lesseq_amd as L
from bench import WORKLOADS
W = WORKLOADS["c3"]
spec = L.SynthSpec(W["seed"], W["n_events"], W["n_reads"], W["R"], W["n_chrom"], L.EVENT_TYPES)
os.makedirs("/dev/shm/lsqc3", exist_ok=True)
L.synth_write(spec, "/dev/shm/lsqc3", "w", write_mrf=True)
print("written")
PY
D=/dev/shm/lsqc3
for i in 1 2 3 4 5 6; do
if [ $i -gt 3 ]; then export LSQ_CLI_TEARDOWN=1; fi
T0=$(date +%s.%N); LSQ_CLI_TIMING=1 lesseq_amd/bin/solve 0 w ./ LH_GENE_TXT $D/w.interval UCSC_GENE2ISOFORM $D/w.map 0 100000000 MRF_SINGLE SHORT_READ 100 $D/w.mrf 10000000000 > /dev/null 2> gpurun_out/cli_t$i.log || true
T1=$(date +%s.%N); tail -3 gpurun_out/cli_t$i.log; python3 -c "print(\"wall\", $T1 - $T0)"
done
rm -rf /dev/shm/lsqc3
