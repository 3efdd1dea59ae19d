#!/bin/bash
# quick kernel check on the box: parity subset + kbench timing
mkdir -p gpurun_out/kb
python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "synthetic_parity or adversarial or event_shaped or geometry or many_seg or many_block or long_introns" > gpurun_out/kb/test.log 2>&1 || { tail -20 gpurun_out/kb/test.log; exit 1; }
tail -1 gpurun_out/kb/test.log
KB_ABLATE=0 python tools/kbench.py c3 8192 2 2>&1 | grep "^abl"
KB_ABLATE=0 python tools/kbench.py c2 8192 2 2>&1 | grep "^abl"
KB_ABLATE=0 python tools/kbench.py c5s 8192 4 2>&1 | grep "^abl"
