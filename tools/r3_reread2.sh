#!/bin/bash
# the variant without window history (H == 0: the symbols that leave the window are read again) against eight blocks of history in
# registers (numAvg 513 ... 1024):  tools/r3_reread2.sh  (4096 channels, several samplesPerBaud)
cd $GRAFT_REPO_ROOT
for sm in "2 4" "4 4" "6 4" "10 8" "12 4" "16 4"; do set -- $sm; for v in 0 1; do
  PSK_SOFT_REREAD=$v python bench.py --S $1 --M $2 --numAvg 600 --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('samplesPerBaud $1 M $2 numAvg 600 reread $v: %.3f ms, check %s' % (d['ms_per_step'], d['check']['soft_phase_bit_identical']))"
done; done
