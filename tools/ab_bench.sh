#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <reps> "<EXTRA flags A>" "<EXTRA flags B>" ...
# Builds each variant of the headline instantiations once, then runs bench.py on them in ALTERNATING
# order <reps> times and prints the per-variant mean: run-to-run drift on a box (1-2 %) is larger
# than most effects worth measuring.
reps=$1; shift
cd $GRAFT_REPO_ROOT/psk_soft_amd/csrc
i=0
for v in "$@"; do
  rm -f obj/psk_fast_S8_H1_E0.o obj/psk_fast_S8_H1_E1.o
  make -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; exit 1; }
  cp ../libpsk_soft_hip.so /tmp/lib_variant_$i.so
  i=$((i+1))
done
n=$i
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $reps); do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/lib_variant_$i.so psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $i %.4f'%d['roofline']['launch_ms_avg'])"
  done
done | tee /tmp/ab.log
python - "$@" <<'PY'
import sys
from collections import defaultdict
acc=defaultdict(list)
for l in open('/tmp/ab.log'):
    _,i,ms=l.split(); acc[int(i)].append(float(ms))
for i,v in sorted(acc.items()):
    print('VARIANT %d [%s]: mean %.4f ms  min %.4f  max %.4f  n=%d'%(i, sys.argv[1+i], sum(v)/len(v), min(v), max(v), len(v)))
PY
