#!/bin/bash
# usage (GPU box): tools/ab_bench_args.sh <reps> "<bench args>" "<EXTRA flags A>" "<EXTRA flags B>" ...
# like ab_bench.sh, with extra bench.py arguments (a configuration other than the headline)
reps=$1; shift
bargs=$1; shift
cd $GRAFT_REPO_ROOT/psk_soft_amd/csrc
i=0
for v in "$@"; do
  rm -f obj/psk_fast_S8_H1_E0.o
  make -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; exit 1; }
  cp ../libpsk_soft_hip.so /tmp/lib_variant_$i.so
  i=$((i+1))
done
n=$i
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $reps); do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/lib_variant_$i.so psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check --no-few $bargs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $i %.4f %d'%(d['roofline']['launch_ms_avg'], d['kernel_stats']['fit_chain_blocks']))"
  done
done | tee /tmp/ab.log
python - "$@" <<'PY'
import sys
from collections import defaultdict
acc=defaultdict(list); ch={}
for l in open('/tmp/ab.log'):
    _,i,ms,c=l.split(); acc[int(i)].append(float(ms)); ch[int(i)]=c
for i,v in sorted(acc.items()):
    print('VARIANT %d [%s]: mean %.4f ms  min %.4f  max %.4f  n=%d chain_blocks=%s'%(i, sys.argv[1+i], sum(v)/len(v), min(v), max(v), len(v), ch[i]))
PY
