#!/bin/bash
# usage (GPU box): tools/ab_stamps.sh <tag> <reps> "<bench args>" "<EXTRA flags A>" "<EXTRA flags B>" ...
# A/B of builds of the headline instantiation (psk_fast_S8_H1_E0), each with -DPSK_DIAG_STAMP: launch time AND when the
# waves of the last launch ended, by dispatch quarter.  Output: gpurun_out/<tag>/
tag=$1; shift
reps=$1; shift
bargs=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cp $R/psk_soft_amd/libpsk_soft_hip.so /tmp/lib_orig.so
cd $R/psk_soft_amd/csrc
i=0
for v in "$@"; do
  rm -f obj/psk_fast_S8_H1_E0.o
  make -j16 EXTRA="-DPSK_DIAG_STAMP $v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; exit 1; }
  cp ../libpsk_soft_hip.so /tmp/lib_variant_$i.so
  i=$((i+1))
done
n=$i
cd $R
for r in $(seq 1 $reps); do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/lib_variant_$i.so psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-few $bargs --stamps $out/stamps_$i.npy 2>/dev/null | tail -1 | python -c "
import sys,json,numpy as np
d=json.loads(sys.stdin.read())
a=np.load('$out/stamps_$i.npy'); t0,t1=a[:,0],a[:,1]
e=((t1-t0.min())%(1<<32))/100.0
q=len(e)//4
print('RUN $i %.4f %d %s | end us by dispatch quarter %s max %.0f' % (d['roofline']['launch_ms_avg'], d['kernel_stats']['fit_chain_blocks'], d.get('check',{}).get('soft_phase_bit_identical'), [int(e[k*q:(k+1)*q].mean()) for k in range(4)], e.max()))"
  done
done | tee $out/ab.log
python - "$@" <<PY
import sys
from collections import defaultdict
acc=defaultdict(list)
for l in open('$out/ab.log'):
    f=l.split()
    acc[int(f[1])].append(float(f[2]))
for i,v in sorted(acc.items()):
    print('VARIANT %d [%s]: mean %.4f ms  min %.4f  max %.4f  n=%d'%(i, sys.argv[1+i], sum(v)/len(v), min(v), max(v), len(v)))
PY
cp /tmp/lib_orig.so $R/psk_soft_amd/libpsk_soft_hip.so
