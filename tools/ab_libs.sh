#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <reps> <libA.so> <libB.so> ...   (paths relative to the repo root)
# Alternates bench.py between prebuilt copies of libpsk_soft_hip.so (e.g. one built from HEAD and one
# from the working tree, both placed under psk_soft_amd/ab/ before the gpurun call).
reps=$1; shift
cd $GRAFT_REPO_ROOT
cp psk_soft_amd/libpsk_soft_hip.so /tmp/lib_keep.so
for r in $(seq 1 $reps); do
  i=0
  for lib in "$@"; do
    cp $lib psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --check $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RUN $i %.4f %s %.4f'%(d['roofline']['launch_ms_avg'], d['check']['bits_index_exact'], d['ms_per_step']))"
    i=$((i+1))
  done
done | tee /tmp/ab.log
cp /tmp/lib_keep.so psk_soft_amd/libpsk_soft_hip.so
python - "$@" <<'PY'
import sys
from collections import defaultdict
acc=defaultdict(list); wall=defaultdict(list)
for l in open('/tmp/ab.log'):
    p=l.split(); acc[int(p[1])].append(float(p[2])); wall[int(p[1])].append(float(p[4]))
for i,v in sorted(acc.items()):
    w=wall[i]
    print('LIB %d [%s]: launch mean %.4f ms  min %.4f  max %.4f | wall per step mean %.4f ms  n=%d'%(i, sys.argv[1+i], sum(v)/len(v), min(v), max(v), sum(w)/len(w), len(v)))
PY
