#!/bin/bash
# When do the waves of the two window classes of the mixed batch run?  (in-kernel 100 MHz stamps, no profiler)
# usage (GPU box): tools/r3_mixed_stamps.sh <tag> [env assignments for the bench, e.g. GPU_MAX_HW_QUEUES=8]
tag=${1:-mixed_stamps}; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cp $R/psk_soft_amd/libpsk_soft_hip.so /tmp/lib_orig.so
cd $R/psk_soft_amd/csrc
rm -f obj/psk_fast_S8_H1_E0.o obj/psk_fast_S8_H4_E0.o
make -j16 EXTRA="-DPSK_DIAG_STAMP" > /tmp/make.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/make.log; exit 1; }
cd $R
env "$@" python bench.py --mixed --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra --no-check --stamps $out/stamps.npy 2>/dev/null | tail -1 | python -c "
import sys,json,numpy as np
d=json.loads(sys.stdin.read())
a=np.load('$out/stamps.npy'); t0,t1=a[:,0],a[:,1]
base=t0.min()
s=((t0-base)%(1<<32))/100.0; e=((t1-base)%(1<<32))/100.0
c=np.arange(len(s)); wide=((c//9)%3)==2
print('launch_ms_avg %.3f' % d['roofline']['launch_ms_avg'])
for name,m in (('numAvg 400 (H=4)',wide),('numAvg <= 128 (H=1)',~wide)):
    print('%-20s waves %4d  start us: min %7.0f median %7.0f p90 %7.0f max %7.0f | end us: min %7.0f median %7.0f max %7.0f | life us: median %6.0f' % (name, m.sum(), s[m].min(), np.median(s[m]), np.percentile(s[m],90), s[m].max(), e[m].min(), np.median(e[m]), e[m].max(), np.median(e[m]-s[m])))
" | tee $out/summary.txt
cp /tmp/lib_orig.so $R/psk_soft_amd/libpsk_soft_hip.so
rm -f $R/psk_soft_amd/csrc/obj/psk_fast_S8_H1_E0.o $R/psk_soft_amd/csrc/obj/psk_fast_S8_H4_E0.o
