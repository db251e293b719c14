#!/bin/bash
# Where the time of a short call goes inside the wave-scan kernel (100 MHz stamps: start, after the prologue, after the loop, end)
# usage (GPU box): tools/r3_small_stamps.sh <tag> [nsamp]
tag=${1:-small_stamps}; ns=${2:-4096}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cp $R/psk_soft_amd/libpsk_soft_hip.so /tmp/lib_orig.so
cd $R/psk_soft_amd/csrc
rm -f obj/psk_fast_S8_H1_E0.o
make -j16 EXTRA="-DPSK_DIAG_STAMP -DPSK_DIAG_STAMP2" > /tmp/make.log 2>&1 || { echo "BUILD FAILED"; tail -5 /tmp/make.log; exit 1; }
cd $R
python bench.py --nsamp $ns --steps 50 --warmup 20 --no-cpu-baseline --no-few --no-extra --no-check --stamps $out/stamps.npy 2>/dev/null | tail -1 | python -c "
import sys,json,numpy as np
d=json.loads(sys.stdin.read())
a=np.load('$out/stamps.npy').astype(np.int64); t0,t3,t2,t1=a[:,0],a[:,1],a[:,2],a[:,3]
base=t0.min()
f=lambda t:((t-base)%(1<<32))/100.0
s,p,l,e=f(t0),f(t1),f(t2),f(t3)
print('ms_per_step %.4f launch_ms_avg %.4f' % (d['ms_per_step'], d['roofline']['launch_ms_avg']))
q=lambda v:'min %6.1f med %6.1f p90 %6.1f max %6.1f' % (v.min(), np.median(v), np.percentile(v,90), v.max())
print('wave start      us:', q(s)); print('prologue        us:', q(p-s)); print('loop            us:', q(l-p)); print('epilogue        us:', q(e-l)); print('wave end        us:', q(e))
" | tee $out/summary.txt
cp /tmp/lib_orig.so $R/psk_soft_amd/libpsk_soft_hip.so
rm -f $R/psk_soft_amd/csrc/obj/psk_fast_S8_H1_E0.o
