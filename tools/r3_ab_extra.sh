#!/bin/bash
# usage (GPU box): tools/r3_ab_extra.sh "<bench args>" "<EXTRA flags A>" "<EXTRA flags B>" ...  -- A/B of builds of psk_fast_S8_H1_E0, 3 rounds
bargs=$1; shift
R=$GRAFT_REPO_ROOT
cp $R/psk_soft_amd/libpsk_soft_hip.so /tmp/lib_orig.so
cd $R/psk_soft_amd/csrc
i=0
for v in "$@"; do
  rm -f obj/${OBJ:-psk_fast_S8_H1_E0}.o
  make -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; exit 1; }
  cp ../libpsk_soft_hip.so /tmp/lib_variant_$i.so
  i=$((i+1))
done
n=$i
cd $R
for r in 1 2 3; do
  for i in $(seq 0 $((n-1))); do
    cp /tmp/lib_variant_$i.so psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-few --no-extra $bargs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('RUN $i ms_per_step %.4f launch %.4f check %s' % (d['ms_per_step'], d['roofline']['launch_ms_avg'], d.get('check',{}).get('soft_phase_bit_identical')))"
  done
done
cp /tmp/lib_orig.so $R/psk_soft_amd/libpsk_soft_hip.so
rm -f $R/psk_soft_amd/csrc/obj/${OBJ:-psk_fast_S8_H1_E0}.o
