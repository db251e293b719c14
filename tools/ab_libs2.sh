#!/bin/bash
# usage (GPU box): tools/ab_libs2.sh <libA.so> <libB.so> -- "<bench args 1>" "<bench args 2>" ...   (A/B of two prebuilt libraries)
A=$1; B=$2; shift 3
cd $GRAFT_REPO_ROOT
cp psk_soft_amd/libpsk_soft_hip.so /tmp/lib_cur.so
for args in "$@"; do
  for v in A B; do
    [ $v = A ] && cp $A psk_soft_amd/libpsk_soft_hip.so || cp $B psk_soft_amd/libpsk_soft_hip.so
    python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v [$args] ms %.3f check %s chain %d' % (d['roofline']['launch_ms_avg'], d['check']['soft_phase_bit_identical'], d['kernel_stats']['fit_chain_blocks']))"
  done
done
cp /tmp/lib_cur.so psk_soft_amd/libpsk_soft_hip.so
