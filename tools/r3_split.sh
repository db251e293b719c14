#!/bin/bash
# one joined call of the mixed batch (configs[4]) cut into n pieces in time: PSK_SOFT_SPLIT_CLASSES=n
cd $GRAFT_REPO_ROOT
for n in 0 2 3 4 6 8; do
  PSK_SOFT_SPLIT_CLASSES=$n python bench.py --mixed --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pieces $n: %.3f ms per step (every call joined), %.1f %% of the read roofline, check %s' % (d['ms_per_step'], 8.0*4096*262144/(d['ms_per_step']*1e-3)/8e12*100, d['check']['soft_phase_bit_identical']))"
done
