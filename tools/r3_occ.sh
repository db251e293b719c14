#!/bin/bash
# wave-scan kernel alone (no time tiling) at 1, 2, 3, 4 waves per SIMD: launch time vs channels
cd $GRAFT_REPO_ROOT
for c in 256 1024 2048 3072 4096; do
  PSK_SOFT_TIME_TILED=0 python bench.py --channels $c --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('channels $c  launch ms %.3f' % d['roofline']['launch_ms_avg'])"
done
