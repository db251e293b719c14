#!/bin/bash
# tiled (forced, PSK_SOFT_TIME_TILED=2) against untiled (0) over channels x samples per call:
#   tools/tiled_sweep2.sh "256 1024 2048" "16384 131072"
for n in $2; do for c in $1; do
  for t in 0 2; do
    PSK_SOFT_TIME_TILED=$t python bench.py --channels $c --nsamp $n --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-few 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('nsamp %8d channels %5d  tiled %d  %.3f ms  %.1f Msamples/s' % ($n, $c, $t, d['ms_per_step'], d['value']))"
  done
done; done
