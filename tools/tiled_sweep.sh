#!/bin/bash
# Few-channel calls with and without the time-tiled kernels (PSK_SOFT_TIME_TILED = 0 / 1):
#   tools/tiled_sweep.sh "1 4 16 64 256 512 1024" 1048576
# prints channels, ms per call and Msamples/s for both.
CH=${1:-"1 16 64 256"}
N=${2:-1048576}
for c in $CH; do
  for t in 0 1; do
    PSK_SOFT_TIME_TILED=$t python bench.py --channels $c --nsamp $N --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-few 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('channels %5d  tiled %d  %.3f ms  %.1f Msamples/s' % ($c, $t, d['ms_per_step'], d['value']))"
  done
done
