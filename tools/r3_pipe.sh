#!/bin/bash
# pipelined mode of the time-tiled path against the other ways of running C channels x N samples:  tools/r3_pipe.sh N C [C ...]
cd $GRAFT_REPO_ROOT
N=$1; shift
for C in "$@"; do
  for v in "PSK_SOFT_PIPELINED=1" "PSK_SOFT_PIPELINED=0" "PSK_SOFT_TIME_TILED=0"; do
    env $v python bench.py --channels $C --nsamp $N --steps 10 --warmup 3 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%5d channels x %8d samples  %-22s %7.3f ms per call  %5.1f %% of the read roofline  check %s  tiled %d pfit %d' % ($C, $N, '$v', d['ms_per_step'], 8.0*$C*$N/(d['ms_per_step']*1e-3)/8e12*100, d['check']['soft_phase_bit_identical'], d['kernel_stats']['channels_tiled'], d['kernel_stats']['channels_parallel_fit']))"
  done
done
