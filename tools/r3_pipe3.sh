#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "X=0" "PSK_SOFT_PIPE_CUMASK=2" "PSK_SOFT_PIPE_CUMASK=4" "PSK_SOFT_PIPE_CUMASK=8"; do
  env $v python bench.py --channels 512 --nsamp 1048576 --steps 10 --warmup 3 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v: %.3f ms per call, check %s' % (d['ms_per_step'], d['check']['soft_phase_bit_identical']))"
done
