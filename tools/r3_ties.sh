#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "--M 8 --S 10 --numAvg 200" "--M 8 --S 10 --numAvg 400" "--S 12 --numAvg 400" "--S 6 --numAvg 200" "--S 8 --numAvg 400 --sigma 0.1"; do
  python bench.py $args --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_stats']; print('$args: %.3f ms, exact-timing channels %d, exact blocks %d, check %s' % (d['ms_per_step'], k['channels_exact_timing'], k['timing_exact_blocks'], d['check']['soft_phase_bit_identical']))"
done
for args in "--numAvg 400" "--numAvg 200" "--numAvg 600" "--mixed" "--numAvg 600 --sigma 0.1" "--S 12 --numAvg 600"; do
  python bench.py $args --steps 10 --warmup 5 --no-cpu-baseline --no-few --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_stats']; print('$args: %.3f ms, exact-timing channels %d, exact blocks %d, check %s' % (d['ms_per_step'], k['channels_exact_timing'], k['timing_exact_blocks'], d['check']['soft_phase_bit_identical']))"
done
