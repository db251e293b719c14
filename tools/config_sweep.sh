#!/bin/bash
# usage (GPU box): tools/config_sweep.sh > gpurun_out/sweep.txt -- the configurations of DESIGN.md section 5,
# one bench.py run each (--check: two channels replayed through the oracle), one line per configuration
cd $GRAFT_REPO_ROOT
run() {
  python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-few --no-extra --check "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; k = d['kernel_stats']; c = d['check']
print('%-44s ms %.3f  frac %.3f  of-ceiling %.3f  exact-blocks %d  extra-passes %d  seq %d  check bits %s soft %.1e' % (
  ' '.join(sys.argv[1:]) or '(headline)', r['launch_ms_avg'], r['frac'], r['frac_of_empirical_read_ceiling'],
  k['timing_exact_blocks'], k['unwrap_extra_passes'], k['channels_sequential'], c['bits_index_exact'], c['soft_max_rel_err']))" "$@"
}
run
run --M 2
run --M 8
run --M 8 --S 10
run --S 16
run --S 4
run --S 3
run --S 6
run --S 7
run --S 12
run --numAvg 200
run --numAvg 400
run --mixed
run --cfo 0.1
run --sigma 0.1
run --sigma 0.3
run --M 8 --S 10 --sigma 0.1
run --nsamp 4096
run --nsamp 16384
run --nsamp 65536
run --S 2
run --S 5
run --S 9
run --S 11
run --S 13
run --S 14
run --S 15
run --S 20
run --S 24
run --S 32
run --numAvg 600
run --numAvg 1024
run --phaseAvg 500
run --phaseAvg 1000
run --phase0
