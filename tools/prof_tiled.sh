#!/bin/bash
# Per-kernel time of few-channel calls (GPU box): tools/prof_tiled.sh "1 64 512" 1048576 [tag]
R=$GRAFT_REPO_ROOT
CH=${1:-"1 64"}
N=${2:-1048576}
tag=${3:-prof_tiled}
cd /tmp && export TMPDIR=/tmp
for c in $CH; do
  out=$R/gpurun_out/$tag/c$c
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --channels $c --nsamp $N --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-few > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $c channels x $N samples"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60.60s calls %4s  avg %10.1f us  %5s %%" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  fi
done
