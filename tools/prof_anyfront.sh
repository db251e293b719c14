#!/bin/bash
# Per-kernel time of calls that go through the run-time front stage (GPU box): samplesPerBaud 40 and numAvg 2000, one channel and a batch.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "--S 40 --channels 1 --nsamp 1048576" "--S 8 --numAvg 2000 --channels 1 --nsamp 1048576" "--S 40 --channels 4096 --nsamp 65536"; do
  out=$R/gpurun_out/anyfront/c$i
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-few > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== bench.py $cfg --steps 10 --warmup 3  ($(grep -o '"ms_per_step": [0-9.]*' $out/log.txt) under the profiler)"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "psk" in r["Name"] or "pf_" in r["Name"]:
        print("%-60.60s calls %4s  avg %10.1f us  %5s %%" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  fi
  i=$((i+1))
done
