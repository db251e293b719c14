#!/bin/bash
# Per-kernel MEDIAN time of few-channel calls in the steady state, and the timeline of the last call (GPU box):
#   tools/prof_tiled2.sh "64 512" 1048576 [tag]
R=$GRAFT_REPO_ROOT
CH=${1:-"64"}
N=${2:-1048576}
tag=${3:-prof_tiled2}
cd /tmp && export TMPDIR=/tmp
for c in $CH; do
  out=$R/gpurun_out/$tag/c$c
  mkdir -p $out
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --channels $c --nsamp $N --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-few --no-extra > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  echo "== $c channels x $N samples"
  python3 - "$f" <<'PY'
import csv, sys, statistics, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "psk" in r["Kernel_Name"] and "read_probe" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows: d[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1])):
    print("%-44.44s calls %4d  median %8.1f us  min %8.1f" % (k.replace("void psk::", "").replace("psk::", ""), len(v), statistics.median(v), min(v)))
    tot += statistics.median(v)
print("sum of medians %.1f us" % tot)
# last call: from its front kernel to its last kernel
idx = [i for i, r in enumerate(rows) if "tile_front" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("   %-40.40s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0].replace("void psk::", "").replace("psk::", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
  rm -rf $out
done
