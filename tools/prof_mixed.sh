#!/bin/bash
# Kernel trace of the mixed batch (GPU box): per-launch start / end of each class's kernel, classes side by side and one after the other.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in "" "--serial-classes"; do
  tag=mixed${mode:+_serial}
  out=$R/gpurun_out/$tag
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --mixed $mode --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-few --no-extra > $out/log.txt 2>&1
  tail -c 600 $out/log.txt | head -c 400; echo
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  echo "== $tag"
  python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "psk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-12:]:
    print("%-48.48s grid %7s  start %9.1f us  dur %8.1f us" % (r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
done
