#!/bin/bash
# A/B of the re-read variant (H == 0) on the mixed configuration and on a machine-filling numAvg = 400 batch
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-reread}
mkdir -p $out
cd $R
for v in 0 1; do
  PSK_SOFT_REREAD=$v python3 bench.py --mixed --check --no-cpu-baseline --no-few --no-extra --steps 20 --warmup 10 > $out/mixed_$v.json 2> $out/mixed_$v.err
  PSK_SOFT_REREAD=$v python3 bench.py --numAvg 400 --check --no-cpu-baseline --no-few --no-extra --steps 20 --warmup 10 > $out/a400_$v.json 2> $out/a400_$v.err
  PSK_SOFT_REREAD=$v python3 bench.py --numAvg 200 --check --no-cpu-baseline --no-few --no-extra --steps 20 --warmup 10 > $out/a200_$v.json 2> $out/a200_$v.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["ms_per_step"], d["roofline"]["frac"], d.get("check"))
    except Exception as e: print(f, "ERR", e, open(f.replace('.json','.err')).read()[-500:])
PY
