#!/bin/bash
# does the bench line's launch time agree with the profiler's kernel time on the same box?  (unprofiled, profiled, unprofiled)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-agree}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
one() { python3 $R/bench.py --no-cpu-baseline --no-few --no-extra --no-check 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unprofiled: launch_ms_avg %.4f min %.4f ms_per_step %.4f' % (d['roofline']['launch_ms_avg'], d['roofline']['launch_ms_min'], d['ms_per_step']))"; }
one; one
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-cpu-baseline --no-few --no-extra --no-check > $out/stats.log 2>&1
python3 - $out <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/*/*_kernel_trace.csv')[0]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f)) if 'psk_fast_kernel<8, 1, false>' in r['Kernel_Name']]
print('profiled: last 50 launches mean %.4f min %.4f' % (sum(d[-50:])/50, min(d[-50:])))
PY
rm -rf $out/stats
one
