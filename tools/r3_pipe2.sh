#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --channels 512 --nsamp 1048576 --steps 10 --warmup 3 --no-cpu-baseline --no-few --no-extra --no-check 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1: %.3f ms per call (launch avg %.3f)' % (d['ms_per_step'], d['roofline']['launch_ms_avg']))"; }
run plain
GPU_MAX_HW_QUEUES=8 run hwq8
rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $R/bench.py --channels 512 --nsamp 1048576 --steps 10 --warmup 3 --no-cpu-baseline --no-few --no-extra --no-check 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('under rocprofv3: %.3f ms per call' % d['ms_per_step'])"
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pp/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'psk' in r['Kernel_Name'] and 'probe' not in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last call: from the first front kernel after the previous seq kernel
seq=[i for i,r in enumerate(rows) if 'seq_kernel' in r['Kernel_Name']]
last=rows[seq[-2]+1:seq[-1]+1]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print('  %-28.28s start %8.1f  end %8.1f' % (r['Kernel_Name'].split('(')[0].replace('void psk::','').replace('psk::',''), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3))
PY
