#!/bin/bash
# small-packet anatomy on the GPU box: call_cost + a kernel trace of the same loop
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${1:-small}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/call_cost.py 4096 4096 > $out/call_cost.txt 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/call_cost.py 4096 4096 > $out/trace.log 2>&1
f=$(ls $out/trace/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv
f=$(ls $out/trace/*/*memory_copy_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/memcpy_stats.csv
rm -rf $out/trace
cat $out/call_cost.txt; head -8 $out/kernel_stats.csv | cut -c1-200
