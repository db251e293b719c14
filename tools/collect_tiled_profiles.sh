#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics of the time-tiled path (1, 64 and 256 channels x 2^20 samples per call),
# the channel sweep with and without it, and the bench line with `few_channels`.  Outputs under gpurun_out/<tag>/.
tag=${1:-tiled}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd $R
tools/tiled_sweep.sh "1 16 64 256 512" 1048576 > $out/sweep_channels.txt 2>&1
tools/tiled_sweep2.sh "1 64 512" "16384 131072" > $out/sweep_short_calls.txt 2>&1
python3 bench.py > $out/bench.json 2> $out/bench.err
tools/prof_tiled.sh "1 64 256" 1048576 $tag/prof > $out/kernel_stats.txt 2>&1
for c in 1 64 256; do
  f=$(find $out/prof/c$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_c$c.csv
done
rm -rf $out/prof
tail -c 300 $out/bench.json
