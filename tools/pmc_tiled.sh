#!/bin/bash
# issue counters of the time-tiled path's launches (GPU box): tools/pmc_tiled.sh [channels] [tag]; summary under gpurun_out/<tag>/
c=${1:-64}
tag=${2:-pmc_tiled}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/p1 -- python3 $R/bench.py --channels $c --nsamp 1048576 --steps 4 --warmup 3 --no-cpu-baseline --no-check --no-few --no-extra > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python3 $R/bench.py --channels $c --nsamp 1048576 --steps 4 --warmup 3 --no-cpu-baseline --no-check --no-few --no-extra > $out/p2.log 2>&1
cd $R && python3 tools/pmc_summary.py $out/p1 $out/p2 > $out/summary.txt
rm -rf $out/p1 $out/p2
