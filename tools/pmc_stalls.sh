#!/bin/bash
# stall-oriented counter passes for one kernel (GPU box); output under gpurun_out/<tag>/
# usage: pmc_stalls.sh [tag] [kernel pattern, default "8, 1, false"] [extra bench.py arguments]
tag=${1:-stalls}
pat=${2:-8, 1, false}
extra=${3:-}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-few --no-extra $extra > $out/$name.log 2>&1; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run p2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH
run p3 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM
run p4 TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE
cd $R && python3 tools/pmc_summary.py $out/p1 $out/p2 $out/p3 $out/p4 | grep -A9 "$pat" > $out/summary.txt; cat $out/summary.txt
# (gpurun copies at most 64 MiB back: the raw per-dispatch rows stay on the box, the summary travels)
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
