#!/bin/bash
# dynamic instruction mix of the headline kernel by class (GPU box); output under gpurun_out/<tag>/
tag=${1:-valu_mix}; shift
bargs="$@"
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-few --no-extra --no-check $bargs > $out/$name.log 2>&1; }
run m1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT
run m2 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SALU SQ_INSTS_LDS
run m3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_IOPS SQ_INSTS_BRANCH SQ_INSTS_SMEM
cd $R && python3 tools/pmc_summary.py $out/m1 $out/m2 $out/m3 | grep -A9 "8, 1, false" | grep -v "8, 1, true" > $out/summary.txt; cat $out/summary.txt
rm -rf $out/m1 $out/m2 $out/m3
