#!/bin/bash
# Runs on the GPU box: headline bench + rocprofv3 kernel stats + PMC passes (separate runs, as the
# microarch guide prescribes).  Outputs under gpurun_out/<tag>/.
tag=${1:-final}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --check > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $R/bench.py --no-cpu-baseline --no-few --no-extra > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-few --no-extra > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-few --no-extra > $out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-few --no-extra > $out/pmc_sq.log 2>&1
cd $R/tools/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_floor stream_floor.hip && /tmp/stream_floor > $out/stream_floor.txt 2>&1
tail -c 600 $out/bench.json
