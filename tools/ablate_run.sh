#!/bin/bash
# usage (GPU box only): tools/ablate_run.sh "<sections variant 1>" "<sections variant 2>" ...
# per variant: instruction counters (one PMC pass) and launch time of the headline kernel
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ablate
mkdir -p $out
cp $R/psk_soft_amd/csrc/psk_fast_loop.h /tmp/loop_orig.h
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  cp /tmp/loop_orig.h $R/psk_soft_amd/csrc/psk_fast_loop.h
  (cd $R && python3 tools/ablate.py $v) || continue
  (cd $R/psk_soft_amd/csrc && rm -f obj/psk_fast_S8_H1_E0.o obj/psk_fast_S8_H1_E1.o && make -j16 > /tmp/make.log 2>&1) || { echo "BUILD FAILED $v"; tail -5 /tmp/make.log; continue; }
  (cd $R && python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ABLATE [$v]: launch_ms=%.3f stats=%s'%(d['roofline']['launch_ms_avg'], d['kernel_stats']))")
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM --output-format csv -d $out/v$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/v$i.log 2>&1)
  (cd $R && python3 tools/pmc_summary.py $out/v$i | grep -A8 "8, 1, false" | grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_ACTIVE_INST_VALU|SQ_INSTS_LDS" | awk '{printf "   %s %s\n", $1, $3}')
done
cp /tmp/loop_orig.h $R/psk_soft_amd/csrc/psk_fast_loop.h
