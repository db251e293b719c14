#!/bin/bash
# usage: tools/variant_bench.sh "<EXTRA flags variant 1>" "<variant 2>" ...   (runs on the GPU box)
# rebuilds only the headline instantiations (S=8, H=1, screened + exact) with the given flags
cd $GRAFT_REPO_ROOT/psk_soft_amd/csrc
for v in "$@"; do
  rm -f obj/psk_fast_S8_H1_E0.o obj/psk_fast_S8_H1_E1.o
  make -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I. -DPSK_INST_S=8 -DPSK_INST_H=1 -DPSK_INST_E=0 $v -c psk_fast_inst.hip -o /tmp/v.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs:|Spill|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
  echo
  (cd $GRAFT_REPO_ROOT && python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$v]: ms_per_step=%.3f launch_ms=%.3f frac=%.3f check=%s exact=%s'%(d['ms_per_step'], d['roofline']['launch_ms_avg'], d['roofline']['frac'], d.get('check'), d['kernel_stats']['channels_exact_timing']))")
done
