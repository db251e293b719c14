#!/bin/bash
# usage: tools/variant_bench.sh "<EXTRA flags variant 1>" "<variant 2>" ...   (runs on the GPU box)
cd $GRAFT_REPO_ROOT/psk_soft_amd/csrc
for v in "$@"; do
  rm -f psk_kernels.o
  make EXTRA="-DPSK_ONLY_S8H1 $v" > /dev/null 2>&1 || { echo "BUILD FAILED: $v"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I. -DPSK_ONLY_S8H1 $v -c psk_kernels.hip -o /tmp/v.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "ILi8ELi1" | grep -E "VGPRs:|Spill|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '
  echo
  (cd $GRAFT_REPO_ROOT && python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VARIANT [$v]: ms_per_step=%.3f launch_ms=%.3f frac=%.3f check=%s'%(d['ms_per_step'], d['roofline']['launch_ms_avg'], d['roofline']['frac'], d.get('check')))")
done
