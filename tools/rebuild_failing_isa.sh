#!/bin/bash
# Rebuilds the ISA of round 2's failing translation unit -- psk_fast_kernel<30,4,true> of the commit before the load-path change
# (a7a7c2b~1), with that commit's compiler flags -- and of the same source with the flag that hid the failure, then runs the two
# lints over them ("The mechanism, named").  CPU only.
#   tools/rebuild_failing_isa.sh [outdir, default /tmp/tail]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/tmp/tail}
wt=$(mktemp -d)
mkdir -p $out
git -C $R worktree add -f $wt a7a7c2b~1 > /dev/null 2>&1
cd $wt/psk_soft_amd/csrc
FLAGS="-O3 -std=c++17 -ffp-contract=off -Wno-unused-function -I../../include -I. -DPSK_INST_S=30 -DPSK_INST_H=4 -DPSK_INST_E=1 --cuda-device-only -S"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -o $out/s30h4e1_bad.s psk_fast_inst.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -mllvm -amdgpu-spill-sgpr-to-vgpr=0 -o $out/s30h4e1_good.s psk_fast_inst.hip 2>/dev/null
cd $R
git worktree remove --force $wt
python3 tools/isa_exec_spills.py $out/s30h4e1_bad.s $out/s30h4e1_good.s --sites 0 || true
python3 tools/isa_lane_loss.py $out/s30h4e1_bad.s $out/s30h4e1_good.s --sites 4 || true
