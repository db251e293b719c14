#!/bin/bash
# ISA of one wave-scan instantiation:  tools/isa_dump.sh S H E out.s [extra hipcc flags]
# (same flags as psk_soft_amd/csrc/Makefile; device code only)
set -e
S=$1; H=$2; E=$3; OUT=$4; shift 4
cd "$(dirname "$0")/../psk_soft_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -I../../include -I. \
  -DPSK_INST_S=$S -DPSK_INST_H=$H -DPSK_INST_E=$E --cuda-device-only -S "$@" -o "$OUT" psk_fast_inst.hip 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" "$OUT" | tr -s ' ' | tr '\n' ' '
echo
