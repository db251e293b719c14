#!/bin/bash
# tools/isa_exec_spills.py over every instantiation of the current build (CPU only: hipcc -S per object, 8 at a time).
# usage: tools/lint_all.sh out.txt
out=${1:-/tmp/lint_all.txt}
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
list=""
for s in $(seq 2 32); do for h in 1 2 4; do for e in 0 1; do list="$list $s,$h,$e"; done; done; done
for s in $(seq 2 16); do for e in 0 1; do list="$list $s,8,$e"; done; done
for s in $(seq 2 16); do list="$list $s,0,0"; done  # (H = 0: the screened variant without window history)
one() { IFS=, read s h e <<< "$1"; extra=""; 
  if [ $e = 1 ] && { [ $h = 8 ] || { [ $s -ge 17 ] && [ $h -ge 2 ]; } || { [ $s -ge 11 ] && [ $h = 4 ]; }; }; then extra="-mllvm -amdgpu-spill-sgpr-to-vgpr=0"; fi
  tools/isa_dump.sh $s $h $e $2/k_${s}_${h}_${e}.s $extra > /dev/null 2>&1
  python3 tools/isa_exec_spills.py $2/k_${s}_${h}_${e}.s --sites 0 | sed "s/^k_/S,H,E = /"
  python3 tools/isa_lane_loss.py $2/k_${s}_${h}_${e}.s --sites 3 | sed "s/^k_/LANES S,H,E = /" >> $2/lanes.txt; rm -f $2/k_${s}_${h}_${e}.s; }
export -f one
echo $list | tr ' ' '\n' | grep . | xargs -P 8 -I{} bash -c "one {} $tmp" > $out.unsorted
sort -t_ -k2,2n -k3,3n -k4,4n $out.unsorted > $out; rm -f $out.unsorted; cp $tmp/lanes.txt $out.lanes; rm -f $tmp/lanes.txt
# the other translation units: the time-tiled kernels and the parallel fit (psk_tile.hip), the reference-order kernel (psk_kernels.hip),
# the instantiated front stages (psk_tile_inst.hip, samplesPerBaud 2 ... 16)
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Iinclude -Ipsk_soft_amd/csrc --cuda-device-only -S"
others() { f=$1; shift; /opt/rocm/bin/hipcc $FL "$@" -o $tmp/$f.s psk_soft_amd/csrc/${f%%@*}.hip 2>/dev/null
  python3 tools/isa_exec_spills.py $tmp/$f.s --sites 0 >> $out; python3 tools/isa_lane_loss.py $tmp/$f.s --sites 3 >> $out.lanes; rm -f $tmp/$f.s; }
others psk_tile; others psk_kernels
for s in $(seq 2 16); do others psk_tile_inst@S$s -DPSK_INST_S=$s -DPSK_INST_H=1; done
echo "instantiations with D sites / with E sites (tools/isa_lane_loss.py):"; grep -c "no covering save) [1-9]" $out.lanes; grep -c "mask restore) [1-9]" $out.lanes
grep -c . $out; grep -v "C (lane carrier moved under a partial mask) 0" $out | wc -l
rmdir $tmp
