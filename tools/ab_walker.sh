#!/bin/bash
# usage (GPU box): tools/ab_walker.sh "<channels ...>" <libA.so> <libB.so> ...  -- per-kernel averages of the parallel fit's launches
# (rocprofv3 kernel statistics of few-channel calls) with prebuilt copies of libpsk_soft_hip.so swapped in, e.g. the timing
# variants of the walker (-DPSK_DIAG_WALK_NOCHAIN / _NOLOAD, psk_pfit.h; their sums are wrong, pf_verify refuses the calls)
CH=$1; shift
R=$GRAFT_REPO_ROOT
cp $R/psk_soft_amd/libpsk_soft_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $R/$lib $R/psk_soft_amd/libpsk_soft_hip.so
  echo "#### $lib"
  $R/tools/prof_tiled.sh "$CH" ${NSAMP:-1048576} abw/$(basename $lib .so) | grep -E "==|xwalk"
done
cp /tmp/lib_keep.so $R/psk_soft_amd/libpsk_soft_hip.so
