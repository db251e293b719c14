#!/bin/bash
# usage (GPU box): tools/variant_test.sh "<EXTRA flags>" ...  -- rebuilds everything per variant and runs a quick parity subset
cd $GRAFT_REPO_ROOT/psk_soft_amd/csrc
for v in "$@"; do
  rm -rf obj
  make -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $v"; tail -5 /tmp/make.log; continue; }
  (cd $GRAFT_REPO_ROOT && python -m pytest tests/test_gpu_parity.py -q -m gpu -k "single_channel_parity" 2>&1 | tail -1 | sed "s|^|VARIANT [$v]: |")
done
