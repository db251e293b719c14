#!/bin/bash
# CPU-only: the host side of libpsk_soft_hip.so (psk_capi.cpp: C ABI, control plane, ingest pipeline
# bookkeeping) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer, device code untouched
# (-fno-gpu-sanitize; GPU sanitizers are not available on this pool), and the CPU test tier run on it.
# The regular library is put back afterwards.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd $R/psk_soft_amd/csrc
make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -I../../include -I. \
  -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c psk_capi.cpp -o $T/psk_capi.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize \
  -o $T/libpsk_soft_hip.so $T/psk_capi.o $(ls obj/*.o | grep -v psk_capi.o)
cp ../libpsk_soft_hip.so $T/orig.so
trap "cp $T/orig.so $R/psk_soft_amd/libpsk_soft_hip.so; rm -rf $T" EXIT
cp $T/libpsk_soft_hip.so ../libpsk_soft_hip.so
cd $R
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1) \
  python -m pytest tests -x -q -m "not gpu"
