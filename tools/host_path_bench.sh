#!/bin/bash
# builds and runs tools/host_path_bench.cpp on the GPU box; args: channels samples calls
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 -std=c++17 -I include tools/host_path_bench.cpp -L psk_soft_amd -lpsk_soft_hip -Wl,-rpath,$PWD/psk_soft_amd -o /tmp/host_path_bench || exit 1
/tmp/host_path_bench "$@"
