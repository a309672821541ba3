#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -DGWEN_HOPS_STAMPS -Iinclude -Igwen_amd/csrc gwen_amd/csrc/hops.hip -o tools/experiments/libhops_stamps.so 2>&1 | grep -E "error" 
for a in "100 1 1" "100 1 3" "100 4 3"; do timeout -k 10 100 python3 tools/experiments/hops_stamps.py $a 2>&1 | grep -v amdgpu.ids; done
