#!/bin/bash
# on the GPU box: relink the library with a stamped interact_rows.o (scratch copy of the repo) and run the reader
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -DGWEN_K6R_STAMPS -Iinclude -Igwen_amd/csrc -c gwen_amd/csrc/interact_rows.hip -o /tmp/ir_st.o 2>&1 | grep error
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gwen_amd/libgwen_hip.so $(ls gwen_amd/build/*.o | grep -v interact_rows) /tmp/ir_st.o
timeout -k 10 200 python3 tools/experiments/k6r_stamps.py ${F:-256} 2>&1 | grep -v amdgpu.ids
