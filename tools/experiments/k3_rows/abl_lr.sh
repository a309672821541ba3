#!/bin/bash
cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -DLR_ABL=$abl -Iinclude -Igwen_amd/csrc -c gwen_amd/csrc/linear_rows.hip -o /tmp/lr_v.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gwen_amd/libgwen_hip.so $(ls gwen_amd/build/*.o | grep -v linear_rows) /tmp/lr_v.o
  echo "== LR_ABL=$abl"; timeout -k 10 200 python3 tools/experiments/k3_time.py 2>&1 | grep -E "^ *(100002|400008) +256 +(256|768) " | cut -c1-60
done
