#!/bin/bash
# ablations of the row-stationary K6 (timing only; results are wrong by construction): ABL=1 one W fragment
# pair per step from LDS, ABL=2 no per-step barrier
cd $GRAFT_REPO_ROOT
for abl in 3 4; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -DABL=$abl -Iinclude -Igwen_amd/csrc -c gwen_amd/csrc/interact_rows.hip -o /tmp/ir_$abl.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gwen_amd/libgwen_hip.so $(ls gwen_amd/build/*.o | grep -v interact_rows) /tmp/ir_$abl.o
  echo "== ABL=$abl"; timeout -k 10 200 python3 tools/inet_bench.py --channels 256 --reorder hilbert 2>&1 | tail -5 | head -4
done
