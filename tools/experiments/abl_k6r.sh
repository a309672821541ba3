#!/bin/bash
# variants of the row-stationary K6 (timing): each line of VARIANTS = extra -D flags
cd $GRAFT_REPO_ROOT
while read -r flags; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $flags -Iinclude -Igwen_amd/csrc -c gwen_amd/csrc/interact_rows.hip -o /tmp/ir_v.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gwen_amd/libgwen_hip.so $(ls gwen_amd/build/*.o | grep -v interact_rows) /tmp/ir_v.o
  echo "== $flags"; timeout -k 10 200 python3 tools/inet_bench.py --channels ${CH:-256} --reorder hilbert 2>&1 | tail -5 | head -4 | cut -c1-48
done <<VARIANTS
-DABL=0
-DABL=5
VARIANTS
