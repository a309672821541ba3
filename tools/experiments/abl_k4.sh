#!/bin/bash
# K4 store-pattern ablation (timing only): whole-row stores of the wrong data
cd $GRAFT_REPO_ROOT
for fl in "" "-DK4_ABL"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math $fl -Iinclude -Igwen_amd/csrc -c gwen_amd/csrc/layer.hip -o /tmp/layer_v.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gwen_amd/libgwen_hip.so $(ls gwen_amd/build/*.o | grep -v "/layer.o") /tmp/layer_v.o
  echo "== [$fl]"; timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-exact --no-edge-mlp --no-hbm-leg 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['all_kernels_us'])"
done
