#!/bin/bash
# K8 build variants (gwen_amd/build.py::build_variant -> gwen_amd/variants/libgwen_hip.<name>.so, selected by
# GWEN_HIP_LIB; the product library is never touched): parity tests of the wide layer, then its timings.
#   tools/experiments/k8_variants.sh "<name> <name> ..." "<nu:F:M points>"
cd ${GRAFT_REPO_ROOT:-.}
names=${1:-"product"}
pts=${2:-"100:256:4 100:256:1 100:64:16 100:128:4"}
mkdir -p gpurun_out
for n in $names; do
  if [ "$n" = product ]; then unset GWEN_HIP_LIB GWEN_ALLOW_VARIANT_TESTS; else export GWEN_HIP_LIB=$PWD/gwen_amd/variants/libgwen_hip.$n.so GWEN_ALLOW_VARIANT_TESTS=1; fi
  echo "=== $n" | tee -a gpurun_out/k8_variants.log
  timeout -k 10 300 python -m pytest tests/test_gpu_wide.py -m gpu -x -q 2>&1 | tail -2 | tee -a gpurun_out/k8_variants.log
  KB_WHICH=k8 timeout -k 10 300 python tools/hbm_regime.py $pts 2>&1 | grep -E "K8|nu=" | tee -a gpurun_out/k8_variants.log
done
