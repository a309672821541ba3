#!/bin/bash
# bench.py's HBM legs (the real 4-layer stacks, event-timed) under K8 build variants (GWEN_HIP_LIB)
#   tools/experiments/bench_variants.sh "<name> ..."
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for n in ${1:-product}; do
  if [ "$n" = product ]; then unset GWEN_HIP_LIB; else export GWEN_HIP_LIB=$PWD/gwen_amd/variants/libgwen_hip.$n.so; fi
  echo "=== $n" | tee -a gpurun_out/bench_variants.log
  timeout -k 10 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-edge-mlp --no-exact --allow-variant --no-live-traffic 2>&1 | tail -1 > gpurun_out/bv_$n.json
  python - "$n" <<'PY' | tee -a gpurun_out/bench_variants.log
import json, sys
d = json.load(open(f"gpurun_out/bv_{sys.argv[1]}.json"))
print("c2 ms/step", round(d["ms_per_step"], 4))
for k in ("hbm_leg", "hbm_leg_64ch"):
    r = d[k]["roofline"]
    print(k, d[k]["contraction"][:6], r["kernel"], "avg us", r["avg_launch_us"], "frac", r["frac"], "ms/step", d[k]["ms_per_step"])
    for kk in ("default_precision_bf16x6",):
        if kk in d[k]:
            r2 = d[k][kk]["roofline"]
            print("    ", kk, r2["kernel"], "avg us", r2["avg_launch_us"], "frac", r2["frac"])
PY
done
