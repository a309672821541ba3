#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + the two PMC passes over bench.py, outputs under gpurun_out/$1
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; counters never together with --kernel-trace)
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $B --steps 100 --warmup 10 > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $B --steps 20 --warmup 2 --event-stride 1000 --hbm-steps 3 --edge-mlp-members 0 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $B --steps 20 --warmup 2 --event-stride 1000 --hbm-steps 3 --edge-mlp-members 0 > $out/pmc_write.log 2>&1
# the 8-member edge-MLP block alone (k_mlp2r is one kernel name for 1 and 8 members: its own pair of passes)
E8="$B --steps 2 --warmup 1 --event-stride 1000 --no-hbm-leg --edge-mlp-members 8 --edge-mlp-skip-single"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc8_fetch -- python3 $E8 > $out/pmc8_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc8_write -- python3 $E8 > $out/pmc8_write.log 2>&1
tail -1 $out/trace.log | cut -c1-400
