#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats (+ PMC traffic) of the c3 and c5-shaped workloads -> gpurun_out/$1
tag=${1:-cfg}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${2:-f16x3}       # precision of the c3 legs: f16x3 = the library default (K8, one launch per layer), 3xbf16 = the 17-bit tier
for m in 1 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3_m$m -- python3 $R/tools/c3_bench.py 256 4 $m --no-parity --precision $P > $out/c3_m$m.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/c3_m${m}_fetch -- python3 $R/tools/c3_bench.py 256 4 $m --no-parity --precision $P > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/c3_m${m}_write -- python3 $R/tools/c3_bench.py 256 4 $m --no-parity --precision $P > /dev/null 2>&1
done
python3 $R/tools/c3_bench.py 256 4 1 --precision $P > $out/c3_m1_parity.json 2>/dev/null
python3 $R/tools/c3_bench.py 256 4 4 --precision $P > $out/c3_m4_parity.json 2>/dev/null
python3 $R/tools/c3_bench.py 256 4 1 --precision bf16x6 > $out/c3_m1_bf16x6.json 2>/dev/null
python3 $R/tools/c3_bench.py 256 4 1 --precision 3xbf16 > $out/c3_m1_3xbf16.json 2>/dev/null
python3 $R/tools/c3_bench.py 256 4 4 --precision 3xbf16 > $out/c3_m4_3xbf16.json 2>/dev/null
for h in 64 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/fc_$h -- python3 $R/tools/forecaster_bench.py 8 $h 4 4 > $out/fc_$h.log 2>&1
done
tail -1 $out/c3_m1_parity.json $out/c3_m4_parity.json | cut -c1-700
