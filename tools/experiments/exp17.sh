cd /tmp; export TMPDIR=/tmp
for m in alone zeros small denorm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/e17_$m -- python3 $GRAFT_REPO_ROOT/tools/exp17.py $m > /dev/null 2>&1
  echo "== $m"; grep -E "k_propagate|k_chain|k_layer" $GRAFT_REPO_ROOT/gpurun_out/e17_$m/*/*kernel_stats.csv | sed 's/.*"void (anonymous namespace):://; s/(int const[^"]*"//' | cut -d, -f1-4
done
