cd /tmp; export TMPDIR=/tmp
for abl in 0 1 11; do
  export GWEN_K4_ABL=$abl
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr9_$abl -- python3 $GRAFT_REPO_ROOT/tools/kbench.py all 64 > $GRAFT_REPO_ROOT/gpurun_out/tr9_$abl.log 2>&1
  echo "== ABL=$abl"; grep -E "K4|K2|K3" $GRAFT_REPO_ROOT/gpurun_out/tr9_$abl.log
  cut -d, -f1-8 $GRAFT_REPO_ROOT/gpurun_out/tr9_$abl/*/*kernel_stats.csv | grep -E "k_layer|k_propagate|k_linear" | sed 's/(anonymous namespace):://; s/(int const.*)"/"/' | cut -c1-150
done
