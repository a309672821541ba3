cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or stack or model or propagate" 2>&1 | tail -3
for nw in 4 8; do for mw in 1 4; do
  echo "== WAVES=$nw MINW=$mw"; GWEN_K4_WAVES=$nw GWEN_K4_MINW=$mw python tools/kbench.py k4 16 32 64 128 2>&1 | grep K4
done; done
for abl in 1 2 3; do echo "== ABL=$abl (W4 M1)"; GWEN_K4_ABL=$abl python tools/kbench.py k4 64 2>&1 | grep K4; done
