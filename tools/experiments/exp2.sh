cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or stack or model" 2>&1 | tail -3
for nw in 4 8; do for mw in 1 4; do
  echo "== WAVES=$nw MINW=$mw"; GWEN_K4_WAVES=$nw GWEN_K4_MINW=$mw python tools/kbench.py k4 16 32 64 128 2>&1 | grep K4
done; done
echo "== WAVES=8 MINW=6"; GWEN_K4_WAVES=8 GWEN_K4_MINW=6 python tools/kbench.py k4 16 32 64 128 2>&1 | grep K4
echo "== K2 (remap default)"; python tools/kbench.py k2 16 32 64 128 256 2>&1 | grep K2
