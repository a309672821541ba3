cd $GRAFT_REPO_ROOT
for st in 0 4 8 16 32 64; do echo "== STAGGER=$st (W4 M1)"; GWEN_K4_STAGGER=$st python tools/kbench.py k4 64 2>&1 | grep K4; done
for st in 8 32; do echo "== STAGGER=$st (W8 M1)"; GWEN_K4_WAVES=8 GWEN_K4_STAGGER=$st python tools/kbench.py k4 64 2>&1 | grep K4; done
