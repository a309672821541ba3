cd $GRAFT_REPO_ROOT
for nw in 4 8; do for rpw in 0 16 32 48 64; do echo "== WAVES=$nw RPW=$rpw"; GWEN_K4_WAVES=$nw GWEN_K4_RPW=$rpw python tools/kbench.py k4 32 64 128 2>&1 | grep K4; done; done
