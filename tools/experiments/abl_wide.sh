for nw in 0 1; do echo "== reorder=hilbert NW16=$nw"; GWEN_WIDE_NW16=$nw KB_WHICH=k8 python tools/hbm_regime.py 100:256:4 100:256:1 100:128:4 100:64:4 300:64:1 2>&1 | grep -E "K8"; done
