KB_WHICH=k8 python tools/hbm_regime.py 100:256:4 100:256:1 100:128:4 100:128:1 100:64:4 300:64:1 2>&1 | grep -E "K8|nu="
KB_REORDER=morton KB_WHICH=k8 python tools/hbm_regime.py 100:256:4 100:128:4 2>&1 | grep -E "K8|nu="
