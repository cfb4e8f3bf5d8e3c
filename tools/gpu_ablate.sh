#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_ABLATE -o /tmp/libreina_ab.so reina_model_amd/csrc/reina_hip.hip 2>&1 | grep -E "error" | head
for b in 0 1 2 3 4; do REINA_HIP_LIB=/tmp/libreina_ab.so python tools/ablate_day.py $b 2>/dev/null; done
for b in 0 4; do ABLATE_START=300 REINA_HIP_LIB=/tmp/libreina_ab.so python tools/ablate_day.py $b 2>/dev/null; done
