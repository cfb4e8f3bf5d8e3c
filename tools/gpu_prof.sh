#!/bin/bash
# where a wave of k_day spends its cycles: one -DREINA_DAY_PROF=k build per part of the loop (built here, on the GPU box), each
# run on a peak and a quiet day.  usage: bash tools/gpu_prof.sh [agents]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
for k in ${PARTS:-1 2 3 4 5 6 8 9}; do /opt/rocm/bin/hipcc $F -DREINA_DAY_PROF=$k -o /tmp/libreina_prof$k.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null & done; wait
for k in ${PARTS:-1 2 3 4 5 6 8 9}; do
  REINA_PROF_WHAT=$k REINA_HIP_LIB=/tmp/libreina_prof$k.so python tools/day_prof.py ${1:-100000000} 93 200 2>&1 | grep "^part\|workgroups"
done
