#!/bin/bash
# in-kernel phase stamps of k_hosp_install (-DREINA_INSTALL_STAMPS) at HUS and 10^8 agents
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-st}; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_INSTALL_STAMPS -o /tmp/libreina_is.so reina_model_amd/csrc/reina_hip.hip 2>&1 | tail -3
for n in ${2:-1e8 1.7e6}; do
REINA_HIP_LIB=/tmp/libreina_is.so python tools/stamps_install.py $n 2>&1 | grep days | tee $OUT/${TAG}_stamps_inst_$n.txt
done
