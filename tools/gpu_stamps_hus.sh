#!/bin/bash
# in-kernel phase stamps at the HUS size: the small ordered event walk, the install roles, the day-opening launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
F="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DREINA_HOSP_STAMPS -o /tmp/libreina_hs.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null
/opt/rocm/bin/hipcc $F -DREINA_INSTALL_STAMPS -o /tmp/libreina_is.so reina_model_amd/csrc/reina_hip.hip 2>/dev/null
REINA_HIP_LIB=/tmp/libreina_hs.so timeout 300 python tools/stamps_peak.py 1685983 hosp 2>&1 | tee $OUT/stamps_hosp_hus.txt
REINA_HIP_LIB=/tmp/libreina_is.so timeout 300 python tools/stamps_peak.py 1685983 inst 2>&1 | tee $OUT/stamps_inst_hus.txt
