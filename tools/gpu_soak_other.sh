#!/bin/bash
# a second randomised soak of the final binary on OTHER scenarios than gpu_final's (REINA_SOAK_OFFSET): HIP vs oracle B bit for bit
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
export REINA_SOAK_OFFSET=${3:-100000}
sha256sum reina_model_amd/csrc/libreina_hip.so | cut -d' ' -f1 > $OUT/soak_other_lib_sha256.txt
timeout ${1:-600} python tools/parity_soak.py 100000 > $OUT/soak_unsharded_other_scenarios.txt 2>&1; tail -2 $OUT/soak_unsharded_other_scenarios.txt
timeout ${2:-480} python tools/parity_soak.py 100000 sharded > $OUT/soak_sharded_other_scenarios.txt 2>&1; tail -2 $OUT/soak_sharded_other_scenarios.txt
grep -c MISMATCH $OUT/soak_unsharded_other_scenarios.txt $OUT/soak_sharded_other_scenarios.txt
